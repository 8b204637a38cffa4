// rt_mesh_variants.h -- round 1's form of the mesh path, the wave-scheduled state machine, kept selectable (knob "kernel" = 2) in the tests' reference build as a
// bit-identity reference.  (Round 2's LDS walk pool lived here too until round 5: it reported a stall once in round 3 whose cause was never
// established, so it is no bit-identity reference to trust and was removed -- DESIGN.md 4.1d "Watchdogs"; git show ae77408:<this file> has it.)
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once

namespace mi355rt {

// ===================================================================================================
// k_render_ctr_sm -- the same path tracer as a wave-scheduled state machine, for scenes with meshes.
// A per-lane BVH walk makes a lockstep wave run as long as its slowest ray (measured: 14 % VALU lane
// utilisation on semesterbild).  Here every lane is in one of three states and each loop iteration the
// wave VOTES (ballot + popcount) which block to run:
//   TRAV   one "while-while" round of the threaded BVH walk (inner-node steps until every walking lane has a
//          leaf pending or is done, then the leaf triangle tests) -- cheap, run while >= trav_min lanes walk;
//   TOP    the top-level list from each lane's own cursor (records still come through scalar loads: the
//          list index is wave-uniform, lanes join when it reaches their cursor); a mesh primitive either
//          starts a walk (-> TRAV) or, when its walk is done, finalises the hit and moves on;
//   SHADE  shade_and_regenerate() for lanes whose list is finished (and idle lanes).
// Lanes that finish a walk early wait in TOP until enough of them have gathered, instead of idling inside
// a divergent while loop.  Results are bit-identical to the lockstep kernel: every lane executes exactly the
// same arithmetic in the same per-lane order.
// ===================================================================================================
enum : uint32_t { ST_IDLE = 0, ST_TOP = 1, ST_TRAV = 2, ST_SHADE = 3 };

template <bool FIXED_AABB>
DI void render_ctr_state_machine(const RenderParams& P) {
    cprim_t prims = (cprim_t)(P.prims);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(P.nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(P.tris);
    const uint32_t lane = threadIdx.x & 63u;
    // The workgroup (all 16 waves of the CU) copies the hot top of the node array -- the whole array when it fits -- into
    // LDS once; from then on a box test costs two ds_read_b128 instead of two L2 round trips.
    __shared__ float4 s_nodes[2u * LDS_NODE_CAP];
    const uint32_t lds_count = P.lds_nodes;
    for (uint32_t i = threadIdx.x; i < 2u * lds_count; i += blockDim.x) s_nodes[i] = n4[i];
    __syncthreads();
    lds_nodes_t lds = (lds_nodes_t)s_nodes;                       // explicit cast into the LDS address space: ds_read, not flat_load
    WorkCursor wc; wc.init();
    PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
    ps.rng.clear();
    uint32_t state = ST_IDLE, cursor = 0;
    bool walk_done = false;
    Cand best; cand_reset(best);                                       // the list's running winner (4 registers; the record is built at SHADE)
    MeshTrav mt; mt.ro = mk(0, 0, 0); mt.rd = mk(0, 0, 1); mt.ix = mt.iy = mt.iz = 0.f; mt.len_raw = 0.f; mt.node = NODE_END; mt.best_t = 0.f;
    mt.best_tri = 0xFFFFFFFFu; mt.leaf_a = mt.leaf_b = 0;
    uint32_t n_paths = 0, n_rays = 0;
    Prof prof; prof.begin();
    const uint32_t trav_min = P.trav_min;
#ifdef MI355RT_STAMPS
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_dry = 0ull; uint32_t drain_iters = 0, live_at_dry = 0;
    unsigned long long c_exec[4] = {0, 0, 0, 0}, c_lanes[4] = {0, 0, 0, 0};    // 0 inner steps, 1 leaf phases, 2 TOP passes, 3 SHADE passes
#define MI355RT_COUNT(i, mask) do { c_exec[i] += 1; c_lanes[i] += (unsigned long long)__popcll(mask); } while (0)
#else
#define MI355RT_COUNT(i, mask) do {} while (0)
#endif

    for (;;) {
        const uint32_t nT = (uint32_t)__popcll(__ballot(state == ST_TRAV));
        const uint32_t nP = (uint32_t)__popcll(__ballot(state == ST_TOP));
        const uint32_t nS = (uint32_t)__popcll(__ballot(state == ST_SHADE));
        const uint32_t nI = wc.exhausted() ? 0u : (uint32_t)__popcll(__ballot(state == ST_IDLE));
        if (nT + nP + nS + nI == 0u) break;
#ifdef MI355RT_STAMPS
        if (wc.exhausted()) {                              // all work dealt: from here on the wave only drains its own paths
            if (t_dry == 0ull) { t_dry = __builtin_amdgcn_s_memrealtime(); live_at_dry = nT + nP + nS; }
            ++drain_iters;
        }
#endif

        if (nT != 0u && (nT >= trav_min || nP + nS + nI == 0u)) {
            // ---- TRAV: one while-while round.  Inner-node steps and the leaf phase are themselves voted: step
            //      while at least as many lanes are walking as have a leaf pending, then test the leaves ----
#ifndef MI355RT_TRAV_STEPS
#define MI355RT_TRAV_STEPS 16
#endif
#ifndef MI355RT_TRAV_UNROLL
#define MI355RT_TRAV_UNROLL 4                              // box tests per vote (the vote costs a third of a step; A/B: 1 -> 4 = -5 %, 8 and 16 lose again)
#endif
            for (int it = 0; it < MI355RT_TRAV_STEPS; it += MI355RT_TRAV_UNROLL) {
                const bool walking = (state == ST_TRAV) && mt.leaf_b == 0u && mt.node != NODE_END;
                const uint64_t wm = __ballot(walking);
                const uint64_t lm = __ballot(state == ST_TRAV && mt.leaf_b != 0u);
                if (wm == 0ull || __popcll(wm) * MI355RT_TRAV_BIAS < __popcll(lm)) break;
                MI355RT_COUNT(0, wm);
                if (walking) {
                    mesh_step<FIXED_AABB, 1>(n4, lds, lds_count, EPS, mt);
#pragma unroll
                    for (int u = 1; u < MI355RT_TRAV_UNROLL; ++u)
                        if (mt.leaf_b == 0u && mt.node != NODE_END) mesh_step<FIXED_AABB, 1>(n4, lds, lds_count, EPS, mt);
                }
            }
            MI355RT_COUNT(1, __ballot(state == ST_TRAV && mt.leaf_b != 0u));
            if (state == ST_TRAV && mt.leaf_b != 0u) mesh_leaf(t4, EPS, mt);
            if (state == ST_TRAV && mt.leaf_b == 0u && mt.node == NODE_END) { state = ST_TOP; walk_done = true; }
            prof.mark(0);
            continue;
        }
        if (nP != 0u && nP >= nS + nI) {
            // ---- TOP: hittable.rs:45-58 from each lane's cursor ----
            // (Serving one list segment per pass -- the cursor most lanes wait at -- was measured and dropped: the passes are
            // already homogeneous on semesterbild, 44.6 lanes either way, and it fragments teapot's passes: 27.7 -> 33.1 ms.)
            MI355RT_COUNT(2, __ballot(state == ST_TOP));
            for (uint32_t i = 0; i < P.n_prims; ++i) {
                const bool mine = (state == ST_TOP) && cursor == i;
                if (__ballot(mine) == 0ull) continue;
                cprim_t pr = prims + i;
                if (mine) {
                    bool advance = true;
                    switch (pr->kind) {                                       // wave-uniform: scalar branch
                        case MI355RT_PRIM_SPHERE: hit_sphere(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_PLANE:  hit_plane(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_QUAD:   hit_quad(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_CUBE:   hit_cube(pr, i, ps.ro, ps.rd, EPS, best); break;
                        default:
                            if (!walk_done) {
                                // Most rays leave a mesh within a few box tests (they miss its root or upper boxes):
                                // take those steps right here so that only long walks pay a TRAV / TOP round trip.
                                mesh_setup(pr, ps.ro, ps.rd, best.t, mt);
#pragma unroll 1
                                for (uint32_t k = 0; k < P.inline_steps; ++k) {
                                    if (mt.leaf_b != 0u || mt.node == NODE_END) break;
                                    mesh_step<FIXED_AABB, 1>(n4, lds, lds_count, EPS, mt);
                                }
                                if (mt.leaf_b == 0u && mt.node == NODE_END) walk_done = true;   // walked off the tree without meeting a leaf
                            }
                            if (walk_done) { mesh_accept(i, mt, ps.rd, EPS, best); walk_done = false; }
                            else { state = ST_TRAV; advance = false; }
                            break;
                    }
                    if (advance) ++cursor;
                }
            }
            if (state == ST_TOP && cursor == P.n_prims) state = ST_SHADE;
            prof.mark(1);
            continue;
        }
        // ---- SHADE + regeneration (lanes in TOP / TRAV are left untouched) ----
        bool live = (state == ST_SHADE);
        const bool any_hit = live && best.idx != CAND_NONE;
        Hit h; h.t = 0.f; h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.mat_ff = 0;
        if (any_hit) finish_hit<true>(P.prims, P.tris, best, ps.ro, ps.rd, h);             // the winner's HitRecord, once per ray
        const bool part = live || state == ST_IDLE;
        MI355RT_COUNT(3, __ballot(part));
        shade_and_regenerate<MATS_ALL>(P, wc, lane, live, part, any_hit, h, ps, n_paths, n_rays, prof);
        if (part) {
            if (live) { state = ST_TOP; cursor = 0; cand_reset(best); walk_done = false; }
            else state = ST_IDLE;
        }
        prof.mark(4);
    }
#ifdef MI355RT_STAMPS
    if (lane == 0 && P.stats) {
        for (int i = 0; i < 6; ++i) atomicAdd(&P.stats[2 + i], prof.acc[i]);
        for (int i = 0; i < 4; ++i) { atomicAdd(&P.stats[8 + 2 * i], c_exec[i]); atomicAdd(&P.stats[9 + 2 * i], c_lanes[i]); }
    }
    if (P.wave_times) {
        const unsigned long long t_wave1 = __builtin_amdgcn_s_memrealtime();
        const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const uint32_t np = wave_sum(n_paths);
        if (lane == 0) {
            unsigned long long* w = P.wave_times + WAVE_TIME_WORDS * (size_t)wid;
            w[0] = t_wave0; w[1] = t_wave1; w[2] = np; w[3] = t_dry ? t_dry : t_wave1; w[4] = drain_iters; w[5] = live_at_dry;
        }
    }
#endif
    const uint32_t wp = wave_sum(n_paths), wr = wave_sum(n_rays);
    if (lane == 0 && P.stats) { atomicAdd(&P.stats[0], (unsigned long long)wp); atomicAdd(&P.stats[1], (unsigned long long)wr); }
}
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_sm(const RenderParams P) { render_ctr_state_machine<false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_sm_fixaabb(const RenderParams P) { render_ctr_state_machine<true>(P); }


}  // namespace mi355rt
