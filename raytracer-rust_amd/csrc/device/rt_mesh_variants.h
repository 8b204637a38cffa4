// rt_mesh_variants.h -- the two earlier forms of the mesh path, kept selectable (MI355RT_KERNEL=2 / 5) as bit-identity references: state machine, walk pool
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
    ps.rng.k0 = ps.rng.k1 = ps.rng.x = ps.rng.s = ps.rng.ray = 0; ps.rng.b0[0] = ps.rng.b0[1] = ps.rng.b0[2] = ps.rng.b0[3] = 0;
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

// ===================================================================================================
// k_render_ctr_pool -- the state machine with its BVH walks handed to dedicated WALKER waves through LDS.
//
// Measured on k_render_ctr_sm (profiles/, stamps): its BVH rounds run at ~40 % of the lanes -- walks end at different
// lengths and a finished lane can only be refilled by its own path, which first needs a TOP and a SHADE pass.  Here the 16
// waves of the workgroup (one per CU, sharing LDS) split into roles:
//   producers (16 - W waves)  the state machine without its TRAV block: TOP / SHADE passes over their own paths.  A lane
//                             that reaches a mesh writes a walk REQUEST (object-space ray, 1/d, t_max, root node: 12 dwords)
//                             into its fixed LDS slot, publishes the slot number in a ring, and waits (state WAIT) until the
//                             slot's flag says the RESULT (best_t, best triangle) is there; then it goes on exactly where the
//                             in-wave walk would have returned (mesh_accept).
//   walkers   (W waves)       persistent loops: every lane without a walk takes the next ring ticket (one ds_add per wave) and
//                             picks its request up when the ticket's entry is filled; four box tests + the pending leaves per
//                             iteration, refill in between -- a finished lane is refilled with ANY path's walk, so the walk
//                             instructions run near full lanes.
// Per lane the walk is the same mesh_step / mesh_leaf sequence on the same inputs, so images are bit-identical to
// k_render_ctr_sm.  No barrier after start-up; every spin is bounded (a watchdog count sets an error word and every wave
// leaves), and the exit conditions do not depend on scheduling order: producers finish when their paths are done and count
// themselves out; walkers leave when no producer is left (no request can be outstanding then).
// LDS: control 64 B | ring 4 KB | flags 3 KB | results 6 KB | requests 36 KB | node copy (<= POOL_NODE_CAP nodes).
// ===================================================================================================
constexpr uint32_t POOL_MAX_PRODUCER_LANES = 768;          // 12 producer waves (W >= 4)
constexpr uint32_t POOL_RING = 1024;                        // > POOL_MAX_PRODUCER_LANES: a path has at most one request in flight
constexpr uint32_t POOL_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t POOL_CTRL_WORDS = 16, POOL_REQ_WORDS = 12;
constexpr uint32_t POOL_FIXED_BYTES = 4u * (POOL_CTRL_WORDS + POOL_RING + POOL_MAX_PRODUCER_LANES + 2u * POOL_MAX_PRODUCER_LANES + POOL_REQ_WORDS * POOL_MAX_PRODUCER_LANES);
static_assert(POOL_NODE_CAP * 32u + POOL_FIXED_BYTES <= 163840u, "pool kernel LDS budget");
enum : uint32_t { ST_WAIT = 2 };                            // a producer lane whose walk is with the walkers (the slot of ST_TRAV)

template <bool FIXED_AABB>
DI void render_ctr_pool(const RenderParams& P) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[POOL_FIXED_BYTES / 4u + 8u * POOL_NODE_CAP];
    uint32_t* const ctrl = s_pool;                                          // [0] ring tail, [1] ring head, [2] producers still running, [3] error
    uint32_t* const ring = ctrl + POOL_CTRL_WORDS;
    uint32_t* const flags = ring + POOL_RING;
    uint32_t* const results = flags + POOL_MAX_PRODUCER_LANES;              // 2 words per slot
    uint32_t* const requests = results + 2u * POOL_MAX_PRODUCER_LANES;      // POOL_REQ_WORDS per slot, 16-byte aligned
    float4* const s_nodes = reinterpret_cast<float4*>(requests + POOL_REQ_WORDS * POOL_MAX_PRODUCER_LANES);
    cprim_t prims = (cprim_t)(P.prims);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(P.nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(P.tris);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t n_walkers = P.walker_waves;                               // host guarantees 4 <= W < n_waves
    const uint32_t lds_count = P.lds_nodes;
    for (uint32_t i = threadIdx.x; i < 2u * lds_count; i += blockDim.x) s_nodes[i] = n4[i];
    for (uint32_t i = threadIdx.x; i < POOL_RING; i += blockDim.x) ring[i] = POOL_EMPTY;
    for (uint32_t i = threadIdx.x; i < POOL_MAX_PRODUCER_LANES; i += blockDim.x) flags[i] = 0u;
    if (threadIdx.x < POOL_CTRL_WORDS) ctrl[threadIdx.x] = threadIdx.x == 2u ? (n_waves - n_walkers) : 0u;
    __syncthreads();
    lds_nodes_t lds = (lds_nodes_t)s_nodes;

    if (wave < n_walkers) {
        // ------------------------------------------------ walker ------------------------------------------------
        // Every lane carries POOL_WALKS independent walks: their node / triangle loads are in flight together (twice the
        // memory-level parallelism per wave slot -- the walkers are the only waves that load nodes) and their box tests interleave.
#ifndef MI355RT_POOL_WALKS
#define MI355RT_POOL_WALKS 1                               // measured: 2 walks per lane 15.4 -> 20.1 ms on semesterbild -- the requests in flight cannot fill more walk slots
#endif
#ifndef MI355RT_POOL_STEPS
#define MI355RT_POOL_STEPS 8                               // box tests per walker iteration
#endif
        constexpr int NW = MI355RT_POOL_WALKS;
        MeshTrav m[NW]; bool has[NW]; uint32_t ticket[NW], slot[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            m[w].ro = mk(0, 0, 0); m[w].rd = mk(0, 0, 1); m[w].ix = m[w].iy = m[w].iz = 0.f; m[w].len_raw = 0.f; m[w].node = NODE_END; m[w].best_t = 0.f;
            m[w].best_tri = 0xFFFFFFFFu; m[w].leaf_a = m[w].leaf_b = 0; has[w] = false; ticket[w] = POOL_EMPTY; slot[w] = 0;
        }
        uint32_t spins = 0, seen_progress = 0;
        // The watchdogs of this kernel count polls without PROGRESS in the workgroup, as the wavefront kernel's does (rt_wavefront.h): the sum of the
        // ring counters ([0] requests published, [1] tickets drawn), [4] producer passes run and [5] results handed back moves whenever any wave
        // of the workgroup gets something done; a wave that sees it move starts counting again.
        auto progress = [&]() { return __hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + __hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) +
                                         __hip_atomic_load(&ctrl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + __hip_atomic_load(&ctrl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        for (;;) {
            // refill: a walk slot with neither a walk nor a ticket draws the next ticket (one ds_add per wave for all of them);
            // a ticketed slot takes its request once the ring entry is filled
            uint64_t wm[NW]; uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { wm[w] = __ballot(!has[w] && ticket[w] == POOL_EMPTY); total += (uint32_t)__popcll(wm[w]); }
            if (total != 0u) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctrl[1], total);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (!has[w] && ticket[w] == POOL_EMPTY) ticket[w] = base + mbcnt64(wm[w]);
                    base += (uint32_t)__popcll(wm[w]);
                }
            }
            bool any = false;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (!has[w] && ticket[w] != POOL_EMPTY) {
                    const uint32_t got = atomicExch(&ring[ticket[w] & (POOL_RING - 1u)], POOL_EMPTY);
                    if (got != POOL_EMPTY) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        slot[w] = got; ticket[w] = POOL_EMPTY; has[w] = true;
                        const float4* rq = reinterpret_cast<const float4*>(requests + POOL_REQ_WORDS * got);
                        const float4 a = rq[0], b = rq[1], c = rq[2];
                        m[w].ro = mk(a.x, a.y, a.z); m[w].rd = mk(a.w, b.x, b.y); m[w].ix = b.z; m[w].iy = b.w; m[w].iz = c.x;
                        m[w].best_t = c.y; m[w].node = __float_as_uint(c.z); m[w].best_tri = 0xFFFFFFFFu; m[w].leaf_a = m[w].leaf_b = 0;
                    }
                }
                any = any || has[w];
            }
            if (__ballot(any) != 0ull) {
                spins = 0;
                // POOL_STEPS box tests, then the pending leaves, then refill.  Measured on semesterbild (800x600x64, 4 walkers): 1 step per
                // refill 29.7 ms, 2 -> 20.6, 4 -> 15.8, 8 -> 14.4, 16 -> 15.5, 32 -> 18.4; the state machine's voted rounds (leaf phase as
                // soon as twice as many lanes wait for one as walk) 14.8 -- the refill / ticket logic is what the steps amortise.
#pragma unroll
                for (int u = 0; u < MI355RT_POOL_STEPS; ++u) {
#pragma unroll
                    for (int w = 0; w < NW; ++w)
                        if (has[w] && m[w].leaf_b == 0u && m[w].node != NODE_END) mesh_step<FIXED_AABB, 1>(n4, lds, lds_count, EPS, m[w]);
                }
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (has[w] && m[w].leaf_b != 0u) mesh_leaf(t4, EPS, m[w]);
                    if (has[w] && m[w].leaf_b == 0u && m[w].node == NODE_END) {          // walk over: hand the result back
                        results[2u * slot[w]] = __float_as_uint(m[w].best_t); results[2u * slot[w] + 1u] = m[w].best_tri;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(&flags[slot[w]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&ctrl[5], 1u);
                        has[w] = false;
                    }
                }
            } else {
                if (__hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;     // no producer left: nothing can be outstanding
                if (__hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;
                __builtin_amdgcn_s_sleep(2);
                { const uint32_t pr = progress(); if (pr != seen_progress) { seen_progress = pr; spins = 0; } }
                if (++spins > P.spin_limit_idle) { if (lane == 0) atomicOr(&ctrl[3], (uint32_t)WAIT_POOL_WALKER_IDLE); break; }
            }
        }
        if (lane == 0 && P.err) {
            const uint32_t waits = __hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (waits != 0u) { atomicAdd(P.err, 1ull); atomicOr(P.err, (unsigned long long)waits << 32); }
        }
        return;
    }

    // ------------------------------------------------ producer ------------------------------------------------
    const uint32_t my_slot = (wave - n_walkers) * 64u + lane;
    WorkCursor wc; wc.init();
    PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
    ps.rng.k0 = ps.rng.k1 = ps.rng.x = ps.rng.s = ps.rng.ray = 0; ps.rng.b0[0] = ps.rng.b0[1] = ps.rng.b0[2] = ps.rng.b0[3] = 0;
    uint32_t state = ST_IDLE, cursor = 0, spins = 0, stall = 0, seen_progress = 0;
    bool walk_done = false;
    Cand best; cand_reset(best);
    float len_raw = 0.f, res_t = 0.f; uint32_t res_tri = 0xFFFFFFFFu;        // what mesh_accept needs of the walk once it is back
    uint32_t n_paths = 0, n_rays = 0;
    Prof prof; prof.begin();
    const uint32_t min_ready = P.trav_min;
    bool failed = false;
    for (;;) {
        // results that have arrived
        if (state == ST_WAIT && __hip_atomic_load(&flags[my_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            res_t = __uint_as_float(results[2u * my_slot]); res_tri = results[2u * my_slot + 1u];
            __hip_atomic_store(&flags[my_slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            state = ST_TOP; walk_done = true;
        }
        const uint32_t nW = (uint32_t)__popcll(__ballot(state == ST_WAIT));
        const uint32_t nP = (uint32_t)__popcll(__ballot(state == ST_TOP));
        const uint32_t nS = (uint32_t)__popcll(__ballot(state == ST_SHADE));
        const uint32_t nI = wc.exhausted() ? 0u : (uint32_t)__popcll(__ballot(state == ST_IDLE));
        if (nW + nP + nS + nI == 0u) break;
        // Lanes are out with the walkers: unless enough of the others are ready, wait for more results to come back, so that the
        // TOP / SHADE passes run well filled (their instructions are the larger half of the kernel).
        if (nW != 0u && nP + nS + nI < min_ready) {
            if (__hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) { failed = true; break; }
            __builtin_amdgcn_s_sleep(2);
            {   // (progress anywhere in the workgroup -- a request taken, a result handed back, a pass run -- restarts the count; see the walkers)
                const uint32_t pr = __hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + __hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) +
                                    __hip_atomic_load(&ctrl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + __hip_atomic_load(&ctrl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pr != seen_progress) { seen_progress = pr; stall = 0; } }
            ++spins;
            if (++stall > P.spin_limit_idle) { if (lane == 0) atomicOr(&ctrl[3], (uint32_t)WAIT_POOL_RESULTS); failed = true; break; }
            if (spins < P.pool_patience || nP + nS + nI == 0u) continue;         // waited long enough: run what is there
        }
        spins = 0;
        if (lane == 0) atomicAdd(&ctrl[4], 1u);                                    // a pass is about to run: progress

        if (nP != 0u && nP >= nS + nI) {
            // ---- TOP: hittable.rs:45-58 from each lane's cursor ----
            for (uint32_t i = 0; i < P.n_prims; ++i) {
                const bool mine = (state == ST_TOP) && cursor == i;
                if (__ballot(mine) == 0ull) continue;
                cprim_t pr = prims + i;
                bool submit = false;
                if (mine) {
                    bool advance = true;
                    switch (pr->kind) {                                       // wave-uniform: scalar branch
                        case MI355RT_PRIM_SPHERE: hit_sphere(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_PLANE:  hit_plane(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_QUAD:   hit_quad(pr, i, ps.ro, ps.rd, EPS, best); break;
                        case MI355RT_PRIM_CUBE:   hit_cube(pr, i, ps.ro, ps.rd, EPS, best); break;
                        default:
                            if (!walk_done) {
                                MeshTrav mt; mesh_setup(pr, ps.ro, ps.rd, best.t, mt);
                                len_raw = mt.len_raw;
                                bool gone = false;
                                if (P.inline_steps != 0u) {
                                    // several meshes share the list: most rays miss a mesh's root box -- test it here and spare them the round trip
                                    // (the walker tests the root again for the others: same inputs, same result)
                                    const uint32_t root = mt.node;
                                    mesh_step<FIXED_AABB, 1>(n4, lds, lds_count, EPS, mt);
                                    gone = mt.leaf_b == 0u && mt.node == NODE_END;
                                    mt.node = root; mt.leaf_b = 0u; mt.leaf_a = 0u;
                                }
                                if (gone) { res_t = mt.best_t; res_tri = 0xFFFFFFFFu; walk_done = true; }
                                else {
                                    float4* rq = reinterpret_cast<float4*>(requests + POOL_REQ_WORDS * my_slot);
                                    rq[0] = make_float4(mt.ro.x, mt.ro.y, mt.ro.z, mt.rd.x);
                                    rq[1] = make_float4(mt.rd.y, mt.rd.z, mt.ix, mt.iy);
                                    rq[2] = make_float4(mt.iz, mt.best_t, __uint_as_float(mt.node), 0.f);
                                    submit = true;
                                }
                            }
                            if (walk_done) {
                                MeshTrav mt; mt.best_t = res_t; mt.best_tri = res_tri; mt.len_raw = len_raw;
                                mesh_accept(i, mt, ps.rd, EPS, best); walk_done = false;
                            } else { state = ST_WAIT; advance = false; }
                            break;
                    }
                    if (advance) ++cursor;
                }
                const uint64_t sm = __ballot(submit);                          // publish the new requests: one ring reservation per wave
                if (sm != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    uint32_t base = 0;
                    if (lane == (uint32_t)__builtin_ctzll(sm)) base = atomicAdd(&ctrl[0], (uint32_t)__popcll(sm));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(sm));
                    if (submit) __hip_atomic_store(&ring[(base + mbcnt64(sm)) & (POOL_RING - 1u)], my_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (state == ST_TOP && cursor == P.n_prims) state = ST_SHADE;
            prof.mark(1);
            continue;
        }
        // ---- SHADE + regeneration (lanes in TOP / WAIT are left untouched) ----
        bool live = (state == ST_SHADE);
        const bool any_hit = live && best.idx != CAND_NONE;
        Hit h; h.t = 0.f; h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.mat_ff = 0;
        if (any_hit) finish_hit<true>(P.prims, P.tris, best, ps.ro, ps.rd, h);
        const bool part = live || state == ST_IDLE;
        shade_and_regenerate<MATS_ALL>(P, wc, lane, live, part, any_hit, h, ps, n_paths, n_rays, prof);
        if (part) {
            if (live) { state = ST_TOP; cursor = 0; cand_reset(best); walk_done = false; }
            else state = ST_IDLE;
        }
        prof.mark(4);
    }
    if (lane == 0) atomicSub(&ctrl[2], 1u);                                    // this producer is done (also when it gave up)
    const uint32_t wp = wave_sum(n_paths), wr = wave_sum(n_rays);
    if (lane == 0 && P.stats) {
        atomicAdd(&P.stats[0], (unsigned long long)wp); atomicAdd(&P.stats[1], (unsigned long long)wr);
    }
    if (lane == 0 && P.err && failed) { atomicAdd(P.err, 1ull); atomicOr(P.err, (unsigned long long)__hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << 32); }
}
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_pool(const RenderParams P) { render_ctr_pool<false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_pool_fixaabb(const RenderParams P) { render_ctr_pool<true>(P); }


}  // namespace mi355rt
