// rt_wavefront.h -- k_render_ctr_wf: path state in LDS slots, stages as queues (the default for scenes with meshes)
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once

// shade_and_regenerate's form in the wavefront kernel (rt_kernels.hip): without default values and with the generator's multiplies as 64-bit products,
// like the lockstep kernels.  Round 2 measured both the other way round (38 spilled registers with defaults, 120 without; the wide
// multiply 2-3 % slower) -- on a kernel whose SHADE spilled 50 registers.  With the metal's loop gone and the material sets
//  the allocation has room: semesterbild / teapot 800x600x256 (ms): defaults + narrow 28.45 / 16.88, defaults + wide
// 28.17 / 16.66, no defaults + narrow 28.17 / 16.83, no defaults + wide 27.91 / 16.62.
// The forms this kernel settled on, each measured against its alternative (the A/B builds and their numbers: docs/kernels/wavefront_round3.md,
// docs/rounds/r04.md, r05.md; the alternatives are in git history, not in this file):
//   * shade_and_regenerate without default values for its temporaries, Philox / pcg4d on 64-bit products, and (round 5) with the generator state
//     derived inside it (REKEY) -- round 2 measured the first two the other way round on a kernel whose SHADE spilled 50 registers;
//   * the short reciprocal / square root of rt_math.h in mesh_setup (where TOP meets a mesh, where a WALK pass re-enters one) and in TOP's cube and
//     quad tests: semesterbild 27.13 -> 26.77 ms, teapot 16.31 -> 16.09; in SHADE's normalisations only in the mesh-free form (the forms with the BVH
//     walk lose a third of that gain again there: one more spilled register);
//   * every kind finishes its own hit record in SHADE (the shared set_face_normal tail of the mesh-free kernels costs this kernel 3-4 %);
//   * wave priority by phase: 2 while a wave chooses its stage, pops and pushes (chains of dependent LDS round trips), 3 inside a WALK pass (a chain
//     of node loads with ~35 instructions per link), 1 over the top-level list (primitive reads), 0 in SHADE's arithmetic.

namespace mi355rt {

constexpr int WF_PRIO_SCHED = 2, WF_PRIO_WALK = 3, WF_PRIO_TOP = 1;
constexpr bool WF_FAST_MESH = true;                         // mesh_setup / TOP's cube and quad tests on the short reciprocal and square root (rt_math.h)

// ===================================================================================================
// k_render_ctr_wf -- the path tracer as a WAVEFRONT inside one workgroup: path state lives in LDS, stages are queues.
//
// The lockstep loop and the state machine keep a path in the registers of ONE lane for its whole life, so every pass of every
// stage runs with whatever lanes of that wave happen to be in that stage (measured: 0.40 of the lanes on semesterbild).  Here
// the CU's workgroup owns WF_PATHS path slots in LDS (20 dwords each) and queues of slot numbers -- FREE, WALK (a BVH walk in
// progress), TOP1 (a ray whose walk is back), SHADE x 4 material classes.  Every wave loops: look at the queue lengths, choose a
// stage, pop up to 64 of its slots, load what that stage needs, run the stage with (nearly) all lanes busy, store what changed,
// push each slot to the queue of its next stage.  (A new ray has no queue of its own: the SHADE pass that generates it walks the
// head of the list for it right away.)  A path therefore migrates between waves;
// per path the arithmetic is exactly that of the other kernels (same device functions, same inputs, same order), so images
// are bit-identical.  Regeneration stays in SHADE: a finished path's slot is refilled from the wave's own work cursor in the
// same pass, and SHADE passes top themselves up from the FREE queue.
// Queues: one ring of 1 024 u16 per stage (> WF_PATHS, a slot is in at most one queue), `tail` reserved by ds_add, `head`
// advanced by ds_cmpst so that a pop never takes more than is there; an entry is written after its ticket is reserved, so a
// popper may have to wait a few cycles for it (bounded spin) and writes EMPTY back; a pusher whose entry is still occupied (the
// popper of the previous ring revolution has reserved it but not read it yet) waits for that popper, so no slot number is ever lost.
// No barrier after start-up.  A wave leaves when its work cursor is exhausted and no path is alive in the workgroup.
// ===================================================================================================
// Two workgroups of 12 waves per CU (24 waves = 6 per SIMD at 80 VGPRs), 832 slots each: the passes begin with a chain of
// dependent LDS round trips (pop, ring entry, slot) and the walk reads its nodes from L1/L2, so waves to switch to are worth more
// than registers.  Measured when the kernel still had its TOP0 queue (semesterbild / teapot, 800x600x64, ms): 1 x 16 waves,
// 1 728 slots 11.60 / 7.42;  2 x 12 waves, 832 slots each 10.65 / 6.65;  3 x 8 waves, 512 each 11.33 / 6.86;  2 x 14 at 72 VGPRs 14.8 / 9.8 and 2 x 16 at 64 VGPRs
// 15.3 / 9.0 (spills);  2 x 10 at 96 VGPRs 15.2 / 9.7;  1 x 16 waves with 960 fat slots (36 dwords) 11.6 / 7.3.
constexpr uint32_t WF_EMPTY = 0xFFFFu, WF_WALK_DONE = 0x80000000u;
// SHADE is four queues, one per material class of the hit: a pass whose slots all take the same branch of Material::scatter pays
// for that branch only (a mixed pass pays for the sum of all branches that any of its lanes takes).
enum : uint32_t { WQ_FREE = 0, WQ_WALK = 1, WQ_TOP1 = 2,
                  WQ_SHADE = 3,      // + class: 0 terminal (miss / emissive / null: the path ends, the slot regenerates), 1 diffuse (Lambert,
                                     //          checker, texture, plastic), 2 rough conductor, 3 specular (metal, dielectric)
                  WQ_NONE = 15 };
DI uint32_t shade_class(uint32_t kind) {
    return (kind == MI355RT_MAT_EMISSIVE || kind == MI355RT_MAT_NULL) ? 0u
         : (kind == MI355RT_MAT_ROUGH_GGX || kind == MI355RT_MAT_ROUGH_BECKMANN) ? 2u
         : (kind == MI355RT_MAT_METAL || kind == MI355RT_MAT_DIELECTRIC) ? 3u : 1u;
}
constexpr uint32_t WF_PATHS_MESHFREE = 1023;                // 64-byte slots beside the seven rings: 79 936 of the 81 920 bytes
static_assert(WF_PATHS < WF_RING && WF_PATHS < WF_EMPTY, "a ring holds every slot number");
// (The less a path carries, the more paths fit, and the fill of every pass follows from their number: 960 slots of 36 dwords ran SHADE at 37 of 64 lanes.
// Recomputed instead of stored: the generator state (from sidx), the walk's object-space ray and 1/d (mesh_setup per WALK pass: +3 % instructions),
// |w2o d| for the (sic) t_world.)

// ---------------------------------------------------------------------------------------------------
// Slot layout, 20 words.  What a path carries between passes: the ray, the throughput, its sample and ray index, the list cursor, the
// candidate (closest hit so far) and -- while a BVH walk is parked -- the walk (next node, best t, best triangle):
//   q0 ro.xyz thr.x | q1 rd.xyz thr.y | q2 thr.z sidx ray cursor(+WALK_DONE) | q3 cand t idx aux aux2 | q4 node best_t best_tri -
// (mesh-free lists use q0..q3 of it: 16 words).  A packed 16-word form (1 023 slots instead of 832) was built in round 3 and measured
// -0.6 % / +0.9 % (semesterbild / teapot): the packing arithmetic costs what the extra slots buy; it is in git history (ae77408), not here.
// ---------------------------------------------------------------------------------------------------
struct WalkRec { uint32_t node; float best_t; uint32_t best_tri; };
struct WalkKeep { uint32_t cursor_word; };                             // what a WALK pass writes back unchanged
struct SlotIO {
    DI static void load_shade(const uint32_t* sl, f3& ro, f3& rd, f3& thr, uint32_t& sidx, uint32_t& ray, Cand& c) {
        const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1];
        const float4 d = reinterpret_cast<const float4*>(sl)[2], g = reinterpret_cast<const float4*>(sl)[3];
        ro = mk(a.x, a.y, a.z); rd = mk(b.x, b.y, b.z); thr = mk(a.w, b.w, d.x);
        sidx = __float_as_uint(d.y); ray = __float_as_uint(d.z);
        c.t = g.x; c.idx = __float_as_uint(g.y); c.aux = g.z; c.aux2 = __float_as_uint(g.w);
    }
    DI static void store_shade(uint32_t* sl, f3 ro, f3 rd, f3 thr, uint32_t sidx, uint32_t ray) {       // a new ray: cursor 0, no candidate
        reinterpret_cast<float4*>(sl)[0] = make_float4(ro.x, ro.y, ro.z, thr.x);
        reinterpret_cast<float4*>(sl)[1] = make_float4(rd.x, rd.y, rd.z, thr.y);
        reinterpret_cast<float4*>(sl)[2] = make_float4(thr.z, __uint_as_float(sidx), __uint_as_float(ray), __uint_as_float(0u));
        reinterpret_cast<float4*>(sl)[3] = make_float4(__builtin_inff(), __uint_as_float(CAND_NONE), 0.f, 0.f);
    }
    DI static void store_top(uint32_t* sl, const Cand& c, uint32_t cursor, uint32_t, const WalkRec& w, bool parked) {
        reinterpret_cast<float4*>(sl)[3] = make_float4(c.t, __uint_as_float(c.idx), c.aux, __uint_as_float(c.aux2));
        sl[11] = cursor;
        if (parked) reinterpret_cast<float4*>(sl)[4] = make_float4(__uint_as_float(w.node), w.best_t, __uint_as_float(w.best_tri), 0.f);
    }
    DI static void load_walk(const uint32_t* sl, f3& ro, f3& rd, uint32_t& cursor, WalkRec& w, WalkKeep& k) {
        const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1], q = reinterpret_cast<const float4*>(sl)[4];
        ro = mk(a.x, a.y, a.z); rd = mk(b.x, b.y, b.z);
        k.cursor_word = sl[11]; cursor = k.cursor_word & ~WF_WALK_DONE;
        w.node = __float_as_uint(q.x); w.best_t = q.y; w.best_tri = __float_as_uint(q.z);
    }
    DI static void store_walk(uint32_t* sl, const WalkRec& w, bool done, const WalkKeep& k) {
        reinterpret_cast<float4*>(sl)[4] = make_float4(__uint_as_float(w.node), w.best_t, __uint_as_float(w.best_tri), 0.f);
        if (done) sl[11] = k.cursor_word | WF_WALK_DONE;
    }
    DI static void load_top1(const uint32_t* sl, f3& ro, f3& rd, Cand& c, uint32_t& cursor, bool& done, uint32_t& ray, WalkRec& w) {
        const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1], g = reinterpret_cast<const float4*>(sl)[3];
        const float4 q = reinterpret_cast<const float4*>(sl)[4];
        ro = mk(a.x, a.y, a.z); rd = mk(b.x, b.y, b.z);
        c.t = g.x; c.idx = __float_as_uint(g.y); c.aux = g.z; c.aux2 = __float_as_uint(g.w);
        const uint32_t cw = sl[11];
        cursor = cw & ~WF_WALK_DONE; done = (cw & WF_WALK_DONE) != 0u; ray = 0u;
        w.node = __float_as_uint(q.x); w.best_t = q.y; w.best_tri = __float_as_uint(q.z);
    }
};

// Ring entries are polled, hence volatile -- and the compiler's address-space inference leaves volatile accesses alone: through a plain pointer they were
// FLAT loads / stores (nine of each in the kernel), every one followed by s_waitcnt vmcnt(0) lgkmcnt(0), i.e. a wait for ALL of the wave's outstanding
// global memory traffic in every pop and every push.  The pointer says where the rings are: ds_read_u16 / ds_write_b16, counted in lgkmcnt alone.
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
struct WfQueues {
    uint32_t* ctrl;        // [q] head, [8 + q] tail, [16] live paths, [17] error
    uint16_t* rings;       // WF_QUEUES x WF_RING slot numbers
    uint32_t entry_spins;  // bound for the wait on one ring entry (RenderParams.spin_limit_entry)
    // Pop up to `want` entries of queue q for lanes [lane0, lane0 + n): returns n; those lanes get their slot in `id`.
    // `at_least`: take nothing if fewer are there by now -- every wave reads the same queue lengths, so several decide for the
    // same stage at once and all but the first would get scraps (measured: SHADE at 38 of 64 lanes); they look again instead.
    DI uint32_t pop(uint32_t q, uint32_t want, uint32_t at_least, uint32_t lane, uint32_t lane0, uint32_t& id, bool& failed) const {
        uint32_t h = 0, n = 0;
        if (lane == 0) {
            for (;;) {
                h = __hip_atomic_load(&ctrl[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t t = __hip_atomic_load(&ctrl[8u + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                n = min(t - h, want);
                if (n < at_least) { n = 0u; break; }
                if (n == 0u || atomicCAS(&ctrl[q], h, h + n) == h) break;
            }
        }
        h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h); n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
        if (lane >= lane0 && lane < lane0 + n) {
            volatile lds_u16_t* e = (volatile lds_u16_t*)(rings + q * WF_RING + ((h + lane - lane0) & (WF_RING - 1u)));
            uint32_t v = WF_EMPTY, spins = 0;
            for (;;) {                                                   // the pusher reserved this ticket and is about to write it
                v = *e;
                if (v != WF_EMPTY) break;
                if (++spins > entry_spins) { failed = true; break; }
            }
            *e = (uint16_t)WF_EMPTY;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            id = (v == WF_EMPTY) ? 0u : v;                               // after a failed wait the wave leaves; keep the address in range until then
        }
        return n;
    }
    DI void push(uint32_t q, bool pred, uint32_t id, uint32_t lane, bool& failed) const {
        const uint64_t m = __ballot(pred);
        if (m == 0ull) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the slot's stores are visible before its number is
        const uint32_t first = (uint32_t)__builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&ctrl[8u + q], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
        if (pred) {
            volatile lds_u16_t* e = (volatile lds_u16_t*)(rings + q * WF_RING + ((base + mbcnt64(m)) & (WF_RING - 1u)));
            // The entry of ticket T is free once the popper of ticket T - WF_RING has read it and written EMPTY back.  That popper
            // exists (a ring holds more entries than there are slots, so ticket T - WF_RING was popped before T could be reserved);
            // if it has been held up between reserving and reading, wait for it instead of overwriting its entry.
            uint32_t spins = 0;
            while (*e != WF_EMPTY) { if (++spins > entry_spins) { failed = true; break; } }
            *e = (uint16_t)id;
        }
    }
    // Every lane with `pred` pushes its slot to ITS queue `q` (lanes may name different queues), in ONE reservation: the pass's slots go to at most NQ queues (WALK and
    // the four SHADE classes), so lane j counts the slots bound for the j-th of them and a single ds_add_rtn_u32 -- per-lane address, per-lane count -- draws the tickets
    // of all queues at once: one LDS round trip instead of one per queue present (a loop over the queues present, typically three, until round 5: veach-mis -1.0 %,
    // the mesh scenes +-0; profiles/r05/ab_scalar_diet.txt).  Ticket order between queues is free (a path's arithmetic does not depend on which pass it rides in).
    template <bool MESH>
    DI void push_all(bool pred, uint32_t q, uint32_t id, uint32_t lane, bool& failed) const {
        if (__ballot(pred) == 0ull) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the slot's stores are visible before its number is
        constexpr int NQ = MESH ? 5 : 4;
        auto queue_at = [](uint32_t j) { return MESH ? (j == 0u ? (uint32_t)WQ_WALK : (uint32_t)WQ_SHADE + j - 1u) : (uint32_t)WQ_SHADE + j; };
        uint64_t m[NQ]; uint32_t bound = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) { m[j] = __ballot(pred && q == queue_at((uint32_t)j)); bound = (lane == (uint32_t)j) ? (uint32_t)__popcll(m[j]) : bound; }
        uint32_t base = 0;
        if (lane < (uint32_t)NQ) base = atomicAdd(&ctrl[8u + queue_at(lane)], bound);    // (a count of zero leaves that tail where it is)
        uint32_t ticket = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)base, j);
            ticket = (q == queue_at((uint32_t)j)) ? bj + mbcnt64(m[j]) : ticket;
        }
        if (pred) {
            volatile lds_u16_t* e = (volatile lds_u16_t*)(rings + q * WF_RING + (ticket & (WF_RING - 1u)));
            uint32_t spins = 0;
            while (*e != WF_EMPTY) { if (++spins > entry_spins) { failed = true; break; } }      // (see push(): the popper of the previous revolution)
            *e = (uint16_t)id;
        }
    }
    DI uint32_t count(uint32_t q) const {
        return __hip_atomic_load(&ctrl[8u + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - __hip_atomic_load(&ctrl[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// HAS_MESH = false: the same wavefront for lists without a mesh (picked when the materials of such a scene diverge): no
// WALK / TOP1 stages, 16-word slots (no walk state), 1 023 of them.
// MESH_IDENT: every mesh of the list is untransformed (the host picks this instantiation then): see ray_nonzero_finite() in rt_intersect.h.
// INLINE_STEPS: box tests of a walk that TOP runs itself when at least half the wave is inside the root box (8; 12 measured better for the deep trees of the
// scene the MESH_IDENT form serves: teapot -0.7 %, and worse for semesterbild's shallower one: +0.8 %).
template <bool FIXED_AABB, uint32_t MATS, bool HAS_MESH = true, bool MESH_IDENT = false, int INLINE_STEPS = 8, int WF_ROUNDS = 3, int WF_STEPS = 8>
DI void render_ctr_wavefront(const RenderParams& P) {
    typedef SlotIO Slot;
    constexpr uint32_t WF_PATHS = HAS_MESH ? mi355rt::WF_PATHS : WF_PATHS_MESHFREE, WF_SLOT_WORDS = HAS_MESH ? mi355rt::WF_SLOT_WORDS : 16u;
    constexpr uint32_t WF_LDS_WORDS = WF_CTRL_WORDS + WF_QUEUES * WF_RING / 2u + WF_PATHS * WF_SLOT_WORDS;
    static_assert(WF_LDS_WORDS <= WF_LDS_BUDGET_WORDS && WF_PATHS < WF_RING && WF_PATHS < WF_EMPTY, "wavefront kernel LDS budget / ring size");
    __shared__ __attribute__((aligned(16))) uint32_t s_wf[WF_LDS_WORDS];
    WfQueues Q; Q.ctrl = s_wf; Q.rings = reinterpret_cast<uint16_t*>(s_wf + WF_CTRL_WORDS); Q.entry_spins = P.spin_limit_entry;
    uint32_t* const slots = s_wf + WF_CTRL_WORDS + WF_QUEUES * WF_RING / 2u;
    // (The walk reads its nodes from L1 / L2.  A copy of the top of the node array in what the slots leave of the LDS budget was built in round 3 and
    // measured: worth nothing -- the walk is bound by instruction issue at low lane fill, not by node latency -- while every slot given up for it costs time.)
    cprim_t prims = (cprim_t)(P.prims);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(P.nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(P.tris);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < WF_QUEUES * WF_RING; i += blockDim.x) Q.rings[i] = (uint16_t)((i < WF_PATHS) ? i : WF_EMPTY);   // FREE holds every slot
    if (threadIdx.x < WF_CTRL_WORDS) Q.ctrl[threadIdx.x] = (threadIdx.x == 8u + WQ_FREE) ? WF_PATHS : 0u;
    __syncthreads();

    WorkCursorWf wc; wc.init();
    uint32_t n_paths = 0, n_rays = 0, spins = 0, seen_passes = 0;
    Prof prof; prof.begin();
    bool failed = false;
#ifdef MI355RT_STAMPS
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_dry = 0ull; uint32_t passes_after_dry = 0, alive_at_dry = 0;
    unsigned long long w_exec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w_lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // passes and slots per pass: 0 WALK, 1 TOP1, 2 walks parked by a TOP pass, 3 SHADE (+ free fill),
                                                                                                        // 4 box-test steps of WALK passes (lanes stepping), 5 leaf phases (lanes with a leaf), 6 WALK passes (walks finished), 7 inline steps in TOP
#define MI355RT_WFCOUNT(i, n) do { w_exec[i] += 1; w_lanes[i] += (n); } while (0)
#else
#define MI355RT_WFCOUNT(i, n) do {} while (0)
#endif
    // WALK geometry (template arguments): WF_ROUNDS rounds of WF_STEPS box tests + the pending leaves per pass.  Measured (ms, semesterbild / teapot at 64 spp): 1x8 11.8 / 7.5,
    // 2x8 10.7 / 6.6, 3x8 10.4 / 6.4, 4x8 10.4 / 6.3, 8x8 10.7 / 6.6, 3x12 10.7 / 6.4; per scene class since round 5: see k_render_ctr_wf_nometal_shallow.  The box test is unrolled
    // WF_STEPS times (the kernel is 47 KB of code; two CUs share a 64 KB instruction cache).
    // TOP: hittable.rs:45-58 from the slot's cursor; a mesh whose root box is hit sends the ray to WALK; at the end of the list the
    // slot is routed by the material class of its hit, so that SHADE passes are homogeneous.  Run by SHADE passes on the rays they
    // have just generated (still in registers) and by TOP1 passes on the slots whose walk is back.
    auto run_top = [&](const bool have, const f3 ro, const f3 rd, Cand c, uint32_t cursor, bool walk_done, uint32_t ray_index, WalkRec wk, uint32_t* sl, const uint32_t id) {
        bool to_walk = false;
        bool ident_ok = false;
        if constexpr (MESH_IDENT) ident_ok = __ballot(have && !ray_nonzero_finite(ro, rd)) == 0ull;
        if constexpr (!HAS_MESH) {
            // A mesh-free list parks nothing: every ray of the pass is at the head of the list and stays in step with the others to its end -- the lockstep
            // kernels' run loops (one dispatch per RUN of a kind, no vote and no cursor test per primitive) do the same arithmetic in the same order.
            // (veach-mis -0.6 %, profiles/r05/ab_scalar_diet.txt)
            if (have) walk_list<false>(prims, P.n_prims, nullptr, nullptr, ro, rd, c);
            cursor = P.n_prims;
        } else
        for (uint32_t i = 0; i < P.n_prims; ++i) {
            const bool mine = have && !to_walk && cursor == i;
            if (__ballot(mine) == 0ull) continue;
            cprim_t pr = prims + i;
            if (mine) {
                bool advance = true;
                switch (pr->kind) {                                       // wave-uniform: scalar branch
                    case MI355RT_PRIM_SPHERE: hit_sphere(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_PLANE:  hit_plane(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_QUAD:   hit_quad<!HAS_MESH || WF_FAST_MESH>(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_CUBE:   hit_cube<!HAS_MESH || WF_FAST_MESH>(pr, i, ro, rd, EPS, c); break;
                    default:
                        if constexpr (!HAS_MESH) break;                // (the host picks this instantiation for mesh-free lists only)
                        else if (!walk_done) {
                            MeshTrav mt; mesh_setup<WF_FAST_MESH>(pr, ro, rd, c.t, mt, ident_ok);
                            const uint32_t root = mt.node;
                            mesh_step<FIXED_AABB>(n4, nullptr, 0u, EPS, mt);                        // the root box, here: most rays miss it
                            if (mt.leaf_b == 0u && mt.node == NODE_END) { /* missed: no hit in this mesh */ }
                            else {
                                // lanes inside the root box for the first INLINE_STEPS steps of the walk to run right here: 32 (thresholds 1 / 32 / 48 lanes and
                                // 4 / 8 / 16 / 24 steps measured in round 2; 12 steps for the deep trees the MESH_IDENT form serves in round 4)
                                constexpr uint32_t WF_INLINE_MIN = 32;
                                bool parked = false;
                                if ((uint32_t)__popcll(__ballot(true)) >= WF_INLINE_MIN) {
#pragma unroll 1
                                    for (int u = 0; u < INLINE_STEPS; ++u) {
                                        if (mt.leaf_b != 0u) mesh_leaf(t4, EPS, mt);
                                        if (mt.node == NODE_END) break;
                                        MI355RT_WFCOUNT(7, (uint32_t)__popcll(__ballot(true)));
                                        mesh_step<FIXED_AABB>(n4, nullptr, 0u, EPS, mt);
                                    }
                                    if (mt.leaf_b != 0u) mesh_leaf(t4, EPS, mt);
                                    if (mt.node == NODE_END) { mesh_accept(i, mt, rd, EPS, c); parked = true; }      // the whole walk fitted: the list goes on
                                    else { wk.node = mt.node; wk.best_t = mt.best_t; wk.best_tri = mt.best_tri; }
                                } else {
                                    wk.node = root; wk.best_t = c.t; wk.best_tri = 0xFFFFFFFFu;
                                }
                                if (!parked) { to_walk = true; advance = false; }
                            }
                        } else {
                            MeshTrav mt; mt.best_t = wk.best_t; mt.best_tri = wk.best_tri; mt.len_raw = len(ident_ok ? rd : xform_w2o_dir(pr, rd));   // mesh_object.rs:288, again
                            mesh_accept(i, mt, rd, EPS, c); walk_done = false;
                        }
                        break;
                }
                if (advance) ++cursor;
            }
        }
        if (have) Slot::store_top(sl, c, cursor, ray_index, wk, to_walk);
        uint32_t cls = 0u;
        if (have && !to_walk && c.idx != CAND_NONE) cls = shade_class(prim_material_kind(P, c.idx));
        MI355RT_WFCOUNT(2, (uint32_t)__popcll(__ballot(have && to_walk)));
        __builtin_amdgcn_s_setprio(WF_PRIO_SCHED);                     // the pushes are LDS round trips again
        Q.template push_all<HAS_MESH>(have, to_walk ? (uint32_t)WQ_WALK : WQ_SHADE + cls, id, lane, failed);
    };
    for (;;) {
        // Wave priority (s_setprio): the stage choice and the pop are a chain of dependent LDS round trips with a handful of instructions
        // between them, and a WALK pass is a chain of node loads with ~35 instructions per link -- waves in these phases should get the
        // issue port the moment their data is back, the long arithmetic of SHADE / TOP fills the gaps.  Priority 2 while choosing and
        // popping and pushing, 3 in WALK, 0 in the arithmetic of SHADE and TOP: semesterbild 28.13 -> 27.60 -> 27.33 ms, teapot 16.74 -> 16.64 -> 16.42,
        // veach-mis 16.14 -> 16.01 -> 15.84
        // (profiles/r03_ab_wavefront_wave_priority.txt; WALK alone -1.4 / -0.3 / 0 %, TOP1 raised to 2 as well: no better).
        // Second pass, after the same idea paid 4 % in the lockstep kernels: SHADE keeps the raised priority until its fresh samples are
        // dealt (material read, radiance store, cursor atomic), and the top-level list runs at 1 rather than 0:
        // semesterbild 27.36 -> 27.25, teapot 16.44 -> 16.34, veach-mis 15.86 -> 15.66 (profiles/r03_ab_wavefront_wave_priority2.txt).
        __builtin_amdgcn_s_setprio(WF_PRIO_SCHED);
#ifdef MI355RT_STAMPS
        if (wc.exhausted()) {                              // all work dealt: from here on the workgroup only finishes the paths it holds
            if (t_dry == 0ull) { t_dry = __builtin_amdgcn_s_memrealtime(); alive_at_dry = __hip_atomic_load(&Q.ctrl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            ++passes_after_dry;
        }
#endif
        if (__ballot(failed) != 0ull) { if (lane == 0) atomicOr(&Q.ctrl[17], 1u); break; }
        if (__hip_atomic_load(&Q.ctrl[17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) { failed = true; break; }
        // (all heads and tails with four 16-byte LDS reads instead of fourteen 4-byte ones: +-0 on all three scenes, round 5)
        const uint32_t cT1 = HAS_MESH ? Q.count(WQ_TOP1) : 0u, cW = HAS_MESH ? Q.count(WQ_WALK) : 0u;
        const uint32_t cS0 = Q.count(WQ_SHADE), cS1 = Q.count(WQ_SHADE + 1u), cS2 = Q.count(WQ_SHADE + 2u), cS3 = Q.count(WQ_SHADE + 3u);
        const uint32_t cF = wc.exhausted() ? 0u : Q.count(WQ_FREE);
        // A pass costs its instructions whatever its fill, and the stages differ in price (SHADE ~1 800 instructions + ~700 for the
        // head of the list it goes on with, WALK ~750, TOP1 ~400): run the stage whose pass WASTES the fewest lane-instructions,
        // price x empty lanes.  A full queue wastes nothing; of two thin ones the cheap stage runs and the expensive one keeps
        // filling (measured with "fullest first": SHADE ran at 39 of 64 lanes).  Ties go to the later stage.
        uint32_t stage = WQ_NONE;
        {
            uint32_t waste = 0xFFFFFFFFu;
            auto consider = [&](uint32_t q, uint32_t n, uint32_t price) {
                if (n == 0u) return;
                const uint32_t w = price * (64u - min(n, 64u));
                if (w <= waste) { waste = w; stage = q; }
            };
            consider(WQ_WALK, cW, 11u); consider(WQ_TOP1, cT1, 4u);
            constexpr uint32_t T0 = 7;                                   // a SHADE pass goes on with the head of the list for the rays it generates
            consider(WQ_SHADE + 3u, cS3, 5u + T0); consider(WQ_SHADE + 2u, cS2, 10u + T0); consider(WQ_SHADE + 1u, cS1, 8u + T0);
            consider(WQ_SHADE, cS0 + cF, 5u + T0);                      // terminal class: free slots ride along (both only regenerate)
        }
        if (stage == WQ_NONE) {
            if (wc.exhausted() && __hip_atomic_load(&Q.ctrl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;   // nothing alive, nothing to start
            __builtin_amdgcn_s_sleep(2);
            // The watchdog measures lack of PROGRESS in the workgroup, not how long this wave has been idle: every pass ends with
            // pushes, which move a queue tail, so an idle wave that sees the sum of the tails move starts counting again.  (A few
            // waves tracing the last, very long paths of a band may keep the others idle for any length of time; that is not a
            // failure.  A dedicated progress counter bumped per pass was measured first: +1.1 / +1.6 % -- one more hot LDS word.)
            uint32_t passes = 0;
#pragma unroll
            for (uint32_t q = 0; q < WF_QUEUES; ++q) passes += __hip_atomic_load(&Q.ctrl[8u + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (passes != seen_passes) { seen_passes = passes; spins = 0; }
            if (++spins > P.spin_limit_idle) { failed = true; }
            continue;
        }
        // (Napping -- a wave whose best queue is short sleeps while slots are still in flight in other waves -- was measured in round 2: any napping loses,
        // semesterbild at 64 spp 11.6 -> 12.4..13.0 ms: thin passes still hide latency.)
        auto keep = [](uint32_t seen) { return max(1u, seen * 3u / 4u); };          // a pop must still find 3/4 of what the decision saw
        spins = 0;
        uint32_t id = 0;

        if (stage >= WQ_SHADE) {
            // ---- SHADE (one material class) + regeneration; a terminal-class pass is topped up with free slots (which only regenerate) ----
            const uint32_t seen = stage == WQ_SHADE ? cS0 : stage == WQ_SHADE + 1u ? cS1 : stage == WQ_SHADE + 2u ? cS2 : cS3;
            const uint32_t n = Q.pop(stage, 64u, seen == 0u ? 0u : keep(min(seen, 64u)), lane, 0u, id, failed);
            uint32_t nf = 0;
            if (stage == WQ_SHADE && n < 64u && !wc.exhausted()) nf = Q.pop(WQ_FREE, 64u - n, 0u, lane, n, id, failed);
            if (n + nf == 0u || __ballot(failed) != 0ull) continue;              // a failed wait: no pass on a made-up slot number, the loop head leaves
            const bool have = lane < n, fill = lane >= n && lane < n + nf;
            MI355RT_WFCOUNT(3, n + nf);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            // (the defaults stay: with the path state and the record left uninitialised for the lanes that hold no slot the same
            // kernel ran 50 % slower -- 41.5 instead of 27.9 ms on semesterbild)
            PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
            ps.rng.clear();
            Cand c; cand_reset(c);
            if (have) Slot::load_shade(sl, ps.ro, ps.rd, ps.thr, ps.sidx, ps.ray_index, c);     // (the generator state is a function of sidx / ray_index: shade_and_regenerate derives it, REKEY)
            bool live = have;
            const bool any_hit = have && c.idx != CAND_NONE;
            Hit h; h.t = 0.f; h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.mat_ff = 0;
            if (any_hit) finish_hit<HAS_MESH, !HAS_MESH>(P.prims, P.tris, c, ps.ro, ps.rd, h);
            // (the priority stays raised through the material read, the radiance store and the work cursor's atomic: shade_and_regenerate
            // drops it to 0 where the arithmetic starts, DROP_PRIO; the list walk below reads primitives again and runs at PRIO_TOP)
            shade_and_regenerate<MATS, false, true, true, false, /* FASTN */ !HAS_MESH, 0, /* REKEY */ true>(P, wc, lane, live, have || fill, any_hit, h, ps, n_paths, n_rays, prof);
            if (live) Slot::store_shade(sl, ps.ro, ps.rd, ps.thr, ps.sidx, ps.ray_index);    // a ray to trace: continuing or freshly generated
            const int born = (int)__popcll(__ballot(fill && live)), died = (int)__popcll(__ballot(have && !live));
            if (lane == 0 && born != died) atomicAdd(&Q.ctrl[16], (uint32_t)(born - died));
            Q.push(WQ_FREE, (have || fill) && !live, id, lane, failed);
            // Every ray SHADE produces -- continuing or freshly generated -- starts at the head of the list, so the pass goes straight
            // on with TOP for its live lanes: as homogeneous as a pass over a queue of such rays and at least as full, minus one queue
            // round trip per ray (there was a TOP0 queue: semesterbild 9.73 -> 9.08 ms, teapot 6.49 -> 6.17 ms at 64 spp without it).
            prof.mark(4);
            __builtin_amdgcn_s_setprio(WF_PRIO_TOP);
            {   Cand c0; cand_reset(c0);
                WalkRec w0; w0.node = NODE_END; w0.best_t = 0.f; w0.best_tri = 0xFFFFFFFFu;
                run_top((have || fill) && live, ps.ro, ps.rd, c0, 0u, false, ps.ray_index, w0, sl, id); }
            prof.mark(1);
            continue;
        }

        if constexpr (HAS_MESH) {                                        // (a mesh-free list has only the SHADE stages)
        if (stage == WQ_WALK) {
            // ---- WALK: two rounds of eight box tests + the pending leaves; unfinished walks go round again ----
            const uint32_t n = Q.pop(WQ_WALK, 64u, keep(min(cW, 64u)), lane, 0u, id, failed);
            if (n == 0u || __ballot(failed) != 0ull) continue;
            const bool have = lane < n;
            __builtin_amdgcn_s_setprio(WF_PRIO_WALK);
            MI355RT_WFCOUNT(0, n);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            MeshTrav m; m.ro = mk(0, 0, 0); m.rd = mk(0, 0, 1); m.ix = m.iy = m.iz = 0.f; m.len_raw = 0.f; m.node = NODE_END; m.best_t = 0.f;
            m.best_tri = 0xFFFFFFFFu; m.leaf_a = m.leaf_b = 0;
            WalkKeep wkeep; wkeep.cursor_word = 0u;
            if constexpr (MESH_IDENT) {
                // untransformed meshes: the object-space ray is the slot's ray -- no record read, no matrix products -- for passes whose rays all
                // pass the test TOP applied to the same rays; the general form otherwise
                f3 ro_w = mk(0, 0, 0), rd_w = mk(0, 0, 1); uint32_t cur = 0; WalkRec w; w.node = NODE_END; w.best_t = 0.f; w.best_tri = 0xFFFFFFFFu;
                if (have) Slot::load_walk(sl, ro_w, rd_w, cur, w, wkeep);
                if (__ballot(have && !ray_nonzero_finite(ro_w, rd_w)) == 0ull) { if (have) mesh_setup<WF_FAST_MESH>(P.prims, ro_w, rd_w, 0.f, m, true); }
                else if (have) mesh_setup<WF_FAST_MESH>(P.prims + cur, ro_w, rd_w, 0.f, m);
                if (have) { m.node = w.node; m.best_t = w.best_t; m.best_tri = w.best_tri; }
            }
            else if (have) {
                f3 ro_w, rd_w; uint32_t cur; WalkRec w;
                Slot::load_walk(sl, ro_w, rd_w, cur, w, wkeep);
                const DevPrim* __restrict__ pr = P.prims + cur;                                          // lanes may be in different meshes
                mesh_setup<WF_FAST_MESH>(pr, ro_w, rd_w, 0.f, m);                                                      // the object-space ray, as TOP computed it
                m.node = w.node; m.best_t = w.best_t; m.best_tri = w.best_tri;
            }
            // Speculative walk past a leaf.  In the reference's recursion a hit leaf is tested at once, because a triangle hit shrinks
            // t_max for every box that follows (bvh.rs:148-156).  Most leaf tests MISS, and then the walk goes on exactly as if the leaf
            // had not been there.  So a lane that reaches a leaf leaves it pending (leaf_a / leaf_b; `resume` = the node behind it) and
            // keeps stepping with the unchanged best_t instead of idling until the wave's leaf phase (measured: box-test steps ran at
            // 31 / 40 of 64 lanes on semesterbild / teapot); only a SECOND leaf stalls the lane, in front of that leaf's node.  The leaf
            // phase tests the pending leaf; if it HITS, the speculation beyond it is void: the walk resumes at the node behind the leaf
            // with the new best_t / best_tri.  Per ray this is the reference's sequence of accepted tests with the reference's t_max
            // at each of them -- same results, bit for bit; the box tests beyond a hit and one repeated box test per stall are the
            // only extra work.  (Two queued leaves per lane with a leaf phase per queue slot were measured too: the second slot's
            // phases run nearly empty and cost more than the stalls they avoid -- +3 % / +7 %.)
            uint32_t resume = NODE_END; bool stalled = false;
            for (int round = 0; round < WF_ROUNDS; ++round) {
                if (__ballot(have && (m.leaf_b != 0u || m.node != NODE_END)) == 0ull) break;
#pragma unroll
                for (int u = 0; u < WF_STEPS; ++u) {
                    const bool stepping = have && !stalled && m.node != NODE_END;
                    MI355RT_WFCOUNT(4, (uint32_t)__popcll(__ballot(stepping)));
                    if (stepping) mesh_step<FIXED_AABB, 0, true>(n4, nullptr, 0u, EPS, m, &resume, &stalled);
                }
                MI355RT_WFCOUNT(5, (uint32_t)__popcll(__ballot(have && m.leaf_b != 0u)));
                if (have && m.leaf_b != 0u) {
                    const uint32_t before = m.best_tri;
                    mesh_leaf(t4, EPS, m);                                       // leaves m.leaf_b == 0
                    if (m.best_tri != before) m.node = resume;                   // a hit: resume behind the leaf with the new best_t
                }
                stalled = false;
            }
            MI355RT_WFCOUNT(6, (uint32_t)__popcll(__ballot(have && m.leaf_b == 0u && m.node == NODE_END)));     // walks finished per WALK pass
            const bool done = have && m.leaf_b == 0u && m.node == NODE_END;        // (a pass always ends with its pending leaves tested: leaf_b == 0)
            if (have) { WalkRec w; w.node = m.node; w.best_t = m.best_t; w.best_tri = m.best_tri; Slot::store_walk(sl, w, done, wkeep); }
            __builtin_amdgcn_s_setprio(WF_PRIO_SCHED);
            Q.push(WQ_WALK, have && !done, id, lane, failed);
            prof.mark(0);
            // (Letting the finished walks go on with the rest of the list in this pass -- the WALK -> TOP1 counterpart of the fused
            // SHADE -> TOP0 -- was measured at thresholds of 1 / 24 / 40 finished lanes: +-0.5 %, not kept.)
            Q.push(WQ_TOP1, done, id, lane, failed);
            continue;
        }

        {
            // ---- TOP1: hittable.rs:45-58 goes on from the slot's cursor (the mesh whose walk is back) ----
            const uint32_t n = Q.pop(WQ_TOP1, 64u, keep(min(cT1, 64u)), lane, 0u, id, failed);
            if (n == 0u || __ballot(failed) != 0ull) continue;
            const bool have = lane < n;
            __builtin_amdgcn_s_setprio(WF_PRIO_TOP);
            MI355RT_WFCOUNT(1, n);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            f3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
            Cand c; cand_reset(c);
            uint32_t cursor = 0xFFFFFFFFu, ray_index = 0u; bool walk_done = false;
            WalkRec wk; wk.node = NODE_END; wk.best_t = 0.f; wk.best_tri = 0xFFFFFFFFu;
            if (have) Slot::load_top1(sl, ro, rd, c, cursor, walk_done, ray_index, wk);
            run_top(have, ro, rd, c, cursor, walk_done, ray_index, wk, sl, id);
            prof.mark(1);
        }
        }   // HAS_MESH
    }
    const uint32_t wp = wave_sum(n_paths), wr = wave_sum(n_rays);
#ifdef MI355RT_STAMPS
    if (P.wave_times && lane == 0) {                        // per wave: start, end, paths, the time its work cursor ran dry, loop turns after that, paths alive in the workgroup then
        const unsigned long long t_wave1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* w = P.wave_times + WAVE_TIME_WORDS * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
        w[0] = t_wave0; w[1] = t_wave1; w[2] = wp; w[3] = t_dry ? t_dry : t_wave1; w[4] = passes_after_dry; w[5] = alive_at_dry;
    }
#endif
    if (lane == 0 && P.stats) {
        atomicAdd(&P.stats[0], (unsigned long long)wp); atomicAdd(&P.stats[1], (unsigned long long)wr);
#ifdef MI355RT_STAMPS
        for (int i = 0; i < 6; ++i) atomicAdd(&P.stats[2 + i], prof.acc[i]);
        for (int i = 0; i < 8; ++i) { atomicAdd(&P.stats[8 + 2 * i], w_exec[i]); atomicAdd(&P.stats[9 + 2 * i], w_lanes[i]); }
#endif
    }
    // A wave that gave up a bounded wait left paths unfinished: count it in the context's sticky error word, which the host reads
    // back behind every render (rt_api.cpp) -- the call that sees it returns MI355RT_ERR_HIP, whether or not it asked for stats.
    // The word also says WHICH wait, so that one failure in a log explains itself -- worked out here, behind the loop, from what the wave still
    // holds (anything in the loop itself costs the kernel 2 %, measured): a wave whose idle count ran out gave up for lack of progress; a wave
    // with a lane that failed on a ring entry gave up there; the rest only followed the workgroup's error flag out.
    const uint64_t failed_lanes = __ballot(failed);
    if (failed_lanes != 0ull && lane == 0 && P.err) {
        const uint32_t wait = spins > P.spin_limit_idle ? (uint32_t)WAIT_WF_IDLE : failed_lanes != ~0ull ? (uint32_t)WAIT_WF_RING : (uint32_t)WAIT_WF_FOLLOWED;
        atomicAdd(P.err, 1ull);
        atomicOr(P.err, (unsigned long long)wait << 32);
    }
}
#define MI355RT_OCC_WFK __attribute__((amdgpu_waves_per_eu(6, 6)))         // 80 VGPRs: 2 workgroups of 12 waves per CU (see the top of this file)
// Entry points: one body per material set (rt_device.h) -- and, for the set the mesh scenes use, per transform class of the meshes; the opt-in slab test only in the general form.
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf(const RenderParams P) { render_ctr_wavefront<false, MATS_ALL>(P); }
// WALK geometry by tree size (round 5; template arguments INLINE_STEPS, WF_ROUNDS, WF_STEPS): large trees want LONGER rounds -- teapot (8 191 / 14 161 nodes, depth 12 / 13), kernel ms at 800x600x256
// (profiles/r05/ab_wavefront_walk_geometry_deep.txt): on its own instantiation 3 x 8 15.49, 3 x 9 15.31*, **3 x 10 15.33 (-1.0 %)**, 3 x 11 15.36, 3 x 12 15.59, 4 x 10 15.55, 2 x 12 15.33*; forced onto this general
// form: 8 inline steps + 3 x 8 16.01, 8 + 3 x 10 15.81, **12 + 3 x 10 15.66 (-2.2 %)** (* = another session, base 15.40) -- small trees want SHORTER ones (k_render_ctr_wf_nometal_shallow below).
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf_nometal(const RenderParams P) { render_ctr_wavefront<false, MATS_NO_METAL, true, false, 12, 3, 10>(P); }
// ... and for lists whose meshes all have SMALL trees (every mesh <= WF_SHALLOW_NODES nodes: semesterbild's text mesh, 3 351 nodes, depth 11): WALK passes of 3 x 6 box tests instead of
// 3 x 8 -- a shallow walk ends or stalls sooner, so the last steps of an 8-step round run nearly empty.  Round 5, 800x600x256 kernel ms (profiles/r05/ab_wavefront_walk_geometry_shallow.txt):
// 3x8 26.68, 4x6 26.46, 3x6 26.36 (-1.2 %), 5x5 26.61, 4x5 26.63; teapot's deep trees (8 191 / 14 161 nodes) want 3 x 8 (every other geometry +1 ... +2.8 %, round 4).
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf_nometal_shallow(const RenderParams P) { render_ctr_wavefront<false, MATS_NO_METAL, true, false, 8, 3, 6>(P); }
// ... and for lists whose meshes are all untransformed (teapot -1.9 % at 256 spp, another -0.7 % with 12 inline steps; profiles/r04/ab_wavefront_transform_classes.txt; round 5: rounds of 10 box tests, -1.0 %)
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf_nometal_ident(const RenderParams P) { render_ctr_wavefront<false, MATS_NO_METAL, true, true, 12, 3, 10>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_WF_MESHFREE) __attribute__((amdgpu_waves_per_eu(8, 8))) k_render_ctr_wf_meshfree(const RenderParams P) { render_ctr_wavefront<false, MATS_NO_SPECULAR, false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf_fixaabb(const RenderParams P) { render_ctr_wavefront<true, MATS_ALL>(P); }


}  // namespace mi355rt
