// rt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) replacing the rayon per-pixel loop of
// jackra1n/raytracer-rust (render_scene -> trace_ray, src/renderer.rs:19-123).
//
// Kernels
//   k_render_ctr   persistent wave64 path tracer.  One lane = one path (one sample of one pixel).  A wave
//                  claims BATCH_SAMPLES consecutive sample indices with ONE global atomic and deals them
//                  to its lanes with ballot/mbcnt whenever lanes run dry (path regeneration), so lanes
//                  whose paths end early (miss, emitter, absorption) are refilled on the next iteration
//                  instead of idling until the longest path of the wave finishes.
//                  The top-level primitive list is walked with a wave-uniform index through the constant
//                  address space (scalar loads, records live in SGPRs); the per-mesh BVH is a threaded
//                  pre-order array walked per lane with dwordx4 loads -- the reference always descends
//                  left-then-right (bvh.rs:142-156), so escape links reproduce its visit order and its
//                  shrinking t_max exactly and no traversal stack is needed.
//                  Randomness: a counter-based generator (pcg4d since round 5; Philox4x32-10 / -7 selectable at build time,
//                  rt_rng.h) addressed by (image row key; x, sample, ray index, block): every draw is a pure function of
//                  the path, so any schedule / tiling / GPU count produces bit-identical radiance.
//                  Round 5: a wave keeps the camera rays of its next 64 samples IN STOCK (RayStock, LDS) and the mesh-free kernels
//                  are instantiated per material set AND per primitive-kind set (k_render_ctr_simple_qc).
//                  Output: three floats of radiance per path into the HBM workspace.
//   k_resolve      per pixel, sums its spp radiance values IN SAMPLE ORDER (renderer.rs:100), scales by
//                  1/spp (:103), sqrt-gamma, clamp, pack 0x00RRGGBB (:112-120, color.rs:87-93).
//   k_render_ref   validation mode: one lane per image row replays the reference's sequential
//                  StdRng::seed_from_u64(y) stream (renderer.rs:91) and folds radiance tail-first.
//
// Files (one translation unit; this file includes the rest): rt_math.h (vec3.rs helpers), rt_rng.h (pcg4d / Philox, ChaCha12 replay),
// rt_intersect.h (hit tests, BVH walk, finish_hit), rt_materials.h (scatter, camera, miss colour), here: the work cursor,
// shade_and_regenerate() and the lockstep kernels, then rt_mesh_variants.h (the state machine of the reference build), rt_wavefront.h
// (k_render_ctr_wf), and at the end k_resolve, k_render_ref, the debug kernels and the launchers.
//
// Numerics: compiled with -ffp-contract=off and correctly rounded f32 divide/sqrt; every expression
// keeps the reference's operation order (file:line cited per function), so results agree with the
// CPU oracle bit-for-bit wherever only + - * / sqrt are involved.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rt_device.h"
#include "../../../include/mi355rt.h"

#include "rt_math.h"
#include "rt_rng.h"
#include "rt_intersect.h"
#include "rt_materials.h"

namespace mi355rt {

DI uint32_t mbcnt64(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
DI uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ===================================================================================================
// Shared pieces of the two counter-mode kernels
// ===================================================================================================
// n / d for n < 2^31 with a host-computed magic pair (mul == 0 means d == 1): q = umulhi(n, mul) >> shift.
DI uint32_t fastdiv(uint32_t n, uint32_t mul, uint32_t shift) { return mul ? (__umulhi(n, mul) >> shift) : n; }

// Wave-uniform cursor over the band's sample indices.  A wave claims a run of consecutive indices with one
// global atomic and deals them to idle lanes with ballot + mbcnt.
//  * The band is cut into WORK_SHARDS contiguous ranges, each with its own counter on its own cache line.
//    A wave starts on the shard of its XCD (HW_REG_XCC_ID; used for speed only) and moves to the next shard
//    when one runs dry (work stealing), so one hot word never serialises all 6 k waves of the chip
//    (measured: a single counter saturates near 88 atomics/us and made short runs 30 % slower).
//  * Run length follows guided self-scheduling: (what was left in the shard at the wave's previous claim)
//    / guided_div, clamped to [run_min, run_max] -- long runs while there is plenty of work (few atomics,
//    coherent primary rays), short runs at the end so that all waves drain together.  This matters when one
//    image is split across 8 GPUs and a launch lasts only a few ms.
DI uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & (WORK_SHARDS - 1u); }
// RMIN .. RMAX: the run lengths, compile-time (as kernel arguments they cost the Lambert-only kernel two more live scalar registers
// in its hot loop and 5.7 % of its time).  RMIN == RMAX: fixed runs, no guided division.
template <uint32_t RMIN, uint32_t RMAX>
struct WorkCursorT {
    uint32_t next = 0, end = 0; bool no_more = false;
    uint32_t shard = 0, dry = 0, seen = 0;      // current shard, consecutive dry shards, last counter value seen in it
    DI void init() { shard = xcc_id(); }
    DI bool exhausted() const { return no_more && next == end; }
    // One global atomic: the next run of the wave's shard (or of the next shard that still has work).  Wave-uniform.
    DI void claim(const RenderParams& P, uint32_t lane) {
        while (next == end && !no_more) {
            const uint32_t base = shard * P.shard_samples;
            const uint32_t len = min(P.shard_samples, P.band_samples > base ? P.band_samples - base : 0u);
            const uint32_t rem = len > seen ? len - seen : 0u;
            const uint32_t size = RMIN == RMAX ? RMAX : min(max(rem / P.guided_div, RMIN), RMAX);
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(P.batch_counter + shard * WORK_SHARD_STRIDE, size);
            start = __builtin_amdgcn_readfirstlane(start);
            if (start >= len) {                              // shard is dry: steal from the next one
                shard = (shard + 1u) & (WORK_SHARDS - 1u); seen = 0;
                if (++dry == WORK_SHARDS) no_more = true;
            } else { next = base + start; end = base + min(start + size, len); seen = start + size; dry = 0; }
        }
    }
    // Lanes with want == true get a sample index (returns true and sets sidx); others / surplus stay idle.
    DI bool deal(const RenderParams& P, bool want, uint32_t lane, uint32_t& sidx) {
        const uint64_t idle = __ballot(want);
        if (idle == 0ull) return false;
        claim(P, lane);
        const uint32_t take = min((uint32_t)__popcll(idle), end - next);
        bool got = false;
        if (take != 0u) {
            const uint32_t rank = mbcnt64(idle);
            if (want && rank < take) { sidx = next + rank; got = true; }
            next += take;
        }
        return got;
    }
    // The same, for a kernel that keeps camera rays IN STOCK (RayStock below): samples are dealt out of [next, stock_end), the part of the
    // wave's run whose rays `refill(first, count)` has prepared -- up to 64 at a time, by the whole wave, whenever the stock runs out
    // (also in the middle of a deal: the lanes the old stock could not serve are served from the new one in the same call).
    uint32_t stock_end = 0;                                  // next == stock_end: the stock is empty
    template <class Refill, class Take>
    DI bool deal_stocked(const RenderParams& P, bool want, uint32_t lane, uint32_t& sidx, Refill&& refill, Take&& take_entry) {
        uint64_t idle = __ballot(want);
        if (idle == 0ull) return false;
        uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle <= stock_end - next) {                    // the common case: the stock serves every idle lane
            if (want) { sidx = next + mbcnt64(idle); take_entry(sidx); }
            next += n_idle;
            return want;
        }
        bool got = false;                                    // the stock runs out in this deal (once per 64 samples)
        for (;;) {
            if (next == stock_end) {
                claim(P, lane);
                if (next == end) break;                      // no work left anywhere
                stock_end = next + min(64u, end - next);
                refill(next, stock_end - next);
            }
            const uint32_t take = min(n_idle, stock_end - next);
            const uint32_t rank = mbcnt64(idle);
            if (want && !got && rank < take) { sidx = next + rank; got = true; take_entry(sidx); }   // (read before a refill reuses the entry)
            next += take;
            idle = __ballot(want && !got);
            if (idle == 0ull) break;
            n_idle = (uint32_t)__popcll(idle);
        }
        return got;
    }
};
typedef WorkCursorT<BATCH_MIN, BATCH_MAX> WorkCursor;                    // lockstep kernels (and the reference build's mesh kernels)
typedef WorkCursorT<RUN_WAVEFRONT_MIN, RUN_WAVEFRONT> WorkCursorWf;          // wavefront kernel: fixed, aligned runs (rt_device.h)

// Decode a band-local sample index into (x, y, s) and key the path's RNG (renderer.rs:91-97).
DI void start_path(const RenderParams& P, uint32_t sidx, RngCtr& rng, uint32_t& px, uint32_t& py) {
    const uint32_t pix_local = fastdiv(sidx, P.spp_mul, P.spp_shift);
    const uint32_t s = sidx - pix_local * P.spp;
    const uint32_t pix = P.band_pixel0 + pix_local;
    const uint32_t jrow = fastdiv(pix, P.width_mul, P.width_shift);
    px = pix - jrow * P.width;
    py = P.rows[jrow];
    const uint64_t ykey = (uint64_t)py + (((uint64_t)P.seed_hi << 32) | (uint64_t)P.seed_lo);
    rng.start((uint32_t)ykey, (uint32_t)(ykey >> 32), px, s + P.sample0);
}

// Per-lane path state shared by both kernels.
struct PathState {
    f3 ro, rd, thr;
    uint32_t sidx, ray_index;
    uint32_t px, py;               // only meaningful while `fresh`
    RngCtr rng;
};

// Camera rays IN STOCK (lockstep kernels, round 5).  A path that ends hands its lane to a fresh sample in the same iteration, so every iteration
// started a few paths -- 11 of 64 lanes on cornell, in 93 % of the iterations (profiles/r05/stamps_final_kernels.txt) -- and paid the whole
// price of starting one for them: the sample index decoded, its row looked up (a global load the arithmetic half then waited for), the path's
// generator base, the jitter block, u / v, Camera::get_ray: ~100 instructions at a sixth of the lanes.  Every one of those values is a pure function
// of the sample index.  So the wave prepares the rays of the next (up to) 64 samples of its run AT ONCE, with all lanes, whenever its stock runs out
// (WorkCursorT::deal_stocked), and keeps them in 2 KB of LDS: per sample the direction before its two normalisations (camera.rs:33-42; the shared
// tail of the iteration normalises it with the scattered rays, as before) and the generator base.  A lane that is dealt sample i reads entry i mod 64:
// two ds_read_b128.  Same arithmetic per path, same bits.  LDS operations of one wave execute in order: no barrier between refill and take.
struct RayStock {
    float4* dir; uint4* key;                                   // [64] each, this wave's
    template <bool FASTN>
    DI void refill(const RenderParams& P, uint32_t first, uint32_t count, uint32_t lane) const {
        if (lane < count) {
            const uint32_t sidx = first + lane;
            RngCtr rng; uint32_t px, py;
            start_path(P, sidx, rng, px, py);
            rng.template load_block0<true>();                                                   // ray 0, block 0: the jitter (words 0 / 1)
            const float un = (float)px + rng.jitter_u(), vn = (float)py + rng.jitter_v();
            const float u = FASTN ? div_by_rn(un, P.width_f, P.inv_width_rn) : un / (float)P.width;       // renderer.rs:96
            const float v = FASTN ? div_by_rn(vn, P.height_f, P.inv_height_rn) : vn / (float)P.height;    // renderer.rs:97
            const f3 raw = camera_raw(P.cam, u, v);                                              // renderer.rs:99
            dir[sidx & 63u] = make_float4(raw.x, raw.y, raw.z, 0.f);
            if constexpr (RngCtr::NW == 4) key[sidx & 63u] = make_uint4(rng.w[0], rng.w[1], rng.w[2], rng.w[3]);
        }
    }
    DI void take(const RenderParams& P, uint32_t sidx, f3& raw, PathState& ps) const {
        const float4 d = dir[sidx & 63u];
        raw = mk(d.x, d.y, d.z);
        // (handing the key over only after the iteration's shared generator call -- a fresh path does not use that call's block -- so that the wave need not
        //  wait for this read in front of it was measured: four more live registers, cornell +0.5 %, the general kernels +0.5 ... 2 %)
        if constexpr (RngCtr::NW == 4) { const uint4 k = key[sidx & 63u]; ps.rng.w[0] = k.x; ps.rng.w[1] = k.y; ps.rng.w[2] = k.z; ps.rng.w[3] = k.w; }
        else start_path(P, sidx, ps.rng, ps.px, ps.py);                                          // (the Philox builds address by (key, x, s): nothing to keep)
    }
};
struct NoStock {};

// random_in_unit_sphere (vec3.rs:54-61) for the whole wave at once, counter mode.  Try 0 comes from the event's
// block 0 (already in rng.b0).  Lanes whose try 0 failed become OWNERS of a retry request; then every lane of
// the wave -- busy or not -- is a WORKER: with n owners, G = 2^floor(log2(64 / n)) workers serve each owner and
// evaluate its tries jbase .. jbase+G-1 in parallel (the owner's counters arrive through ds_bpermute; a draw is a
// pure function of (key, x, s, ray, try), so any lane can compute it).  The owner takes the FIRST accepted try of
// its segment of the ballot, i.e. exactly the try the sequential loop would have stopped at.  Typically two
// rounds instead of E[max over 64 lanes of a geometric(0.52)] ~ 7.7 iterations of a mostly idle wave.
// Must be called in wave-uniform control flow.
DI int lane_shfl(int v, uint32_t src_lane) { return __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), v); }
DI float lane_shfl(float v, uint32_t src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v))); }
template <bool WIDE = false, int TRY1 = 0>
DI f3 unit_ball_cooperative(bool diffuse, const RngCtr& rng, uint32_t lane, const uint32_t* b1 = nullptr) {
    f3 p = mk(u32_to_range11(rng.b0[1]), u32_to_range11(rng.b0[2]), u32_to_range11(rng.b0[3]));   // try 0
    bool need = diffuse && !(len2(p) < 1.0f);
    uint32_t jbase = 1;
    if constexpr (TRY1 >= 1) {                                                     // try 1 was drawn in the lane itself, next to the event's block
        const f3 p1 = mk(u32_to_range11(b1[1]), u32_to_range11(b1[2]), u32_to_range11(b1[3]));
        if (need && len2(p1) < 1.0f) { p = p1; need = false; }
        jbase = 2;                                                                 // (tries 1 AND 2 in the lane: the same kernel spills, 14.8 -> 17.9 ms)
    }
    if constexpr (TRY1 >= 2) {
        if (__ballot(need) != 0ull) {
            uint32_t b2[4]; RngCtr::block<WIDE>(rng.w, 2u, b2);
            const f3 p2 = mk(u32_to_range11(b2[1]), u32_to_range11(b2[2]), u32_to_range11(b2[3]));
            if (need && len2(p2) < 1.0f) { p = p2; need = false; }
        }
        jbase = 3;
    }
    for (;;) {
        const uint64_t m = __ballot(need);
        if (m == 0ull) break;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t lg = 6u - (n <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(n - 1u));   // G = 2^lg = largest power of two <= 64 / n workers per owner
        const uint32_t r = mbcnt64(m);                                             // owners below this lane
        // table: lane t holds the lane id of owner number t (a full permutation keeps every lane enabled)
        const int tab = __builtin_amdgcn_ds_permute((int)((need ? r : n + (lane - r)) << 2), (int)lane);
        const uint32_t orank = lane >> lg;
        const bool worker = orank < n;
        const uint32_t olane = (uint32_t)lane_shfl(tab, worker ? orank : 0u);
        uint32_t ow[RngCtr::NW];                                                   // the owner's draw address (rt_rng.h)
#pragma unroll
        for (int i = 0; i < RngCtr::NW; ++i) ow[i] = (uint32_t)lane_shfl((int)rng.w[i], olane);
        const uint32_t oj = (uint32_t)lane_shfl((int)jbase, olane);
        uint32_t w[4];
        RngCtr::block<WIDE>(ow, oj + (lane & ((1u << lg) - 1u)), w);
        const f3 q = mk(u32_to_range11(w[1]), u32_to_range11(w[2]), u32_to_range11(w[3]));
        const uint64_t acc = __ballot(worker && (len2(q) < 1.0f));
        const uint32_t seg_lo = r << lg;                                           // my segment of the ballot (owners only)
        const uint64_t segmask = (lg == 6u) ? ~0ull : ((1ull << (1u << lg)) - 1ull);
        const uint64_t seg = need ? ((acc >> seg_lo) & segmask) : 0ull;
        const bool found = seg != 0ull;
        const uint32_t src = found ? seg_lo + (uint32_t)__builtin_ctzll(seg) : lane;
        const float qx = lane_shfl(q.x, src), qy = lane_shfl(q.y, src), qz = lane_shfl(q.z, src);
        if (need) { if (found) { p = mk(qx, qy, qz); need = false; } else jbase += (1u << lg); }
    }
    return p;
}

// The shading half of one trace_ray level (renderer.rs:26-36) plus path regeneration, for every lane of the
// wave at once.  On entry `live` lanes carry a finished intersection (`hit`, `h`); on exit `live` lanes carry
// the next ray to trace.  Order: finish paths that end without scattering (miss / emitter / null) ->
// deal fresh samples to idle lanes -> ONE generator call for all lanes -> camera ray (fresh) or BSDF (continuing).
// Must be called by the whole wave in uniform control flow (it ballots): lanes that are busy elsewhere
// pass live = false and can_take = false and are left untouched.
// Returns false when no lane is live afterwards and no work is left to deal.
// DEFAULTS: give the per-lane temporaries default values.  The lockstep kernels run without (every value is read only on the
// path that wrote it, and the defaults cost ~30 v_mov per iteration: cornell -1.5 %), and since round 3 so does the wavefront
// kernel (rt_wavefront.h); the reference build's state machine keeps them (its other lanes'
// state must not be touched).  WIDE: the generator's multiplies as 64-bit products (rt_rng.h).  DROP_PRIO: lower the wave's priority to 0 once the
// fresh samples are dealt (the caller raised it for the memory-bound half of the iteration).  Q0_IN_HIT: see struct Hit.
// FASTN: see normalized() (rt_math.h).  TRY1: try 1 of the unit-ball draw comes from a second generator block drawn in the lane itself, right after the
// event's block, so the cooperative rounds start at try 2 and a second round is needed in 44 % of the iterations instead of all: cornell -0.9 %
// on the Lambert-only kernel; every other kernel pays for the three more live registers with spills (+1.5 ... +23 %: profiles/r03_ab_inlane_try1.txt).
// REKEY (wavefront kernel: a path's generator state is not carried in its slot): the state is derived from the sample index HERE, once, for continuing
// and freshly dealt lanes together, right in front of the event's block -- instead of by the caller when the slot is loaded (live through the hit record,
// the material read and the radiance store) and a second time where fresh samples are dealt.
template <uint32_t MATS, bool DEFAULTS = true, bool WIDE = !DEFAULTS, bool DROP_PRIO = false, bool Q0_IN_HIT = false, bool FASTN = false, int TRY1 = 0, bool REKEY = false, class WC, class STOCK = NoStock>
DI bool shade_and_regenerate(const RenderParams& P, WC& wc, uint32_t lane, bool& live, bool can_take, bool hit, const Hit& h,
                             PathState& ps, uint32_t& n_paths, uint32_t& n_rays, Prof& prof, const STOCK& stock = STOCK()) {
    constexpr bool STOCKED = std::is_same<STOCK, RayStock>::value;                           // camera rays come out of the wave's stock (lockstep kernels)
    static_assert(!STOCKED || (!DEFAULTS && !REKEY), "the stock serves the lockstep form");
    struct Rad { float x, y, z; };                                                        // 12 bytes per path: global_store_dwordx3
    Rad* __restrict__ radiance = reinterpret_cast<Rad*>(P.radiance);
    float4 q0;                                                                            // first 16 bytes of the hit material; read by lanes that loaded it
    if (DEFAULTS) q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        f3 term = mk(0.f, 0.f, 0.f); bool fin = false;
        if (!hit) { term = miss_colour(P.sky, P.sky_w, P.sky_h, P.miss, ps.rd); fin = true; }   // renderer.rs:38-63
        else {
            if constexpr (Q0_IN_HIT) q0 = h.q0;                                              // (finish_hit read it with the hit record)
            else q0 = reinterpret_cast<const float4*>(P.mats + (h.mat_ff & 0x7FFFFFFFu))[0];
            const uint32_t kind = __float_as_uint(q0.x);
            if (kind == MI355RT_MAT_EMISSIVE) { term = mk(q0.y, q0.z, q0.w); fin = true; }   // scatter -> None, emitted = colour
            else if (kind == MI355RT_MAT_NULL) fin = true;
        }
        if (fin) { const f3 L = ps.thr * term; radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false; }
    }
    prof.mark(2);
    bool fresh = false;
    f3 stocked_raw = mk(0.f, 0.f, 1.f);
    if constexpr (STOCKED) {
        if (wc.deal_stocked(P, can_take && !live, lane, ps.sidx, [&](uint32_t first, uint32_t count) { stock.template refill<FASTN>(P, first, count, lane); },
                            [&](uint32_t sidx) { stock.take(P, sidx, stocked_raw, ps); })) { fresh = true; live = true; ++n_paths; }
    } else
    if (wc.deal(P, can_take && !live, lane, ps.sidx)) { if constexpr (!REKEY) start_path(P, ps.sidx, ps.rng, ps.px, ps.py); fresh = true; live = true; ++n_paths; }
    if constexpr (DROP_PRIO) __builtin_amdgcn_s_setprio(0);                               // the lockstep kernels' arithmetic half (see render_ctr_lockstep)
    if (__ballot(live) == 0ull) return !wc.exhausted();
    uint32_t ball_use = BALL_NONE; bool scattered = false;
    f3 raw, atten, emitted; float side, fuzz = 0.f;  // written by the branch a lane takes below, read only on that lane's own path
    if constexpr (DEFAULTS) {
        raw = mk(0.f, 0.f, 1.f); atten = mk(0.f, 0.f, 0.f); emitted = mk(0.f, 0.f, 0.f); side = EPS;
        if (live) {
            if (!fresh) ps.rng.next_event();
            ps.rng.template load_block0<WIDE>();
            if (fresh) {
                const float u = ((float)ps.px + ps.rng.jitter_u()) / (float)P.width;         // renderer.rs:96
                const float v = ((float)ps.py + ps.rng.jitter_v()) / (float)P.height;        // renderer.rs:97
                raw = camera_raw(P.cam, u, v);                                               // renderer.rs:99; normalised below with the scattered rays
                ps.ro = mk(P.cam.position[0], P.cam.position[1], P.cam.position[2]);
                ps.thr = mk(1.f, 1.f, 1.f); ps.ray_index = 0;
                if (P.max_depth == 0u) { radiance[ps.sidx] = Rad{0.f, 0.f, 0.f}; live = false; }   // depth == 0 -> BLACK
            } else {
                scattered = scatter_pre<MATS, WIDE>(P.mats, P.textures, q0, h, ps.rd, ps.rng, side, raw, atten, emitted, ball_use, fuzz);
            }
        }
        prof.mark(5);
        const f3 ball = unit_ball_cooperative<WIDE>(scattered && ball_use != BALL_NONE, ps.rng, lane);   // whole wave, uniform control flow
        if (live && !fresh) {
            if (scattered) scattered = ball_finish(ball_use, h, ball, fuzz, raw);            // (a fuzzed metal reflection may still be absorbed)
            if (scattered) {
                ps.thr = ps.thr * atten; ps.ro = scatter_origin(h, side); ++ps.ray_index;
                if (ps.ray_index == P.max_depth) {                                           // next level has depth == 0 (renderer.rs:20-22)
                    const f3 L = ps.thr * mk(0.f, 0.f, 0.f);
                    radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false;
                }
            } else {                                                                         // absorbed: scatter -> None (renderer.rs:35)
                const f3 L = ps.thr * emitted;
                radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false;
            }
        }
        if (live) { ps.rd = ray_direction(raw); ++n_rays; }                                  // fresh and scattered lanes together
    } else {
        // Lockstep kernels: every lane of the wave passes through here in every iteration, so a lane that is not live afterwards
        // is idle until it is dealt a fresh path (which sets all of its state) or for good -- its path state may hold anything.
        // The branches therefore only PRODUCE the next state (fresh values, nothing carried through them) and the state is
        // overwritten for all lanes at the end: no conditional updates of loop-carried registers, no copies to merge them.
        f3 n_ro, n_thr; uint32_t n_ri;
        if constexpr (REKEY) { if (live) { start_path(P, ps.sidx, ps.rng, ps.px, ps.py); ps.rng.set_ray(fresh ? 0u : ps.ray_index + 1u); } }
        else if (!fresh) ps.rng.next_event();
        ps.rng.template load_block0<WIDE>();
        uint32_t blk1[4] = {0u, 0u, 0u, 0u};
        if constexpr (TRY1 >= 1) RngCtr::block<WIDE>(ps.rng.w, 1u, blk1);
        if (live) {
            if (fresh) {
                // (FASTN: div_bounded with the host's RN(1/width) -- the dividend is 0 or in [2^-24, 2^24), the divisor an image dimension in [1, 2^24);
                //  with the reciprocal computed in the kernel the compiler hoisted it into four vector registers that the loop then spilled)
                if constexpr (STOCKED) raw = stocked_raw;                                    // RayStock::refill computed it, with all lanes at once
                else {
                const float un = (float)ps.px + ps.rng.jitter_u(), vn = (float)ps.py + ps.rng.jitter_v();
                const float u = FASTN ? div_by_rn(un, P.width_f, P.inv_width_rn) : un / (float)P.width;       // renderer.rs:96
                const float v = FASTN ? div_by_rn(vn, P.height_f, P.inv_height_rn) : vn / (float)P.height;    // renderer.rs:97
                raw = camera_raw(P.cam, u, v);                                               // renderer.rs:99; normalised below with the scattered rays
                }
                n_ro = mk(P.cam.position[0], P.cam.position[1], P.cam.position[2]);
                n_thr = mk(1.f, 1.f, 1.f); n_ri = 0;
                if (P.max_depth == 0u) { radiance[ps.sidx] = Rad{0.f, 0.f, 0.f}; live = false; }   // depth == 0 -> BLACK
            } else {
                scattered = scatter_pre<MATS, WIDE, FASTN, /* TERMINAL_DONE */ true>(P.mats, P.textures, q0, h, ps.rd, ps.rng, side, raw, atten, emitted, ball_use, fuzz);
            }
        }
        prof.mark(5);
        prof.classes(live && !fresh, live && fresh, __float_as_uint(q0.x));
        const f3 ball = unit_ball_cooperative<WIDE, TRY1>(scattered && ball_use != BALL_NONE, ps.rng, lane, blk1);   // whole wave, uniform control flow
        if (live && !fresh) {
            if (scattered) scattered = ball_finish<FASTN>(ball_use, h, ball, fuzz, raw);     // (a fuzzed metal reflection may still be absorbed)
            if (scattered) {
                n_thr = ps.thr * atten; n_ro = scatter_origin(h, side); n_ri = ps.ray_index + 1u;
                if (n_ri == P.max_depth) {                                                   // next level has depth == 0 (renderer.rs:20-22)
                    const f3 L = n_thr * mk(0.f, 0.f, 0.f);
                    radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false;
                }
            } else {                                                                         // absorbed: scatter -> None (renderer.rs:35)
                const f3 L = ps.thr * emitted;
                radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false;
            }
        }
        ps.ro = n_ro; ps.thr = n_thr; ps.ray_index = n_ri;
        ps.rd = ray_direction<FASTN>(raw);                                                   // fresh and scattered lanes together
        if (live) ++n_rays;
    }
    prof.mark(3);
    return true;
}

// Register budget per kernel, as waves per SIMD (A/B: tools/ab.py).  The lockstep kernel is VALU-issue bound
// and gains from 7 waves/SIMD (72 VGPRs) even with a few spills; the state-machine kernel keeps its hot BVH state in
// registers and loses when capped.
#define MI355RT_TRAV_BIAS 2                                 // (reference build's state machine)
#define MI355RT_OCC_LOCKSTEP 6                               // general mesh-free kernel, veach-mis 64 spp.  Round 1: 4 -> 6.46 ms, 5 -> 6.03, 6 -> 5.79,
                                                            // 7 -> 5.73.  After the instruction diet: 7 -> 4.94, 6 -> 4.88, and with the cube hit point
                                                            // carried (3 more registers): 7 -> 5.03 (spills), 6 -> 4.81, 5 -> 5.04
#define MI355RT_OCC_LS __attribute__((amdgpu_waves_per_eu(MI355RT_OCC_LOCKSTEP, MI355RT_OCC_LOCKSTEP)))
#define MI355RT_OCC_SIMPLE __attribute__((amdgpu_waves_per_eu(7, 7)))
#define MI355RT_OCC_SMK __attribute__((amdgpu_waves_per_eu(4, 4)))      // (reference build's state machine)

// ===================================================================================================
// k_render_ctr<HAS_MESH> -- persistent, path-regenerating wave64 path tracer, lockstep form: every live lane
// traces one full ray per loop iteration.  Used for scenes whose top level has no mesh (cornell, veach-mis):
// all lanes walk the same primitive list, so the iteration is divergence-free up to the hit tests.
// ===================================================================================================
// The lockstep kernels' forms (each measured against its alternative; docs/kernels/lockstep_round3.md): the candidate carries the cube's object-space hit
// point (also in the general kernel: pays at 80 VGPRs); the short reciprocal / square root / division of rt_math.h; in the Lambert-only kernel ONE try of the
// unit-ball draw in the lane itself before the cooperative rounds (0 tries: +1.9 %, 2 tries: +-0 with pcg4d, +23 % with Philox: profiles/r05/ab_counter_generator.txt).
template <bool HAS_MESH, uint32_t MATS, uint32_t KINDS = PRIMS_ALL>
DI void render_ctr_lockstep(const RenderParams& P) {
    constexpr bool SIMPLE = (MATS & ~MATS_LAMBERT) == 0u;
    // The mesh-free kernels are compiled for lists that hold something and for paths that may take a step: the host sends an empty list or max_depth == 0 to
    // k_render_ctr_mesh, the plain loop (rt_api.cpp render_samples).  Both are launch constants, and the compiler had turned `n_prims != 0` and `max_depth == 0`
    // into lane masks tested in every iteration (the masks spilled: four v_readlane, two mask operations, two branches): cornell -1.9 %.
    // (k_render_ctr_nomesh -1.2 %; k_render_ctr_nospec pays for the different allocation with 6 more scratch instructions, +0.4 %: it keeps the tests.)
    if constexpr (!HAS_MESH && MATS != MATS_NO_SPECULAR) { __builtin_assume(P.n_prims != 0u); __builtin_assume(P.max_depth != 0u); }
    cprim_t prims = (cprim_t)(P.prims);
    const uint32_t lane = threadIdx.x & 63u;
    WorkCursor wc; wc.init();
    __shared__ __attribute__((aligned(16))) float4 s_stock_dir[BLOCK_THREADS];                // RayStock: 64 entries per wave
    __shared__ __attribute__((aligned(16))) uint4 s_stock_key[BLOCK_THREADS];
    RayStock stock; stock.dir = s_stock_dir + (threadIdx.x & ~63u); stock.key = s_stock_key + (threadIdx.x & ~63u);
    PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
    ps.rng.clear();
    bool live = false;
    uint32_t n_paths = 0, n_rays = 0;
    Prof prof; prof.begin();
#ifdef MI355RT_STAMPS
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_dry = 0ull; uint32_t drain_iters = 0, live_at_dry = 0;
#endif
    for (;;) {
        Hit h; bool hit = false;                           // h is read only where `hit` says it was written: no default values to copy around
        // Wave priority by phase (round 3; profiles/r03_ab_lockstep_priority.txt): the half of an iteration that waits on memory -- the
        // list walk's primitive reads, the material read, the radiance store and the work cursor's atomic -- runs at priority 1, the
        // arithmetic half (generator, BSDF, the cooperative unit-ball draw, the next ray) at 0, so a SIMD's issue slots go first to the
        // wave whose loads can then be in flight under the others' arithmetic.  cornell 16.49 -> 15.82 ms, veach-mis on these kernels -1.7 %;
        // priority 2 or 3 measure the same; keeping it through the generator call (15.96) or only over the walk (16.10-16.17) gains less.
        // (Round 5, after the scalar diet: all four schemes -- 1/0, none, inverted, 3/0 -- are within 0.2 % of each other; the list walk no longer waits.
        //  profiles/r05/ab_scalar_diet.txt r05_q13.)
        __builtin_amdgcn_s_setprio(1);
        if (live) hit = hit_scene<HAS_MESH, true, KINDS>(prims, P.n_prims, P.nodes, P.tris, ps.ro, ps.rd, h);     // renderer.rs:24
        prof.mark(1);
        if (!shade_and_regenerate<MATS, false, true, true, !HAS_MESH, /* FASTN */ true, /* TRY1 */ SIMPLE ? 1 : 0>(P, wc, lane, live, true, hit, h, ps, n_paths, n_rays, prof, stock)) break;
        prof.mark(4);
#ifdef MI355RT_STAMPS
        if (wc.exhausted()) {                              // all work dealt: from here on the wave only drains its own paths
            if (t_dry == 0ull) { t_dry = __builtin_amdgcn_s_memrealtime(); live_at_dry = (uint32_t)__popcll(__ballot(live)); }
            ++drain_iters;
        }
#endif
    }
#ifdef MI355RT_STAMPS
    if (lane == 0 && P.stats) { for (int i = 0; i < 6; ++i) atomicAdd(&P.stats[2 + i], prof.acc[i]); for (int i = 0; i < 10; ++i) atomicAdd(&P.stats[24 + i], prof.cls[i]); }
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

// Entry points: one body, instantiated per scene class so that each gets its own register budget.
//   k_render_ctr_nomesh     any materials, no mesh in the list                      (any other mesh-free list)                        6 waves/SIMD
//   k_render_ctr_simple     Lambertian/Emissive/Null only, no mesh                  (-3 % vs nomesh on cornell)                       7 waves/SIMD
//   k_render_ctr_simple_qc  the same for lists of quads and cubes only              (cornell: another -1.1 %)                         7 waves/SIMD
//   k_render_ctr_nospec     no metal, no dielectric, no mesh                        (-2.5 % vs nomesh on veach-mis; the probe kernel) 7 waves/SIMD
//   k_render_ctr_mesh       lockstep with the per-lane BVH walk inlined             (diagnostic knob; the two degenerate renders: an empty list, max_depth 0)
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_LS k_render_ctr_nomesh(const RenderParams P) { render_ctr_lockstep<false, MATS_ALL>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_SIMPLE k_render_ctr_simple(const RenderParams P) { render_ctr_lockstep<false, MATS_LAMBERT>(P); }
// ... and the Lambert-only kernel for lists of quads and cubes (round 5: cornell -1.1 %, profiles/r05/ab_scalar_diet.txt r05_q16)
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_SIMPLE k_render_ctr_simple_qc(const RenderParams P) { render_ctr_lockstep<false, MATS_LAMBERT, PRIMS_QUAD_CUBE>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_SIMPLE k_render_ctr_nospec(const RenderParams P) { render_ctr_lockstep<false, MATS_NO_SPECULAR>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) k_render_ctr_mesh(const RenderParams P) { render_ctr_lockstep<true, MATS_ALL>(P); }

}  // namespace mi355rt

// Round 1's form of the mesh path -- the in-wave state machine -- is retired from the product library: it is compiled only into the
// tests' reference build (-DMI355RT_REFS, build.build_device_variant("refs")), where it serves as a bit-identity reference for the
// wavefront kernel.  k_render_ctr_mesh above stays in the product library: the plain
// per-lane loop is the simplest statement of the BVH walk and what the diagnostic knob "kernel" = 1 selects.
#ifdef MI355RT_REFS
#include "rt_mesh_variants.h"   // k_render_ctr_sm (uses the shared pieces above)
#endif
#include "rt_wavefront.h"       // k_render_ctr_wf

namespace mi355rt {

// ===================================================================================================
// k_resolve -- ordered per-pixel sum, 1/spp, sqrt gamma, pack (renderer.rs:100-120), without LDS.
// One pixel's samples are contiguous in HBM (spp * 16 B apart from the next pixel's) and must be added in sample order.
// A 16-lane DPP row owns one pixel; lane s of the row loads sample
// c + s (64 lanes = 4 pixels x 16 samples = four fully coalesced 256-byte segments), and the sequential sum
// ((acc + x0) + x1) + ... + x15 runs ALONG the row: T = row_shr:1(T) + x, fifteen times.  Lane k's value is final after
// step k and every later step recomputes exactly the same sum (its left neighbour no longer changes), so no select is
// needed; lane 0 reads 0 from outside the row (bound_ctrl) and 0 + (acc + x0) is exact (a running sum that started at
// +0 is never -0).  Samples past spp enter as +0, which leaves a sum unchanged.  No LDS, 14 VGPRs: enough waves in
// flight to keep the HBM read stream busy (6.4 TB/s; the LDS-transpose version it replaces reached 3.3).
// ===================================================================================================
DI float dpp_row_shr1_zero(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true)); }
DI float dpp_row_ror1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, true)); }
__global__ void __launch_bounds__(256) k_resolve(const ResolveParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t s = lane & 15u;
    const uint32_t p = wave * 4u + (lane >> 4);                  // this row's pixel within the band
    const bool valid = p < P.band_pixels;
    size_t o = (size_t)P.band_pixel0 + p;                        // the pixel's number in processing order ...
    if (P.out_row && valid) {                                    // ... and where it belongs in the output (rows are processed dearest first, rt_api.cpp)
        const uint32_t jp = fastdiv((uint32_t)o, P.width_mul, P.width_shift);
        o = (size_t)P.out_row[jp] * P.width + ((uint32_t)o - jp * P.width);
    }
    const float* __restrict__ rad = P.radiance + 3u * (size_t)(valid ? p : 0u) * P.spp;     // 3 floats per sample
    float4* __restrict__ accum = reinterpret_cast<float4*>(P.accum);
    f3 acc = mk(0.f, 0.f, 0.f);                                  // meaningful in lane 0 of the row
    if (accum && P.accum_load && valid) { const float4 a = accum[o]; acc = mk(a.x, a.y, a.z); }
    for (uint32_t c = 0; c < P.spp; c += 16u) {
        f3 x = mk(0.f, 0.f, 0.f);
        if (valid && c + s < P.spp) {
            const float* q = rad + 3u * (c + s);
            x = mk(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2));
        }
        if (s == 0u) x = acc + x;                                 // renderer.rs:100 goes on where the previous 16 samples stopped
        f3 t = x;
#pragma unroll
        for (int k = 0; k < 15; ++k) t = mk(dpp_row_shr1_zero(t.x) + x.x, dpp_row_shr1_zero(t.y) + x.y, dpp_row_shr1_zero(t.z) + x.z);
        acc = mk(dpp_row_ror1(t.x), dpp_row_ror1(t.y), dpp_row_ror1(t.z));   // lane 0 <- lane 15: the sum so far
    }
    if (!valid || s != 0u) return;
    const f3 pixel = acc * P.inv_spp;                            // renderer.rs:103
    if (accum) accum[o] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    if (P.out_linear) { P.out_linear[3 * o] = pixel.x; P.out_linear[3 * o + 1] = pixel.y; P.out_linear[3 * o + 2] = pixel.z; }
    P.out_packed[o] = color_to_u32(sqrt3(pixel));                // renderer.rs:112-120
}

// ===================================================================================================
// k_render_ref -- validation: replay of the reference's per-row sequential stream, one lane per row
// ===================================================================================================
__global__ void __launch_bounds__(64) k_render_ref(const RefParams P) {
    cprim_t prims = (cprim_t)(P.prims);
    // One row per WAVE, carried by lane 0: rows consume their streams at data-dependent rates, so 64 rows in one wave
    // would run in lockstep through 64 different control flows; one active lane per wave has no divergence and the
    // rows spread over all CUs (the chip is otherwise idle in this validation mode).
    if ((threadIdx.x & 63u) != 0u) return;
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= P.n_rows) return;
    const uint32_t y = P.rows[j];
    RngRef rng;
    rng.seed_from_u64((uint64_t)y + (((uint64_t)P.seed_hi << 32) | (uint64_t)P.seed_lo));      // renderer.rs:91
    float* __restrict__ stack = P.fold_stack + (size_t)j * P.max_depth * 3;
    const float inv_spp = 1.0f / (float)P.spp;                                                  // renderer.rs:85
    unsigned long long n_rays = 0;
    for (uint32_t x = 0; x < P.width; ++x) {                                                    // renderer.rs:93
        f3 acc = mk(0.f, 0.f, 0.f);
        for (uint32_t s = 0; s < P.spp; ++s) {                                                  // renderer.rs:95
            const float u = ((float)x + rng.jitter_u()) / (float)P.width;
            const float v = ((float)y + rng.jitter_v()) / (float)P.height;
            f3 ro, rd; camera_ray(P.cam, u, v, ro, rd);
            f3 term = mk(0.f, 0.f, 0.f);
            uint32_t depth = 0;
            for (;;) {
                if (depth == P.max_depth) break;
                ++n_rays;
                Hit h;
                if (!hit_scene<true>(prims, P.n_prims, P.nodes, P.tris, ro, rd, h)) { term = miss_colour(P.sky, P.sky_w, P.sky_h, P.miss, rd); break; }
                f3 no, nd, atten, emitted;
                const float4 q0 = reinterpret_cast<const float4*>(P.mats + (h.mat_ff & 0x7FFFFFFFu))[0];
                if (!surface_scatter(P.mats, P.textures, q0, h, rd, rng, no, nd, atten, emitted)) { term = emitted; break; }
                stack[3 * depth] = atten.x; stack[3 * depth + 1] = atten.y; stack[3 * depth + 2] = atten.z;
                ro = no; rd = nd; ++depth;
            }
            f3 L = term;                                                                        // fold tail-first: emitted + atten * scattered (renderer.rs:33)
            for (uint32_t d = depth; d-- > 0;) L = mk(0.f, 0.f, 0.f) + mk(stack[3 * d], stack[3 * d + 1], stack[3 * d + 2]) * L;
            acc = acc + L;                                                                      // renderer.rs:100-101
        }
        const f3 pixel = acc * inv_spp;                                                         // renderer.rs:103
        const size_t o = (size_t)j * P.width + x;
        if (P.out_linear) { P.out_linear[3 * o] = pixel.x; P.out_linear[3 * o + 1] = pixel.y; P.out_linear[3 * o + 2] = pixel.z; }
        P.out_packed[o] = color_to_u32(sqrt3(pixel));
    }
    if (P.stats) { atomicAdd(&P.stats[0], (unsigned long long)P.width * P.spp); atomicAdd(&P.stats[1], n_rays); }
}

// ===================================================================================================
// Diagnostic kernels: one Material::scatter / one HittableList::hit per lane through the device functions above
// (tests/test_kat_functions.py compares them with independent numpy float32 known answers).
// ===================================================================================================
__global__ void __launch_bounds__(64) k_debug_scatter(const DevMat* __restrict__ mats, const DevTexture* __restrict__ texs, const DebugScatterIn* __restrict__ in, DebugScatterOut* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DebugScatterIn r = in[i];
    Hit h; h.t = 0.f; h.p = mk(r.p[0], r.p[1], r.p[2]); h.n = mk(r.n[0], r.n[1], r.n[2]);
    h.mat_ff = r.material | (r.front_face ? 0x80000000u : 0u);
    RngCtr rng; rng.start(r.k0, r.k1, r.x, r.s); rng.set_ray(r.ray); rng.load_block0();
    const float4 q0 = reinterpret_cast<const float4*>(mats + r.material)[0];
    f3 no = mk(0, 0, 0), nd = mk(0, 0, 0), atten = mk(0, 0, 0), emitted = mk(0, 0, 0);
    const bool ok = surface_scatter(mats, texs, q0, h, mk(r.rd[0], r.rd[1], r.rd[2]), rng, no, nd, atten, emitted);
    DebugScatterOut o{};
    o.scattered = ok ? 1.0f : 0.0f;
    o.o[0] = no.x; o.o[1] = no.y; o.o[2] = no.z; o.d[0] = nd.x; o.d[1] = nd.y; o.d[2] = nd.z;
    o.atten[0] = atten.x; o.atten[1] = atten.y; o.atten[2] = atten.z; o.emitted[0] = emitted.x; o.emitted[1] = emitted.y; o.emitted[2] = emitted.z;
    out[i] = o;
}
__global__ void __launch_bounds__(64) k_debug_hit(const DevPrim* prims_, uint32_t n_prims, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris,
                                                  const DebugHitIn* __restrict__ in, DebugHitOut* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DebugHitIn r = in[i];
    const f3 ro = mk(r.o[0], r.o[1], r.o[2]), rd = normalized(mk(r.d[0], r.d[1], r.d[2]));       // Ray::new, ray.rs:12-17
    Hit h;
    const bool hit = hit_scene<true>((cprim_t)prims_, n_prims, nodes, tris, ro, rd, h);
    DebugHitOut o{};
    o.hit = hit ? 1.0f : 0.0f;
    if (hit) {
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z; o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z; o.t = h.t;
        o.material = (float)(h.mat_ff & 0x7FFFFFFFu); o.front_face = (h.mat_ff >> 31) ? 1.0f : 0.0f;
    }
    out[i] = o;
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
int launch_debug_scatter(const DevMat* mats, const DevTexture* textures, const DebugScatterIn* in, DebugScatterOut* out, uint32_t n, void* stream) {
    hipLaunchKernelGGL(k_debug_scatter, dim3((n + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, mats, textures, in, out, n);
    return (int)hipGetLastError();
}
int launch_debug_hit(const DevPrim* prims, uint32_t n_prims, const DevNode* nodes, const DevTri* tris, const DebugHitIn* in, DebugHitOut* out, uint32_t n, void* stream) {
    hipLaunchKernelGGL(k_debug_hit, dim3((n + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, prims, n_prims, nodes, tris, in, out, n);
    return (int)hipGetLastError();
}
int launch_render_ctr(const RenderParams& p, uint32_t variant, uint32_t grid_blocks, void* stream) {
    switch (variant) {
        case KERNEL_LOCKSTEP:        hipLaunchKernelGGL(k_render_ctr_nomesh, dim3(grid_blocks), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, p); break;
        case KERNEL_LOCKSTEP_MESH:   hipLaunchKernelGGL(k_render_ctr_mesh, dim3(grid_blocks), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, p); break;
        case KERNEL_LOCKSTEP_SIMPLE: hipLaunchKernelGGL(k_render_ctr_simple, dim3(grid_blocks), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, p); break;
        case KERNEL_LOCKSTEP_SIMPLE_QC: hipLaunchKernelGGL(k_render_ctr_simple_qc, dim3(grid_blocks), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, p); break;
        case KERNEL_LOCKSTEP_NOSPEC: hipLaunchKernelGGL(k_render_ctr_nospec, dim3(grid_blocks), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_MESHFREE: hipLaunchKernelGGL(k_render_ctr_wf_meshfree, dim3(grid_blocks), dim3(BLOCK_THREADS_WF_MESHFREE), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_NOMETAL: hipLaunchKernelGGL(k_render_ctr_wf_nometal, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_NOMETAL_IDENT: hipLaunchKernelGGL(k_render_ctr_wf_nometal_ident, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_NOMETAL_SHALLOW: hipLaunchKernelGGL(k_render_ctr_wf_nometal_shallow, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT:       hipLaunchKernelGGL(k_render_ctr_wf, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_FIXAABB: hipLaunchKernelGGL(k_render_ctr_wf_fixaabb, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
#ifdef MI355RT_REFS
        case KERNEL_STATE_MACHINE_FIXAABB: hipLaunchKernelGGL(k_render_ctr_sm_fixaabb, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
        case KERNEL_STATE_MACHINE:   hipLaunchKernelGGL(k_render_ctr_sm, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
#endif
        default: return -1;                                  // a variant this library was not built with (render_ctr_variant_built)
    }
    return (int)hipGetLastError();
}
bool render_ctr_variant_built(uint32_t variant) {
#ifdef MI355RT_REFS
    return variant < KERNEL_VARIANTS && variant != KERNEL_RETIRED_5 && variant != KERNEL_RETIRED_6;
#else
    return variant == KERNEL_LOCKSTEP || variant == KERNEL_LOCKSTEP_MESH || variant == KERNEL_LOCKSTEP_SIMPLE || variant == KERNEL_LOCKSTEP_SIMPLE_QC || variant == KERNEL_LOCKSTEP_NOSPEC ||
           is_wavefront(variant);
#endif
}
int launch_resolve(const ResolveParams& p, void* stream) {
    const uint32_t blocks = (p.band_pixels + 15u) / 16u;         // 4 waves x 4 pixels per block
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
int launch_render_ref(const RefParams& p, void* stream) {
    hipLaunchKernelGGL(k_render_ref, dim3(p.n_rows), dim3(64), 0, (hipStream_t)stream, p);      // one wave per row
    return (int)hipGetLastError();
}
int query_render_ctr_occupancy(uint32_t variant, int* blocks_per_cu, int* vgprs, int* sgprs) {
    if (!render_ctr_variant_built(variant)) return -1;
    const void* fn = variant == KERNEL_LOCKSTEP ? reinterpret_cast<const void*>(k_render_ctr_nomesh)
                   : variant == KERNEL_LOCKSTEP_MESH ? reinterpret_cast<const void*>(k_render_ctr_mesh)
                   : variant == KERNEL_LOCKSTEP_SIMPLE ? reinterpret_cast<const void*>(k_render_ctr_simple)
                   : variant == KERNEL_LOCKSTEP_SIMPLE_QC ? reinterpret_cast<const void*>(k_render_ctr_simple_qc)
                   : variant == KERNEL_LOCKSTEP_NOSPEC ? reinterpret_cast<const void*>(k_render_ctr_nospec)
                   : variant == KERNEL_WAVEFRONT_NOMETAL ? reinterpret_cast<const void*>(k_render_ctr_wf_nometal)
                   : variant == KERNEL_WAVEFRONT_NOMETAL_IDENT ? reinterpret_cast<const void*>(k_render_ctr_wf_nometal_ident)
                   : variant == KERNEL_WAVEFRONT_NOMETAL_SHALLOW ? reinterpret_cast<const void*>(k_render_ctr_wf_nometal_shallow)
                   : variant == KERNEL_WAVEFRONT_MESHFREE ? reinterpret_cast<const void*>(k_render_ctr_wf_meshfree)
                   : variant == KERNEL_WAVEFRONT ? reinterpret_cast<const void*>(k_render_ctr_wf)
#ifdef MI355RT_REFS
                   : variant == KERNEL_STATE_MACHINE_FIXAABB ? reinterpret_cast<const void*>(k_render_ctr_sm_fixaabb)
                   : variant == KERNEL_STATE_MACHINE ? reinterpret_cast<const void*>(k_render_ctr_sm)
#endif
                                                       : reinterpret_cast<const void*>(k_render_ctr_wf_fixaabb);
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (int)block_threads_of(variant), 0);
    if (e != hipSuccess) return (int)e;
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return (int)e;
    *blocks_per_cu = nb; *vgprs = fa.numRegs; *sgprs = 0;
    return 0;
}

}  // namespace mi355rt
