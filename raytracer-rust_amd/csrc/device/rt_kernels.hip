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
//                  Randomness: counter-based Philox4x32-10 keyed by the image row, counters
//                  (x, sample, ray index, block): every draw is a pure function of the path, so any
//                  schedule / tiling / GPU count produces bit-identical radiance.
//                  Output: one float4 radiance per path into the HBM workspace.
//   k_resolve      per pixel, sums its spp radiance values IN SAMPLE ORDER (renderer.rs:100), scales by
//                  1/spp (:103), sqrt-gamma, clamp, pack 0x00RRGGBB (:112-120, color.rs:87-93).
//   k_render_ref   validation mode: one lane per image row replays the reference's sequential
//                  StdRng::seed_from_u64(y) stream (renderer.rs:91) and folds radiance tail-first.
//
// Numerics: compiled with -ffp-contract=off and correctly rounded f32 divide/sqrt; every expression
// keeps the reference's operation order (file:line cited per function), so results agree with the
// CPU oracle bit-for-bit wherever only + - * / sqrt are involved.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rt_device.h"
#include "../../../include/mi355rt.h"

namespace mi355rt {

#define DI __device__ __forceinline__

constexpr float EPS = 1e-4f;                      // renderer.rs:17
constexpr float PI_F = 3.14159265358979323846f;   // std::f32::consts::PI

typedef const __attribute__((address_space(4))) DevPrim* cprim_t;   // wave-uniform reads -> s_load

// Diagnostic-only cycle stamps (build with -DMI355RT_STAMPS into a separate library; the product build
// compiles Prof to nothing).  Shares of wave time per section, summed over waves, land in stats[2..].
#ifdef MI355RT_STAMPS
struct Prof {
    unsigned long long acc[6]; unsigned long long last;
    DI void begin() { for (int i = 0; i < 6; ++i) acc[i] = 0; last = now(); }
    DI static unsigned long long now() {
        unsigned long long t; __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0); return t;
    }
    DI void mark(int i) { unsigned long long t = now(); acc[i] += t - last; last = t; }
};
#else
struct Prof { DI void begin() {} DI void mark(int) {} };
#endif


// ---------------------------------------------------------------------------------------------------
// vec3.rs
// ---------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
DI f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DI f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DI f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DI f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DI f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }     // Color * Color
DI f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }     // Color / Color
DI f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
DI f3 divf(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // vec3.rs:17-19
DI f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // :21-27
DI float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                 // :29-31
DI float len(f3 a) { return sqrtf(len2(a)); }                                     // :33-35
DI f3 normalized(f3 a) { float l = len(a); if (l < EPS) return a; return a * (1.0f / l); }   // :37-44
DI bool near_zero(f3 a) { const float S = 1e-8f; return fabsf(a.x) < S && fabsf(a.y) < S && fabsf(a.z) < S; }  // :63-66
DI bool has_nan(f3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
DI bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
DI f3 nan3() { float n = __builtin_nanf(""); return mk(n, n, n); }
DI f3 splat(float v) { return mk(v, v, v); }
DI f3 sqrt3(f3 a) { return mk(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
DI f3 to_world(f3 local, f3 normal) {                                             // vec3.rs:72-81
    f3 up = (fabsf(normal.z) < 0.999f) ? mk(0.f, 0.f, 1.f) : mk(0.f, 1.f, 0.f);
    f3 tangent = normalized(cross(normal, up));
    f3 bitangent = cross(normal, tangent);
    return (tangent * local.x + bitangent * local.y) + normal * local.z;
}
DI float clamp01(float v) { if (v < 0.0f) return 0.0f; if (v > 1.0f) return 1.0f; return v; }   // f32::clamp, NaN stays
DI uint32_t as_u32_sat(float v) {                                                 // Rust `as u32`
    if (!(v == v) || v <= 0.0f) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
DI int32_t as_i32_sat(float v) {                                                  // Rust `as i32`
    if (!(v == v)) return 0;
    if (v <= -2147483648.0f) return (int32_t)0x80000000;
    if (v >= 2147483648.0f) return 0x7FFFFFFF;
    return (int32_t)v;
}
DI uint32_t color_to_u32(f3 c) {                                                  // color.rs:87-93
    c.x = clamp01(c.x); c.y = clamp01(c.y); c.z = clamp01(c.z);
    return (as_u32_sat(c.x * 255.0f) << 16) | (as_u32_sat(c.y * 255.0f) << 8) | as_u32_sat(c.z * 255.0f);
}

// ---------------------------------------------------------------------------------------------------
// RNG: float conversions of rand 0.9.1 (StandardUniform<f32>, UniformFloat::sample_single(-1..1))
// ---------------------------------------------------------------------------------------------------
DI float u32_to_f01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
DI float u32_to_range11(uint32_t w) { float v12 = __uint_as_float((w >> 9) | 0x3F800000u); float v01 = v12 - 1.0f; return v01 * 2.0f + -1.0f; }

// Philox4x32-10: counter-based, no state.  10 x (2 x 32x32->64 multiplies + 4 xor + 2 add).
// WIDE: one 64-bit product per multiplier -- v_mad_u64_u32 issues like ONE v_mul_hi_u32 (2.1 add slots, tools/microbench/int_mul.hip)
// and yields both halves, where __umulhi() and `*` written separately compile to two such instructions: 20 instead of 40
// slow multiplies per call.  The VALU-bound lockstep kernels use it (cornell -6.1 %, veach-mis -2.4 %); the latency-bound
// wavefront kernel is 2-3 % faster on the two independent multiplies (measured), so it keeps them.  Same bits either way.
template <bool WIDE = false>
DI void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (WIDE) {
            const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        } else {
            hi0 = __umulhi(M0, c0); lo0 = M0 * c0; hi1 = __umulhi(M1, c2); lo1 = M1 * c2;
        }
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Counter mode sampler: draws are addressed, not consumed (slots documented in oracle/rt_oracle.cpp
// and DESIGN.md): jitter = (ray 0, block 0, words 0/1); scatter event after ray r uses ray r+1:
// random::<f32>() number k -> block 0 word k; rejection try j -> block j words 1..3.
struct RngCtr {
    uint32_t k0, k1, x, s, ray;
    uint32_t b0[4];
    // One Philox call per loop iteration serves BOTH kinds of lanes: a freshly dealt path reads its camera
    // jitter from (ray 0, block 0); a continuing path reads its scatter draws from (ray r+1, block 0).
    DI void start(uint32_t k0_, uint32_t k1_, uint32_t x_, uint32_t s_) { k0 = k0_; k1 = k1_; x = x_; s = s_; ray = 0; }
    DI void next_event() { ++ray; }
    template <bool WIDE = false> DI void load_block0() { philox4x32_10<WIDE>(k0, k1, x, s, ray, 0u, b0); }
    DI float jitter_u() { return u32_to_f01(b0[0]); }
    DI float jitter_v() { return u32_to_f01(b0[1]); }
    DI void begin_scatter() {}
    DI float uniform01_0() { return u32_to_f01(b0[0]); }
    DI float uniform01_1() { return u32_to_f01(b0[1]); }
    template <bool WIDE = false> DI f3 cube_point(uint32_t j) {
        if (j == 0) return mk(u32_to_range11(b0[1]), u32_to_range11(b0[2]), u32_to_range11(b0[3]));
        uint32_t b[4]; philox4x32_10<WIDE>(k0, k1, x, s, ray, j, b);
        return mk(u32_to_range11(b[1]), u32_to_range11(b[2]), u32_to_range11(b[3]));
    }
};

// Reference mode sampler: rand_chacha ChaCha12 with the BlockRng 64-word buffer, seeded by
// rand_core's seed_from_u64 (PCG32 expansion).  SURVEY.md Appendix A.
DI uint32_t rotl(uint32_t v, int n) { return __builtin_rotateleft32(v, n); }
struct RngRef {
    uint32_t key[8]; uint32_t ctr_lo, ctr_hi; uint32_t idx; uint32_t buf[64];
    DI void seed_from_u64(uint64_t state) {
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ULL + 11634580027462260723ULL;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = __builtin_rotateright32(xs, rot);
        }
        ctr_lo = 0; ctr_hi = 0; idx = 64;
    }
    DI void block(uint32_t* out) {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                           key[4], key[5], key[6], key[7], ctr_lo, ctr_hi, 0u, 0u};
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = in[i];
#define MI_QR(a, b, c, d) \
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16); x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12); \
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);  x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
        for (int r = 0; r < 6; ++r) {
            MI_QR(0, 4, 8, 12) MI_QR(1, 5, 9, 13) MI_QR(2, 6, 10, 14) MI_QR(3, 7, 11, 15)
            MI_QR(0, 5, 10, 15) MI_QR(1, 6, 11, 12) MI_QR(2, 7, 8, 13) MI_QR(3, 4, 9, 14)
        }
#undef MI_QR
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
        if (++ctr_lo == 0) ++ctr_hi;
    }
    DI uint32_t next_u32() {
        if (idx >= 64) { for (int b = 0; b < 4; ++b) block(buf + 16 * b); idx = 0; }
        return buf[idx++];
    }
    DI float jitter_u() { return u32_to_f01(next_u32()); }
    DI float jitter_v() { return u32_to_f01(next_u32()); }
    DI void begin_scatter() {}
    DI float uniform01_0() { return u32_to_f01(next_u32()); }
    DI float uniform01_1() { return u32_to_f01(next_u32()); }
    template <bool WIDE = false> DI f3 cube_point(uint32_t) { float x = u32_to_range11(next_u32()); float y = u32_to_range11(next_u32()); float z = u32_to_range11(next_u32()); return mk(x, y, z); }
};

// ---------------------------------------------------------------------------------------------------
// Intersection.  The list walk (hittable.rs:45-58) only needs to know WHICH primitive is closest so far and at what t; the
// HitRecord (hittable.rs:10-27) of all but the last winner is never looked at.  So the walk carries a 4-register candidate
// (Cand) instead of the 8-register record, and finish_hit() builds the record of the winner once per ray, with exactly the
// arithmetic the reference's hit() performs for it (same inputs, same operations, same order -> same bits).  Measured
// reason: the compiler keeps a loop-carried record in two register sets and copies it at every nesting level of every
// primitive test -- 24-33 v_mov per quad, a third of its instructions; the copies scale with the size of the state.
// ---------------------------------------------------------------------------------------------------
struct Hit { float t; f3 p; f3 n; uint32_t mat_ff; };            // the finished record; `mat_ff` = material | front_face << 31
constexpr uint32_t CAND_NONE = 0xFFFFFFFFu;
struct Cand {
    float t;            // closest hit distance so far (world), +inf while idx == CAND_NONE
    uint32_t idx;       // list index of the primitive that owns it
    float aux;          // cube: the object-space slab distance t_hit (cube.rs:98); mesh: the walk's object-space best_t
    uint32_t aux2;      // mesh: the winning triangle (index into the leaf-ordered array)
};
DI void cand_reset(Cand& c) { c.t = __builtin_inff(); c.idx = CAND_NONE; c.aux = 0.f; c.aux2 = 0u; }
// The lockstep kernels of mesh-free scenes carry the cube's object-space hit point as well (3 more registers that only the cube
// loop touches): finish_cube() then needs neither the object-space ray nor `aux` again (-42 instructions per shaded cube hit).
struct CandP : Cand { f3 po; };
DI void cand_reset(CandP& c) { cand_reset(static_cast<Cand&>(c)); c.po = mk(0.f, 0.f, 0.f); }

DI void set_face(Hit& h, f3 rd, f3 outward, uint32_t material) {                 // hittable.rs:19-26
    bool front = dot(rd, outward) < 0.0f;
    h.n = front ? outward : -outward;
    h.mat_ff = material | (front ? 0x80000000u : 0u);
}

// objects/sphere.rs:15-53 (rejections folded into one predicate; sqrt of a negative discriminant is discarded)
// Every test below PROBES: it reads the candidate only as t_max and hands back fresh values (accepted?, t, aux); the one
// place that changes the loop-carried candidate is cand_take()'s selects.  (A test that assigned the candidate inside its own
// branches made the compiler carry two copies of it through the structurised switch: ~10 v_mov per quad, ~25 per cube.)
struct Probe { float t, aux; f3 po; };
DI bool cand_take(Cand& c, bool acc, uint32_t i, float t) { c.t = acc ? t : c.t; c.idx = acc ? i : c.idx; return acc; }
DI bool cand_take(Cand& c, bool acc, uint32_t i, const Probe& o) { c.aux = acc ? o.aux : c.aux; return cand_take(c, acc, i, o.t); }
DI bool cand_take(CandP& c, bool acc, uint32_t i, const Probe& o) {
    c.po.x = acc ? o.po.x : c.po.x; c.po.y = acc ? o.po.y : c.po.y; c.po.z = acc ? o.po.z : c.po.z;
    return cand_take(static_cast<Cand&>(c), acc, i, o.t);
}
DI bool hit_sphere(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    f3 center = mk(pr->d[0], pr->d[1], pr->d[2]); float radius = pr->d[3];
    f3 oc = ro - center;
    float a = dot(rd, rd);
    float half_b = dot(oc, rd);
    float cc = dot(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * cc;
    bool acc = false; float t = 0.f;
    if (!(disc < 0.0f)) {                                   // a wave whose lanes all miss the sphere skips the sqrt and the two divisions
        float sqrtd = sqrtf(disc);
        float r0 = (-half_b - sqrtd) / a, r1 = (-half_b + sqrtd) / a;
        const float t_max = c.t;
        const bool ok0 = !(r0 <= t_min || r0 >= t_max), ok1 = !(r1 <= t_min || r1 >= t_max);
        acc = ok0 || ok1; t = ok0 ? r0 : r1;
    }
    return cand_take(c, acc, i, t);
}

// objects/plane.rs:26-56
DI bool hit_plane(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    f3 p1 = mk(pr->d[0], pr->d[1], pr->d[2]), n = mk(pr->d[3], pr->d[4], pr->d[5]);
    float denom = dot(n, rd);
    float t = dot(n, p1 - ro) / denom;
    return cand_take(c, !(fabsf(denom) < EPS) && !(t <= t_min || t >= c.t), i, t);
}

// tungsten/objects/quad.rs:83-132.  The two cheap rejections (parallel ray, t out of range) are folded into one
// predicate so the wave takes a single branch into the parallelogram test; the arithmetic is unchanged (the
// division also runs for |denom| < EPS lanes, whose result is discarded).
DI bool hit_quad(cprim_t pr, uint32_t i, f3 ro, f3 rd, float t_min, Cand& c) {
    f3 n = mk(pr->d[9], pr->d[10], pr->d[11]);
    float denom = dot(n, rd);
    float t = (pr->d[12] - dot(n, ro)) / denom;
    const bool candidate = !(fabsf(denom) < EPS) && !(t <= t_min || t >= c.t);
#ifndef MI355RT_QUAD_BRANCHY                               // branch-free form: cornell 19.74 -> 19.61 ms, veach-mis +-0 (with the 4-register candidate)
    // every lane runs the parallelogram test; the candidate is updated by two selects (no exec-mask region, no copies per level)
    f3 hit_pos = ro + rd * t;
    f3 v = hit_pos - mk(pr->d[0], pr->d[1], pr->d[2]);
    float l0 = dot(v, mk(pr->d[3], pr->d[4], pr->d[5])) * pr->d[13];
    float l1 = dot(v, mk(pr->d[6], pr->d[7], pr->d[8])) * pr->d[14];
    const float lo = -EPS, hi = 1.0f + EPS;
    return cand_take(c, candidate && ((l0 >= lo && l0 <= hi) && (l1 >= lo && l1 <= hi)), i, t);
#else
    if (!candidate) return false;
    f3 hit_pos = ro + rd * t;
    f3 v = hit_pos - mk(pr->d[0], pr->d[1], pr->d[2]);
    float l0 = dot(v, mk(pr->d[3], pr->d[4], pr->d[5])) * pr->d[13];
    float l1 = dot(v, mk(pr->d[6], pr->d[7], pr->d[8])) * pr->d[14];
    const float lo = -EPS, hi = 1.0f + EPS;
    if (!((l0 >= lo && l0 <= hi) && (l1 >= lo && l1 <= hi))) return false;
    c.t = t; c.idx = i;
    return true;
#endif
}

// glam Mat4 * Vec4 pieces on the DevPrim cube/mesh record (see rt_device.h for the layout).  PrimPtr is the wave-uniform
// constant-address-space pointer of the list walk (scalar loads) or a per-lane global pointer in finish_hit().
template <class PrimPtr> DI f3 xform_w2o_point(PrimPtr pr, f3 p) {            // (w2o * (p, 1)).xyz
    const auto* m = pr->d;
    return mk(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
template <class PrimPtr> DI f3 xform_w2o_dir(PrimPtr pr, f3 v) {              // (w2o * (v, 0)).xyz ; zd = w_axis * 0.0f keeps -0.0 behaviour
    const auto* m = pr->d;
    return mk(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[28], ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[29],
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[30]);
}
template <class PrimPtr> DI f3 xform_o2w_point(PrimPtr pr, f3 p) {            // (o2w * (p, 1)).xyz
    const auto* m = pr->d + 16;
    return mk(((m[0] * p.x + m[3] * p.y) + m[6] * p.z) + m[9], ((m[1] * p.x + m[4] * p.y) + m[7] * p.z) + m[10],
              ((m[2] * p.x + m[5] * p.y) + m[8] * p.z) + m[11]);
}
template <class PrimPtr> DI f3 xform_normal(PrimPtr pr, f3 n) {               // (w2o.transpose() * (n, 0)).xyz
    const auto* m = pr->d;
    return mk(((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[31], ((m[4] * n.x + m[5] * n.y) + m[6] * n.z) + m[32],
              ((m[8] * n.x + m[9] * n.y) + m[10] * n.z) + m[33]);
}
DI float glam_signum(float v) { if (v != v) return v; return copysignf(1.0f, v); }

// objects/cube.rs:59-158, the part that decides whether and where the cube is hit; the face normal (cube.rs:105-143) is
// computed by finish_hit() for the winner only.
DI uint32_t cube_axis(f3 po) {                                                          // cube.rs:112-133 as selects
    const float ax = fabsf(po.x), ay = fabsf(po.y), az = fabsf(po.z);
    const float tol = 1e-4f;
    return (fabsf(ax - 0.5f) < tol) ? 0u : (fabsf(ay - 0.5f) < tol) ? 1u : (fabsf(az - 0.5f) < tol) ? 2u
         : (ax > ay && ax > az) ? 0u : (ay > az) ? 1u : 2u;
}
template <class C>
DI bool hit_cube(cprim_t pr, uint32_t i, f3 ro_w, f3 rd_w, float t_min, C& c) {
    f3 ro = xform_w2o_point(pr, ro_w);
    f3 rd = xform_w2o_dir(pr, rd_w);
    float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;
    float t1x = (-0.5f - ro.x) * ix, t2x = (0.5f - ro.x) * ix;
    float t1y = (-0.5f - ro.y) * iy, t2y = (0.5f - ro.y) * iy;
    float t1z = (-0.5f - ro.z) * iz, t2z = (0.5f - ro.z) * iz;
    float t_enter = fmaxf(fminf(t1x, t2x), fmaxf(fminf(t1y, t2y), fminf(t1z, t2z)));
    float t_exit = fminf(fmaxf(t1x, t2x), fminf(fmaxf(t1y, t2y), fmaxf(t1z, t2z)));
    const float t_hit = (t_enter > 0.0f) ? t_enter : t_exit;
    const float t_max = c.t;
    const bool candidate = !(t_exit < t_enter || t_exit <= 0.0f) && !(t_hit >= t_max || t_hit <= t_min || t_hit < EPS);   // cube.rs:90-103, one branch
    Probe o; o.t = 0.f; o.aux = t_hit;
    bool acc = false;
    if (candidate) {
        f3 po = ro + rd * t_hit;
        o.po = po;
        f3 pw = xform_o2w_point(pr, po);
        o.t = dot(pw - ro_w, rd_w);                                                         // cube.rs:145-153: the same dot product twice
        acc = !((o.t < 0.0f) || (o.t < t_min || o.t > t_max));
    }
    return cand_take(c, acc, i, o);
}
// The record of a cube hit.  The object-space normal is +-e_axis, normalize_or_zero() of such a vector is the vector
// itself (1/sqrt(1) == 1), and the world normal normalized(w2o^T * (n, 0)) therefore takes one of 6 values per cube, which
// the host precomputed with the same f32 operations (DevPrim.d[34..51], rt_api.cpp cube_normal_table).
template <class PrimPtr>
DI f3 cube_po(PrimPtr pr, const Cand& c, f3 ro_w, f3 rd_w) {                                // cube.rs:104, from the slab distance the candidate kept
    const f3 ro = xform_w2o_point(pr, ro_w), rd = xform_w2o_dir(pr, rd_w);
    return ro + rd * c.aux;
}
template <class PrimPtr> DI f3 cube_po(PrimPtr, const CandP& c, f3, f3) { return c.po; }  // ... or the point itself
template <class PrimPtr, class C>
DI void finish_cube(PrimPtr pr, const C& c, f3 ro_w, f3 rd_w, f3& p, f3& outward) {
    const f3 po = cube_po(pr, c, ro_w, rd_w);
    p = xform_o2w_point(pr, po);
    const uint32_t axis = cube_axis(po);
    const float cc = (axis == 0u) ? po.x : ((axis == 1u) ? po.y : po.z);
    f3 nw;
    if (cc != cc) nw = normalized(xform_normal(pr, mk(0.f, 0.f, 0.f)));                     // NaN signum -> normalize_or_zero -> the zero vector (cube.rs:134)
    else {
        const uint32_t code = 2u * axis + (__float_as_uint(cc) >> 31);                      // glam signum: the sign bit decides, also for +-0
        const auto* t = pr->d + 34u + 3u * code;
        nw = mk(t[0], t[1], t[2]);
    }
    outward = nw;
}

// mesh/mesh_object.rs:263-329 + acceleration/bvh.rs:78-170 + acceleration/aabb.rs:27-45.
// Stackless walk over two-link nodes (rt_device.h): hit inner -> left child, everything else -> the escape link, which
// reproduces the reference's left-then-right recursion exactly; `best_t` plays the role of the recursion's shrinking
// t_max.  The walk is split into setup / node step / leaf / finalize so that the state-machine kernel can interleave
// the traversals of different lanes; hit_mesh() composes them into the plain per-lane loop.
struct MeshTrav {
    f3 ro, rd;                 // object-space ray (direction normalised twice, mesh_object.rs:289)
    float ix, iy, iz;          // 1/d, aabb.rs:29 (same value at every node)
    float len_raw;             // |w2o * d_world| for the (sic) t_world formula
    uint32_t node;             // next node to visit; NODE_END: the walk is over
    float best_t; uint32_t best_tri;
    uint32_t leaf_a, leaf_b;   // pending leaf (first triangle, count); leaf_b == 0: none
};
template <class PrimPtr>
DI void mesh_setup(PrimPtr pr, f3 ro_w, f3 rd_w, float t_max, MeshTrav& m) {   // mesh_object.rs:264-291
    m.ro = xform_w2o_point(pr, ro_w);
    f3 rd_raw = xform_w2o_dir(pr, rd_w);
    m.len_raw = len(rd_raw);
    m.rd = normalized(normalized(rd_raw));
    m.ix = 1.0f / m.rd.x; m.iy = 1.0f / m.rd.y; m.iz = 1.0f / m.rd.z;
    m.node = pr->node_begin;
    m.best_t = t_max; m.best_tri = 0xFFFFFFFFu; m.leaf_b = 0; m.leaf_a = 0;
}
// Visit m.node (box test, aabb.rs:27-45).  Afterwards m.node is the next node to visit and, when a leaf was hit, its
// triangles are pending (m.leaf_b > 0) and must be tested before the walk goes on.
// FIXED_AABB: MI355RT_FLAG_FIXED_AABB -- a box is missed only when t_max < t_min (the reference misses on <=, aabb.rs:41).
// USE_LDS: nodes below `lds_count` are read from the workgroup's LDS copy (ds_read_b128), the rest from global memory.
typedef float lds_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lds_v4f* lds_nodes_t;
template <bool FIXED_AABB = false, bool USE_LDS = false>
DI void mesh_step(const float4* __restrict__ n4, lds_nodes_t lds, uint32_t lds_count, float t_min, MeshTrav& m) {
    // 32-bit byte offset from the uniform base: the load takes the base from SGPRs instead of a 64-bit per-lane address
    // The choice between the LDS copy and global memory is made for the WAVE (the LDS copy is a copy: global memory holds every
    // node): a per-lane choice would make the LDS readers wait for the other lanes' global loads (both paths fill the
    // same registers) and serialise the two latencies.  Whole array in LDS (semesterbild): always the LDS path.
    float4 q0, q1;
    if (USE_LDS && __ballot(m.node >= lds_count) == 0ull) {
        lds_nodes_t lq = reinterpret_cast<lds_nodes_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(lds) + (m.node << 5));
        const lds_v4f l0 = lq[0], l1 = lq[1];
        q0 = make_float4(l0.x, l0.y, l0.z, l0.w); q1 = make_float4(l1.x, l1.y, l1.z, l1.w);
    } else {
        const float4* __restrict__ nq = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(n4) + (m.node << 5));
        q0 = nq[0]; q1 = nq[1];
    }
    const uint32_t a = __float_as_uint(q0.w), b = __float_as_uint(q1.w);
    // aabb.rs:31-44 returns false at the first axis whose interval is empty.  tmin only grows and tmax only shrinks from axis
    // to axis (f32::max / f32::min ignore a NaN operand, so they never move the other way), hence an interval that is empty
    // after some axis is still empty after the last one and vice versa: ONE test after the z axis decides the same.
    float tmin = t_min, tmax = m.best_t;
    {   float t0 = (q0.x - m.ro.x) * m.ix, t1 = (q1.x - m.ro.x) * m.ix; if (m.ix < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    {   float t0 = (q0.y - m.ro.y) * m.iy, t1 = (q1.y - m.ro.y) * m.iy; if (m.iy < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    {   float t0 = (q0.z - m.ro.z) * m.iz, t1 = (q1.z - m.ro.z) * m.iz; if (m.iz < 0.0f) { float s = t0; t0 = t1; t1 = s; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1); }
    const bool ok = !(FIXED_AABB ? (tmax < tmin) : (tmax <= tmin));
    // Branch-free successor: hit inner node -> its left child `a`; missed node or leaf -> the escape link (a hit leaf's
    // triangles are tested first: leaf_b > 0 holds the walk until mesh_leaf() has run).
    const uint32_t count = b >> NODE_LINK_BITS, esc = b & NODE_END;
    const bool take_leaf = ok && count != 0u;
    m.node = (ok && count == 0u) ? a : esc;
    m.leaf_a = take_leaf ? a : m.leaf_a;
    m.leaf_b = take_leaf ? count : m.leaf_b;
}
// Moeller-Trumbore over the pending leaf, bvh.rs:91-138
DI void mesh_leaf(const float4* __restrict__ t4, float t_min, MeshTrav& m) {
    for (uint32_t k = 0; k < m.leaf_b; ++k) {
        const float4* __restrict__ tq = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(t4) + (m.leaf_a + k) * 48u);
        const float4 r0 = tq[0], r1 = tq[1], r2 = tq[2];
        const f3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
        f3 hh = cross(m.rd, e2);
        float aa = dot(e1, hh);
        float f = 1.0f / aa;
        f3 s = m.ro - v0;
        float u = f * dot(s, hh);
        f3 q = cross(s, e1);
        float v = f * dot(m.rd, q);
        float t = f * dot(e2, q);
        // bvh.rs:99-116, the four `continue`s as one predicate (same values, one branch)
        const bool hit = !(fabsf(aa) < EPS) && (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t > t_min && t < m.best_t);
        if (hit) { m.best_t = t; m.best_tri = m.leaf_a + k; }
    }
    m.leaf_b = 0;
}
// The end of Mesh::hit that decides acceptance (mesh_object.rs:312-318); the record is built by finish_mesh() for the winner.
DI bool mesh_accept(uint32_t i, const MeshTrav& m, f3 rd_w, float t_min, Cand& c) {
    float t_world = m.best_t * m.len_raw / len(rd_w);                   // (sic) mesh_object.rs:312-314
    const bool acc = (m.best_tri != 0xFFFFFFFFu) && !(t_world < t_min || t_world > c.t);
    c.aux2 = acc ? m.best_tri : c.aux2;
    Probe o; o.t = t_world; o.aux = m.best_t;
    return cand_take(c, acc, i, o);
}
// mesh_object.rs:264-310 for the winning triangle: the object-space ray is recomputed exactly as mesh_setup() computed it.
template <class PrimPtr>
DI void finish_mesh(PrimPtr pr, const float4* __restrict__ t4, const Cand& c, f3 ro_w, f3 rd_w, f3& p, f3& outward) {
    const f3 ro = xform_w2o_point(pr, ro_w);
    const f3 rd = normalized(normalized(xform_w2o_dir(pr, rd_w)));
    const float4 r2 = t4[3 * (size_t)c.aux2 + 2];
    f3 tn = mk(r2.y, r2.z, r2.w);
    f3 pos_obj = ro + rd * c.aux;
    f3 n_obj = (dot(rd, tn) < 0.0f) ? tn : -tn;                         // bvh.rs:118-124
    p = xform_o2w_point(pr, pos_obj);
    outward = normalized(xform_normal(pr, n_obj));
}
DI bool hit_mesh(cprim_t pr, uint32_t i, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris, f3 ro_w, f3 rd_w,
                 float t_min, Cand& c) {
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(tris);
    MeshTrav m; mesh_setup(pr, ro_w, rd_w, c.t, m);
    while (m.node != NODE_END) {
        mesh_step(n4, nullptr, 0u, t_min, m);
        if (m.leaf_b) mesh_leaf(t4, t_min, m);
    }
    return mesh_accept(i, m, rd_w, t_min, c);
}

// The HitRecord of the list's winner (hittable.rs:10-27), once per ray.  Lanes of a wave may have different winners, so
// the primitive record is read per lane here (global loads; L1/L2 resident).
template <bool HAS_MESH, class C>
DI void finish_hit(const DevPrim* __restrict__ prims, const DevTri* __restrict__ tris, const C& c, f3 ro, f3 rd, Hit& h) {
    const DevPrim* __restrict__ pr = prims + c.idx;
    const uint32_t kind = pr->kind;
    // Mesh-free lists: each kind only says where the hit is and which way its surface faces; HitRecord::set_face_normal
    // (hittable.rs:19-26) then runs once for all lanes of the wave, whatever their winners are (cornell -2.5 %).  With meshes in
    // the list every kind finishes its own record (measured: the shared tail costs the wavefront kernel 3-4 %).
#ifndef MI355RT_FINISH_SHARED
#define MI355RT_FINISH_SHARED (!HAS_MESH)
#endif
    if (MI355RT_FINISH_SHARED) {
        f3 p = ro + rd * c.t, outward;                                        // sphere.rs:35, plane.rs:40, quad.rs:103
        if (kind == MI355RT_PRIM_QUAD) {                                      // quad.rs:103-131
            outward = mk(pr->d[9], pr->d[10], pr->d[11]);                     // dot(ray.direction, normal): the same sum of the same products as `denom`
        } else if (kind == MI355RT_PRIM_CUBE) {
            finish_cube(pr, c, ro, rd, p, outward);
        } else if (kind == MI355RT_PRIM_SPHERE) {                             // sphere.rs:35-52
            outward = divf(p - mk(pr->d[0], pr->d[1], pr->d[2]), pr->d[3]);
        } else if (kind == MI355RT_PRIM_PLANE) {                              // plane.rs:40-55
            outward = mk(pr->d[3], pr->d[4], pr->d[5]);
        } else if (HAS_MESH) {
            finish_mesh(pr, reinterpret_cast<const float4*>(tris), c, ro, rd, p, outward);
        }
        h.t = c.t; h.p = p;
        set_face(h, rd, outward, pr->material);
    } else {
        h.t = c.t;
        if (kind == MI355RT_PRIM_QUAD) {
            h.p = ro + rd * c.t;
            set_face(h, rd, mk(pr->d[9], pr->d[10], pr->d[11]), pr->material);
        } else if (kind == MI355RT_PRIM_CUBE) {
            f3 outward; finish_cube(pr, c, ro, rd, h.p, outward);
            set_face(h, rd, outward, pr->material);
        } else if (kind == MI355RT_PRIM_SPHERE) {
            h.p = ro + rd * c.t;
            set_face(h, rd, divf(h.p - mk(pr->d[0], pr->d[1], pr->d[2]), pr->d[3]), pr->material);
        } else if (kind == MI355RT_PRIM_PLANE) {
            h.p = ro + rd * c.t;
            set_face(h, rd, mk(pr->d[3], pr->d[4], pr->d[5]), pr->material);
        } else if (HAS_MESH) {
            f3 outward; finish_mesh(pr, reinterpret_cast<const float4*>(tris), c, ro, rd, h.p, outward);
            set_face(h, rd, outward, pr->material);
        }
    }
}

// hittable.rs:45-58 -- HittableList::hit with t_min = EPSILON, t_max = INFINITY (renderer.rs:24)
template <bool HAS_MESH, class C>
DI void walk_list(cprim_t prims, uint32_t n_prims, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris, f3 ro, f3 rd, C& c) {
    // Same order as the list, but the dispatch on the kind (wave-uniform: a scalar branch) is taken once per RUN of equal kinds
    // (DevPrim.run_end, host-computed) and each kind has its own tight loop: the structurised switch inside one loop carried the
    // candidate through a chain of merge blocks with register copies at every one of them.
    // The kinds are tried in a fixed cyclic order, each as `if (the run at i is of this kind) loop over the run`: plain nested
    // structured control flow (a `switch` here is lowered to a chain of flow blocks, each with its own copies of the candidate).
    uint32_t i = 0;
    while (i < n_prims) {
#define MI_RUN(KIND, CALL) if (i < n_prims && prims[i].kind == (KIND)) { const uint32_t end = min(prims[i].run_end, n_prims); do { CALL; } while (++i < end); }
        MI_RUN(MI355RT_PRIM_QUAD,   hit_quad(prims + i, i, ro, rd, EPS, c))
        MI_RUN(MI355RT_PRIM_CUBE,   hit_cube(prims + i, i, ro, rd, EPS, c))
        MI_RUN(MI355RT_PRIM_SPHERE, hit_sphere(prims + i, i, ro, rd, EPS, c))
        MI_RUN(MI355RT_PRIM_PLANE,  hit_plane(prims + i, i, ro, rd, EPS, c))
        if (HAS_MESH) { MI_RUN(MI355RT_PRIM_MESH, hit_mesh(prims + i, i, nodes, tris, ro, rd, EPS, c)) }
        else if (i < n_prims && prims[i].kind >= MI355RT_PRIM_MESH) ++i;          // cannot happen (the host picks this kernel only for mesh-free lists); keeps the loop finite
#undef MI_RUN
    }
}
// CARRY_PO: the candidate keeps the cube's object-space hit point (CandP).  On for the kernel of Lambert-only scenes (cornell
// -1.8 %); the general mesh-free kernel has no registers to spare for it (veach-mis +1.9 % with it: spills).
template <bool HAS_MESH, bool CARRY_PO = false>
DI bool hit_scene(cprim_t prims, uint32_t n_prims, const DevNode* __restrict__ nodes, const DevTri* __restrict__ tris,
                  f3 ro, f3 rd, Hit& best) {
    typename std::conditional<CARRY_PO && !HAS_MESH, CandP, Cand>::type c; cand_reset(c);
    walk_list<HAS_MESH>(prims, n_prims, nodes, tris, ro, rd, c);
    if (c.idx == CAND_NONE) return false;
    finish_hit<HAS_MESH>((const DevPrim*)prims, tris, c, ro, rd, best);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Materials
// ---------------------------------------------------------------------------------------------------
DI f3 mat_reflect(f3 v, f3 n) {                                                   // material.rs:194-206
    if (has_nan(v)) return nan3();
    if (has_nan(n) || is_zero(n)) return nan3();
    return v - (n * 2.0f) * dot(v, n);
}
DI float powi5(float x) { return x * ((x * x) * (x * x)); }                       // llvm.powi.f32(x, 5)
DI float schlick(float cosine, float ref_idx) {                                   // material.rs:221-227 == tungsten/materials.rs:23-27
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}
DI f3 fresnel_conductor(float cos_theta, f3 eta, f3 k) {                          // tungsten/materials.rs:184-202
    cos_theta = clamp01(cos_theta);
    f3 cos2 = splat(cos_theta * cos_theta);
    f3 sin2 = splat(1.0f) - cos2;
    f3 eta2 = eta * eta, k2 = k * k;
    f3 t0 = eta2 - k2 - sin2;
    f3 a2plusb2 = sqrt3(t0 * t0 + splat(4.0f) * eta2 * k2);
    f3 t1 = a2plusb2 + cos2;
    f3 a = sqrt3((a2plusb2 + t0) * splat(0.5f));
    f3 t2 = splat(2.0f * cos_theta) * a;
    f3 rs = (t1 - t2) / (t1 + t2);
    f3 t3 = cos2 * a2plusb2 + sin2 * sin2;
    f3 rp = rs * ((t3 - t2) / (t3 + t2));
    return (rs + rp) * splat(0.5f);
}
DI float ggx_g1(float n_dot_x, float roughness) {                                 // tungsten/materials.rs:205-216
    if (n_dot_x <= 0.0f) return 0.0f;
    float a = roughness * roughness;
    float k = a / 2.0f;
    float denom = n_dot_x * (1.0f - k) + k;
    if (denom < EPS) return 1.0f;
    return n_dot_x / denom;
}
DI float beckmann_lambda(float a, float x) {                                      // tungsten/materials.rs:225-232
    float t = 1.0f / (a * x);
    if (t < 1.6f) return (1.0f - 1.259f * t + 0.396f * t * t) / (3.535f * t + 2.181f * t * t);
    return 0.0f;
}

// Result of one surface interaction (renderer.rs:26-36): either the path goes on (scattered ray +
// attenuation) or it ends with `emitted` (scatter -> None).
// Split in two so that the counter-mode kernels can run the unit-ball rejection of the Lambert-style bounce
// wave-cooperatively between the halves: scatter_pre() decides everything except that direction (it sets
// `diffuse`), diffuse_finish() turns the accepted unit-ball point into the scattered ray (material.rs:54-62).
// SIMPLE: the scene's materials are only Lambertian (solid) / Emissive / Null (checked on the host), so every
// scattering material is the Lambert bounce and the other BSDFs -- which set the register peak -- are compiled out.
// tungsten/parser.rs:222-240: TextureMaterial's texel, looked up by the hit NORMAL (equirectangular, nearest), as a colour in [0, 1]
DI f3 texture_lookup(const DevTexture* __restrict__ texs, uint32_t index, float h_offset, f3 n) {
    const DevTexture t = texs[index];
    const float theta = acosf(n.y);                                                 // :223
    const float phi = atan2f(n.z, n.x) + PI_F;                                      // :224
    float u = phi / (2.0f * PI_F);                                                  // :225
    const float v = theta / PI_F;                                                   // :226
    u = fmodf(u + h_offset, 1.0f);                                                  // :227  (f32 % f32)
    const uint32_t xp = as_u32_sat(fmaxf(u, 0.0f) * (float)(t.width - 1u));        // :231
    const uint32_t yp = as_u32_sat(fmaxf(v, 0.0f) * (float)(t.height - 1u));       // :232
    const uint32_t px = t.rgba8[(size_t)min(yp, t.height - 1u) * t.width + min(xp, t.width - 1u)];   // :234-236
    return mk((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f);   // :237-241
}

template <bool SIMPLE, bool WIDE = false, class Rng>
DI bool scatter_pre(const DevMat* __restrict__ mats, const DevTexture* __restrict__ texs, const float4 q0, const Hit& h, f3 rd_in, Rng& rng, float& side, f3& raw_d, f3& atten, f3& emitted, bool& diffuse_out) {
    const float4* __restrict__ m4 = reinterpret_cast<const float4*>(mats + (h.mat_ff & 0x7FFFFFFFu));
    const uint32_t kind = __float_as_uint(q0.x);
    const f3 albedo = mk(q0.y, q0.z, q0.w);
    const bool front_face = (h.mat_ff >> 31) != 0;
    emitted = mk(0.f, 0.f, 0.f);
    diffuse_out = false;
    side = EPS;                                                                    // every material but the dielectric leaves on the normal's side
    if (kind == MI355RT_MAT_EMISSIVE) { emitted = albedo; return false; }         // material.rs:179-191
    if (kind == MI355RT_MAT_NULL) return false;                                   // material.rs:239-251
    rng.begin_scatter();
    bool diffuse = false;                                                          // Lambert-style bounce shared by 3 materials
    atten = albedo;
    if (SIMPLE || kind == MI355RT_MAT_LAMBERT_SOLID) {                             // material.rs:47-71
        diffuse = true;
    } else if (kind == MI355RT_MAT_LAMBERT_CHECKER) {                              // tungsten/materials.rs:89-99
        const float4 q1 = m4[1];
        float inv_scale = q1.w;
        int32_t sum = (int32_t)((uint32_t)as_i32_sat(floorf(h.p.x * inv_scale)) + (uint32_t)as_i32_sat(floorf(h.p.y * inv_scale)) +
                                (uint32_t)as_i32_sat(floorf(h.p.z * inv_scale)));
        if ((sum & 1) != 0) atten = mk(q1.x, q1.y, q1.z);
        diffuse = true;
    } else if (kind == MI355RT_MAT_TEXTURE) {                                      // tungsten/parser.rs:205-243
        const float4 q1 = m4[1];
        atten = albedo * texture_lookup(texs, __float_as_uint(m4[3].w), q1.w, h.n);
        diffuse = true;
    } else if (kind == MI355RT_MAT_PLASTIC) {                                      // tungsten/materials.rs:29-65
        float ior = m4[1].w;
        float dn = dot(rd_in, h.n);
        float cosine = (dn > 0.0f) ? ior * dn / len(rd_in) : -dn / len(rd_in);
        float reflect_prob = schlick(cosine, ior);
        if (rng.uniform01_0() < reflect_prob) {
            raw_d = rd_in - (h.n * 2.0f) * dot(rd_in, h.n);                        // Vec3::reflect, vec3.rs:68-70: .normalized(), then Ray::new
            atten = mk(0.9f, 0.9f, 0.9f);
        } else {
            diffuse = true;
        }
    } else if (kind == MI355RT_MAT_METAL) {                                        // material.rs:87-110
        float fuzz = m4[1].w;
        f3 reflected = mat_reflect(normalized(rd_in), h.n);
        f3 fuzzed = reflected;
        if (fuzz > 0.0f) {
            f3 p; uint32_t j = 0;
            do { p = rng.template cube_point<WIDE>(j); ++j; } while (!(len2(p) < 1.0f));   // vec3.rs:54-61
            fuzzed = reflected + p * fuzz;
        }
        if (!(dot(fuzzed, h.n) > 0.0f)) return false;
        raw_d = fuzzed;
    } else if (kind == MI355RT_MAT_DIELECTRIC) {                                   // material.rs:122-162
        float ri = m4[1].w;
        float ratio = front_face ? (1.0f / ri) : (ri / 1.0f);
        f3 unit = normalized(rd_in);
        float cos_theta = fminf(dot(-unit, h.n), 1.0f);
        float sin2 = 1.0f - cos_theta * cos_theta;
        bool cannot_refract = ratio * ratio * sin2 > 1.0f;
        float reflectance = schlick(cos_theta, 1.0f / ratio);
        f3 dir;
        if (cannot_refract || reflectance > rng.uniform01_0()) {                   // no draw under TIR (material.rs:145)
            dir = mat_reflect(unit, h.n);
        } else {                                                                   // refract(), material.rs:208-219
            float ct = fminf(dot(-unit, h.n), 1.0f);
            f3 perp = (unit + h.n * ct) * ratio;
            float par2 = 1.0f - len2(perp);
            dir = (par2 < 0.0f) ? mat_reflect(unit, h.n) : perp + h.n * (-sqrtf(par2));
        }
        side = (dot(dir, h.n) > 0.0f) ? EPS : -EPS;                                // p - n*EPS == p + n*(-EPS) bit for bit
        raw_d = dir;
        atten = mk(1.f, 1.f, 1.f);
    } else {                                                                       // RoughConductor, tungsten/materials.rs:306-377
        const bool ggx = (kind == MI355RT_MAT_ROUGH_GGX);
        if (has_nan(rd_in)) return false;
        if (has_nan(h.n) || is_zero(h.n)) return false;
        f3 n = h.n;
        f3 v = -normalized(rd_in);
        if (has_nan(v)) return false;
        const float4 q1 = m4[1], q2 = m4[2], q3 = m4[3];
        float rough = q1.w;
        f3 eta = mk(q2.y, q2.z, q2.w), kk = mk(q3.x, q3.y, q3.z);
        // sample_ggx / sample_beckmann, tungsten/materials.rs:236-290
        float u1 = fmaxf(rng.uniform01_0(), 1e-6f);
        float u2 = rng.uniform01_1();
        float theta_arg;
        if (ggx) { float a = rough * rough; theta_arg = a * a * (-logf(u1)) / (1.0f - u1); }
        else { theta_arg = -(rough * rough * logf(u1)); }
        f3 hv;
        if ((theta_arg != theta_arg) || isinf(theta_arg) || theta_arg < 0.0f) {
            hv = to_world(mk(0.f, 0.f, 1.f), n);
        } else {
            float theta = atanf(sqrtf(theta_arg));
            float phi = 2.0f * PI_F * u2;
            float st, ct, sp, cp;                          // sin_cos(): one argument reduction serves both values
            sincosf(theta, &st, &ct); sincosf(phi, &sp, &cp);
            f3 hl = mk(st * cp, st * sp, ct);
            hv = has_nan(hl) ? to_world(mk(0.f, 0.f, 1.f), n) : to_world(hl, n);
        }
        if (has_nan(hv)) return false;
        f3 l = mat_reflect(-v, hv);
        if (has_nan(l)) return false;
        if (dot(l, n) <= 0.0f) return false;
        float n_dot_l = fmaxf(dot(n, l), 0.0f), n_dot_v = fmaxf(dot(n, v), 0.0f);
        float n_dot_h = fmaxf(dot(n, hv), 0.0f), v_dot_h = fmaxf(dot(v, hv), 0.0f);
        float g = ggx ? ggx_g1(n_dot_v, rough) * ggx_g1(n_dot_l, rough)
                      : 1.0f / (1.0f + beckmann_lambda(rough, n_dot_v) + beckmann_lambda(rough, n_dot_l));
        f3 f = fresnel_conductor(v_dot_h, eta, kk);
        f3 num = f * g * v_dot_h;
        float den = n_dot_v * n_dot_h + EPS;
        atten = (den > EPS) ? albedo * divf(num, den) : mk(0.f, 0.f, 0.f);
        raw_d = l;
    }
    diffuse_out = diffuse;
    return true;
}
DI f3 diffuse_finish(const Hit& h, f3 p) {                                          // material.rs:54-62
    f3 dir = h.n + normalized(p);
    return near_zero(dir) ? h.n : dir;
}
// What every scatter() and Camera::get_ray end with: `.normalized()` of the direction, then Ray::new normalises again
// (ray.rs:12-17) -- and the origin offset along the normal.  The callers run it ONCE for all lanes of the wave, whatever
// branch produced the raw direction (it was the tail of every material branch and of the camera ray: ~66 instructions each).
DI f3 ray_direction(f3 raw) { return normalized(normalized(raw)); }
DI f3 scatter_origin(const Hit& h, float side) { return h.p + h.n * side; }
// Sequential composition (reference-stream replay kernel): random_in_unit_sphere as the plain loop, vec3.rs:54-61.
template <class Rng>
DI bool surface_scatter(const DevMat* __restrict__ mats, const DevTexture* __restrict__ texs, const float4 q0, const Hit& h, f3 rd_in, Rng& rng, f3& new_o, f3& new_d, f3& atten, f3& emitted) {
    bool diffuse = false; float side = EPS; f3 raw = mk(0.f, 0.f, 1.f);
    if (!scatter_pre<false>(mats, texs, q0, h, rd_in, rng, side, raw, atten, emitted, diffuse)) return false;
    if (diffuse) {
        f3 p; uint32_t j = 0;
        do { p = rng.cube_point(j); ++j; } while (!(len2(p) < 1.0f));
        raw = diffuse_finish(h, p);
    }
    new_o = scatter_origin(h, side); new_d = ray_direction(raw);
    return true;
}

// renderer.rs:38-63: the colour a missing ray returns -- equirectangular HDR lookup (nearest texel) when a skybox
// is loaded, Color::GRAY (passed in as `miss`) otherwise.
DI f3 miss_colour(const float* __restrict__ sky, uint32_t sky_w, uint32_t sky_h, const float (&miss)[3], f3 rd) {
    if (sky == nullptr) return mk(miss[0], miss[1], miss[2]);                      // renderer.rs:61
    const f3 dir = normalized(rd);                                                  // :41
    const float theta = acosf(dir.y);                                               // :42
    const float phi = atan2f(dir.z, dir.x) + PI_F;                                  // :43
    const float u = phi / (2.0f * PI_F);                                            // :44
    const float v = theta / PI_F;                                                   // :45
    const uint32_t xp = as_u32_sat(fmaxf(u * (float)(sky_w - 1u), 0.0f));           // :47  (f32::max ignores NaN, `as u32` saturates)
    const uint32_t yp = as_u32_sat(fmaxf(v * (float)(sky_h - 1u), 0.0f));           // :48
    const size_t o = 3 * ((size_t)min(yp, sky_h - 1u) * sky_w + min(xp, sky_w - 1u));   // :50-53
    return mk(sky[o], sky[o + 1], sky[o + 2]);
}

// camera.rs:33-42 + ray.rs:12-17
DI f3 camera_raw(const DevCamera& cam, float u, float v) {                         // the direction before its two normalisations
    float ndc_x = 2.0f * u - 1.0f;
    float ndc_y = 1.0f - 2.0f * v;
    f3 right = mk(cam.right[0], cam.right[1], cam.right[2]), up = mk(cam.true_up[0], cam.true_up[1], cam.true_up[2]);
    f3 offset = right * (ndc_x * cam.half_width) + up * (ndc_y * cam.half_height);
    return mk(cam.forward[0], cam.forward[1], cam.forward[2]) + offset;
}
DI void camera_ray(const DevCamera& cam, float u, float v, f3& ro, f3& rd) {
    ro = mk(cam.position[0], cam.position[1], cam.position[2]);
    rd = ray_direction(camera_raw(cam, u, v));
}

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
//    / guided_div, clamped to [BATCH_MIN, BATCH_MAX] -- long runs while there is plenty of work (few atomics,
//    coherent primary rays), short runs at the end so that all waves drain together.  This matters when one
//    image is split across 8 GPUs and a launch lasts only a few ms.
DI uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & (WORK_SHARDS - 1u); }
struct WorkCursor {
    uint32_t next = 0, end = 0; bool no_more = false;
    uint32_t shard = 0, dry = 0, seen = 0;      // current shard, consecutive dry shards, last counter value seen in it
    DI void init() { shard = xcc_id(); }
    DI bool exhausted() const { return no_more && next == end; }
    // Lanes with want == true get a sample index (returns true and sets sidx); others / surplus stay idle.
    DI bool deal(const RenderParams& P, bool want, uint32_t lane, uint32_t& sidx) {
        const uint64_t idle = __ballot(want);
        if (idle == 0ull) return false;
        while (next == end && !no_more) {
            const uint32_t base = shard * P.shard_samples;
            const uint32_t len = min(P.shard_samples, P.band_samples > base ? P.band_samples - base : 0u);
            const uint32_t rem = len > seen ? len - seen : 0u;
            const uint32_t size = min(max(rem / P.guided_div, BATCH_MIN), BATCH_MAX);
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(P.batch_counter + shard * WORK_SHARD_STRIDE, size);
            start = __builtin_amdgcn_readfirstlane(start);
            if (start >= len) {                              // shard is dry: steal from the next one
                shard = (shard + 1u) & (WORK_SHARDS - 1u); seen = 0;
                if (++dry == WORK_SHARDS) no_more = true;
            } else { next = base + start; end = base + min(start + size, len); seen = start + size; dry = 0; }
        }
        const uint32_t take = min((uint32_t)__popcll(idle), end - next);
        bool got = false;
        if (take != 0u) {
            const uint32_t rank = mbcnt64(idle);
            if (want && rank < take) { sidx = next + rank; got = true; }
            next += take;
        }
        return got;
    }
};

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
template <bool WIDE = false>
DI f3 unit_ball_cooperative(bool diffuse, const RngCtr& rng, uint32_t lane) {
    f3 p = mk(u32_to_range11(rng.b0[1]), u32_to_range11(rng.b0[2]), u32_to_range11(rng.b0[3]));   // try 0
    bool need = diffuse && !(len2(p) < 1.0f);
    uint32_t jbase = 1;
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
        const uint32_t ok0 = (uint32_t)lane_shfl((int)rng.k0, olane), ok1 = (uint32_t)lane_shfl((int)rng.k1, olane);
        const uint32_t ox = (uint32_t)lane_shfl((int)rng.x, olane), os = (uint32_t)lane_shfl((int)rng.s, olane);
        const uint32_t oray = (uint32_t)lane_shfl((int)rng.ray, olane), oj = (uint32_t)lane_shfl((int)jbase, olane);
        uint32_t w[4];
        philox4x32_10<WIDE>(ok0, ok1, ox, os, oray, oj + (lane & ((1u << lg) - 1u)), w);
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
// deal fresh samples to idle lanes -> ONE Philox call for all lanes -> camera ray (fresh) or BSDF (continuing).
// Must be called by the whole wave in uniform control flow (it ballots): lanes that are busy elsewhere
// pass live = false and can_take = false and are left untouched.
// Returns false when no lane is live afterwards and no work is left to deal.
// DEFAULTS: give the per-lane temporaries default values.  The lockstep kernels run without (every value is read only on the
// path that wrote it, and the defaults cost ~30 v_mov per iteration: cornell -1.5 %); the register allocation of the
// state-machine / pool / wavefront kernels is better WITH them (wavefront: 38 spilled registers with, 120 without).
template <bool SIMPLE, bool DEFAULTS = true>
DI bool shade_and_regenerate(const RenderParams& P, WorkCursor& wc, uint32_t lane, bool& live, bool can_take, bool hit, const Hit& h,
                             PathState& ps, uint32_t& n_paths, uint32_t& n_rays, Prof& prof) {
    struct Rad { float x, y, z; };                                                        // 12 bytes per path: global_store_dwordx3
    Rad* __restrict__ radiance = reinterpret_cast<Rad*>(P.radiance);
    float4 q0;                                                                            // first 16 bytes of the hit material; read by lanes that loaded it
    if (DEFAULTS) q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        f3 term = mk(0.f, 0.f, 0.f); bool fin = false;
        if (!hit) { term = miss_colour(P.sky, P.sky_w, P.sky_h, P.miss, ps.rd); fin = true; }   // renderer.rs:38-63
        else {
            q0 = reinterpret_cast<const float4*>(P.mats + (h.mat_ff & 0x7FFFFFFFu))[0];
            const uint32_t kind = __float_as_uint(q0.x);
            if (kind == MI355RT_MAT_EMISSIVE) { term = mk(q0.y, q0.z, q0.w); fin = true; }   // scatter -> None, emitted = colour
            else if (kind == MI355RT_MAT_NULL) fin = true;
        }
        if (fin) { const f3 L = ps.thr * term; radiance[ps.sidx] = Rad{L.x, L.y, L.z}; live = false; }
    }
    prof.mark(2);
    bool fresh = false;
    if (wc.deal(P, can_take && !live, lane, ps.sidx)) { start_path(P, ps.sidx, ps.rng, ps.px, ps.py); fresh = true; live = true; ++n_paths; }
    if (__ballot(live) == 0ull) return !wc.exhausted();
    bool diffuse = false, scattered = false;
    f3 raw, atten, emitted; float side;              // written by the branch a lane takes below, read only on that lane's own path
    if constexpr (DEFAULTS) {
        raw = mk(0.f, 0.f, 1.f); atten = mk(0.f, 0.f, 0.f); emitted = mk(0.f, 0.f, 0.f); side = EPS;
        if (live) {
            if (!fresh) ps.rng.next_event();
            ps.rng.load_block0();
            if (fresh) {
                const float u = ((float)ps.px + ps.rng.jitter_u()) / (float)P.width;         // renderer.rs:96
                const float v = ((float)ps.py + ps.rng.jitter_v()) / (float)P.height;        // renderer.rs:97
                raw = camera_raw(P.cam, u, v);                                               // renderer.rs:99; normalised below with the scattered rays
                ps.ro = mk(P.cam.position[0], P.cam.position[1], P.cam.position[2]);
                ps.thr = mk(1.f, 1.f, 1.f); ps.ray_index = 0;
                if (P.max_depth == 0u) { radiance[ps.sidx] = Rad{0.f, 0.f, 0.f}; live = false; }   // depth == 0 -> BLACK
            } else {
                scattered = scatter_pre<SIMPLE>(P.mats, P.textures, q0, h, ps.rd, ps.rng, side, raw, atten, emitted, diffuse);
            }
        }
        prof.mark(5);
        const f3 ball = unit_ball_cooperative(diffuse, ps.rng, lane);                        // whole wave, uniform control flow
        if (live && !fresh) {
            if (scattered) {
                if (diffuse) raw = diffuse_finish(h, ball);
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
        if (!fresh) ps.rng.next_event();
        ps.rng.load_block0<true>();
        if (live) {
            if (fresh) {
                const float u = ((float)ps.px + ps.rng.jitter_u()) / (float)P.width;         // renderer.rs:96
                const float v = ((float)ps.py + ps.rng.jitter_v()) / (float)P.height;        // renderer.rs:97
                raw = camera_raw(P.cam, u, v);                                               // renderer.rs:99; normalised below with the scattered rays
                n_ro = mk(P.cam.position[0], P.cam.position[1], P.cam.position[2]);
                n_thr = mk(1.f, 1.f, 1.f); n_ri = 0;
                if (P.max_depth == 0u) { radiance[ps.sidx] = Rad{0.f, 0.f, 0.f}; live = false; }   // depth == 0 -> BLACK
            } else {
                scattered = scatter_pre<SIMPLE, true>(P.mats, P.textures, q0, h, ps.rd, ps.rng, side, raw, atten, emitted, diffuse);
            }
        }
        prof.mark(5);
        const f3 ball = unit_ball_cooperative<true>(diffuse, ps.rng, lane);                        // whole wave, uniform control flow
        if (live && !fresh) {
            if (scattered) {
                if (diffuse) raw = diffuse_finish(h, ball);
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
        ps.rd = ray_direction(raw);                                                          // fresh and scattered lanes together
        if (live) ++n_rays;
    }
    prof.mark(3);
    return true;
}

// Register budget per kernel, as waves per SIMD (A/B: tools/ab.py).  The lockstep kernel is VALU-issue bound
// and gains from 7 waves/SIMD (72 VGPRs) even with a few spills; the state-machine kernel keeps its hot BVH state in
// registers and loses when capped.
#ifndef MI355RT_TRAV_BIAS
#define MI355RT_TRAV_BIAS 2
#endif
#ifndef MI355RT_OCC_LOCKSTEP
#define MI355RT_OCC_LOCKSTEP 7                               // veach-mis: 4 -> 6.46 ms, 5 -> 6.03, 6 -> 5.79, 7 -> 5.73 (64 spp)
#endif
#if MI355RT_OCC_LOCKSTEP > 0
#define MI355RT_OCC_LS __attribute__((amdgpu_waves_per_eu(MI355RT_OCC_LOCKSTEP, MI355RT_OCC_LOCKSTEP)))
#else
#define MI355RT_OCC_LS
#endif
#ifndef MI355RT_OCC_LOCKSTEP_SIMPLE
#define MI355RT_OCC_LOCKSTEP_SIMPLE 7
#endif
#define MI355RT_OCC_SIMPLE __attribute__((amdgpu_waves_per_eu(MI355RT_OCC_LOCKSTEP_SIMPLE, MI355RT_OCC_LOCKSTEP_SIMPLE)))
#ifndef MI355RT_OCC_SM
#define MI355RT_OCC_SM 4
#endif
#if MI355RT_OCC_SM > 0
#define MI355RT_OCC_SMK __attribute__((amdgpu_waves_per_eu(MI355RT_OCC_SM, MI355RT_OCC_SM)))
#else
#define MI355RT_OCC_SMK
#endif

// ===================================================================================================
// k_render_ctr<HAS_MESH> -- persistent, path-regenerating wave64 path tracer, lockstep form: every live lane
// traces one full ray per loop iteration.  Used for scenes whose top level has no mesh (cornell, veach-mis):
// all lanes walk the same primitive list, so the iteration is divergence-free up to the hit tests.
// ===================================================================================================
template <bool HAS_MESH, bool SIMPLE>
DI void render_ctr_lockstep(const RenderParams& P) {
    cprim_t prims = (cprim_t)(P.prims);
    const uint32_t lane = threadIdx.x & 63u;
    WorkCursor wc; wc.init();
    PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
    ps.rng.k0 = ps.rng.k1 = ps.rng.x = ps.rng.s = ps.rng.ray = 0; ps.rng.b0[0] = ps.rng.b0[1] = ps.rng.b0[2] = ps.rng.b0[3] = 0;
    bool live = false;
    uint32_t n_paths = 0, n_rays = 0;
    Prof prof; prof.begin();
#ifdef MI355RT_STAMPS
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_dry = 0ull; uint32_t drain_iters = 0, live_at_dry = 0;
#endif
    for (;;) {
        Hit h; bool hit = false;                           // h is read only where `hit` says it was written: no default values to copy around
        if (live) hit = hit_scene<HAS_MESH, SIMPLE>(prims, P.n_prims, P.nodes, P.tris, ps.ro, ps.rd, h);     // renderer.rs:24
        prof.mark(1);
        if (!shade_and_regenerate<SIMPLE, false>(P, wc, lane, live, true, hit, h, ps, n_paths, n_rays, prof)) break;
        prof.mark(4);
#ifdef MI355RT_STAMPS
        if (wc.exhausted()) {                              // all work dealt: from here on the wave only drains its own paths
            if (t_dry == 0ull) { t_dry = __builtin_amdgcn_s_memrealtime(); live_at_dry = (uint32_t)__popcll(__ballot(live)); }
            ++drain_iters;
        }
#endif
    }
#ifdef MI355RT_STAMPS
    if (lane == 0 && P.stats) for (int i = 0; i < 6; ++i) atomicAdd(&P.stats[2 + i], prof.acc[i]);
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
//   k_render_ctr_nomesh  any materials, no mesh in the list            (veach-mis)                7 waves/SIMD
//   k_render_ctr_simple  Lambertian/Emissive/Null only, no mesh        (cornell: -3 % vs nomesh)   7 waves/SIMD
//   k_render_ctr_mesh    lockstep with the per-lane BVH walk inlined   (A/B reference for the state machine)
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_LS k_render_ctr_nomesh(const RenderParams P) { render_ctr_lockstep<false, false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS) MI355RT_OCC_SIMPLE k_render_ctr_simple(const RenderParams P) { render_ctr_lockstep<false, true>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) k_render_ctr_mesh(const RenderParams P) { render_ctr_lockstep<true, false>(P); }

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
                    mesh_step<FIXED_AABB, true>(n4, lds, lds_count, EPS, mt);
#pragma unroll
                    for (int u = 1; u < MI355RT_TRAV_UNROLL; ++u)
                        if (mt.leaf_b == 0u && mt.node != NODE_END) mesh_step<FIXED_AABB, true>(n4, lds, lds_count, EPS, mt);
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
                                    mesh_step<FIXED_AABB, true>(n4, lds, lds_count, EPS, mt);
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
        shade_and_regenerate<false>(P, wc, lane, live, part, any_hit, h, ps, n_paths, n_rays, prof);
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
constexpr uint32_t POOL_SPIN_LIMIT = 1u << 22;              // watchdog: polls without progress before a wave gives up (seconds)
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
        uint32_t spins = 0;
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
                        if (has[w] && m[w].leaf_b == 0u && m[w].node != NODE_END) mesh_step<FIXED_AABB, true>(n4, lds, lds_count, EPS, m[w]);
                }
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (has[w] && m[w].leaf_b != 0u) mesh_leaf(t4, EPS, m[w]);
                    if (has[w] && m[w].leaf_b == 0u && m[w].node == NODE_END) {          // walk over: hand the result back
                        results[2u * slot[w]] = __float_as_uint(m[w].best_t); results[2u * slot[w] + 1u] = m[w].best_tri;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(&flags[slot[w]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        has[w] = false;
                    }
                }
            } else {
                if (__hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;     // no producer left: nothing can be outstanding
                if (__hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > POOL_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctrl[3], 1u); break; }
            }
        }
        if (lane == 0 && P.stats && __hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) atomicAdd(&P.stats[15], 1ull);
        return;
    }

    // ------------------------------------------------ producer ------------------------------------------------
    const uint32_t my_slot = (wave - n_walkers) * 64u + lane;
    WorkCursor wc; wc.init();
    PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
    ps.rng.k0 = ps.rng.k1 = ps.rng.x = ps.rng.s = ps.rng.ray = 0; ps.rng.b0[0] = ps.rng.b0[1] = ps.rng.b0[2] = ps.rng.b0[3] = 0;
    uint32_t state = ST_IDLE, cursor = 0, spins = 0;
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
            if (++spins > POOL_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctrl[3], 1u); failed = true; break; }
            if (spins < P.pool_patience || nP + nS + nI == 0u) continue;         // waited long enough: run what is there
        }
        spins = 0;

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
                                    mesh_step<FIXED_AABB, true>(n4, lds, lds_count, EPS, mt);
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
        shade_and_regenerate<false>(P, wc, lane, live, part, any_hit, h, ps, n_paths, n_rays, prof);
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
        if (failed) atomicAdd(&P.stats[15], 1ull);
    }
}
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_pool(const RenderParams P) { render_ctr_pool<false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_SM) MI355RT_OCC_SMK k_render_ctr_pool_fixaabb(const RenderParams P) { render_ctr_pool<true>(P); }

// ===================================================================================================
// k_render_ctr_wf -- the path tracer as a WAVEFRONT inside one workgroup: path state lives in LDS, stages are queues.
//
// The state machine and the pool kernel keep a path in the registers of ONE lane for its whole life, so every pass of every
// stage runs with whatever lanes of that wave happen to be in that stage (measured: 0.40 of the lanes on semesterbild).  Here
// the CU's workgroup owns WF_PATHS path slots in LDS (20 dwords each) and queues of slot numbers -- FREE, WALK (a BVH walk in
// progress), TOP1 (a ray whose walk is back), SHADE x 4 material classes.  Every wave loops: look at the queue lengths, choose a
// stage, pop up to 64 of its slots, load what that stage needs, run the stage with (nearly) all lanes busy, store what changed,
// push each slot to the queue of its next stage.  (A new ray has no queue of its own: the SHADE pass that generates it walks the
// head of the list for it right away.)  A path therefore migrates between waves;
// per path the arithmetic is exactly that of the other kernels (same device functions, same inputs, same order), so images
// are bit-identical.  Regeneration stays in SHADE: a finished path's slot is refilled from the wave's own work cursor in the
// same pass, and SHADE passes top themselves up from the FREE queue.
// Queues: one ring of 1 024 u32 per stage (> WF_PATHS, a slot is in at most one queue), `tail` reserved by ds_add, `head`
// advanced by ds_cmpst so that a pop never takes more than is there; an entry is written after its ticket is reserved, so a
// popper may have to wait a few cycles for it (bounded spin) and writes EMPTY back; a pusher whose entry is still occupied (the
// popper of the previous ring revolution has reserved it but not read it yet) waits for that popper, so no slot number is ever lost.
// No barrier after start-up.  A wave leaves when its work cursor is exhausted and no path is alive in the workgroup.
// ===================================================================================================
// Two workgroups of 12 waves per CU (24 waves = 6 per SIMD at 80 VGPRs), 832 slots each: the passes begin with a chain of
// dependent LDS round trips (pop, ring entry, slot) and the walk reads its nodes from L1/L2, so waves to switch to are worth more
// than registers.  Measured (semesterbild / teapot, 800x600x64, ms): 1 x 16 waves, 1 728 slots 11.60 / 7.42;  2 x 12 waves,
// 832 slots each 10.65 / 6.65;  3 x 8 waves, 512 each 11.33 / 6.86;  2 x 14 at 72 VGPRs 14.8 / 9.8 and 2 x 16 at 64 VGPRs
// 15.3 / 9.0 (spills);  2 x 10 at 96 VGPRs 15.2 / 9.7;  1 x 16 waves with 960 fat slots (36 dwords) 11.6 / 7.3.
#ifndef MI355RT_WF_PATHS
#define MI355RT_WF_PATHS 832                                // what fits beside seven rings (768 beside the eight there were: semesterbild +2.4 %, teapot +1.6 %)
#endif
#ifndef MI355RT_WF_RING
#define MI355RT_WF_RING 1024
#endif
constexpr uint32_t WF_PATHS = MI355RT_WF_PATHS, WF_SLOT_WORDS = 20, WF_RING = MI355RT_WF_RING, WF_QUEUES = 7, WF_CTRL_WORDS = 32;
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
constexpr uint32_t WF_LDS_WORDS = WF_CTRL_WORDS + WF_QUEUES * WF_RING / 2u + WF_PATHS * WF_SLOT_WORDS;
static_assert(WF_LDS_WORDS * 4u <= 163840u / 2u, "wavefront kernel LDS budget: two workgroups per CU");
static_assert(WF_PATHS < WF_RING && WF_PATHS < WF_EMPTY, "a ring holds every slot number");
// Slot layout, 5 x 16 bytes (the less a path carries, the more paths fit, and the fill of every pass follows from their number:
// 960 slots of 36 dwords ran SHADE at 37 of 64 lanes):  q0 ro.xyz thr.x | q1 rd.xyz thr.y | q2 thr.z sidx ray_index cursor(+WALK_DONE) |
// q3 cand.t idx aux aux2 | q4 walk node, best_t, best_tri, -.   Recomputed instead of stored: the RNG key (from sidx), the walk's
// object-space ray and 1/d (mesh_setup per WALK pass: +3 % instructions), |w2o d| for the (sic) t_world.

struct WfQueues {
    uint32_t* ctrl;        // [q] head, [8 + q] tail, [16] live paths, [17] error
    uint16_t* rings;       // WF_QUEUES x WF_RING slot numbers
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
            volatile uint16_t* e = rings + q * WF_RING + ((h + lane - lane0) & (WF_RING - 1u));
            uint32_t v = WF_EMPTY, spins = 0;
            for (;;) {                                                   // the pusher reserved this ticket and is about to write it
                v = *e;
                if (v != WF_EMPTY) break;
                if (++spins > (1u << 20)) { failed = true; break; }
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
            volatile uint16_t* e = rings + q * WF_RING + ((base + mbcnt64(m)) & (WF_RING - 1u));
            // The entry of ticket T is free once the popper of ticket T - WF_RING has read it and written EMPTY back.  That popper
            // exists (a ring holds more entries than there are slots, so ticket T - WF_RING was popped before T could be reserved);
            // if it has been held up between reserving and reading, wait for it instead of overwriting its entry.
            uint32_t spins = 0;
            while (*e != WF_EMPTY) { if (++spins > (1u << 20)) { failed = true; break; } }
            *e = (uint16_t)id;
        }
    }
    // Every lane with `pred` pushes its slot to ITS queue `q` (lanes may name different queues): one reservation per queue present.
    DI void push_each(bool pred, uint32_t q, uint32_t id, uint32_t lane, bool& failed) const {
        uint64_t rem = __ballot(pred);
        while (rem != 0ull) {
            const uint32_t qq = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)__builtin_ctzll(rem));
            const bool mine = pred && q == qq;
            push(qq, mine, id, lane, failed);
            rem &= ~__ballot(mine);
        }
    }
    DI uint32_t count(uint32_t q) const {
        return __hip_atomic_load(&ctrl[8u + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - __hip_atomic_load(&ctrl[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

template <bool FIXED_AABB>
DI void render_ctr_wavefront(const RenderParams& P) {
    __shared__ __attribute__((aligned(16))) uint32_t s_wf[WF_LDS_WORDS];
    WfQueues Q; Q.ctrl = s_wf; Q.rings = reinterpret_cast<uint16_t*>(s_wf + WF_CTRL_WORDS);
    uint32_t* const slots = s_wf + WF_CTRL_WORDS + WF_QUEUES * WF_RING / 2u;
    cprim_t prims = (cprim_t)(P.prims);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(P.nodes);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(P.tris);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < WF_QUEUES * WF_RING; i += blockDim.x) Q.rings[i] = (uint16_t)((i < WF_PATHS) ? i : WF_EMPTY);   // FREE holds every slot
    if (threadIdx.x < WF_CTRL_WORDS) Q.ctrl[threadIdx.x] = (threadIdx.x == 8u + WQ_FREE) ? WF_PATHS : 0u;
    __syncthreads();

    WorkCursor wc; wc.init();
    uint32_t n_paths = 0, n_rays = 0, spins = 0, naps = 0;
    Prof prof; prof.begin();
    bool failed = false;
#ifdef MI355RT_STAMPS
    unsigned long long w_exec[4] = {0, 0, 0, 0}, w_lanes[4] = {0, 0, 0, 0};    // passes and slots per pass: 0 WALK, 1 TOP1, 2 (unused: there was a TOP0 stage), 3 SHADE (+ free fill)
#define MI355RT_WFCOUNT(i, n) do { w_exec[i] += 1; w_lanes[i] += (n); } while (0)
#else
#define MI355RT_WFCOUNT(i, n) do {} while (0)
#endif
#ifndef MI355RT_WF_ROUNDS
#define MI355RT_WF_ROUNDS 3                                 // rounds x steps (ms, semesterbild / teapot 64 spp): 1x8 11.8 / 7.5, 2x8 10.7 / 6.6, 3x8 10.4 / 6.4, 4x8 10.4 / 6.3, 8x8 10.7 / 6.6, 3x12 10.7 / 6.4
#endif
#ifndef MI355RT_WF_STEPS
#define MI355RT_WF_STEPS 8
#endif
    // TOP: hittable.rs:45-58 from the slot's cursor; a mesh whose root box is hit sends the ray to WALK; at the end of the list the
    // slot is routed by the material class of its hit, so that SHADE passes are homogeneous.  Run by SHADE passes on the rays they
    // have just generated (still in registers) and by TOP1 passes on the slots whose walk is back.
    auto run_top = [&](const bool have, const f3 ro, const f3 rd, Cand c, uint32_t cursor, bool walk_done, uint32_t* sl, const uint32_t id) {
        bool to_walk = false;
        for (uint32_t i = 0; i < P.n_prims; ++i) {
            const bool mine = have && !to_walk && cursor == i;
            if (__ballot(mine) == 0ull) continue;
            cprim_t pr = prims + i;
            if (mine) {
                bool advance = true;
                switch (pr->kind) {                                       // wave-uniform: scalar branch
                    case MI355RT_PRIM_SPHERE: hit_sphere(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_PLANE:  hit_plane(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_QUAD:   hit_quad(pr, i, ro, rd, EPS, c); break;
                    case MI355RT_PRIM_CUBE:   hit_cube(pr, i, ro, rd, EPS, c); break;
                    default:
                        if (!walk_done) {
                            MeshTrav mt; mesh_setup(pr, ro, rd, c.t, mt);
                            const uint32_t root = mt.node;
                            mesh_step<FIXED_AABB, false>(n4, nullptr, 0u, EPS, mt);       // the root box, here: most rays miss it
                            if (mt.leaf_b == 0u && mt.node == NODE_END) { /* missed: no hit in this mesh */ }
                            else {
#ifndef MI355RT_WF_INLINE_MIN
#define MI355RT_WF_INLINE_MIN 32                            // lanes inside the root box for the first steps of the walk to run right here (65: never)
#endif
#ifndef MI355RT_WF_INLINE_STEPS
#define MI355RT_WF_INLINE_STEPS 8
#endif
                                bool parked = false;
                                if (MI355RT_WF_INLINE_MIN <= 64 && (uint32_t)__popcll(__ballot(true)) >= (uint32_t)MI355RT_WF_INLINE_MIN) {
#pragma unroll 1
                                    for (int u = 0; u < MI355RT_WF_INLINE_STEPS; ++u) {
                                        if (mt.leaf_b != 0u) mesh_leaf(t4, EPS, mt);
                                        if (mt.node == NODE_END) break;
                                        mesh_step<FIXED_AABB, false>(n4, nullptr, 0u, EPS, mt);
                                    }
                                    if (mt.leaf_b != 0u) mesh_leaf(t4, EPS, mt);
                                    if (mt.node == NODE_END) { mesh_accept(i, mt, rd, EPS, c); parked = true; }      // the whole walk fitted: the list goes on
                                    else reinterpret_cast<float4*>(sl)[4] = make_float4(__uint_as_float(mt.node), mt.best_t, __uint_as_float(mt.best_tri), 0.f);
                                } else {
                                    reinterpret_cast<float4*>(sl)[4] = make_float4(__uint_as_float(root), c.t, __uint_as_float(0xFFFFFFFFu), 0.f);
                                }
                                if (!parked) { to_walk = true; advance = false; }
                            }
                        } else {
                            const float4 w = reinterpret_cast<const float4*>(sl)[4];
                            MeshTrav mt; mt.best_t = w.y; mt.best_tri = __float_as_uint(w.z); mt.len_raw = len(xform_w2o_dir(pr, rd));   // mesh_object.rs:288, again
                            mesh_accept(i, mt, rd, EPS, c); walk_done = false;
                        }
                        break;
                }
                if (advance) ++cursor;
            }
        }
        if (have) {
            reinterpret_cast<float4*>(sl)[3] = make_float4(c.t, __uint_as_float(c.idx), c.aux, __uint_as_float(c.aux2));
            sl[11] = cursor;
        }
        uint32_t cls = 0u;
        if (have && !to_walk && c.idx != CAND_NONE) cls = shade_class(P.mats[P.prims[c.idx].material].kind);
        Q.push_each(have, to_walk ? (uint32_t)WQ_WALK : WQ_SHADE + cls, id, lane, failed);
    };
    for (;;) {
        if (__ballot(failed) != 0ull) { if (lane == 0) atomicOr(&Q.ctrl[17], 1u); break; }
        if (__hip_atomic_load(&Q.ctrl[17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) { failed = true; break; }
        const uint32_t cT1 = Q.count(WQ_TOP1), cW = Q.count(WQ_WALK);
        const uint32_t cS0 = Q.count(WQ_SHADE), cS1 = Q.count(WQ_SHADE + 1u), cS2 = Q.count(WQ_SHADE + 2u), cS3 = Q.count(WQ_SHADE + 3u);
        const uint32_t cS = cS0 + cS1 + cS2 + cS3;
        const uint32_t cF = wc.exhausted() ? 0u : Q.count(WQ_FREE);
        // A pass costs its instructions whatever its fill, and the stages differ in price (SHADE ~1 800 instructions, WALK ~750,
        // TOP0 ~700, TOP1 ~400): run the stage whose pass WASTES the fewest lane-instructions, price x empty lanes.  A full queue
        // wastes nothing; of two thin ones the cheap stage runs and the expensive one keeps filling (measured with "fullest
        // first": SHADE ran at 39 of 64 lanes while TOP0 ran at 61).  Ties go to the later stage.
#ifndef MI355RT_WF_POLICY
#define MI355RT_WF_POLICY 1
#endif
        uint32_t stage = WQ_NONE, best = 0;
        {
            uint32_t waste = 0xFFFFFFFFu;
            auto consider = [&](uint32_t q, uint32_t n, uint32_t price) {
                if (n == 0u) return;
                const uint32_t w = price * (64u - min(n, 64u));
                if (w <= waste) { waste = w; stage = q; best = n; }
            };
            consider(WQ_WALK, cW, 11u); consider(WQ_TOP1, cT1, 4u);
#ifndef MI355RT_WF_T0PRICE
#define MI355RT_WF_T0PRICE 7
#endif
            constexpr uint32_t T0 = MI355RT_WF_T0PRICE;                  // a SHADE pass goes on with the head of the list for the rays it generates
            consider(WQ_SHADE + 3u, cS3, 5u + T0); consider(WQ_SHADE + 2u, cS2, 10u + T0); consider(WQ_SHADE + 1u, cS1, 8u + T0);
            consider(WQ_SHADE, cS0 + cF, 5u + T0);                      // terminal class: free slots ride along (both only regenerate)
        }
        if (stage == WQ_NONE) {
            if (wc.exhausted() && __hip_atomic_load(&Q.ctrl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;   // nothing alive, nothing to start
            __builtin_amdgcn_s_sleep(2);
            if (++spins > POOL_SPIN_LIMIT) { failed = true; }
            continue;
        }
        // There are fewer path slots (960) than lanes in the workgroup (1 024), so with every wave busy the queues stay short and
        // the passes run under-filled (measured: SHADE at 37 of 64).  A pass costs its instructions whatever its fill, and the
        // kernel is issue-bound: while slots are still in flight in OTHER waves (they will land in a queue soon) a wave whose best
        // queue is short sleeps instead of running a thin pass.  Bounded: after WF_PATIENCE naps it runs what there is.
#ifndef MI355RT_WF_MINFILL
#define MI355RT_WF_MINFILL 48
#endif
#ifndef MI355RT_WF_PATIENCE
#define MI355RT_WF_PATIENCE 0                              // measured: any napping loses (semesterbild 64 spp 11.6 -> 12.4..13.0 ms): thin passes still hide latency
#endif
        if (best < MI355RT_WF_MINFILL && naps < MI355RT_WF_PATIENCE) {
            const uint32_t alive = __hip_atomic_load(&Q.ctrl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (alive > cS + cT1 + cW) { ++naps; __builtin_amdgcn_s_sleep(4); continue; }
        }
        naps = 0;
#ifndef MI355RT_WF_KEEP
#define MI355RT_WF_KEEP 3                                   // a pop must still find 3/4 of what the decision saw
#endif
        auto keep = [](uint32_t seen) { return MI355RT_WF_KEEP == 0 ? 0u : max(1u, seen * MI355RT_WF_KEEP / 4u); };
        spins = 0;
        uint32_t id = 0;

        if (stage >= WQ_SHADE) {
            // ---- SHADE (one material class) + regeneration; a terminal-class pass is topped up with free slots (which only regenerate) ----
            const uint32_t seen = stage == WQ_SHADE ? cS0 : stage == WQ_SHADE + 1u ? cS1 : stage == WQ_SHADE + 2u ? cS2 : cS3;
            const uint32_t n = Q.pop(stage, 64u, seen == 0u ? 0u : keep(min(seen, 64u)), lane, 0u, id, failed);
            uint32_t nf = 0;
            if (stage == WQ_SHADE && n < 64u && !wc.exhausted()) nf = Q.pop(WQ_FREE, 64u - n, 0u, lane, n, id, failed);
            if (n + nf == 0u) continue;
            const bool have = lane < n, fill = lane >= n && lane < n + nf;
            MI355RT_WFCOUNT(3, n + nf);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            PathState ps; ps.ro = mk(0, 0, 0); ps.rd = mk(0, 0, 1); ps.thr = mk(1, 1, 1); ps.sidx = 0; ps.ray_index = 0; ps.px = ps.py = 0;
            ps.rng.k0 = ps.rng.k1 = ps.rng.x = ps.rng.s = ps.rng.ray = 0; ps.rng.b0[0] = ps.rng.b0[1] = ps.rng.b0[2] = ps.rng.b0[3] = 0;
            Cand c; cand_reset(c);
            if (have) {
                const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1];
                const float4 d = reinterpret_cast<const float4*>(sl)[2], g = reinterpret_cast<const float4*>(sl)[3];
                ps.ro = mk(a.x, a.y, a.z); ps.rd = mk(b.x, b.y, b.z); ps.thr = mk(a.w, b.w, d.x);
                ps.sidx = __float_as_uint(d.y); ps.ray_index = __float_as_uint(d.z);
                start_path(P, ps.sidx, ps.rng, ps.px, ps.py);                   // the RNG key is a function of the sample index
                ps.rng.ray = ps.ray_index;
                c.t = g.x; c.idx = __float_as_uint(g.y); c.aux = g.z; c.aux2 = __float_as_uint(g.w);
            }
            bool live = have;
            const bool any_hit = have && c.idx != CAND_NONE;
            Hit h; h.t = 0.f; h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.mat_ff = 0;
            if (any_hit) finish_hit<true>(P.prims, P.tris, c, ps.ro, ps.rd, h);
            shade_and_regenerate<false>(P, wc, lane, live, have || fill, any_hit, h, ps, n_paths, n_rays, prof);
            if (live) {                                                          // a ray to trace: continuing or freshly generated
                reinterpret_cast<float4*>(sl)[0] = make_float4(ps.ro.x, ps.ro.y, ps.ro.z, ps.thr.x);
                reinterpret_cast<float4*>(sl)[1] = make_float4(ps.rd.x, ps.rd.y, ps.rd.z, ps.thr.y);
                reinterpret_cast<float4*>(sl)[2] = make_float4(ps.thr.z, __uint_as_float(ps.sidx), __uint_as_float(ps.ray_index), __uint_as_float(0u));
                reinterpret_cast<float4*>(sl)[3] = make_float4(__builtin_inff(), __uint_as_float(CAND_NONE), 0.f, 0.f);
            }
            const int born = (int)__popcll(__ballot(fill && live)), died = (int)__popcll(__ballot(have && !live));
            if (lane == 0 && born != died) atomicAdd(&Q.ctrl[16], (uint32_t)(born - died));
            Q.push(WQ_FREE, (have || fill) && !live, id, lane, failed);
            // Every ray SHADE produces -- continuing or freshly generated -- starts at the head of the list, so the pass goes straight
            // on with TOP for its live lanes: as homogeneous as a pass over a queue of such rays and at least as full, minus one queue
            // round trip per ray (there was a TOP0 queue: semesterbild 9.73 -> 9.08 ms, teapot 6.49 -> 6.17 ms at 64 spp without it).
            prof.mark(4);
            {   Cand c0; cand_reset(c0);
                run_top((have || fill) && live, ps.ro, ps.rd, c0, 0u, false, sl, id); }
            prof.mark(1);
            continue;
        }

        if (stage == WQ_WALK) {
            // ---- WALK: two rounds of eight box tests + the pending leaves; unfinished walks go round again ----
            const uint32_t n = Q.pop(WQ_WALK, 64u, keep(min(cW, 64u)), lane, 0u, id, failed);
            if (n == 0u) continue;
            const bool have = lane < n;
            MI355RT_WFCOUNT(0, n);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            MeshTrav m; m.ro = mk(0, 0, 0); m.rd = mk(0, 0, 1); m.ix = m.iy = m.iz = 0.f; m.len_raw = 0.f; m.node = NODE_END; m.best_t = 0.f;
            m.best_tri = 0xFFFFFFFFu; m.leaf_a = m.leaf_b = 0;
            uint32_t cursor_word = 0;
            if (have) {
                const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1], w = reinterpret_cast<const float4*>(sl)[4];
                cursor_word = sl[11];
                const DevPrim* __restrict__ pr = P.prims + (cursor_word & ~WF_WALK_DONE);                 // lanes may be in different meshes
                mesh_setup(pr, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), 0.f, m);                            // the object-space ray, as TOP computed it
                m.node = __float_as_uint(w.x); m.best_t = w.y; m.best_tri = __float_as_uint(w.z);
            }
            for (int round = 0; round < MI355RT_WF_ROUNDS; ++round) {
                if (__ballot(have && (m.leaf_b != 0u || m.node != NODE_END)) == 0ull) break;
#pragma unroll
                for (int u = 0; u < MI355RT_WF_STEPS; ++u)
                    if (have && m.leaf_b == 0u && m.node != NODE_END) mesh_step<FIXED_AABB, false>(n4, nullptr, 0u, EPS, m);
                if (have && m.leaf_b != 0u) mesh_leaf(t4, EPS, m);
            }
            const bool done = have && m.leaf_b == 0u && m.node == NODE_END;        // (a pass always ends with its pending leaves tested: leaf_b == 0)
            if (have) {
                reinterpret_cast<float4*>(sl)[4] = make_float4(__uint_as_float(m.node), m.best_t, __uint_as_float(m.best_tri), 0.f);
                if (done) sl[11] = cursor_word | WF_WALK_DONE;
            }
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
            if (n == 0u) continue;
            const bool have = lane < n;
            MI355RT_WFCOUNT(1, n);
            uint32_t* sl = slots + WF_SLOT_WORDS * id;
            f3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
            Cand c; cand_reset(c);
            uint32_t cursor = 0xFFFFFFFFu; bool walk_done = false;
            if (have) {
                const float4 a = reinterpret_cast<const float4*>(sl)[0], b = reinterpret_cast<const float4*>(sl)[1], g = reinterpret_cast<const float4*>(sl)[3];
                ro = mk(a.x, a.y, a.z); rd = mk(b.x, b.y, b.z);
                c.t = g.x; c.idx = __float_as_uint(g.y); c.aux = g.z; c.aux2 = __float_as_uint(g.w);
                const uint32_t cw = sl[11];
                cursor = cw & ~WF_WALK_DONE; walk_done = (cw & WF_WALK_DONE) != 0u;
            }
            run_top(have, ro, rd, c, cursor, walk_done, sl, id);
            prof.mark(1);
        }
    }
    const uint32_t wp = wave_sum(n_paths), wr = wave_sum(n_rays);
    if (lane == 0 && P.stats) {
        atomicAdd(&P.stats[0], (unsigned long long)wp); atomicAdd(&P.stats[1], (unsigned long long)wr);
#ifdef MI355RT_STAMPS
        for (int i = 0; i < 6; ++i) atomicAdd(&P.stats[2 + i], prof.acc[i]);
        for (int i = 0; i < 4; ++i) { atomicAdd(&P.stats[8 + 2 * i], w_exec[i]); atomicAdd(&P.stats[9 + 2 * i], w_lanes[i]); }
#else
        if (failed) atomicAdd(&P.stats[15], 1ull);
#endif
    }
}
#ifndef MI355RT_OCC_WF
#define MI355RT_OCC_WF 6
#endif
#define MI355RT_OCC_WFK __attribute__((amdgpu_waves_per_eu(MI355RT_OCC_WF, MI355RT_OCC_WF)))
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf(const RenderParams P) { render_ctr_wavefront<false>(P); }
__global__ void __launch_bounds__(BLOCK_THREADS_WF) MI355RT_OCC_WFK k_render_ctr_wf_fixaabb(const RenderParams P) { render_ctr_wavefront<true>(P); }

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
    const size_t o = (size_t)P.band_pixel0 + p;
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
    RngCtr rng; rng.start(r.k0, r.k1, r.x, r.s); rng.ray = r.ray; rng.load_block0();
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
        case KERNEL_WAVEFRONT:       hipLaunchKernelGGL(k_render_ctr_wf, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_WAVEFRONT_FIXAABB: hipLaunchKernelGGL(k_render_ctr_wf_fixaabb, dim3(grid_blocks), dim3(BLOCK_THREADS_WF), 0, (hipStream_t)stream, p); break;
        case KERNEL_POOL:            hipLaunchKernelGGL(k_render_ctr_pool, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
        case KERNEL_POOL_FIXAABB:    hipLaunchKernelGGL(k_render_ctr_pool_fixaabb, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
        case KERNEL_STATE_MACHINE_FIXAABB: hipLaunchKernelGGL(k_render_ctr_sm_fixaabb, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
        default:                     hipLaunchKernelGGL(k_render_ctr_sm, dim3(grid_blocks), dim3(BLOCK_THREADS_SM), 0, (hipStream_t)stream, p); break;
    }
    return (int)hipGetLastError();
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
    const void* fn = variant == KERNEL_LOCKSTEP ? reinterpret_cast<const void*>(k_render_ctr_nomesh)
                   : variant == KERNEL_LOCKSTEP_MESH ? reinterpret_cast<const void*>(k_render_ctr_mesh)
                   : variant == KERNEL_LOCKSTEP_SIMPLE ? reinterpret_cast<const void*>(k_render_ctr_simple)
                   : variant == KERNEL_STATE_MACHINE_FIXAABB ? reinterpret_cast<const void*>(k_render_ctr_sm_fixaabb)
                   : variant == KERNEL_WAVEFRONT ? reinterpret_cast<const void*>(k_render_ctr_wf)
                   : variant == KERNEL_WAVEFRONT_FIXAABB ? reinterpret_cast<const void*>(k_render_ctr_wf_fixaabb)
                   : variant == KERNEL_POOL ? reinterpret_cast<const void*>(k_render_ctr_pool)
                   : variant == KERNEL_POOL_FIXAABB ? reinterpret_cast<const void*>(k_render_ctr_pool_fixaabb)
                                                       : reinterpret_cast<const void*>(k_render_ctr_sm);
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
