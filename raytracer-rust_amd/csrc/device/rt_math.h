// rt_math.h -- device prelude (DI, EPS, Prof) and the f32 vector helpers of vec3.rs / color.rs, operation for operation
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once
#include "rt_device.h"

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
    unsigned long long cls[10];     // lockstep kernels: per branch of the shading step {iterations in which any lane takes it, lanes that take it}:
                                   // 0 rough conductor, 1 Lambert-style bounce (Lambert, checker, texture, plastic), 2 metal / dielectric, 3 camera ray, 4 every iteration / live lanes
    DI void begin() { for (int i = 0; i < 6; ++i) acc[i] = 0; for (int i = 0; i < 10; ++i) cls[i] = 0; last = now(); }
    DI void classes(bool cont, bool fresh, uint32_t kind) {
        const unsigned long long m[5] = {__ballot(cont && (kind == 6u || kind == 7u)), __ballot(cont && (kind == 0u || kind == 1u || kind == 5u || kind == 9u)),
                                         __ballot(cont && (kind == 2u || kind == 3u)), __ballot(fresh), __ballot(cont || fresh)};
        for (int i = 0; i < 5; ++i) { cls[2 * i] += m[i] != 0ull; cls[2 * i + 1] += (unsigned long long)__popcll(m[i]); }
    }
    DI static unsigned long long now() {
        unsigned long long t; __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0); return t;
    }
    DI void mark(int i) { unsigned long long t = now(); acc[i] += t - last; last = t; }
};
#else
struct Prof { DI void begin() {} DI void mark(int) {} DI void classes(bool, bool, uint32_t) {} };
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
// 1.0f / x for 2^-126 <= |x| < 2^126, +-inf and NaN: v_rcp_f32, one Newton step, v_div_fixup_f32.  Equal to the compiler's correctly
// rounded division for EVERY such x -- compared exhaustively over all 2^32 bit patterns on the MI355X (tools/microbench/recip.hip,
// profiles/r03_microbench_reciprocal.txt: the only inputs where the two differ are denormals and |x| >= 2^126, whose reciprocal is a
// denormal) -- at about half the issue slots (7.5 instead of 14.8).  For callers that can bound their argument.
DI float recip_normal_range(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
// (the length is a square root: at most sqrt(FLT_MAX) = 1.8e19 < 2^126 or +inf / NaN, and at least EPS here)
// FASTN: with length_for_normalize() and recip_normal_range().  Same bits either way; the kernels of mesh-free lists gain 2-3 % from it
// in their shading step, the wavefront kernels with the BVH walk LOSE 0.4-1.5 % there (fewer instructions, one more spilled
// register) and gain only in mesh_setup (profiles/r03_ab_short_reciprocal.txt) -- so it is a template argument set site by site.
// a / b by the exact reciprocal and ONE correction: r = RN(1/b) (above), q0 = a*r, e = fma(-b, q0, a), q = fma(e, r, q0), v_div_fixup_f32 for
// infinities / NaN / zeros.  Equal to the compiler's correctly rounded division for all 2^23 x 2^23 pairs of significands (enumerated on the
// MI355X in 43 s: tools/microbench/div.hip, profiles/r03_microbench_division.txt); rounding commutes with scaling by powers of two while
// everything stays normal, so it IS a / b whenever 2^-25 <= |b| <= 2^25 and (a == 0 or 2^-100 <= |a| <= 2^100) -- 11 + 1.5 issue slots
// instead of 18.5.  For callers that can bound their arguments (or guard them with a ballot) and do not use the result elsewhere.
DI float div_bounded(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    { const float e = __builtin_fmaf(-b, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    float q = a * r;
    { const float e = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(e, r, q); }
    return __builtin_amdgcn_div_fixupf(q, b, a);
}

// div_bounded() with the reciprocal handed in (r must be RN(1/b), e.g. the host's 1.0f / b): for a divisor that is the same for the whole
// launch the reciprocal and its Newton step are not even hoisted into vector registers -- b and r stay scalar operands.
DI float div_by_rn(float a, float b, float r) {
    float q = a * r;
    { const float e = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(e, r, q); }
    return __builtin_amdgcn_div_fixupf(q, b, a);
}

// 1 / x, 1 / y, 1 / z for ANY arguments: the short form where every lane of the wave has all three in its range (zero counts as
// in range: v_div_fixup_f32 returns the infinity of the right sign), the compiler's division for the whole wave otherwise -- a
// wave-uniform branch, so the common case pays three short reciprocals and a range test (two 3-input min / max on the magnitudes).
// NaN components are ignored by the test and give NaN either way.  (The ballot covers the lanes that are active at the call.)
template <bool FASTR>
DI void recip3(float x, float y, float z, float& ix, float& iy, float& iz) {
    if constexpr (FASTR) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    const float mx = fmaxf(fmaxf(ax, ay), az);
    const float mn = fminf(fminf(ax == 0.0f ? 1.0f : ax, ay == 0.0f ? 1.0f : ay), az == 0.0f ? 1.0f : az);
    if (__ballot((mn < 0x1p-126f) || (mx >= 0x1p126f)) == 0ull) { ix = recip_normal_range(x); iy = recip_normal_range(y); iz = recip_normal_range(z); return; }
    }
    ix = 1.0f / x; iy = 1.0f / y; iz = 1.0f / z;
}

// The length normalized() needs: bit-equal to sqrtf(x) for every x >= 2^-100, +inf and NaN; for 0 <= x < 2^-100 (a length below
// 1e-15) it returns 0, which normalized() treats like the true value -- both are below its 1e-4 pass-through threshold.  v_rsq_f32 +
// one residual correction instead of the compiler's expansion (denormal scaling, v_sqrt_f32, both neighbours tested): 9 instead of 17
// issue slots.  The contract was checked for EVERY x >= +0 and every NaN on the MI355X (tools/microbench/sqrt.hip, k_contract:
// 0 violations; profiles/r03_microbench_sqrt.txt).  x is a sum of squares here: never negative, never -0.
DI float length_for_normalize(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    s = (x < 0x1p-100f) ? 0.0f : s;
    return (x == __builtin_inff()) ? x : s;
}
template <bool FASTN = false>
DI f3 normalized(f3 a) {                                                          // :37-44
    if constexpr (FASTN) { const float l = length_for_normalize(len2(a)); if (l < EPS) return a; return a * recip_normal_range(l); }
    const float l = len(a); if (l < EPS) return a; return a * (FASTN ? recip_normal_range(l) : 1.0f / l);
}
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


}  // namespace mi355rt
