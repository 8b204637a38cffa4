// rt_materials.h -- material.rs / tungsten/materials.rs: scatter_pre(), the shared ray tail, miss colour, camera ray
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once
#include "rt_intersect.h"
#include "rt_rng.h"

namespace mi355rt {

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
// MATS: bit k set = a material of kind k (MI355RT_MAT_*) may occur.  mi355rt_context_set_scene knows which kinds the scene holds
// and picks a kernel instantiated for a superset; the branches of the other kinds are compiled out.  They set the register
// peak: the metal's fuzz loop and the rough conductor together cost the wavefront kernel 49 of its 50 spilled registers
// (measured per branch), and the Lambert-only lockstep kernel (cornell) runs 7 waves per SIMD because of it.
// MATS_LAMBERT: only Lambertian (solid) / Emissive / Null -- every scattering material is the Lambert bounce, no dispatch at all.
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

enum : uint32_t { BALL_NONE = 0, BALL_DIFFUSE = 1, BALL_METAL = 2 };    // what a scatter event needs a random_in_unit_sphere point for
// TERMINAL_DONE: the caller has already finished the paths that hit an emitter or a null material (shade_and_regenerate classifies them before it deals
// fresh samples), so no lane arrives here with one: the two tests below -- and the mask bookkeeping the compiler builds around them -- are compiled out.
template <uint32_t MATS, bool WIDE = false, bool FASTN = false, bool TERMINAL_DONE = false, class Rng>
DI bool scatter_pre(const DevMat* __restrict__ mats, const DevTexture* __restrict__ texs, const float4 q0, const Hit& h, f3 rd_in, Rng& rng, float& side, f3& raw_d, f3& atten, f3& emitted, uint32_t& ball_use, float& fuzz_out) {
    const float4* __restrict__ m4 = reinterpret_cast<const float4*>(mats + (h.mat_ff & 0x7FFFFFFFu));
    const uint32_t kind = __float_as_uint(q0.x);
    const f3 albedo = mk(q0.y, q0.z, q0.w);
    const bool front_face = (h.mat_ff >> 31) != 0;
    emitted = mk(0.f, 0.f, 0.f);
    ball_use = BALL_NONE;
    side = EPS;                                                                    // every material but the dielectric leaves on the normal's side
    if constexpr (!TERMINAL_DONE) {
    if (kind == MI355RT_MAT_EMISSIVE) { emitted = albedo; return false; }         // material.rs:179-191
    if (kind == MI355RT_MAT_NULL) return false;                                   // material.rs:239-251
    }
    rng.begin_scatter();
    bool diffuse = false;                                                          // Lambert-style bounce shared by 3 materials
    atten = albedo;
    constexpr bool SIMPLE = (MATS & ~MATS_LAMBERT) == 0u;
#define MI_HAS(k) (((MATS >> (k)) & 1u) != 0u)
    if (SIMPLE || kind == MI355RT_MAT_LAMBERT_SOLID) {                             // material.rs:47-71
        diffuse = true;
    } else if (MI_HAS(MI355RT_MAT_LAMBERT_CHECKER) && kind == MI355RT_MAT_LAMBERT_CHECKER) {                              // tungsten/materials.rs:89-99
        const float4 q1 = m4[1];
        float inv_scale = q1.w;
        int32_t sum = (int32_t)((uint32_t)as_i32_sat(floorf(h.p.x * inv_scale)) + (uint32_t)as_i32_sat(floorf(h.p.y * inv_scale)) +
                                (uint32_t)as_i32_sat(floorf(h.p.z * inv_scale)));
        if ((sum & 1) != 0) atten = mk(q1.x, q1.y, q1.z);
        diffuse = true;
    } else if (MI_HAS(MI355RT_MAT_TEXTURE) && kind == MI355RT_MAT_TEXTURE) {                                      // tungsten/parser.rs:205-243
        const float4 q1 = m4[1];
        atten = albedo * texture_lookup(texs, __float_as_uint(m4[3].w), q1.w, h.n);
        diffuse = true;
    } else if (MI_HAS(MI355RT_MAT_PLASTIC) && kind == MI355RT_MAT_PLASTIC) {                                      // tungsten/materials.rs:29-65
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
    } else if (MI_HAS(MI355RT_MAT_METAL) && kind == MI355RT_MAT_METAL) {                                        // material.rs:87-110
        // The fuzz's random_in_unit_sphere (material.rs:97) is the same rejection loop as the Lambert bounce's, with the same
        // addressed draws (try j = block j words 1..3): it is left to the caller's wave-cooperative rejection and ball_finish()
        // adds it.  (As a per-lane loop with a Philox call in it, this branch set the register peak of every general kernel.)
        const float fuzz = m4[1].w;
        raw_d = mat_reflect(normalized<FASTN>(rd_in), h.n);
        if (fuzz > 0.0f) { ball_use = BALL_METAL; fuzz_out = fuzz; diffuse = false; }
        else if (!(dot(raw_d, h.n) > 0.0f)) return false;
    } else if (MI_HAS(MI355RT_MAT_DIELECTRIC) && kind == MI355RT_MAT_DIELECTRIC) {                                   // material.rs:122-162
        float ri = m4[1].w;
        float ratio = front_face ? (1.0f / ri) : (ri / 1.0f);
        f3 unit = normalized<FASTN>(rd_in);
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
    } else if ((MATS & MATS_ROUGH) != 0u) {                                        // RoughConductor, tungsten/materials.rs:306-377 (the two kinds that are left)
        const bool ggx = (kind == MI355RT_MAT_ROUGH_GGX);
        if (has_nan(rd_in)) return false;
        if (has_nan(h.n) || is_zero(h.n)) return false;
        f3 n = h.n;
        f3 v = -normalized<FASTN>(rd_in);
        if (has_nan(v)) return false;
        const float4 q1 = m4[1], q2 = m4[2], q3 = m4[3];
        float rough = q1.w;
        f3 eta = mk(q2.y, q2.z, q2.w), kk = mk(q3.x, q3.y, q3.z);
        // sample_ggx / sample_beckmann, tungsten/materials.rs:236-290
        float u1 = fmaxf(rng.uniform01_0(), 1e-6f);
        float u2 = rng.uniform01_1();
        float theta_arg;
        // The replay of the reference stream (Rng = RngRef) evaluates ln / atan / sin / cos in double and rounds once: practically the
        // correctly rounded f32 value, which is what a good host libm returns.  One ulp in the sampled half vector can flip an
        // absorb / scatter decision and with it the rest of the row's stream; with these the GPU replay reproduces the reference's
        // committed render (tools/gpu_ref_vs_golden.py).  The counter-mode kernels keep the native f32 functions.
        constexpr bool EXACT_LIBM = std::is_same<Rng, RngRef>::value;
        auto ln = [](float x) { return EXACT_LIBM ? (float)log((double)x) : logf(x); };
        if (ggx) { float a = rough * rough; theta_arg = a * a * (-ln(u1)) / (1.0f - u1); }
        else { theta_arg = -(rough * rough * ln(u1)); }
        f3 hv;
        if ((theta_arg != theta_arg) || isinf(theta_arg) || theta_arg < 0.0f) {
            hv = to_world(mk(0.f, 0.f, 1.f), n);
        } else {
            float theta = EXACT_LIBM ? (float)atan((double)sqrtf(theta_arg)) : atanf(sqrtf(theta_arg));
            float phi = 2.0f * PI_F * u2;
            float st, ct, sp, cp;                          // sin_cos(): one argument reduction serves both values
            if (EXACT_LIBM) { st = (float)sin((double)theta); ct = (float)cos((double)theta); sp = (float)sin((double)phi); cp = (float)cos((double)phi); }
            else { sincosf(theta, &st, &ct); sincosf(phi, &sp, &cp); }
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
#undef MI_HAS
    if (diffuse) ball_use = BALL_DIFFUSE;
    return true;
}
template <bool FASTN = false>
DI f3 diffuse_finish(const Hit& h, f3 p) {                                          // material.rs:54-62
    f3 dir = h.n + normalized<FASTN>(p);
    return near_zero(dir) ? h.n : dir;
}
// What the accepted unit-ball point turns into: the Lambert-style direction, or the metal's fuzzed reflection, which may be absorbed
// (material.rs:97-104: `reflected + fuzz * p`, None unless it leaves on the normal's side).  Returns false when absorbed.
template <bool FASTN = false>
DI bool ball_finish(uint32_t ball_use, const Hit& h, f3 p, float fuzz, f3& raw) {
    if (ball_use == BALL_DIFFUSE) raw = diffuse_finish<FASTN>(h, p);
    else if (ball_use == BALL_METAL) { raw = raw + p * fuzz; return dot(raw, h.n) > 0.0f; }
    return true;
}
// What every scatter() and Camera::get_ray end with: `.normalized()` of the direction, then Ray::new normalises again
// (ray.rs:12-17) -- and the origin offset along the normal.  The callers run it ONCE for all lanes of the wave, whatever
// branch produced the raw direction (it was the tail of every material branch and of the camera ray: ~66 instructions each).
template <bool FASTN = false>
DI f3 ray_direction(f3 raw) { return normalized<FASTN>(normalized<FASTN>(raw)); }
DI f3 scatter_origin(const Hit& h, float side) { return h.p + h.n * side; }
// Sequential composition (reference-stream replay kernel): random_in_unit_sphere as the plain loop, vec3.rs:54-61.
template <class Rng>
DI bool surface_scatter(const DevMat* __restrict__ mats, const DevTexture* __restrict__ texs, const float4 q0, const Hit& h, f3 rd_in, Rng& rng, f3& new_o, f3& new_d, f3& atten, f3& emitted) {
    uint32_t ball_use = BALL_NONE; float side = EPS, fuzz = 0.f; f3 raw = mk(0.f, 0.f, 1.f);
    if (!scatter_pre<MATS_ALL>(mats, texs, q0, h, rd_in, rng, side, raw, atten, emitted, ball_use, fuzz)) return false;
    if (ball_use != BALL_NONE) {
        f3 p; uint32_t j = 0;
        do { p = rng.cube_point(j); ++j; } while (!(len2(p) < 1.0f));
        if (!ball_finish(ball_use, h, p, fuzz, raw)) return false;
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


}  // namespace mi355rt
