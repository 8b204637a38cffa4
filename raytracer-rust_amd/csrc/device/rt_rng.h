// rt_rng.h -- the two samplers: counter-mode Philox4x32-10 (RngCtr) and the replay of rand_chacha ChaCha12 (RngRef)
// Part of the device code of libmi355rt.so; included by rt_kernels.hip only (one translation unit: every kernel sees the same
// inlined device functions, and build.kernel_hash() covers every file of this directory).
#pragma once
#include "rt_math.h"

namespace mi355rt {

// ---------------------------------------------------------------------------------------------------
// RNG: float conversions of rand 0.9.1 (StandardUniform<f32>, UniformFloat::sample_single(-1..1))
// ---------------------------------------------------------------------------------------------------
DI float u32_to_f01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
// rand's `random_range(-1.0..1.0)`: (value1_2 - 1.0) * 2.0 + -1.0 with value1_2 = the 23 high bits as the mantissa of a float in [1, 2).
// Every step of that is exact, so the result is k * 2^-22 - 1 for k = w >> 9 -- which is also, exactly, the float in [2, 4) with mantissa k
// minus 3: three instructions instead of five, the same bits for all 2^23 values of k (tests/test_oracle_rng.py checks them all).
DI float u32_to_range11(uint32_t w) {
#ifdef MI355RT_AB_RANGE11_LONG
    float v12 = __uint_as_float((w >> 9) | 0x3F800000u); float v01 = v12 - 1.0f; return v01 * 2.0f + -1.0f;
#else
    return __uint_as_float((w >> 9) | 0x40000000u) - 3.0f;
#endif
}

// Philox4x32-10: counter-based, no state.  10 x (2 x 32x32->64 multiplies + 4 xor + 2 add).
// WIDE: one 64-bit product per multiplier -- v_mad_u64_u32 issues like ONE v_mul_hi_u32 (2.1 add slots, tools/microbench/int_mul.hip)
// and yields both halves, where __umulhi() and `*` written separately compile to two such instructions: 20 instead of 40
// slow multiplies per call.  The VALU-bound lockstep kernels use it (cornell -6.1 %, veach-mis -2.4 %); the wavefront kernel
// was 2-3 % faster on the two independent multiplies while its SHADE spilled 50 registers (round 2) and is 1 % faster on the wide
// form since round 3 (rt_wavefront.h).  Same bits either way.
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


}  // namespace mi355rt
