// rt_rng.h -- the two samplers: the counter-mode generator (RngCtr: pcg4d since round 5, Philox4x32-10 / -7 at build time) and the replay of rand_chacha ChaCha12 (RngRef)
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
DI float u32_to_range11(uint32_t w) { return __uint_as_float((w >> 9) | 0x40000000u) - 3.0f; }

// Philox4x32-R (Salmon et al., SC'11): counter-based, no state.  R x (2 x 32x32->64 multiplies + 4 xor + 2 add).
// WIDE: one 64-bit product per multiplier -- v_mad_u64_u32 issues like ONE v_mul_hi_u32 (2.1 add slots, tools/microbench/int_mul.hip)
// and yields both halves, where __umulhi() and `*` written separately compile to two such instructions: 20 instead of 40
// slow multiplies per call.  Same bits either way.
template <bool WIDE = false, int ROUNDS = 10>
DI void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
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

// pcg4d (Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 9(3), 2020): a bijection of 128 bits -- one LCG step per word, a
// round of word-by-word multiply-adds, xorshift 16, a second round.  12 multiply-adds (a*b+c on 32-bit operands: one v_mad_u64_u32 each,
// the low half is what is used) + 8 shift / xor: about 32 issue slots where Philox4x32-10 takes about 85.
DI uint32_t mad32(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)((uint64_t)a * b + c); }
DI void pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t (&out)[4]) {
    x = mad32(x, 1664525u, 1013904223u); y = mad32(y, 1664525u, 1013904223u); z = mad32(z, 1664525u, 1013904223u); w = mad32(w, 1664525u, 1013904223u);
    x = mad32(y, w, x); y = mad32(z, x, y); z = mad32(x, y, z); w = mad32(y, z, w);
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x = mad32(y, w, x); y = mad32(z, x, y); z = mad32(x, y, z); w = mad32(y, z, w);
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

// Which counter-mode generator the library is built with (the oracle's CTR mode mirrors it: oracle/rt_oracle.cpp, CTR_GEN):
//   0  Philox4x32-10, key = row key (64 bit), counter = (x, sample, ray, block)                      -- rounds 1-4
//   1  Philox4x32-7, same addressing (Random123's Crush-resistant minimum)
//   2  pcg4d: a per-path base = pcg4d(x, sample, key lo, key hi); block j of the event after ray r = pcg4d(base + (0, 0, r, j))
// Shipped since round 5: 2.  Measured on the headline config (cornell 800x600x256, kernel ms, one process, interleaved: profiles/r05/ab_counter_generator.txt):
// Philox4x32-10 14.57, Philox4x32-7 13.71 (-5.9 %), pcg4d 13.38 (-8.1 %); quality gates: tools/rng_battery.py (the addressed stream through a
// SmallCrush-style battery: profiles/r05/rng_battery_*.txt) and tools/rng_image_gate.py (image means / RMSE against the reference-stream oracle).
#ifndef MI355RT_CTR_GEN
#define MI355RT_CTR_GEN 2
#endif
constexpr int CTR_GEN = MI355RT_CTR_GEN;

// Counter mode sampler: draws are addressed, not consumed (slots documented in oracle/rt_oracle.cpp
// and DESIGN.md): jitter = (ray 0, block 0, words 0/1); scatter event after ray r uses ray r+1:
// random::<f32>() number k -> block 0 word k; rejection try j -> block j words 1..3.
// `w` = what addresses a path's draws (NW words; the cooperative rejection hands exactly these to its worker lanes):
//   Philox: k0, k1, x, s, ray;  pcg4d: base.x, base.y, base.z + ray, base.w.
struct RngCtr {
    static constexpr int NW = CTR_GEN == 2 ? 4 : 5;
    uint32_t w[NW];
    uint32_t b0[4];
    template <bool WIDE = false> DI static void block(const uint32_t (&a)[NW], uint32_t j, uint32_t (&out)[4]) {
        if constexpr (CTR_GEN == 2) pcg4d(a[0], a[1], a[2], a[3] + j, out);
        else philox4x32<WIDE, CTR_GEN == 1 ? 7 : 10>(a[0], a[1], a[2], a[3], a[NW - 1], j, out);
    }
    DI void clear() { for (int i = 0; i < NW; ++i) w[i] = 0u; b0[0] = b0[1] = b0[2] = b0[3] = 0u; }
    // One generator call per loop iteration serves BOTH kinds of lanes: a freshly dealt path reads its camera
    // jitter from (ray 0, block 0); a continuing path reads its scatter draws from (ray r+1, block 0).
    DI void start(uint32_t k0_, uint32_t k1_, uint32_t x_, uint32_t s_) {
        if constexpr (CTR_GEN == 2) { uint32_t b[4]; pcg4d(x_, s_, k0_, k1_, b); w[0] = b[0]; w[1] = b[1]; w[2] = b[2]; w[3] = b[3]; }
        else { w[0] = k0_; w[1] = k1_; w[2] = x_; w[3] = s_; w[NW - 1] = 0u; }
    }
    DI void set_ray(uint32_t r) { if constexpr (CTR_GEN == 2) w[2] += r; else w[NW - 1] = r; }     // right after start(): the path is at ray r
    DI void next_event() { if constexpr (CTR_GEN == 2) ++w[2]; else ++w[NW - 1]; }
    template <bool WIDE = false> DI void load_block0() { block<WIDE>(w, 0u, b0); }
    DI float jitter_u() { return u32_to_f01(b0[0]); }
    DI float jitter_v() { return u32_to_f01(b0[1]); }
    DI void begin_scatter() {}
    DI float uniform01_0() { return u32_to_f01(b0[0]); }
    DI float uniform01_1() { return u32_to_f01(b0[1]); }
    template <bool WIDE = false> DI f3 cube_point(uint32_t j) {
        if (j == 0) return mk(u32_to_range11(b0[1]), u32_to_range11(b0[2]), u32_to_range11(b0[3]));
        uint32_t b[4]; block<WIDE>(w, j, b);
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
