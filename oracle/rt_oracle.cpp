// rt_oracle.cpp -- CPU ORACLE for the render loop of jackra1n/raytracer-rust.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under raytracer-rust_amd/ may include, link, import or
// execute this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and only as the checker / the timed CPU baseline -- never as the product path.
//
// What it is: a scalar C++17 restatement of the reference's hot path (SURVEY.md section 8a, rows
// a1-a25), one function per reference function, each citing the reference file:line it follows
// (paths are relative to /root/reference/src).  It consumes the same POD scene the C ABI takes
// (include/mi355rt.h) but IGNORES the caller's flattened BVH: it rebuilds a pointer-based BVH per
// mesh from the triangle soup with its own restatement of BVHNode::new and traverses it
// recursively like the reference, so the product's flattening/skip-link traversal is checked
// independently.
//
// Third-party arithmetic that is NOT under /root/reference (Cargo.lock pins), restated from the
// crates' published algorithms:
//   rand 0.9.1 / rand_chacha 0.9.0 / rand_core 0.9.3  StdRng = ChaCha12, seed_from_u64 (PCG32
//       expansion), StandardUniform<f32>, UniformFloat<f32>::sample_single   
//   glam 0.30.3  Mat4 * Vec4 (column-major, ((x*vx + y*vy) + z*vz) + w*vw), Vec3 min/max/signum
// Parity pinning: tests/test_oracle_golden.py checks this oracle (ref RNG mode) against the
// reference's own committed render docs/semesterbild.png and against the App. A RNG vectors.
//
// RNG modes (mi355rt_options.rng_mode):
//   REF: one ChaCha12 stream per image row, consumed sequentially by every pixel and sample of the
//        row (renderer.rs:91-101).  Radiance folded tail-first exactly like the recursion.
//   CTR: the GPU-native mode.  Same algorithm, but every draw is a pure function of
//        (row y + seed; x, sample, ray index, block) through Philox4x32-10, and the path
//        throughput is accumulated front-to-back.  Draw slots:
//          camera jitter   : ray 0, block 0, words 0 (u) and 1 (v)
//          scatter after ray r (r = 0 is the camera ray): ray r+1,
//              random::<f32>() number k of the event (k = 0,1) -> block 0, word k
//              rejection try j of random_in_unit_sphere       -> block j, words 1,2,3
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread

#include "rust_sort_unstable.hpp"
#include "../include/mi355rt.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

constexpr float EPSILON = 1e-4f;                       // renderer.rs:17
constexpr float PI_F = 3.14159265358979323846264338327950288f;   // std::f32::consts::PI

// ------------------------------------------------------------------------------------------------
// vec3.rs -- Vec3 (hand-rolled f32, not glam)
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }         // vec3.rs:84-93
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }         // vec3.rs:95-104
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }            // vec3.rs:106-115
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }                              // vec3.rs:143-152
inline V3 divf(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }                 // vec3.rs:117-129; its panic for |s| < 1e-4 is reached only through sphere.rs:38: build_scene() refuses such a sphere
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // vec3.rs:17-19
inline V3 cross(V3 a, V3 b) {                                                         // vec3.rs:21-27
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }       // vec3.rs:29-31
inline float length(V3 a) { return std::sqrt(length_squared(a)); }                    // vec3.rs:33-35
inline V3 normalized(V3 a) {                                                          // vec3.rs:37-44
    float len = length(a);
    if (len < EPSILON) return a;
    return a * (1.0f / len);
}
inline bool near_zero(V3 a) {                                                         // vec3.rs:63-66
    const float S = 1e-8f;
    return std::fabs(a.x) < S && std::fabs(a.y) < S && std::fabs(a.z) < S;
}
inline V3 vec_reflect(V3 v, V3 n) { return v - (n * 2.0f) * dot(v, n); }              // vec3.rs:68-70
inline V3 to_world(V3 local, V3 normal) {                                             // vec3.rs:72-81
    V3 up = (std::fabs(normal.z) < 0.999f) ? v3(0, 0, 1) : v3(0, 1, 0);
    V3 tangent = normalized(cross(normal, up));
    V3 bitangent = cross(normal, tangent);
    return (tangent * local.x + bitangent * local.y) + normal * local.z;
}
inline bool has_nan(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
inline bool is_zero_vec(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
inline V3 nan3() { float n = std::numeric_limits<float>::quiet_NaN(); return {n, n, n}; }
inline float idx(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }         // vec3.rs:131-141

// color.rs -- Color
struct Col { float r, g, b; };
inline Col operator+(Col a, Col b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Col operator-(Col a, Col b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Col operator*(Col a, Col b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Col operator/(Col a, Col b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
inline Col operator*(Col a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline Col cdivf(Col a, float s) { return {a.r / s, a.g / s, a.b / s}; }
inline Col splat(float v) { return {v, v, v}; }
inline Col csqrt(Col a) { return {std::sqrt(a.r), std::sqrt(a.g), std::sqrt(a.b)}; }
const Col BLACK = {0, 0, 0}, WHITE = {1, 1, 1};

// f32::clamp (NaN stays NaN) and `as u32` (saturating, NaN -> 0)
inline float rust_clamp01(float v) { if (v < 0.0f) return 0.0f; if (v > 1.0f) return 1.0f; return v; }
inline uint32_t rust_as_u32(float v) {
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
inline int32_t rust_as_i32(float v) {
    if (!(v == v)) return 0;
    if (v <= -2147483648.0f) return INT32_MIN;
    if (v >= 2147483648.0f) return INT32_MAX;
    return (int32_t)v;
}
inline uint32_t color_to_u32(Col c) {                                                // color.rs:87-93
    c.r = rust_clamp01(c.r); c.g = rust_clamp01(c.g); c.b = rust_clamp01(c.b);
    uint32_t r = rust_as_u32(c.r * 255.0f), g = rust_as_u32(c.g * 255.0f), b = rust_as_u32(c.b * 255.0f);
    return (r << 16) | (g << 8) | b;
}

// ray.rs
struct Ray { V3 origin, direction; };
inline Ray ray_new(V3 o, V3 d) { return {o, normalized(d)}; }                          // ray.rs:12-17
inline V3 ray_at(const Ray& r, float t) { return r.origin + r.direction * t; }        // ray.rs:9-11

// ------------------------------------------------------------------------------------------------
// RNG, reference mode: rand 0.9.1 StdRng == ChaCha12Rng
// ------------------------------------------------------------------------------------------------
inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
inline uint32_t rotr32(uint32_t v, uint32_t n) { n &= 31u; return n ? ((v >> n) | (v << (32 - n))) : v; }

struct ChaCha12 {
    uint32_t key[8];
    uint64_t counter;
    uint32_t buf[64];
    uint32_t index;
    uint64_t words_drawn;

    // rand_core SeedableRng::seed_from_u64 default impl: PCG32 output steps fill the 32-byte seed.
    void seed_from_u64(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
        for (int i = 0; i < 8; ++i) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = rotr32(xorshifted, rot);
        }
        counter = 0; index = 64; words_drawn = 0;
    }
    static void block(const uint32_t key[8], uint64_t ctr, uint32_t out[16]) {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                           key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                           (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
#define QR(a, b, c, d)                                   \
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 16); \
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 12); \
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 8);  \
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 7);
        for (int r = 0; r < 6; ++r) {          // 12 rounds = 6 double rounds
            QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
            QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
        }
#undef QR
        for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
    }
    void refill() {                              // 4 consecutive blocks per refill (BlockRng buffer)
        for (int b = 0; b < 4; ++b) block(key, counter + (uint64_t)b, buf + 16 * b);
        counter += 4; index = 0;
    }
    uint32_t next_u32() { if (index >= 64) refill(); ++words_drawn; return buf[index++]; }
};

inline float u32_to_f01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }   // StandardUniform<f32>
inline float u32_to_range11(uint32_t w) {                                               // UniformFloat::sample_single(-1.0..1.0)
    uint32_t bits = (w >> 9) | 0x3F800000u;
    float v12; std::memcpy(&v12, &bits, 4);
    float v01 = v12 - 1.0f;
    return v01 * 2.0f + -1.0f;
}

struct SamplerRef {                     // sequential row stream, call sites renderer.rs:96-97, vec3.rs:48-50,
    ChaCha12* rng;                      // material.rs:145, tungsten/materials.rs:46,247-248,275-276
    // ln / atan / sin / cos of the microfacet sampling: evaluated in double and rounded once (the correctly rounded f32 value in all
    // but ~1e-8 of the cases), not the platform's float functions -- the replay must not depend on whose libm runs it; the device's
    // replay kernel does the same (rt_materials.h), and both reproduce the reference's committed render exactly.
    static constexpr bool exact_libm = true;
    float jitter_u() { return u32_to_f01(rng->next_u32()); }
    float jitter_v() { return u32_to_f01(rng->next_u32()); }
    void begin_scatter() {}
    float uniform01(int /*slot*/) { return u32_to_f01(rng->next_u32()); }
    V3 cube_point(int /*try_index*/) {                                                 // vec3.rs:46-52 (x, y, z order)
        float x = u32_to_range11(rng->next_u32());
        float y = u32_to_range11(rng->next_u32());
        float z = u32_to_range11(rng->next_u32());
        return {x, y, z};
    }
};

// ------------------------------------------------------------------------------------------------
// RNG, counter mode.  The device library is built with ONE generator (rt_rng.h, MI355RT_CTR_GEN); the oracle holds all of them and is
// told which one to check against (oracle_set_ctr_gen; default = what the product library ships):
//   0  Philox4x32-10 (Salmon et al., SC'11), key = row key, counter = (x, sample, ray, block)
//   1  Philox4x32-7, same addressing
//   2  pcg4d (Jarzynski & Olano, JCGT 9(3), 2020): per-path base = pcg4d(x, sample, key lo, key hi); block j of the event after
//      ray r = pcg4d(base.x, base.y, base.z + r, base.w + j)
// One block = 4 words in every case; which word serves which draw is the same for all (SamplerCtr below).
// ------------------------------------------------------------------------------------------------
inline void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int rounds,
                       uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    philox4x32(k0, k1, c0, c1, c2, c3, 10, out);
}
inline void pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t out[4]) {
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}
#ifndef ORACLE_CTR_GEN_DEFAULT
#define ORACLE_CTR_GEN_DEFAULT 2                              // what the product library ships (rt_rng.h MI355RT_CTR_GEN)
#endif
static int g_ctr_gen = ORACLE_CTR_GEN_DEFAULT;
inline void ctr_block(int gen, uint32_t k0, uint32_t k1, uint32_t x, uint32_t s, uint32_t ray, uint32_t j, uint32_t out[4]) {
    if (gen == 2) { uint32_t b[4]; pcg4d(x, s, k0, k1, b); pcg4d(b[0], b[1], b[2] + ray, b[3] + j, out); }
    else philox4x32(k0, k1, x, s, ray, j, gen == 1 ? 7 : 10, out);
}

struct SamplerCtr {
    static constexpr bool exact_libm = false;                  // counter mode: the platform's float functions (the device uses its native ones)
    uint32_t k0, k1, x, s, ray;
    uint64_t words_drawn = 0;
    uint32_t cached_block = 0xFFFFFFFFu, cached_ray = 0xFFFFFFFFu, w[4];
    void load(uint32_t blockno) {
        if (cached_block != blockno || cached_ray != ray) {
            ctr_block(g_ctr_gen, k0, k1, x, s, ray, blockno, w);
            cached_block = blockno; cached_ray = ray;
        }
    }
    float jitter_u() { load(0); ++words_drawn; return u32_to_f01(w[0]); }
    float jitter_v() { load(0); ++words_drawn; return u32_to_f01(w[1]); }
    void begin_scatter() { ++ray; }
    float uniform01(int slot) { load(0); ++words_drawn; return u32_to_f01(w[slot]); }
    V3 cube_point(int try_index) {
        load((uint32_t)try_index); words_drawn += 3;
        return {u32_to_range11(w[1]), u32_to_range11(w[2]), u32_to_range11(w[3])};
    }
};

// ------------------------------------------------------------------------------------------------
// glam::Mat4 (column-major) * Vec4, as cube.rs:63-76 / mesh_object.rs:264-278 use it
// ------------------------------------------------------------------------------------------------
struct M4 { float m[16]; };     // m[4*col + row]
inline void mat_mul_vec4(const M4& a, float vx, float vy, float vz, float vw, float out[4]) {
    for (int r = 0; r < 4; ++r) {
        float acc = a.m[0 + r] * vx;
        acc = acc + a.m[4 + r] * vy;
        acc = acc + a.m[8 + r] * vz;
        acc = acc + a.m[12 + r] * vw;
        out[r] = acc;
    }
}
inline void mat_transpose_mul_vec4(const M4& a, float vx, float vy, float vz, float vw, float out[4]) {
    // (a.transpose()) * v : column c of the transpose is row c of a
    for (int r = 0; r < 4; ++r) {
        float acc = a.m[4 * r + 0] * vx;
        acc = acc + a.m[4 * r + 1] * vy;
        acc = acc + a.m[4 * r + 2] * vz;
        acc = acc + a.m[4 * r + 3] * vw;
        out[r] = acc;
    }
}
inline float glam_signum(float v) { if (std::isnan(v)) return v; return std::signbit(v) ? -1.0f : 1.0f; }

// ------------------------------------------------------------------------------------------------
// hittable.rs:10-27 -- HitRecord
// ------------------------------------------------------------------------------------------------
struct HitRecord { V3 position, normal; float t; uint32_t material; bool front_face; };
inline void set_face_normal(HitRecord& h, const Ray& ray, V3 outward) {               // hittable.rs:19-26
    h.front_face = dot(ray.direction, outward) < 0.0f;
    h.normal = h.front_face ? outward : -outward;
}

struct Counters {
    uint64_t samples = 0, rays = 0, prim_tests = 0, bvh_nodes = 0, tri_tests = 0, rng_words = 0, depth_exhausted = 0;
    void add(const Counters& o) {
        samples += o.samples; rays += o.rays; prim_tests += o.prim_tests; bvh_nodes += o.bvh_nodes;
        tri_tests += o.tri_tests; rng_words += o.rng_words; depth_exhausted += o.depth_exhausted;
    }
};

// ------------------------------------------------------------------------------------------------
// acceleration/aabb.rs, acceleration/bvh.rs, mesh/triangle.rs
// ------------------------------------------------------------------------------------------------
// MI355RT_FLAG_FIXED_AABB (opt-in, not the reference): set for the duration of one oracle_render call.
static bool g_fixed_aabb = false;
// TextureMaterial images of the scene being rendered (set by build_scene / oracle_set_textures before workers start; read-only then)
static const mi355rt_texture* g_textures = nullptr;
static uint32_t g_n_textures = 0;
struct Aabb {
    V3 min, max;
    static Aabb empty() {                                                            // aabb.rs:11-16
        float inf = std::numeric_limits<float>::infinity();
        return {{inf, inf, inf}, {-inf, -inf, -inf}};
    }
    void add_point(V3 p) {                                                            // aabb.rs:18-25
        min.x = std::fmin(min.x, p.x); min.y = std::fmin(min.y, p.y); min.z = std::fmin(min.z, p.z);
        max.x = std::fmax(max.x, p.x); max.y = std::fmax(max.y, p.y); max.z = std::fmax(max.z, p.z);
    }
    bool intersect(const Ray& ray, float t_min, float t_max) const {                  // aabb.rs:27-45
        for (int axis = 0; axis < 3; ++axis) {
            float inv_d = 1.0f / idx(ray.direction, axis);
            float t0 = (idx(min, axis) - idx(ray.origin, axis)) * inv_d;
            float t1 = (idx(max, axis) - idx(ray.origin, axis)) * inv_d;
            if (inv_d < 0.0f) std::swap(t0, t1);
            t_min = std::fmax(t_min, t0);      // f32::max ignores NaN, like fmaxf
            t_max = std::fmin(t_max, t1);
            if (g_fixed_aabb ? (t_max < t_min) : (t_max <= t_min)) return false;  // reference: zero-thickness boxes never hit
        }
        return true;
    }
};

struct Tri { V3 v0, v1, v2, normal; };

struct BVHNode {                                                                      // bvh.rs:7-12
    Aabb bounds;
    std::unique_ptr<BVHNode> left, right;
    std::vector<uint32_t> triangle_indices;
};

std::unique_ptr<BVHNode> bvh_new(const std::vector<Tri>& tris, uint32_t* indices, size_t n, size_t depth,
                                 uint32_t* max_depth_out) {                           // bvh.rs:15-76
    auto node = std::make_unique<BVHNode>();
    if (depth > *max_depth_out) *max_depth_out = (uint32_t)depth;
    Aabb bounds = Aabb::empty();
    for (size_t i = 0; i < n; ++i) {
        const Tri& t = tris[indices[i]];
        bounds.add_point(t.v0); bounds.add_point(t.v1); bounds.add_point(t.v2);
    }
    node->bounds = bounds;
    const size_t MAX_DEPTH = 25, MIN_TRIANGLES_PER_LEAF = 4;
    if (n <= MIN_TRIANGLES_PER_LEAF || depth >= MAX_DEPTH) {
        node->triangle_indices.assign(indices, indices + n);
        return node;
    }
    V3 extent = bounds.max - bounds.min;
    int axis = (extent.x > extent.y && extent.x > extent.z) ? 0 : (extent.y > extent.z ? 1 : 2);
    auto centroid_axis = [&](uint32_t a) {
        V3 c = ((tris[a].v0 + tris[a].v1) + tris[a].v2) * (1.0f / 3.0f);
        return idx(c, axis);
    };
    // sort_unstable_by(partial_cmp, NaN -> Equal), bvh.rs:45-53.  The order of TIES is not specified by Rust but it is
    // deterministic, and it decides which triangles share a leaf -- i.e. which flat leaf boxes exist (App. B-1).  So the sort is
    // restated (rust_sort_unstable.hpp: core::slice::sort::unstable of Rust 1.81+, "ipnsort"); with it the oracle's REF mode
    // matches the reference's committed render with NO pixel further than 20/255 (0.15 % of the pixels with a stable sort).
    rustsort::sort_unstable_by(indices, n, [&](uint32_t a, uint32_t b) { return centroid_axis(a) < centroid_axis(b); });
    size_t mid = n / 2;
    if (mid == 0 || mid == n) {
        node->triangle_indices.assign(indices, indices + n);
        return node;
    }
    node->left = bvh_new(tris, indices, mid, depth + 1, max_depth_out);
    node->right = bvh_new(tris, indices + mid, n - mid, depth + 1, max_depth_out);
    return node;
}

bool bvh_intersect_recursive(const BVHNode* node, const Ray& ray, const std::vector<Tri>& tris, float t_min,
                             float t_max, HitRecord& out, Counters& c) {              // bvh.rs:78-170
    ++c.bvh_nodes;
    if (!node->bounds.intersect(ray, t_min, t_max)) return false;
    if (!node->left) {
        bool any = false;
        for (uint32_t id : node->triangle_indices) {
            ++c.tri_tests;
            const Tri& tr = tris[id];
            V3 edge1 = tr.v1 - tr.v0, edge2 = tr.v2 - tr.v0;
            V3 h = cross(ray.direction, edge2);
            float a = dot(edge1, h);
            if (std::fabs(a) < EPSILON) continue;
            float f = 1.0f / a;
            V3 s = ray.origin - tr.v0;
            float u = f * dot(s, h);
            if (!(u >= 0.0f && u <= 1.0f)) continue;
            V3 q = cross(s, edge1);
            float v = f * dot(ray.direction, q);
            if (v < 0.0f || u + v > 1.0f) continue;
            float t = f * dot(edge2, q);
            if (t > t_min && t < t_max) {
                V3 position = ray_at(ray, t);
                bool front = dot(ray.direction, tr.normal) < 0.0f;
                out.t = t; out.position = position; out.normal = front ? tr.normal : -tr.normal;
                out.front_face = front;
                t_max = t; any = true;
            }
        }
        return any;
    }
    HitRecord l, r;
    bool hl = bvh_intersect_recursive(node->left.get(), ray, tris, t_min, t_max, l, c);
    if (hl) t_max = l.t;
    bool hr = bvh_intersect_recursive(node->right.get(), ray, tris, t_min, t_max, r, c);
    if (hl && hr) { out = (l.t < r.t) ? l : r; return true; }
    if (hl) { out = l; return true; }
    if (hr) { out = r; return true; }
    return false;
}

// ------------------------------------------------------------------------------------------------
// STUDY ONLY (tools/nearfirst_study.py): would a near-child-first walk with an explicit stack return
// the hit the reference's fixed left-then-right recursion returns?  Same box test, same triangle test; an inner node tests
// both child boxes, descends into the one the ray enters first and stacks the other, which is tested AGAIN with the
// current t_max when popped.  Equal-t hits are resolved as the reference resolves them: the reference only ever replaces
// a hit by a strictly closer one while visiting triangles in pre-order, so of several triangles at the minimal t it
// returns the first in pre-order -- here: accept t == best only for a smaller pre-order position.
// Never used for rendering.
// ------------------------------------------------------------------------------------------------
struct WalkStudy {
    unsigned long long walks = 0, differ_tri = 0, differ_hitmiss = 0, differ_t_only = 0, nodes_ref = 0, nodes_ordered = 0,
                       tris_ref = 0, tris_ordered = 0;
};
static WalkStudy* g_walk_study_total = nullptr;              // armed by oracle_walk_study(); summed over the worker threads
static std::mutex g_walk_study_mutex;
static thread_local WalkStudy t_walk_study;
static thread_local WalkStudy* g_walk_study = nullptr;      // per worker thread: &t_walk_study while a study render runs

static bool aabb_entry(const Aabb& b, const Ray& ray, float t_min, float t_max, float& entry) {   // Aabb::intersect + the entry distance it ends with
    for (int axis = 0; axis < 3; ++axis) {
        float inv_d = 1.0f / idx(ray.direction, axis);
        float t0 = (idx(b.min, axis) - idx(ray.origin, axis)) * inv_d;
        float t1 = (idx(b.max, axis) - idx(ray.origin, axis)) * inv_d;
        if (inv_d < 0.0f) std::swap(t0, t1);
        t_min = std::fmax(t_min, t0);
        t_max = std::fmin(t_max, t1);
        if (g_fixed_aabb ? (t_max < t_min) : (t_max <= t_min)) return false;
    }
    entry = t_min;
    return true;
}

// pre-order position of every leaf's first triangle (the reference's visiting order), filled once per mesh for the study
static void preorder_positions(const BVHNode* n, std::vector<uint32_t>& pos_of_tri, uint32_t& next) {
    if (!n->left) { for (uint32_t id : n->triangle_indices) pos_of_tri[id] = next++; return; }
    preorder_positions(n->left.get(), pos_of_tri, next);
    preorder_positions(n->right.get(), pos_of_tri, next);
}

static bool bvh_intersect_ordered(const BVHNode* root, const Ray& ray, const std::vector<Tri>& tris, const std::vector<uint32_t>& pos_of_tri,
                                  float t_min, float t_max, float& best_t, uint32_t& best_tri, WalkStudy& st) {
    best_t = t_max; best_tri = 0xFFFFFFFFu;
    std::vector<const BVHNode*> stack;
    float e;
    ++st.nodes_ordered;
    if (!aabb_entry(root->bounds, ray, t_min, best_t, e)) return false;
    const BVHNode* node = root;
    for (;;) {
        if (!node->left) {
            for (uint32_t id : node->triangle_indices) {
                ++st.tris_ordered;
                const Tri& tr = tris[id];
                V3 edge1 = tr.v1 - tr.v0, edge2 = tr.v2 - tr.v0;
                V3 h = cross(ray.direction, edge2);
                float a = dot(edge1, h);
                if (std::fabs(a) < EPSILON) continue;
                float f = 1.0f / a;
                V3 s = ray.origin - tr.v0;
                float u = f * dot(s, h);
                if (!(u >= 0.0f && u <= 1.0f)) continue;
                V3 q = cross(s, edge1);
                float v = f * dot(ray.direction, q);
                if (v < 0.0f || u + v > 1.0f) continue;
                float t = f * dot(edge2, q);
                if (!(t > t_min)) continue;
                if (t < best_t || (t == best_t && best_tri != 0xFFFFFFFFu && pos_of_tri[id] < pos_of_tri[best_tri])) { best_t = t; best_tri = id; }
            }
            node = nullptr;
        } else {
            float el = 0, er = 0;
            st.nodes_ordered += 2;
            const bool hl = aabb_entry(node->left->bounds, ray, t_min, best_t, el);
            const bool hr = aabb_entry(node->right->bounds, ray, t_min, best_t, er);
            if (hl && hr) {
                const bool left_first = el <= er;
                stack.push_back(left_first ? node->right.get() : node->left.get());
                node = left_first ? node->left.get() : node->right.get();
            } else node = hl ? node->left.get() : hr ? node->right.get() : nullptr;
        }
        while (!node) {
            if (stack.empty()) return best_tri != 0xFFFFFFFFu;
            const BVHNode* cand = stack.back(); stack.pop_back();
            ++st.nodes_ordered;
            // with equal-t tie-breaking a box whose entry EQUALS best_t may still hold the pre-order-earlier triangle: the box
            // test culls it (t_max <= t_min), exactly as the reference culls it when it comes second -- see the study's report
            if (aabb_entry(cand->bounds, ray, t_min, best_t, e)) node = cand;
        }
    }
}

// reference walk that also reports the winning triangle id
static bool bvh_intersect_reference_id(const BVHNode* node, const Ray& ray, const std::vector<Tri>& tris, float t_min, float t_max,
                                       float& out_t, uint32_t& out_tri, WalkStudy& st) {
    ++st.nodes_ref;
    float e;
    if (!aabb_entry(node->bounds, ray, t_min, t_max, e)) return false;
    if (!node->left) {
        bool any = false;
        for (uint32_t id : node->triangle_indices) {
            ++st.tris_ref;
            const Tri& tr = tris[id];
            V3 edge1 = tr.v1 - tr.v0, edge2 = tr.v2 - tr.v0;
            V3 h = cross(ray.direction, edge2);
            float a = dot(edge1, h);
            if (std::fabs(a) < EPSILON) continue;
            float f = 1.0f / a;
            V3 s = ray.origin - tr.v0;
            float u = f * dot(s, h);
            if (!(u >= 0.0f && u <= 1.0f)) continue;
            V3 q = cross(s, edge1);
            float v = f * dot(ray.direction, q);
            if (v < 0.0f || u + v > 1.0f) continue;
            float t = f * dot(edge2, q);
            if (t > t_min && t < t_max) { out_t = t; out_tri = id; t_max = t; any = true; }
        }
        return any;
    }
    float tl = 0, tr_ = 0; uint32_t il = 0, ir = 0;
    const bool hl = bvh_intersect_reference_id(node->left.get(), ray, tris, t_min, t_max, tl, il, st);
    if (hl) t_max = tl;
    const bool hr = bvh_intersect_reference_id(node->right.get(), ray, tris, t_min, t_max, tr_, ir, st);
    if (hl && hr) { if (tl < tr_) { out_t = tl; out_tri = il; } else { out_t = tr_; out_tri = ir; } return true; }
    if (hl) { out_t = tl; out_tri = il; return true; }
    if (hr) { out_t = tr_; out_tri = ir; return true; }
    return false;
}

// ------------------------------------------------------------------------------------------------
// Scene (scene.rs + hittable.rs:29-58), built from the POD input
// ------------------------------------------------------------------------------------------------
struct MeshData { std::vector<Tri> tris; std::unique_ptr<BVHNode> bvh; uint32_t max_depth = 0; std::vector<uint32_t> pos_of_tri; /* study only */ };

struct Scene {
    std::vector<float> sky; uint32_t sky_w = 0, sky_h = 0;      // scene.rs:9 skybox_hdr_image
    std::vector<mi355rt_primitive> prims;
    std::vector<mi355rt_material> mats;
    std::vector<MeshData> meshes;
    Col miss;
};

bool build_scene(const mi355rt_scene* in, Scene& sc) {
    sc.prims.assign(in->primitives, in->primitives + in->n_primitives);
    sc.mats.assign(in->materials, in->materials + in->n_materials);
    sc.miss = {in->miss_color[0], in->miss_color[1], in->miss_color[2]};
    g_textures = in->textures; g_n_textures = in->textures ? in->n_textures : 0u;
    for (uint32_t i = 0; i < g_n_textures; ++i) if (!g_textures[i].rgba8 || g_textures[i].width == 0 || g_textures[i].height == 0) return false;
    for (uint32_t i = 0; i < in->n_materials; ++i) if (in->materials[i].kind == MI355RT_MAT_TEXTURE && in->materials[i].texture >= g_n_textures) return false;
    if (in->sky_rgb) {
        if (!in->sky_width || !in->sky_height) return false;
        sc.sky.assign(in->sky_rgb, in->sky_rgb + (size_t)in->sky_width * in->sky_height * 3);
        sc.sky_w = in->sky_width; sc.sky_h = in->sky_height;
    }
    sc.meshes.resize(in->n_meshes);
    for (uint32_t m = 0; m < in->n_meshes; ++m) {
        const mi355rt_mesh& md = in->meshes[m];
        if ((uint64_t)md.first_triangle + md.triangle_count > in->n_triangles || md.triangle_count == 0) return false;
        MeshData& out = sc.meshes[m];
        out.tris.resize(md.triangle_count);
        for (uint32_t i = 0; i < md.triangle_count; ++i) {
            const mi355rt_triangle& t = in->triangles[md.first_triangle + i];
            out.tris[i] = {{t.v0[0], t.v0[1], t.v0[2]}, {t.v1[0], t.v1[1], t.v1[2]}, {t.v2[0], t.v2[1], t.v2[2]},
                           {t.normal[0], t.normal[1], t.normal[2]}};
        }
        std::vector<uint32_t> indices(md.triangle_count);
        for (uint32_t i = 0; i < md.triangle_count; ++i) indices[i] = i;       // mesh_object.rs:44-45
        out.bvh = bvh_new(out.tris, indices.data(), indices.size(), 0, &out.max_depth);
        if (g_walk_study_total) { out.pos_of_tri.assign(md.triangle_count, 0u); uint32_t next = 0; preorder_positions(out.bvh.get(), out.pos_of_tri, next); }
    }
    for (const auto& p : sc.prims) {
        if (p.kind >= MI355RT_PRIM_KIND_COUNT) return false;
        if (p.material >= sc.mats.size()) return false;
        if (p.kind == MI355RT_PRIM_MESH && p.mesh >= sc.meshes.size()) return false;
        // sphere.rs:38 divides by the radius with `Vec3 / f32`, which PANICS for |radius| < 1e-4 (vec3.rs:120-122) the first time such
        // a sphere is hit: the reference cannot render the scene, so neither side accepts it (the HIP path returns MI355RT_ERR_INVALID).
        if (p.kind == MI355RT_PRIM_SPHERE && std::fabs(p.data[3]) < 1e-4f) return false;
    }
    for (const auto& m : sc.mats) if (m.kind >= MI355RT_MAT_KIND_COUNT) return false;
    return true;
}

// objects/sphere.rs:15-53
bool sphere_hit(const mi355rt_primitive& p, const Ray& ray, float t_min, float t_max, HitRecord& h) {
    V3 center = {p.data[0], p.data[1], p.data[2]}; float radius = p.data[3];
    V3 oc = ray.origin - center;
    float a = dot(ray.direction, ray.direction);
    float half_b = dot(oc, ray.direction);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = std::sqrt(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root <= t_min || root >= t_max) {
        root = (-half_b + sqrtd) / a;
        if (root <= t_min || root >= t_max) return false;
    }
    h.t = root;
    h.position = ray_at(ray, root);
    V3 outward = divf(h.position - center, radius);
    h.front_face = dot(ray.direction, outward) < 0.0f;
    h.normal = h.front_face ? outward : -outward;
    return true;
}

// objects/plane.rs:26-56
bool plane_hit(const mi355rt_primitive& p, const Ray& ray, float t_min, float t_max, HitRecord& h) {
    V3 p1 = {p.data[0], p.data[1], p.data[2]}, n = {p.data[3], p.data[4], p.data[5]};
    float denom = dot(n, ray.direction);
    if (std::fabs(denom) < EPSILON) return false;
    float t = dot(n, p1 - ray.origin) / denom;
    if (t <= t_min || t >= t_max) return false;
    h.t = t; h.position = ray_at(ray, t);
    h.front_face = dot(ray.direction, n) < 0.0f;
    h.normal = h.front_face ? n : -n;
    return true;
}

// tungsten/objects/quad.rs:83-132
bool quad_hit(const mi355rt_primitive& p, const Ray& ray, float t_min, float t_max, HitRecord& h) {
    const float* d = p.data;
    V3 base = {d[0], d[1], d[2]}, edge0 = {d[3], d[4], d[5]}, edge1 = {d[6], d[7], d[8]}, n = {d[9], d[10], d[11]};
    float dd = d[12], inv0 = d[13], inv1 = d[14];
    float denom = dot(n, ray.direction);
    if (std::fabs(denom) < EPSILON) return false;
    float t = (dd - dot(n, ray.origin)) / denom;
    if (t <= t_min || t >= t_max) return false;
    V3 hit_pos = ray_at(ray, t);
    V3 v = hit_pos - base;
    float l0 = dot(v, edge0) * inv0;
    float l1 = dot(v, edge1) * inv1;
    const float lo = -EPSILON, hi = 1.0f + EPSILON;
    if (!((l0 >= lo && l0 <= hi) && (l1 >= lo && l1 <= hi))) return false;
    h.front_face = dot(ray.direction, n) < 0.0f;
    h.normal = h.front_face ? n : -n;
    h.t = t; h.position = hit_pos;
    return true;
}

// objects/cube.rs:59-158
bool cube_hit(const mi355rt_primitive& p, const Ray& ray, float t_min, float t_max, HitRecord& h) {
    M4 o2w, w2o; std::memcpy(o2w.m, p.data, 64); std::memcpy(w2o.m, p.data + 16, 64);
    float oh[4], dh[4];
    mat_mul_vec4(w2o, ray.origin.x, ray.origin.y, ray.origin.z, 1.0f, oh);
    mat_mul_vec4(w2o, ray.direction.x, ray.direction.y, ray.direction.z, 0.0f, dh);
    float ro[3] = {oh[0], oh[1], oh[2]}, rd[3] = {dh[0], dh[1], dh[2]};
    float te[3], tx[3];
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / rd[i];
        float t1 = (-0.5f - ro[i]) * inv, t2 = (0.5f - ro[i]) * inv;
        te[i] = std::fmin(t1, t2); tx[i] = std::fmax(t1, t2);        // glam Vec3::min/max (f32::min/max)
    }
    float t_enter = std::fmax(te[0], std::fmax(te[1], te[2]));
    float t_exit = std::fmin(tx[0], std::fmin(tx[1], tx[2]));
    if (t_exit < t_enter || t_exit <= 0.0f) return false;
    float t_hit_obj = (t_enter > 0.0f) ? t_enter : t_exit;
    if (t_hit_obj >= t_max || t_hit_obj <= t_min || t_hit_obj < EPSILON) return false;
    float po[3] = {ro[0] + rd[0] * t_hit_obj, ro[1] + rd[1] * t_hit_obj, ro[2] + rd[2] * t_hit_obj};
    float n[3] = {0, 0, 0};
    float ax = std::fabs(po[0]), ay = std::fabs(po[1]), az = std::fabs(po[2]);
    const float tol = 1e-4f;
    if (std::fabs(ax - 0.5f) < tol) n[0] = glam_signum(po[0]);
    else if (std::fabs(ay - 0.5f) < tol) n[1] = glam_signum(po[1]);
    else if (std::fabs(az - 0.5f) < tol) n[2] = glam_signum(po[2]);
    else if (ax > ay && ax > az) n[0] = glam_signum(po[0]);
    else if (ay > az) n[1] = glam_signum(po[1]);
    else n[2] = glam_signum(po[2]);
    {   // normalize_or_zero
        float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float rcp = 1.0f / len;
        if (std::isfinite(rcp) && rcp > 0.0f) { n[0] *= rcp; n[1] *= rcp; n[2] *= rcp; } else { n[0] = n[1] = n[2] = 0.0f; }
    }
    float pw[4], nw[4];
    mat_mul_vec4(o2w, po[0], po[1], po[2], 1.0f, pw);
    mat_transpose_mul_vec4(w2o, n[0], n[1], n[2], 0.0f, nw);
    V3 position_world = {pw[0], pw[1], pw[2]};
    V3 normal_world = normalized(v3(nw[0], nw[1], nw[2]));
    V3 p_minus_o = position_world - ray.origin;
    if (dot(p_minus_o, ray.direction) < 0.0f) return false;
    float t_world = dot(position_world - ray.origin, ray.direction);
    if (t_world < t_min || t_world > t_max) return false;
    h.t = t_world; h.position = position_world;
    set_face_normal(h, ray, normal_world);
    return true;
}

// mesh/mesh_object.rs:263-329
// Diagnostic (tools/): histogram of BVH nodes visited per mesh walk, filled when oracle_walk_histogram() armed it.
constexpr uint64_t WALK_HIST_BINS = 512;
static unsigned long long* g_walk_hist = nullptr;
bool mesh_hit(const mi355rt_primitive& p, const MeshData& mesh, const Ray& ray_world, float t_min_world,
              float t_max_world, HitRecord& h, Counters& c) {
    M4 o2w, w2o; std::memcpy(o2w.m, p.data, 64); std::memcpy(w2o.m, p.data + 16, 64);
    float oh[4], dh[4];
    mat_mul_vec4(w2o, ray_world.origin.x, ray_world.origin.y, ray_world.origin.z, 1.0f, oh);
    mat_mul_vec4(w2o, ray_world.direction.x, ray_world.direction.y, ray_world.direction.z, 0.0f, dh);
    V3 ray_origin_obj = {oh[0], oh[1], oh[2]}, ray_direction_obj = {dh[0], dh[1], dh[2]};
    Ray ray_obj = ray_new(ray_origin_obj, normalized(ray_direction_obj));
    HitRecord rec;
    const uint64_t nodes_before = c.bvh_nodes;
    const bool walk_hit = bvh_intersect_recursive(mesh.bvh.get(), ray_obj, mesh.tris, t_min_world, t_max_world, rec, c);
    if (g_walk_hist) __atomic_fetch_add(&g_walk_hist[std::min<uint64_t>(c.bvh_nodes - nodes_before, WALK_HIST_BINS - 1)], 1ull, __ATOMIC_RELAXED);
    if (g_walk_study && !mesh.pos_of_tri.empty()) {      // study: the same walk in both orders (single-threaded runs only)
        WalkStudy& st = *g_walk_study;
        float t_ref = 0, t_ord = 0; uint32_t i_ref = 0xFFFFFFFFu, i_ord = 0xFFFFFFFFu;
        const bool h_ref = bvh_intersect_reference_id(mesh.bvh.get(), ray_obj, mesh.tris, t_min_world, t_max_world, t_ref, i_ref, st);
        const bool h_ord = bvh_intersect_ordered(mesh.bvh.get(), ray_obj, mesh.tris, mesh.pos_of_tri, t_min_world, t_max_world, t_ord, i_ord, st);
        ++st.walks;
        if (h_ref != h_ord) ++st.differ_hitmiss;
        else if (h_ref && i_ref != i_ord) ++st.differ_tri;
        else if (h_ref && std::memcmp(&t_ref, &t_ord, 4) != 0) ++st.differ_t_only;
    }
    if (!walk_hit) return false;
    float pw[4], nw[4];
    mat_mul_vec4(o2w, rec.position.x, rec.position.y, rec.position.z, 1.0f, pw);
    mat_transpose_mul_vec4(w2o, rec.normal.x, rec.normal.y, rec.normal.z, 0.0f, nw);
    V3 pos_world = {pw[0], pw[1], pw[2]};
    V3 normal_world = normalized(v3(nw[0], nw[1], nw[2]));
    float t_world = rec.t * length(ray_direction_obj) / length(ray_world.direction);   // (sic) App. B-3
    if (t_world < t_min_world || t_world > t_max_world) return false;
    h.position = pos_world; h.t = t_world;
    set_face_normal(h, ray_world, normal_world);
    return true;
}

// hittable.rs:45-58 -- HittableList::hit
bool scene_hit(const Scene& sc, const Ray& ray, float t_min, float t_max, HitRecord& out, Counters& c) {
    float closest = t_max; bool any = false; HitRecord tmp;
    for (const auto& p : sc.prims) {
        ++c.prim_tests;
        bool hit = false;
        switch (p.kind) {
            case MI355RT_PRIM_SPHERE: hit = sphere_hit(p, ray, t_min, closest, tmp); break;
            case MI355RT_PRIM_PLANE:  hit = plane_hit(p, ray, t_min, closest, tmp); break;
            case MI355RT_PRIM_QUAD:   hit = quad_hit(p, ray, t_min, closest, tmp); break;
            case MI355RT_PRIM_CUBE:   hit = cube_hit(p, ray, t_min, closest, tmp); break;
            case MI355RT_PRIM_MESH:   hit = mesh_hit(p, sc.meshes[p.mesh], ray, t_min, closest, tmp, c); break;
        }
        if (hit) { closest = tmp.t; tmp.material = p.material; out = tmp; any = true; }
    }
    return any;
}

// ------------------------------------------------------------------------------------------------
// material.rs / tungsten/materials.rs
// ------------------------------------------------------------------------------------------------
inline V3 mat_reflect(V3 v_in, V3 n) {                                                // material.rs:194-206, tungsten/materials.rs:292-304
    if (has_nan(v_in)) return nan3();
    if (has_nan(n) || is_zero_vec(n)) return nan3();
    return v_in - (n * 2.0f) * dot(v_in, n);
}
inline bool mat_refract(V3 uv, V3 n, float etai_over_etat, V3& out) {                 // material.rs:208-219
    float cos_theta = std::fmin(dot(-uv, n), 1.0f);
    V3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    float r_out_parallel_squared = 1.0f - length_squared(r_out_perp);
    if (r_out_parallel_squared < 0.0f) return false;
    V3 r_out_parallel = n * (-std::sqrt(r_out_parallel_squared));
    out = r_out_perp + r_out_parallel;
    return true;
}
inline float powi5(float x) { return x * ((x * x) * (x * x)); }                        // llvm.powi.f32(x, 5) expansion
inline float schlick_reflectance(float cosine, float ref_idx_ratio) {                 // material.rs:221-227
    float r0 = (1.0f - ref_idx_ratio) / (1.0f + ref_idx_ratio);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}
inline float schlick(float cosine, float ref_idx) {                                   // tungsten/materials.rs:23-27
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    float r0_sq = r0 * r0;
    return r0_sq + (1.0f - r0_sq) * powi5(1.0f - cosine);
}
inline Col checker_value(const mi355rt_material& m, V3 p) {                           // tungsten/materials.rs:89-99
    float inv_scale = m.p0;
    int32_t xc = rust_as_i32(std::floor(p.x * inv_scale));
    int32_t yc = rust_as_i32(std::floor(p.y * inv_scale));
    int32_t zc = rust_as_i32(std::floor(p.z * inv_scale));
    int32_t sum = (int32_t)((uint32_t)xc + (uint32_t)yc + (uint32_t)zc);
    if (sum % 2 == 0) return {m.albedo[0], m.albedo[1], m.albedo[2]};
    return {m.aux[0], m.aux[1], m.aux[2]};
}
template <class S> inline V3 random_in_unit_sphere(S& rng) {                          // vec3.rs:54-61
    for (int j = 0;; ++j) {
        V3 p = rng.cube_point(j);
        if (length_squared(p) < 1.0f) return p;
    }
}
template <class S> inline void lambert_direction(const HitRecord& h, S& rng, Ray& scattered) {   // material.rs:54-62
    V3 dir = h.normal + normalized(random_in_unit_sphere(rng));
    if (near_zero(dir)) dir = h.normal;
    V3 origin = h.position + h.normal * EPSILON;
    scattered = ray_new(origin, normalized(dir));
}

inline Col fresnel_conductor(float cos_theta, Col eta, Col k) {                       // tungsten/materials.rs:184-202
    cos_theta = rust_clamp01(cos_theta);
    Col cos2 = splat(cos_theta * cos_theta);
    Col sin2 = splat(1.0f) - cos2;
    Col eta2 = eta * eta, k2 = k * k;
    Col t0 = eta2 - k2 - sin2;
    Col a2plusb2 = csqrt(t0 * t0 + splat(4.0f) * eta2 * k2);
    Col t1 = a2plusb2 + cos2;
    Col a = csqrt((a2plusb2 + t0) * splat(0.5f));
    Col t2 = splat(2.0f * cos_theta) * a;
    Col rs = (t1 - t2) / (t1 + t2);
    Col t3 = cos2 * a2plusb2 + sin2 * sin2;
    Col t4 = t2;
    Col rp = rs * ((t3 - t4) / (t3 + t4));
    return (rs + rp) * splat(0.5f);
}
inline float ggx_g1(float n_dot_x, float roughness) {                                 // tungsten/materials.rs:205-216
    if (n_dot_x <= 0.0f) return 0.0f;
    float a = roughness * roughness;
    float k = a / 2.0f;
    float denom = n_dot_x * (1.0f - k) + k;
    if (denom < EPSILON) return 1.0f;
    return n_dot_x / denom;
}
inline float ggx_g(float roughness, float ndv, float ndl) { return ggx_g1(ndv, roughness) * ggx_g1(ndl, roughness); }  // :218-221
inline float beckmann_lambda(float a, float x) {                                      // tungsten/materials.rs:225-232
    float t = 1.0f / (a * x);
    if (t < 1.6f) return (1.0f - 1.259f * t + 0.396f * t * t) / (3.535f * t + 2.181f * t * t);
    return 0.0f;
}
inline float beckmann_g(float roughness, float ndv, float ndl) {                      // tungsten/materials.rs:223-234
    return 1.0f / (1.0f + beckmann_lambda(roughness, ndv) + beckmann_lambda(roughness, ndl));
}
template <class S> V3 sample_half_vector(bool ggx, V3 normal, float roughness, S& rng) {   // tungsten/materials.rs:236-290
    if (has_nan(normal) || is_zero_vec(normal)) return nan3();
    float u1 = std::fmax(rng.uniform01(0), 1e-6f);
    float u2 = rng.uniform01(1);
    float theta_arg;
    auto ln = [](float x) { return S::exact_libm ? (float)std::log((double)x) : std::log(x); };
    auto at = [](float x) { return S::exact_libm ? (float)std::atan((double)x) : std::atan(x); };
    auto sn = [](float x) { return S::exact_libm ? (float)std::sin((double)x) : std::sin(x); };
    auto cs = [](float x) { return S::exact_libm ? (float)std::cos((double)x) : std::cos(x); };
    if (ggx) { float a = roughness * roughness; theta_arg = a * a * (-ln(u1)) / (1.0f - u1); }
    else     { theta_arg = -(roughness * roughness * ln(u1)); }
    if (std::isnan(theta_arg) || std::isinf(theta_arg) || theta_arg < 0.0f) return to_world(v3(0, 0, 1), normal);
    float theta = at(std::sqrt(theta_arg));
    float phi = 2.0f * PI_F * u2;
    float sin_theta = sn(theta), cos_theta = cs(theta);
    V3 h_local = {sin_theta * cs(phi), sin_theta * sn(phi), cos_theta};
    if (has_nan(h_local)) return to_world(v3(0, 0, 1), normal);
    return to_world(h_local, normal);
}

// TextureMaterial (tungsten/parser.rs:199-243).  The images of the scene being rendered; set by build_scene() /
// oracle_set_textures() before any worker thread starts, read-only afterwards.
static Col texture_value(const mi355rt_material& m, V3 normal_tex) {                  // parser.rs:222-241
    if (m.texture >= g_n_textures) return BLACK;
    const mi355rt_texture& t = g_textures[m.texture];
    float theta = std::acos(normal_tex.y);
    float phi = std::atan2(normal_tex.z, normal_tex.x) + PI_F;
    float u = phi / (2.0f * PI_F);
    float v = theta / PI_F;
    u = std::fmod(u + m.p0, 1.0f);                                                    // f32 % f32
    uint32_t x_pixel = rust_as_u32(std::fmax(u, 0.0f) * (float)(t.width - 1));
    uint32_t y_pixel = rust_as_u32(std::fmax(v, 0.0f) * (float)(t.height - 1));
    const uint8_t* px = t.rgba8 + 4 * ((size_t)std::min(y_pixel, t.height - 1) * t.width + std::min(x_pixel, t.width - 1));
    return {(float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f};
}

// Material::scatter for all kinds.  Returns false for None.
template <class S>
bool scatter(const mi355rt_material& m, const Ray& ray_in, const HitRecord& h, S& rng, Ray& scattered, Col& atten) {
    switch (m.kind) {
    case MI355RT_MAT_LAMBERT_SOLID:                                                   // material.rs:47-71
        lambert_direction(h, rng, scattered);
        atten = {m.albedo[0], m.albedo[1], m.albedo[2]};
        return true;
    case MI355RT_MAT_LAMBERT_CHECKER:
        lambert_direction(h, rng, scattered);
        atten = checker_value(m, h.position);
        return true;
    case MI355RT_MAT_TEXTURE: {                                                       // tungsten/parser.rs:205-243
        lambert_direction(h, rng, scattered);
        Col a = {m.albedo[0], m.albedo[1], m.albedo[2]};
        atten = a * texture_value(m, h.normal);
        return true;
    }
    case MI355RT_MAT_METAL: {                                                         // material.rs:87-110
        V3 reflected = mat_reflect(normalized(ray_in.direction), h.normal);
        float fuzz = m.p0;
        V3 fuzzed = (fuzz > 0.0f) ? reflected + random_in_unit_sphere(rng) * fuzz : reflected;
        if (dot(fuzzed, h.normal) > 0.0f) {
            scattered = ray_new(h.position + h.normal * EPSILON, normalized(fuzzed));
            atten = {m.albedo[0], m.albedo[1], m.albedo[2]};
            return true;
        }
        return false;
    }
    case MI355RT_MAT_DIELECTRIC: {                                                    // material.rs:122-162
        float ri = m.p0;
        float refraction_ratio = h.front_face ? (1.0f / ri) : (ri / 1.0f);
        V3 unit_direction = normalized(ray_in.direction);
        float cos_theta = std::fmin(dot(-unit_direction, h.normal), 1.0f);
        float sin_theta_squared = 1.0f - cos_theta * cos_theta;
        bool cannot_refract = refraction_ratio * refraction_ratio * sin_theta_squared > 1.0f;
        float reflectance = schlick_reflectance(cos_theta, 1.0f / refraction_ratio);
        V3 dir;
        if (cannot_refract || reflectance > rng.uniform01(0)) {      // short-circuit: no draw under TIR (App. B-7)
            dir = mat_reflect(unit_direction, h.normal);
        } else {
            V3 refr;
            dir = mat_refract(unit_direction, h.normal, refraction_ratio, refr) ? refr : mat_reflect(unit_direction, h.normal);
        }
        V3 origin = (dot(dir, h.normal) > 0.0f) ? h.position + h.normal * EPSILON : h.position - h.normal * EPSILON;
        scattered = ray_new(origin, normalized(dir));
        atten = WHITE;
        return true;
    }
    case MI355RT_MAT_EMISSIVE: return false;                                          // material.rs:179-187
    case MI355RT_MAT_NULL: return false;                                              // material.rs:239-247
    case MI355RT_MAT_PLASTIC: {                                                       // tungsten/materials.rs:29-65
        float ior = m.p0;
        float dn = dot(ray_in.direction, h.normal);
        float cosine = (dn > 0.0f) ? ior * dn / length(ray_in.direction) : -dn / length(ray_in.direction);
        float reflect_prob = schlick(cosine, ior);
        if (rng.uniform01(0) < reflect_prob) {
            V3 reflected_dir = normalized(vec_reflect(ray_in.direction, h.normal));
            scattered = ray_new(h.position + h.normal * EPSILON, reflected_dir);
            atten = {0.9f, 0.9f, 0.9f};
        } else {
            lambert_direction(h, rng, scattered);
            atten = {m.albedo[0], m.albedo[1], m.albedo[2]};
        }
        return true;
    }
    case MI355RT_MAT_ROUGH_GGX:
    case MI355RT_MAT_ROUGH_BECKMANN: {                                                // tungsten/materials.rs:306-377
        bool ggx = m.kind == MI355RT_MAT_ROUGH_GGX;
        if (has_nan(ray_in.direction)) return false;
        if (has_nan(h.normal) || is_zero_vec(h.normal)) return false;
        V3 n = h.normal;
        V3 v = -normalized(ray_in.direction);
        if (has_nan(v)) return false;
        Col eta = {m.eta[0], m.eta[1], m.eta[2]}, k = {m.k[0], m.k[1], m.k[2]};
        float rough = m.p0;
        V3 hv = sample_half_vector(ggx, n, rough, rng);
        if (has_nan(hv)) return false;
        V3 l = mat_reflect(-v, hv);
        if (has_nan(l)) return false;
        if (dot(l, n) <= 0.0f) return false;
        float n_dot_l = std::fmax(dot(n, l), 0.0f);
        float n_dot_v = std::fmax(dot(n, v), 0.0f);
        float n_dot_h = std::fmax(dot(n, hv), 0.0f);
        float v_dot_h = std::fmax(dot(v, hv), 0.0f);
        float g = ggx ? ggx_g(rough, n_dot_v, n_dot_l) : beckmann_g(rough, n_dot_v, n_dot_l);
        Col f = fresnel_conductor(v_dot_h, eta, k);
        Col brdf_numerator = f * g * v_dot_h;
        float brdf_denominator = n_dot_v * n_dot_h + EPSILON;
        Col albedo = {m.albedo[0], m.albedo[1], m.albedo[2]};
        atten = (brdf_denominator > EPSILON) ? albedo * cdivf(brdf_numerator, brdf_denominator) : BLACK;
        scattered = ray_new(h.position + n * EPSILON, normalized(l));
        return true;
    }
    }
    return false;
}
inline Col emitted(const mi355rt_material& m) {                                       // material.rs:18-20, :189-191
    if (m.kind == MI355RT_MAT_EMISSIVE) return {m.albedo[0], m.albedo[1], m.albedo[2]};
    return BLACK;
}

// renderer.rs:38-63 -- what a missing ray returns
inline Col miss_colour(const Scene& sc, const Ray& ray_in) {
    if (sc.sky_w == 0) return sc.miss;                                                // :61
    V3 dir = normalized(ray_in.direction);                                            // :41
    float theta = std::acos(dir.y);                                                   // :42
    float phi = std::atan2(dir.z, dir.x) + PI_F;                                      // :43
    float u = phi / (2.0f * PI_F);                                                    // :44
    float v = theta / PI_F;                                                           // :45
    uint32_t xp = rust_as_u32(std::fmax(u * (float)(sc.sky_w - 1), 0.0f));            // :47
    uint32_t yp = rust_as_u32(std::fmax(v * (float)(sc.sky_h - 1), 0.0f));            // :48
    size_t o = 3 * ((size_t)std::min(yp, sc.sky_h - 1) * sc.sky_w + std::min(xp, sc.sky_w - 1));   // :50-53
    return {sc.sky[o], sc.sky[o + 1], sc.sky[o + 2]};                                 // :54
}

// ------------------------------------------------------------------------------------------------
// renderer.rs:19-65 -- trace_ray, exact recursion (tail-first folding)
// ------------------------------------------------------------------------------------------------
template <class S>
Col trace_ray_tail(const Ray& ray_in, const Scene& sc, uint32_t depth, S& rng, Counters& c) {
    if (depth == 0) { ++c.depth_exhausted; return BLACK; }
    ++c.rays;
    HitRecord h;
    if (scene_hit(sc, ray_in, EPSILON, std::numeric_limits<float>::infinity(), h, c)) {
        const mi355rt_material& m = sc.mats[h.material];
        Col emitted_light = emitted(m);
        Ray scattered; Col atten;
        rng.begin_scatter();
        if (scatter(m, ray_in, h, rng, scattered, atten)) {
            Col scattered_color = trace_ray_tail(scattered, sc, depth - 1, rng, c);
            return emitted_light + atten * scattered_color;
        }
        return emitted_light;
    }
    return miss_colour(sc, ray_in);                                                   // renderer.rs:38-63
}

// Same walk, throughput accumulated front-to-back (the GPU's order): L = ((a1*a2)*...*ak) * terminal.
template <class S>
Col trace_ray_fwd(Ray ray, const Scene& sc, uint32_t max_depth, S& rng, Counters& c) {
    Col throughput = WHITE;
    for (uint32_t depth = max_depth;; --depth) {
        if (depth == 0) { ++c.depth_exhausted; return throughput * BLACK; }
        ++c.rays;
        HitRecord h;
        if (!scene_hit(sc, ray, EPSILON, std::numeric_limits<float>::infinity(), h, c)) return throughput * miss_colour(sc, ray);
        const mi355rt_material& m = sc.mats[h.material];
        Ray scattered; Col atten;
        rng.begin_scatter();
        if (!scatter(m, ray, h, rng, scattered, atten)) return throughput * emitted(m);
        throughput = throughput * atten;
        ray = scattered;
    }
}

// camera.rs:33-42
inline Ray camera_get_ray(const mi355rt_camera& cam, float u, float v) {
    V3 position = {cam.position[0], cam.position[1], cam.position[2]};
    V3 forward = {cam.forward[0], cam.forward[1], cam.forward[2]};
    V3 right = {cam.right[0], cam.right[1], cam.right[2]};
    V3 true_up = {cam.true_up[0], cam.true_up[1], cam.true_up[2]};
    float ndc_x = 2.0f * u - 1.0f;
    float ndc_y = 1.0f - 2.0f * v;
    V3 offset = right * (ndc_x * cam.half_width) + true_up * (ndc_y * cam.half_height);
    V3 ray_dir = normalized(forward + offset);
    return ray_new(position, ray_dir);
}

// Row selection shared with the C ABI (mi355rt_options): strips dealt round-robin.
struct RowSel { std::vector<uint32_t> rows; };
bool select_rows(const mi355rt_settings& st, const mi355rt_options* o, RowSel& sel) {
    uint32_t rb = 0, re = st.height, strip = 1, parts = 1, part = 0;
    if (o) {
        rb = o->row_begin; re = o->row_end ? o->row_end : st.height;
        strip = o->strip_rows ? o->strip_rows : 1; parts = o->n_parts ? o->n_parts : 1; part = o->part;
    }
    if (re > st.height || rb > re || part >= parts) return false;
    for (uint32_t y = rb; y < re; ++y) if ((y / strip) % parts == part) sel.rows.push_back(y);
    return true;
}

}  // namespace

extern "C" {

struct oracle_counters {
    uint64_t samples, rays, prim_tests, bvh_nodes, tri_tests, rng_words, depth_exhausted;
    double seconds;
};

// renderer.rs:67-123 -- render_scene.  fold: -1 = mode default (REF -> tail-first recursion,
// CTR -> forward throughput), 0 = tail, 1 = forward.
// Arm (hist != NULL, WALK_HIST_BINS entries, caller-owned) or disarm (NULL) the walk-length histogram.
void oracle_walk_histogram(unsigned long long* hist) { g_walk_hist = hist; }
void oracle_set_textures(const mi355rt_texture* textures, uint32_t n) { g_textures = textures; g_n_textures = n; }   // for oracle_scatter_ctr
// Study hook (tools/nearfirst_study.py): while `out8` is non-null every mesh walk of oracle_render is ALSO run in near-first
// order and compared; out8 = walks, differ_tri, differ_hitmiss, differ_t_only, nodes_ref, nodes_ordered, tris_ref, tris_ordered.
void oracle_walk_study(unsigned long long* out8) { g_walk_study_total = reinterpret_cast<WalkStudy*>(out8); }
int oracle_render(const mi355rt_scene* scene_in, const mi355rt_camera* cam, const mi355rt_settings* st,
                  const mi355rt_options* opt, int n_threads, int fold, uint32_t* out_packed, float* out_linear,
                  oracle_counters* counters_out) {
    Scene sc;
    if (!scene_in || !cam || !st || !build_scene(scene_in, sc)) return MI355RT_ERR_INVALID;
    RowSel sel;
    if (!select_rows(*st, opt, sel)) return MI355RT_ERR_INVALID;
    const uint32_t rng_mode = opt ? opt->rng_mode : MI355RT_RNG_CTR;
    const uint64_t seed = opt ? opt->seed : 0;
    const bool tail = (fold < 0) ? (rng_mode == MI355RT_RNG_REF) : (fold == 0);
    g_fixed_aabb = opt && (opt->flags & MI355RT_FLAG_FIXED_AABB) != 0u;               // workers are started below, joined before returning
    const uint32_t W = st->width, H = st->height, spp = st->samples_per_pixel, max_depth = st->max_depth;
    if (W == 0 || H == 0 || spp == 0) return MI355RT_ERR_INVALID;
    const float inv_spp = 1.0f / (float)spp;                                          // renderer.rs:85
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;

    std::atomic<size_t> next_row{0};
    std::vector<Counters> per_thread((size_t)n_threads);
    auto t_begin = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Counters c;
        if (g_walk_study_total) { t_walk_study = WalkStudy(); g_walk_study = &t_walk_study; }
        for (;;) {
            size_t j = next_row.fetch_add(1);
            if (j >= sel.rows.size()) break;
            const uint32_t y = sel.rows[j];                                            // par_chunks_mut(width).enumerate(), :87-90
            ChaCha12 chacha; chacha.seed_from_u64((uint64_t)y + seed);                // renderer.rs:91
            for (uint32_t x = 0; x < W; ++x) {                                          // :93
                Col acc = BLACK;
                for (uint32_t s = 0; s < spp; ++s) {                                    // :95
                    ++c.samples;
                    Col L;
                    if (rng_mode == MI355RT_RNG_REF) {
                        SamplerRef rng{&chacha};
                        float u = ((float)x + rng.jitter_u()) / (float)W;              // :96
                        float v = ((float)y + rng.jitter_v()) / (float)H;              // :97
                        Ray ray = camera_get_ray(*cam, u, v);                          // :99
                        L = tail ? trace_ray_tail(ray, sc, max_depth, rng, c) : trace_ray_fwd(ray, sc, max_depth, rng, c);
                    } else {
                        uint64_t ykey = (uint64_t)y + seed;
                        SamplerCtr rng; rng.k0 = (uint32_t)ykey; rng.k1 = (uint32_t)(ykey >> 32); rng.x = x; rng.s = s; rng.ray = 0;
                        float u = ((float)x + rng.jitter_u()) / (float)W;
                        float v = ((float)y + rng.jitter_v()) / (float)H;
                        Ray ray = camera_get_ray(*cam, u, v);
                        L = tail ? trace_ray_tail(ray, sc, max_depth, rng, c) : trace_ray_fwd(ray, sc, max_depth, rng, c);
                        c.rng_words += rng.words_drawn;
                    }
                    acc = acc + L;                                                      // :100-101
                }
                Col pixel = acc * inv_spp;                                              // :103
                size_t o = j * (size_t)W + x;
                if (out_linear) { out_linear[3 * o] = pixel.r; out_linear[3 * o + 1] = pixel.g; out_linear[3 * o + 2] = pixel.b; }
                if (out_packed) out_packed[o] = color_to_u32(csqrt(pixel));           // :112-120
            }
            if (rng_mode == MI355RT_RNG_REF) c.rng_words += chacha.words_drawn;
        }
        per_thread[(size_t)tid] = c;
        if (g_walk_study) {
            std::lock_guard<std::mutex> lock(g_walk_study_mutex);
            WalkStudy& T = *g_walk_study_total; const WalkStudy& w = t_walk_study;
            T.walks += w.walks; T.differ_tri += w.differ_tri; T.differ_hitmiss += w.differ_hitmiss; T.differ_t_only += w.differ_t_only;
            T.nodes_ref += w.nodes_ref; T.nodes_ordered += w.nodes_ordered; T.tris_ref += w.tris_ref; T.tris_ordered += w.tris_ordered;
            g_walk_study = nullptr;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    auto t_end = std::chrono::steady_clock::now();
    if (counters_out) {
        Counters tot; for (auto& c : per_thread) tot.add(c);
        counters_out->samples = tot.samples; counters_out->rays = tot.rays; counters_out->prim_tests = tot.prim_tests;
        counters_out->bvh_nodes = tot.bvh_nodes; counters_out->tri_tests = tot.tri_tests; counters_out->rng_words = tot.rng_words;
        counters_out->depth_exhausted = tot.depth_exhausted;
        counters_out->seconds = std::chrono::duration<double>(t_end - t_begin).count();
    }
    g_fixed_aabb = false;
    return MI355RT_OK;
}

// ---- unit-level probes for the tests ------------------------------------------------------------
void oracle_chacha_key(uint64_t seed, uint32_t out8[8]) { ChaCha12 r; r.seed_from_u64(seed); std::memcpy(out8, r.key, 32); }
void oracle_chacha_words(uint64_t seed, uint32_t n, uint32_t* out) {
    ChaCha12 r; r.seed_from_u64(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next_u32();
}
float oracle_u32_to_f01(uint32_t w) { return u32_to_f01(w); }
float oracle_u32_to_range11(uint32_t w) { return u32_to_range11(w); }
void oracle_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    philox4x32_10(k0, k1, c0, c1, c2, c3, out);
}
void oracle_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int rounds, uint32_t out[4]) { philox4x32(k0, k1, c0, c1, c2, c3, rounds, out); }
void oracle_pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t out[4]) { pcg4d(x, y, z, w, out); }
// The 4 words of block j of the event after ray `ray` of path (k0, k1; x, s) under generator `gen` (-1: the one in force)
void oracle_ctr_block(int gen, uint32_t k0, uint32_t k1, uint32_t x, uint32_t s, uint32_t ray, uint32_t j, uint32_t out[4]) { ctr_block(gen < 0 ? g_ctr_gen : gen, k0, k1, x, s, ray, j, out); }
// Which counter-mode generator oracle_render / oracle_scatter_ctr use from now on (0 / 1 / 2, see above); returns the previous one.  Not
// thread-safe against a render in progress: set it between renders.
int oracle_set_ctr_gen(int gen) { const int old = g_ctr_gen; if (gen >= 0 && gen <= 2) g_ctr_gen = gen; return old; }
int oracle_get_ctr_gen(void) { return g_ctr_gen; }
uint32_t oracle_color_to_u32(float r, float g, float b) { return color_to_u32(csqrt(Col{r, g, b})); }

// Closest hit of one ray against the scene (HittableList::hit with t_min = EPSILON, t_max = inf).
// out9 = position[3], normal[3], t, material, front_face.  Returns 1 on hit.
int oracle_scene_hit(const mi355rt_scene* scene_in, const float* origin, const float* dir_unnormalised, float* out9) {
    Scene sc; if (!build_scene(scene_in, sc)) return MI355RT_ERR_INVALID;
    Ray ray = ray_new(v3(origin[0], origin[1], origin[2]), v3(dir_unnormalised[0], dir_unnormalised[1], dir_unnormalised[2]));
    HitRecord h; Counters c;
    if (!scene_hit(sc, ray, EPSILON, std::numeric_limits<float>::infinity(), h, c)) return 0;
    out9[0] = h.position.x; out9[1] = h.position.y; out9[2] = h.position.z;
    out9[3] = h.normal.x; out9[4] = h.normal.y; out9[5] = h.normal.z;
    out9[6] = h.t; out9[7] = (float)h.material; out9[8] = h.front_face ? 1.0f : 0.0f;
    return 1;
}

// One Material::scatter call in CTR mode with explicit counters: (k0,k1,x,s,ray) identify the event.
// out10 = did_scatter, origin[3], direction[3], attenuation[3]
int oracle_scatter_ctr(const mi355rt_material* m, const float* ray_o, const float* ray_d, const float* pos, const float* nrm,
                       int front_face, uint32_t k0, uint32_t k1, uint32_t x, uint32_t s, uint32_t ray_index, float* out10) {
    Ray rin{v3(ray_o[0], ray_o[1], ray_o[2]), v3(ray_d[0], ray_d[1], ray_d[2])};
    HitRecord h; h.position = v3(pos[0], pos[1], pos[2]); h.normal = v3(nrm[0], nrm[1], nrm[2]); h.t = 0; h.material = 0;
    h.front_face = front_face != 0;
    SamplerCtr rng; rng.k0 = k0; rng.k1 = k1; rng.x = x; rng.s = s; rng.ray = ray_index;
    Ray sc; Col at;
    bool ok = scatter(*m, rin, h, rng, sc, at);
    out10[0] = ok ? 1.0f : 0.0f;
    if (ok) {
        out10[1] = sc.origin.x; out10[2] = sc.origin.y; out10[3] = sc.origin.z;
        out10[4] = sc.direction.x; out10[5] = sc.direction.y; out10[6] = sc.direction.z;
        out10[7] = at.r; out10[8] = at.g; out10[9] = at.b;
    }
    return MI355RT_OK;
}

// BVH topology dump of the oracle's own builder, preorder.  Per node: bmin[3], bmax[3] in `bounds`
// (6 floats), and in `info` (3 u32): is_leaf, n_tris, depth.  Leaf triangle ids are appended to
// `leaf_ids` in visiting order.  Call with NULL arrays to get the counts.
static void dump_rec(const BVHNode* n, uint32_t depth, float* bounds, uint32_t* info, uint32_t* leaf_ids, uint32_t& ni, uint32_t& li) {
    if (bounds) { float* b = bounds + 6 * (size_t)ni; b[0] = n->bounds.min.x; b[1] = n->bounds.min.y; b[2] = n->bounds.min.z;
                  b[3] = n->bounds.max.x; b[4] = n->bounds.max.y; b[5] = n->bounds.max.z; }
    if (info) { info[3 * (size_t)ni] = n->left ? 0u : 1u; info[3 * (size_t)ni + 1] = (uint32_t)n->triangle_indices.size(); info[3 * (size_t)ni + 2] = depth; }
    ++ni;
    if (!n->left) { for (uint32_t id : n->triangle_indices) { if (leaf_ids) leaf_ids[li] = id; ++li; } return; }
    dump_rec(n->left.get(), depth + 1, bounds, info, leaf_ids, ni, li);
    dump_rec(n->right.get(), depth + 1, bounds, info, leaf_ids, ni, li);
}
// Branch counters of the sort restatement (rust_sort_unstable.hpp g_paths): out[rustsort::P_COUNT]; reset != 0 clears them afterwards.
void oracle_sort_paths(uint64_t* out, int reset) {
    for (int i = 0; i < rustsort::P_COUNT; ++i) { if (out) out[i] = rustsort::g_paths[i]; if (reset) rustsort::g_paths[i] = 0; }
}
int oracle_bvh_dump(const mi355rt_triangle* tris_in, uint32_t n, float* bounds, uint32_t* info, uint32_t* leaf_ids,
                    uint32_t* n_nodes_out, uint32_t* n_leaf_ids_out, uint32_t* max_depth_out) {
    if (!tris_in || n == 0) return MI355RT_ERR_INVALID;
    std::vector<Tri> tris(n);
    for (uint32_t i = 0; i < n; ++i) {
        const mi355rt_triangle& t = tris_in[i];
        tris[i] = {{t.v0[0], t.v0[1], t.v0[2]}, {t.v1[0], t.v1[1], t.v1[2]}, {t.v2[0], t.v2[1], t.v2[2]}, {t.normal[0], t.normal[1], t.normal[2]}};
    }
    std::vector<uint32_t> indices(n); for (uint32_t i = 0; i < n; ++i) indices[i] = i;
    uint32_t md = 0;
    auto root = bvh_new(tris, indices.data(), n, 0, &md);
    uint32_t ni = 0, li = 0;
    dump_rec(root.get(), 0, bounds, info, leaf_ids, ni, li);
    if (n_nodes_out) *n_nodes_out = ni;
    if (n_leaf_ids_out) *n_leaf_ids_out = li;
    if (max_depth_out) *max_depth_out = md;
    return MI355RT_OK;
}

}  // extern "C"
