// rt_device.h -- device-resident scene layout and kernel launch interface (internal to libmi355rt.so).
//
// HBM layout (all arrays 16-byte aligned, record sizes multiples of 16 bytes so that the
// wave-uniform top-level walk lowers to s_load_dwordx4/x8/x16 and the per-lane BVH walk to
// global_load_dwordx4):
//   DevPrim[n_prims]   240 B  top-level list in caller order (hittable.rs:45-58), read through the
//                             constant address space with a wave-uniform index -> SGPRs
//   DevMat[n_mats]      64 B  per-lane gather by material index (4 x dwordx4)
//   DevNode[n_nodes]    32 B  two-link BVH nodes, stored level by level (all meshes' roots, then their children, ...):
//                             `b` = escape index (bits 0..25; NODE_END ends the walk) | leaf triangle count << 26
//                             (0: inner node); `a` = left child (inner) / first triangle (leaf).  Hit inner -> a,
//                             hit leaf -> its triangles then the escape, miss -> the escape: exactly the
//                             left-then-right order of bvh.rs:142-156, whatever the storage order.  The first
//                             LDS_NODE_CAP nodes (the top levels) are what k_render_ctr_sm keeps in LDS
//   DevTri[n_tris]      48 B  triangles re-ordered into leaf order, stored as v0, e1=v1-v0, e2=v2-v0,
//                             normal (the same f32 subtractions bvh.rs:95-96 performs per test)
//   rows[3 * n_rows]     4 B  three tables: local output row -> absolute image row y; PROCESSING row -> absolute y (the RNG key; what the
//                             counter-mode kernels read); processing row -> local output row (what the resolve kernel reads)
//   sky[h*w*3]           4 B  optional equirect HDR skybox, per-lane nearest-texel gather on miss
//   radiance[band]      12 B  three floats per path (sample-major inside a pixel), written once by
//                             the path tracer and read once by the resolve kernel
#pragma once
#include <stdint.h>
#include "../../../include/mi355rt.h"

namespace mi355rt {

struct DevPrim {                 // 60 words = 240 B
    uint32_t kind, material, node_begin, run_end;    // node_begin: root node of a MI355RT_PRIM_MESH; run_end: list index one past the
                                                     // run of consecutive primitives of this kind that this one belongs to (> own index)
    // sphere: c[3], r | plane: p1[3], n[3] | quad: n[3], d, base[3], e0[3], e1[3], inv0, inv1 (15; normal and plane constant first: every ray needs them)
    // cube / mesh: d[0..11] = rows 0..2 of the 4 columns of w2o {c0.xyz, c1.xyz, c2.xyz, c3.xyz}, d[12..14] = zd[3] = w2o.w_axis.xyz * 0.0f -- the 15 words
    //              of the hit test in one run (one scalar load) --, d[16..27] = o2w likewise, d[31..33] = zn[3] = {w2o[3], w2o[7], w2o[11]} * 0.0f,
    //              cube only: d[34..51] = the 6 possible world normals normalized(w2o^T * (+-e_k, 0)), k = x,y,z, + then -
    float d[52];
    float mat0[4];                                   // a copy of the first 16 bytes of the primitive's material record (kind, albedo):
                                                     // the hit record and the head of its material arrive with ONE memory round trip (finish_hit)
};
static_assert(sizeof(DevPrim) == 240, "DevPrim must stay 16-byte granular");

struct DevMat {                  // 64 B, same field order as mi355rt_material
    uint32_t kind; float albedo[3];
    float aux[3];  float p0;
    float p1;      float eta[3];
    float k[3];    uint32_t texture;   // MI355RT_MAT_TEXTURE: index into RenderParams.textures
};
static_assert(sizeof(DevMat) == 64, "DevMat");

struct DevTexture { const uint32_t* rgba8; uint32_t width, height; };   // one u32 per texel: r | g << 8 | b << 16 | a << 24 (little-endian RGBA8 bytes)

struct DevNode { float bmin[3]; uint32_t a; float bmax[3]; uint32_t b; };
static_assert(sizeof(DevNode) == 32, "DevNode");
constexpr uint32_t NODE_LINK_BITS = 26;
constexpr uint32_t NODE_END = (1u << NODE_LINK_BITS) - 1u;       // escape link of the last nodes of a walk
constexpr uint32_t NODE_MAX_LEAF = 63;                           // triangles per leaf record (6-bit count); fatter leaves become chunk chains
// LDS copy of the hot top of the node array in the state-machine kernel: one 1024-thread workgroup per CU owns the
// CU's whole 160 KiB (163 840 B); 5 104 nodes x 32 B = 163 328 B.
constexpr uint32_t LDS_NODE_CAP = 5104;

struct DevTri { float v0[3], e1[3], e2[3], n[3]; };
static_assert(sizeof(DevTri) == 48, "DevTri");

struct DevCamera { float position[3], forward[3], right[3], true_up[3], half_width, half_height; };

// 256 = one pixel's samples at the headline 256 spp: the lanes of a wave then share the camera ray and the first
// hit, so whole accept blocks are skipped wave-wide (measured: 2048 -> 27.1 ms, 512 -> 26.1, 256 -> 25.8 on cornell).
constexpr uint32_t BATCH_MIN = 128, BATCH_MAX = 256;   // paths a wave claims per global atomic (guided self-scheduling)
// The wavefront kernel claims runs of a FIXED 256 samples, aligned to 256 within the band: at the headline 256 spp a run is exactly
// one pixel, at 64 spp four whole pixels, at 4096 spp a sixteenth of one -- the rays a workgroup holds stay coherent, its passes
// less divergent.  Measured (profiles/r03_ab_wavefront_run_length.txt, 800x600x256): [128, 256] guided 28.96 / 17.76 ms
// (semesterbild / teapot), fixed 256 28.63 / 17.34, fixed 512 28.95 / 17.30, [256, 512] 29.31 / 17.62, fixed 1024 30.5 / 17.4.
constexpr uint32_t RUN_WAVEFRONT = 256, RUN_WAVEFRONT_MIN = RUN_WAVEFRONT, RUN_LIMIT = 2048;    // (MIN < RUN_WAVEFRONT would mean guided shrinking at the end of a shard)
constexpr uint32_t BLOCK_THREADS = 256;         // lockstep kernels
constexpr uint32_t BLOCK_THREADS_SM = 1024;     // state-machine kernels: 16 waves = 4 per SIMD = one workgroup per CU, sharing the LDS node copy
constexpr uint32_t WORK_SHARDS = 8;            // one work counter per XCD (power of two)
constexpr uint32_t WORK_SHARD_STRIDE = 32;     // u32 words between counters: one 128-B line each
constexpr uint32_t WAVE_TIME_WORDS = 6;        // diagnostic builds: per wave {start, end, paths, time work ran dry, iterations after, live lanes then}

struct RenderParams {
    const DevPrim* prims; const DevMat* mats; const DevNode* nodes; const DevTri* tris;
    const uint32_t* rows;        // processing row (the band's pixels are numbered in processing order) -> absolute y
    const float* sky; uint32_t sky_w, sky_h;   // equirect HDR skybox (RGB f32), null = constant miss colour
    const DevTexture* textures;  // images of the MI355RT_MAT_TEXTURE materials (indices validated at upload)
    float* radiance;             // 3 floats per band sample
    uint32_t* batch_counter;     // WORK_SHARDS counters (WORK_SHARD_STRIDE words apart): next unclaimed sample of each shard; zeroed per band
    unsigned long long* stats;   // [0] = paths started, [1] = rays traced
    unsigned long long* err;     // sticky per-context failure word (never reset by a render; the host compares it with what it has already
                                 // reported -- rt_api.cpp, report_device_error): bits 0..31 count the waves that gave up a bounded wait,
                                 // bits 32..39 collect WHICH waits (WAIT_*); the host adds which kernel it had launched
    unsigned long long* wave_times;  // diagnostic builds only: WAVE_TIME_WORDS u64 per wave; null otherwise
    uint32_t n_prims, n_mats;
    float miss[3];
    DevCamera cam;
    uint32_t width, height, spp, max_depth;
    float width_f, height_f, inv_width_rn, inv_height_rn;   // (float)width, (float)height and their correctly rounded reciprocals (host: 1.0f / x): the
                                                            // camera's u = x / width by div_by_rn() with every divisor-side operand in scalar registers
    uint32_t band_pixel0;        // first local pixel (row-major over the selected rows) of this band
    uint32_t band_samples;       // band pixels * spp  (< 2^31)
    uint32_t guided_div;         // run length = (left in the shard) / guided_div, clamped to the kernel's [RMIN, RMAX] (WorkCursorT)
    uint32_t shard_samples;      // samples per shard: ceil(band_samples / WORK_SHARDS), rounded up to a multiple of the kernel's longest run
    uint32_t seed_lo, seed_hi;
    uint32_t sample0;            // index of the first sample of this launch within its pixel (progressive rendering; 0 otherwise)
    uint32_t spp_mul, spp_shift, width_mul, width_shift;   // magic pairs for n / spp and n / width (n < 2^31)
    uint32_t trav_min;           // state-machine kernel: run BVH rounds while at least this many lanes are walking
    uint32_t inline_steps;       // state-machine kernel: box tests taken right at mesh setup (short walks skip the TRAV round trip)
    uint32_t lds_nodes;          // state-machine kernel (reference build): nodes [0, lds_nodes) are read from the workgroup's LDS copy (<= LDS_NODE_CAP)
    uint32_t spin_limit_idle;    // wavefront kernel: polls without progress before a wave gives up (SPIN_LIMIT_IDLE; a diagnostic hook lowers it)
    uint32_t spin_limit_entry;   // wavefront kernel: polls of one ring entry before a lane gives up (SPIN_LIMIT_ENTRY)
};
// The bounded waits a kernel can give up (RenderParams.err): an idle wave of the wavefront kernel that saw no progress in its workgroup; a lane
// whose ring entry was never written (pop) or never emptied (push); a wave that left because another wave of its workgroup had given up.
enum : uint32_t { WAIT_WF_IDLE = 1u, WAIT_WF_RING = 2u, WAIT_WF_FOLLOWED = 4u };
constexpr uint32_t SPIN_LIMIT_IDLE = 1u << 22;    // watchdog bounds: seconds of polling, never reached by a healthy launch
constexpr uint32_t SPIN_LIMIT_ENTRY = 1u << 20;

// Which counter-mode kernel serves a scene
enum : uint32_t {
    KERNEL_LOCKSTEP = 0,         // no mesh at the top level: every lane traces a whole ray per iteration
    KERNEL_LOCKSTEP_MESH = 1,    // same loop with the per-lane BVH walk inlined (A/B reference for the state machine)
    KERNEL_STATE_MACHINE = 2,    // wave-voted TRAV / TOP / SHADE blocks (scenes with meshes)
    KERNEL_LOCKSTEP_SIMPLE = 3,  // KERNEL_LOCKSTEP for scenes whose materials are only Lambertian (solid) / Emissive / Null
    KERNEL_STATE_MACHINE_FIXAABB = 4,   // KERNEL_STATE_MACHINE with the opt-in slab test (MI355RT_FLAG_FIXED_AABB)
    KERNEL_RETIRED_5 = 5,        // (round 2's LDS walk pool and its fixed-AABB form; removed in round 5, numbers kept so that the others stay what
    KERNEL_RETIRED_6 = 6,        //  logs and tests of earlier rounds call them; no library holds them)
    KERNEL_WAVEFRONT = 7,        // path state in LDS, stages as queues: every pass runs with (nearly) full lanes (scenes with meshes)
    KERNEL_WAVEFRONT_FIXAABB = 8,
    KERNEL_LOCKSTEP_NOSPEC = 9,  // KERNEL_LOCKSTEP without the metal and dielectric branches: 72 VGPRs = 7 waves per SIMD (veach-mis)
    KERNEL_WAVEFRONT_NOMETAL = 10,   // KERNEL_WAVEFRONT without the metal branch (teapot, semesterbild)
    KERNEL_WAVEFRONT_MESHFREE = 11,  // the wavefront for lists WITHOUT a mesh, no metal / dielectric: material-sorted SHADE passes for scenes whose materials diverge (veach-mis)
    KERNEL_WAVEFRONT_NOMETAL_IDENT = 12,   // KERNEL_WAVEFRONT_NOMETAL for lists whose meshes are all untransformed (teapot): mesh_setup without its matrix products
    KERNEL_WAVEFRONT_NOMETAL_SHALLOW = 13, // KERNEL_WAVEFRONT_NOMETAL for lists whose meshes all have small trees (semesterbild): WALK passes of 3 x 6 instead of 3 x 8 box tests
    KERNEL_LOCKSTEP_SIMPLE_QC = 14,  // KERNEL_LOCKSTEP_SIMPLE for lists of quads and cubes only (cornell): no sphere / plane run checks in the walk, a two-way finish_hit
    KERNEL_VARIANTS = 15
};
// Primitive-kind sets (bit k = kind MI355RT_PRIM_k may occur), like the material sets below: the run checks and record branches of the other kinds are compiled out.
constexpr uint32_t PRIMS_ALL = 0xFFFFFFFFu;
constexpr uint32_t PRIMS_QUAD_CUBE = (1u << MI355RT_PRIM_QUAD) | (1u << MI355RT_PRIM_CUBE);
inline uint32_t prims_of_variant(uint32_t variant) { return variant == KERNEL_LOCKSTEP_SIMPLE_QC ? PRIMS_QUAD_CUBE : PRIMS_ALL; }
constexpr uint32_t WF_SHALLOW_NODES = 4096;      // "small tree": at most this many BVH nodes per mesh (a median-split tree of <= ~6 000 triangles, depth <= 11)
// Material sets (bit k = kind MI355RT_MAT_k may occur) the kernels are instantiated for; set_scene picks, per kernel family, the
// most pruned instantiation whose set covers the scene's materials.  The branches compiled out set the register peak.
constexpr uint32_t MATBIT(uint32_t kind) { return 1u << kind; }
constexpr uint32_t MATS_ALL = (1u << MI355RT_MAT_KIND_COUNT) - 1u;
constexpr uint32_t MATS_TERMINAL = MATBIT(MI355RT_MAT_EMISSIVE) | MATBIT(MI355RT_MAT_NULL);                       // never scatter: in every set
constexpr uint32_t MATS_LAMBERT = MATS_TERMINAL | MATBIT(MI355RT_MAT_LAMBERT_SOLID);
constexpr uint32_t MATS_DIFFUSE = MATS_LAMBERT | MATBIT(MI355RT_MAT_LAMBERT_CHECKER) | MATBIT(MI355RT_MAT_TEXTURE) | MATBIT(MI355RT_MAT_PLASTIC);
constexpr uint32_t MATS_ROUGH = MATBIT(MI355RT_MAT_ROUGH_GGX) | MATBIT(MI355RT_MAT_ROUGH_BECKMANN);
constexpr uint32_t MATS_NO_METAL = MATS_ALL & ~MATBIT(MI355RT_MAT_METAL);
constexpr uint32_t MATS_NO_SPECULAR = MATS_ALL & ~(MATBIT(MI355RT_MAT_METAL) | MATBIT(MI355RT_MAT_DIELECTRIC));
inline uint32_t mats_of_variant(uint32_t variant) {
    return (variant == KERNEL_LOCKSTEP_SIMPLE || variant == KERNEL_LOCKSTEP_SIMPLE_QC) ? MATS_LAMBERT : (variant == KERNEL_LOCKSTEP_NOSPEC || variant == KERNEL_WAVEFRONT_MESHFREE) ? MATS_NO_SPECULAR
         : (variant == KERNEL_WAVEFRONT_NOMETAL || variant == KERNEL_WAVEFRONT_NOMETAL_IDENT || variant == KERNEL_WAVEFRONT_NOMETAL_SHALLOW) ? MATS_NO_METAL : MATS_ALL;
}

struct ResolveParams {
    const float* radiance;       // 3 floats per band sample
    uint32_t* out_packed;        // local pixels, 0x00RRGGBB
    float* out_linear;           // local pixels * 3, may be null
    float* accum;                // optional running per-pixel sums (float4 per local pixel): progressive rendering
    uint32_t accum_load;         // 1: start from accum[] (samples before this launch), 0: start from zero
    uint32_t band_pixel0, band_pixels, spp;
    float inv_spp;               // 1 / (samples accumulated so far including this launch)
    const uint32_t* out_row;     // processing row -> local output row (rt_api.cpp row_tables()); null: the band's pixels are in output order
    uint32_t width, width_mul, width_shift;   // image width and its magic pair (pixel -> processing row)
};

struct RefParams {               // MI355RT_RNG_REF: one lane per selected row
    const DevPrim* prims; const DevMat* mats; const DevNode* nodes; const DevTri* tris;
    const uint32_t* rows;
    const float* sky; uint32_t sky_w, sky_h;
    const DevTexture* textures;
    uint32_t* out_packed; float* out_linear;
    float* fold_stack;           // n_rows * max_depth * 3 floats (attenuation stack for tail-first folding)
    unsigned long long* stats;
    uint32_t n_prims, n_mats, n_rows;
    float miss[3];
    DevCamera cam;
    uint32_t width, height, spp, max_depth;
    uint32_t seed_lo, seed_hi;
};

// Diagnostic per-function entry points (tests only; mi355rt_debug_scatter / mi355rt_debug_hit in rt_api.cpp): one record per lane
// through exactly the device functions the render kernels call.
struct DebugScatterIn { uint32_t material, front_face; float rd[3], p[3], n[3]; uint32_t k0, k1, x, s, ray; };     // 16 words
struct DebugScatterOut { float scattered, o[3], d[3], atten[3], emitted[3], pad[3]; };                              // 16 words
struct DebugHitIn { float o[3], d[3]; };                                                                            // d is normalised once (Ray::new)
struct DebugHitOut { float p[3], n[3], t, material, front_face, hit, pad[2]; };                                     // 12 words
int launch_debug_scatter(const DevMat* mats, const DevTexture* textures, const DebugScatterIn* in, DebugScatterOut* out, uint32_t n, void* stream);
int launch_debug_hit(const DevPrim* prims, uint32_t n_prims, const DevNode* nodes, const DevTri* tris, const DebugHitIn* in, DebugHitOut* out, uint32_t n, void* stream);

// launchers (rt_kernels.hip); `stream` is a hipStream_t
int launch_render_ctr(const RenderParams& p, uint32_t variant, uint32_t grid_blocks, void* stream);
int launch_resolve(const ResolveParams& p, void* stream);
int launch_render_ref(const RefParams& p, void* stream);
int query_render_ctr_occupancy(uint32_t variant, int* blocks_per_cu, int* vgprs, int* sgprs);
bool render_ctr_variant_built(uint32_t variant);   // false for the retired mesh kernels in the product library (they live in the tests' -DMI355RT_REFS build)
constexpr uint32_t BLOCK_THREADS_WF = 768;                  // wavefront kernel: 12 waves per workgroup (measured against 8 / 10 / 14 / 16: rt_wavefront.h)
// Wavefront kernel, LDS budget of one workgroup (rt_wavefront.h): control words, one ring per queue, the path slots, and -- in what is
// left -- a copy of the first WF_LDS_NODES nodes of the (top-levels-first) node array.
// 832 slots: what fits beside seven rings (768 beside the eight there were: semesterbild +2.4 %, teapot +1.6 %); two workgroups per CU.
// (A copy of the top of the node array in the rest of the budget was measured in round 3, profiles/r03_ab_wavefront_lds_nodes.txt: 348 nodes beside
// 704 slots, or 2 110 nodes with one 16-wave workgroup per CU, run exactly as fast as the same geometry without the copy -- and every slot given up costs time.)
constexpr uint32_t WF_PATHS = 832, WF_SLOT_WORDS = 20, WF_RING = 1024, WF_QUEUES = 7, WF_CTRL_WORDS = 32;
constexpr uint32_t WF_FIXED_WORDS = WF_CTRL_WORDS + WF_QUEUES * WF_RING / 2u + WF_PATHS * WF_SLOT_WORDS;
constexpr uint32_t WF_LDS_BUDGET_WORDS = 163840u / 4u / 2u;
static_assert(WF_FIXED_WORDS <= WF_LDS_BUDGET_WORDS, "wavefront kernel LDS budget");
constexpr uint32_t STATS_WORDS = 40;                         // u64 device counters per render: [0] paths, [1] rays, the rest diagnostic builds only
inline bool is_wavefront(uint32_t variant) { return variant == KERNEL_WAVEFRONT || variant == KERNEL_WAVEFRONT_FIXAABB || variant == KERNEL_WAVEFRONT_NOMETAL || variant == KERNEL_WAVEFRONT_MESHFREE || variant == KERNEL_WAVEFRONT_NOMETAL_IDENT || variant == KERNEL_WAVEFRONT_NOMETAL_SHALLOW; }
// The mesh-free form runs 2 x 16 waves per CU at 64 VGPRs (8 per SIMD): veach-mis 16.71 -> 16.09 ms; the forms with the BVH walk lose a third
// there (42 spilled registers).  Waves per workgroup must be a multiple of 4: a workgroup's waves are dealt round-robin over the CU's
// four SIMDs, and with 10, 13 or 14 of them the second workgroup no longer fits the per-SIMD wave budget (measured: +40 %; this also explains
// round 2's "2 x 10 waves at 96 VGPRs" result).
constexpr uint32_t BLOCK_THREADS_WF_MESHFREE = 1024;
inline uint32_t block_threads_of(uint32_t variant) {
    return variant == KERNEL_WAVEFRONT_MESHFREE ? BLOCK_THREADS_WF_MESHFREE : is_wavefront(variant) ? BLOCK_THREADS_WF
         : (variant == KERNEL_STATE_MACHINE || variant == KERNEL_STATE_MACHINE_FIXAABB) ? BLOCK_THREADS_SM : BLOCK_THREADS;
}

}  // namespace mi355rt
