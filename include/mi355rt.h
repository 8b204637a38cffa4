/*
 * mi355rt.h -- C ABI of the MI355X-native render loop.
 *
 * This is the drop-in boundary for ONE call of the reference:
 *
 *     let buffer = render_scene(&scene, &camera, &render_settings);      // src/main.rs:57
 *     pub fn render_scene(scene: &Scene, camera: &Camera,
 *                         render_settings: &RenderSettings) -> Vec<u32>  // src/renderer.rs:67
 *
 * The reference has no FFI of its own, so the entry points below are
 * what a `#[repr(C)]` / `extern "C"` binding added at src/main.rs:57 would bind (the stub is in
 * INTEGRATION.md).  Every struct is plain-old-data, little-endian f32/u32, no pointers inside
 * arrays, and the caller owns every buffer it passes in.
 *
 * Conventions
 *   - return value 0 = OK, negative = error; nothing aborts, nothing throws across the ABI (every
 *     entry point of both libraries runs inside an exception barrier: a failed host allocation is
 *     MI355RT_ERR_OOM, any other C++ exception MI355RT_ERR_HIP / _IO with its text);
 *     mi355rt_last_error() returns a thread-local message for the last failure.
 *   - output layout == render_scene's Vec<u32>: width*height, row-major, row 0 = top,
 *     0x00RRGGBB (src/color.rs:87-93).
 *   - matrices are column-major 4x4 as glam::Mat4 stores them (x_axis, y_axis, z_axis, w_axis).
 *   - the top-level primitive array is walked in array order, exactly as
 *     HittableList::hit walks `objects` (src/hittable.rs:45-58): order changes tie-breaks.
 *   - the libraries read NO environment variables: everything that selects behaviour is in the structs below.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355RT_ABI_VERSION 4u   /* 2: mi355rt_scene.textures, MI355RT_MAT_TEXTURE; 3: mi355rt_context_check exported, quads must carry a (near-)unit normal;
                                    4: mi355rt_context_set_share exported; MI355RT_RNG_CTR draws from pcg4d (other numbers than versions 1-3, same distribution) */

/* ---- error codes ------------------------------------------------------------------------- */
#define MI355RT_OK               0
#define MI355RT_ERR_INVALID     -1   /* malformed scene / settings / options                    */
#define MI355RT_ERR_NO_DEVICE   -2   /* no usable HIP device (there is NO CPU fallback)          */
#define MI355RT_ERR_HIP         -3   /* a HIP runtime call or a kernel failed                    */
#define MI355RT_ERR_OOM         -4   /* host or device allocation failed                         */
#define MI355RT_ERR_IO          -5   /* file could not be read / parsed (host-side loaders)      */
#define MI355RT_ERR_UNSUPPORTED -6   /* feature flagged in the ABI but not built                   */

/* ---- camera: src/camera.rs:4-11 (the fields of `Camera`, computed by Camera::new on the host) */
typedef struct mi355rt_camera {
    float position[3];
    float forward[3];
    float right[3];
    float true_up[3];
    float half_width;
    float half_height;
} mi355rt_camera;

/* ---- render settings: src/tungsten/parser.rs:191-197 (`RenderSettings`) ---------------------- */
typedef struct mi355rt_settings {
    uint32_t width;
    uint32_t height;
    uint32_t samples_per_pixel;
    uint32_t max_depth;
} mi355rt_settings;

/* ---- materials: the eight `impl Material` types (src/material.rs, src/tungsten/materials.rs) - */
enum {
    MI355RT_MAT_LAMBERT_SOLID   = 0, /* material.rs:47-71, AlbedoKind::Solid     albedo           */
    MI355RT_MAT_LAMBERT_CHECKER = 1, /* material.rs:47-71, AlbedoKind::Checked   albedo=on, aux=off, p0=inv_scale */
    MI355RT_MAT_METAL           = 2, /* material.rs:87-110                       albedo, p0=fuzz  */
    MI355RT_MAT_DIELECTRIC      = 3, /* material.rs:122-162                      p0=refractive_index */
    MI355RT_MAT_EMISSIVE        = 4, /* material.rs:169-192                      albedo=color     */
    MI355RT_MAT_PLASTIC         = 5, /* tungsten/materials.rs:29-65              albedo, p0=ior   */
    MI355RT_MAT_ROUGH_GGX       = 6, /* tungsten/materials.rs:306-377 (Ggx)      albedo, p0=roughness, eta, k */
    MI355RT_MAT_ROUGH_BECKMANN  = 7, /* tungsten/materials.rs:306-377 (Beckmann) albedo, p0=roughness, eta, k */
    MI355RT_MAT_NULL            = 8, /* material.rs:229-252 (never scatters, never emits)         */
    MI355RT_MAT_TEXTURE         = 9, /* tungsten/parser.rs:199-243 TextureMaterial: Lambert bounce, albedo * texel looked up by the
                                        hit NORMAL (equirect, nearest);  albedo, p0=h_offset, texture=index into scene.textures.
                                        No loader path of the reference produces it (parser.rs:315-424 never yields Texture);
                                        a host that builds its Scene in code can. */
    MI355RT_MAT_KIND_COUNT      = 10
};

typedef struct mi355rt_material {      /* 64 bytes */
    uint32_t kind;
    float    albedo[3];
    float    aux[3];
    float    p0;
    float    p1;
    float    eta[3];                   /* MetalType::ior_k().0, tungsten/materials.rs:115-152      */
    float    k[3];                     /* MetalType::ior_k().1                                     */
    uint32_t texture;                  /* MI355RT_MAT_TEXTURE: index into mi355rt_scene.textures; 0 otherwise */
} mi355rt_material;

/* An 8-bit RGBA image as `image::RgbaImage` holds it (parser.rs:201): row-major, row 0 = top, 4 bytes per pixel. */
typedef struct mi355rt_texture {
    const uint8_t* rgba8;
    uint32_t width, height;
} mi355rt_texture;

/* ---- top-level primitives: the five `impl Hittable` types ------------------------------------ */
enum {
    MI355RT_PRIM_SPHERE = 0, /* src/objects/sphere.rs:9-13   data: center[3], radius                */
    MI355RT_PRIM_PLANE  = 1, /* src/objects/plane.rs:9-13    data: p1[3], normal[3] (unit)          */
    MI355RT_PRIM_QUAD   = 2, /* src/tungsten/objects/quad.rs:10-21  data: base[3], edge0[3], edge1[3],
                                normal[3], d, inv_edge0_len_sq, inv_edge1_len_sq                    */
    MI355RT_PRIM_CUBE   = 3, /* src/objects/cube.rs:11-17    data: object_to_world[16], world_to_object[16] */
    MI355RT_PRIM_MESH   = 4, /* src/mesh/mesh_object.rs:17-22 data: object_to_world[16], world_to_object[16];
                                `mesh` indexes mi355rt_scene.meshes                                 */
    MI355RT_PRIM_KIND_COUNT = 5
};

typedef struct mi355rt_primitive {     /* 144 bytes */
    uint32_t kind;
    uint32_t material;                 /* index into mi355rt_scene.materials                       */
    uint32_t mesh;                     /* MI355RT_PRIM_MESH only                                   */
    uint32_t _pad;
    float    data[32];
} mi355rt_primitive;

/* ---- mesh payload: src/mesh/triangle.rs:5-11 and src/acceleration/bvh.rs:7-12 ----------------- */
typedef struct mi355rt_triangle {      /* object space, 48 bytes; `material` lives on the primitive */
    float v0[3];
    float v1[3];
    float v2[3];
    float normal[3];                   /* Triangle::new, triangle.rs:14-25                          */
} mi355rt_triangle;

/* One BVHNode, flattened by any visitor.  Indices are relative to the owning mesh's
 * first_node / first_index.  Inner node: index_count == 0, left/right = child node indices.
 * Leaf: index_count > 0 and [first_index, first_index+index_count) is its `triangle_indices`
 * list (values index the mesh's triangles).  Node 0 of a mesh is its root.                       */
typedef struct mi355rt_bvh_node {      /* 40 bytes */
    float    bmin[3];
    float    bmax[3];
    uint32_t left;
    uint32_t right;
    uint32_t first_index;
    uint32_t index_count;
} mi355rt_bvh_node;

typedef struct mi355rt_mesh {
    uint32_t first_triangle, triangle_count;   /* range in mi355rt_scene.triangles               */
    uint32_t first_node, node_count;           /* range in mi355rt_scene.nodes                    */
    uint32_t first_index, index_count;         /* range in mi355rt_scene.tri_indices              */
    uint32_t max_depth;                        /* depth of the deepest node (root = 0); 0 = unknown */
    uint32_t _pad;
} mi355rt_mesh;

/* ---- the scene: src/scene.rs:6-10 flattened --------------------------------------------------- */
typedef struct mi355rt_scene {
    const mi355rt_primitive* primitives;  uint32_t n_primitives;
    const mi355rt_material*  materials;   uint32_t n_materials;
    const mi355rt_mesh*      meshes;      uint32_t n_meshes;
    const mi355rt_triangle*  triangles;   uint32_t n_triangles;
    const mi355rt_bvh_node*  nodes;       uint32_t n_nodes;
    const uint32_t*          tri_indices; uint32_t n_tri_indices;
    float        miss_color[3];           /* Color::GRAY at HEAD, src/renderer.rs:61             */
    uint32_t     sky_width, sky_height;   /* equirect HDR skybox, src/renderer.rs:40-54; 0 = none */
    const float* sky_rgb;                 /* sky_width*sky_height*3 f32, row 0 = top; NULL = constant miss_color */
    const mi355rt_texture* textures;      uint32_t n_textures;   /* images of the MI355RT_MAT_TEXTURE materials (ABI 2) */
} mi355rt_scene;

/* ---- options that have no counterpart in the reference ---------------------------------------- */
enum {
    MI355RT_RNG_CTR = 0, /* counter-based per-ray generator (pcg4d since round 5, Philox4x32-10 before) addressed by (row y; x, sample, ray, block).
                            GPU-native default; any tiling gives bit-identical images.           */
    MI355RT_RNG_REF = 1  /* replay of the reference stream: StdRng::seed_from_u64(y) shared by a
                            whole row (src/renderer.rs:91). One lane per row -- validation only.   */
};

/* options.flags */
/* Opt-in fix, never the default (it changes the image): the BVH slab test misses a box only when t_max < t_min.
 * The reference tests t_max <= t_min (src/acceleration/aabb.rs:41), so boxes of zero thickness -- every leaf of
 * axis-aligned flat geometry -- are never entered and their triangles are invisible.
 * Counter-mode RNG only. */
#define MI355RT_FLAG_FIXED_AABB 1u

typedef struct mi355rt_options {
    uint32_t abi_version;     /* MI355RT_ABI_VERSION                                               */
    uint32_t rng_mode;        /* MI355RT_RNG_*                                                     */
    uint64_t seed;            /* 0 reproduces the reference: row key = y + seed                    */
    /* Row selection. Rows are dealt in strips of `strip_rows` rows; this call renders the strips
     * with (strip_index % n_parts) == part, restricted to [row_begin, row_end).  The output
     * buffers then hold ONLY those rows, packed in ascending row order.
     * {0, height, 1, 1, 0} renders the whole image.  row_end == 0 means `height`.               */
    uint32_t row_begin, row_end;
    uint32_t strip_rows, n_parts, part;
    uint32_t flags;             /* MI355RT_FLAG_*; 0 reproduces the reference                      */
    uint64_t workspace_bytes; /* cap for the per-sample radiance workspace in HBM; 0 = default (32 GiB,
                                 of which only width*rows*spp*12 bytes are allocated)             */
} mi355rt_options;

typedef struct mi355rt_stats {
    double   render_kernel_ms;   /* sum over bands of the path-tracing kernel, HIP events          */
    double   resolve_kernel_ms;  /* sum over bands of the ordered sum + gamma + pack kernel        */
    double   total_ms;           /* first launch -> last kernel done (device timeline)             */
    uint64_t samples;            /* camera paths started                                           */
    uint64_t rays;               /* trace_ray invocations that intersected the scene               */
    uint32_t rows_rendered;
    uint32_t bands;
    uint32_t grid_blocks, block_threads;
    uint32_t kernel_vgprs, kernel_sgprs;   /* 0 if the runtime does not report them                */
} mi355rt_stats;

/* ---- one-shot call: host buffers in, host buffers out (what src/main.rs:57 would call) -------- */
int mi355rt_render(const mi355rt_scene* scene, const mi355rt_camera* camera,
                   const mi355rt_settings* settings, const mi355rt_options* options_or_null,
                   uint32_t* out_packed_rgb,       /* rows_rendered*width, 0x00RRGGBB              */
                   float*    out_linear_rgb_or_null,/* rows_rendered*width*3, pre-gamma mean       */
                   mi355rt_stats* stats_or_null);

/* ---- resident-scene API: upload once, render many times, device-side outputs ------------------ */
typedef struct mi355rt_context mi355rt_context;

int  mi355rt_context_create(int hip_device, mi355rt_context** out_ctx);
void mi355rt_context_destroy(mi355rt_context* ctx);
/* Uploads the scene (every array is copied: the caller's buffers may be freed afterwards) and picks the kernel for it.  BLOCKING, and it may
 * LAUNCH: for mesh-free scenes that mix a rough conductor with another scattering material a probe render of the same view (<= 64 pixels across,
 * <= 4 samples per pixel; deterministic, a fraction of a millisecond) runs on the NULL stream and is waited for -- its rays per path decide
 * between the lockstep and the wavefront kernel.  The probe stays out of the timing pool (mi355rt_context_set_timing).  Like every entry point
 * that looks at the context's error word, set_scene returns a pending watchdog failure of an EARLIER asynchronous render on this context
 * (MI355RT_ERR_HIP, once) instead of proceeding; call it again.                                                                          */
int  mi355rt_context_set_scene(mi355rt_context* ctx, const mi355rt_scene* scene,
                               const mi355rt_camera* camera, const mi355rt_settings* settings);
/* Number of rows the given options select (so callers can size their buffers).                  */
int  mi355rt_rows_selected(const mi355rt_settings* settings, const mi355rt_options* options_or_null,
                           uint32_t* out_rows);
/* Enqueue a render on `hip_stream` (a hipStream_t, or NULL for the default stream).  Outputs are
 * DEVICE pointers.  The call returns after the work is enqueued unless `stats_or_null` is given,
 * in which case it synchronises the stream to read the timers.                                   */
int  mi355rt_context_render(mi355rt_context* ctx, const mi355rt_options* options_or_null,
                            void* d_out_packed_rgb, void* d_out_linear_rgb_or_null,
                            void* hip_stream, mi355rt_stats* stats_or_null);

/* Frames in flight.  A render's path-tracing kernel is PERSISTENT: it launches as many workgroups as the device holds and each keeps claiming
 * samples until the frame is done -- so the end of every launch is a tail in which ever fewer paths keep the device busy, and a second frame
 * enqueued on another stream only trickles in as the first frame's workgroups retire.  For a large frame the tail is noise; for a small one -- the
 * eighth of an 800 x 600 image that one of 8 GPUs renders, src/renderer.rs:87-103 sharded by rows -- it is a third of the launch.  A caller that
 * renders a SEQUENCE of frames (an animation, progressive refinement, the bench) can hide it: keep F frames in flight, each on its own context
 * (own workspace) and its own stream, and tell every one of these contexts that it has 1 / share_of of the device: its kernels then launch that
 * fraction of the resident grid, F launches are co-resident, and a draining frame shares every SIMD with frames in their steady state.
 * share_of = 1 (the default) is the whole device; valid: 1 .. 16.  F > 4 streams buy nothing (HIP multiplexes streams onto 4 hardware queues by
 * default); F = 4 with share_of = 4 (mesh scenes with short walks: 2) measured best -- DESIGN.md section 7.  The image does not depend on it.
 * Takes effect with the next render on the context.                                                                                     */
int  mi355rt_context_set_share(mi355rt_context* ctx, uint32_t share_of);

/* Completion check of the asynchronous form.  render_scene is infallible (src/renderer.rs:67): it either returns the whole image or
 * panics.  mi355rt_context_render with stats == NULL only ENQUEUES work, so a failure inside a kernel -- a wave of the wavefront
 * kernel that gives up a bounded wait leaves paths unfinished -- cannot be returned by that call.  It is never lost: every render
 * leaves the context's error word behind in stream order, and the next of
 *   - mi355rt_context_check (waits for every render enqueued on this context so far),
 *   - mi355rt_context_read_timing,
 *   - the next mi355rt_context_render* on this context (without waiting: it sees the renders that have finished),
 *   - the same call, when it was given stats and therefore synchronises,
 * returns MI355RT_ERR_HIP once per failed render ("kernel watchdog ... that image is incomplete").  A caller that consumes images
 * from the asynchronous form calls mi355rt_context_check after synchronising its stream and before using them.                     */
int  mi355rt_context_check(mi355rt_context* ctx);

/* Progressive rendering (the sample loop of src/renderer.rs:93-101 cut into chunks): trace samples
 * [sample_begin, sample_end) of every selected pixel and add them, in sample order, to the running
 * sums in `d_accum` (DEVICE, 4 floats per selected pixel, row-major over the selected rows; read only
 * when sample_begin > 0, always written).  The outputs hold the image of the first `sample_end`
 * samples (sum * 1/sample_end, renderer.rs:103).  Because every draw is addressed by (row, x, sample,
 * ray) and the f32 additions happen in the same order, a sequence of calls covering 0..N is
 * bit-identical to one mi355rt_context_render with samples_per_pixel == N -- for any chunking.
 * settings.samples_per_pixel is not consulted.  MI355RT_RNG_CTR only.                              */
int  mi355rt_context_render_progressive(mi355rt_context* ctx, const mi355rt_options* options_or_null,
                                        uint32_t sample_begin, uint32_t sample_end, void* d_accum,
                                        void* d_out_packed_rgb, void* d_out_linear_rgb_or_null,
                                        void* hip_stream, mi355rt_stats* stats_or_null);

/* mi355rt_render over several GPUs from ONE host process (the reference's host is a single `main`):
 * row strips of options.strip_rows rows (0 -> 4) are dealt round-robin over `hip_devices`, each device
 * renders its strips with the full scene resident and copies them into the caller's image; no
 * collective is involved.  options.n_parts / part must be left 0 (row_begin / row_end still select a
 * window, and the outputs then hold only that window).  Bit-identical to the one-device image.  A
 * device may be listed more than once (testing on a one-GPU machine).                                */
int  mi355rt_render_multi(const mi355rt_scene* scene, const mi355rt_camera* camera,
                          const mi355rt_settings* settings, const mi355rt_options* options_or_null,
                          const int* hip_devices, uint32_t n_devices,
                          uint32_t* out_packed_rgb, float* out_linear_rgb_or_null, mi355rt_stats* stats_or_null);

/* One-shot progressive render with HOST buffers: mi355rt_render in chunks of `chunk_spp` samples.  After
 * every chunk `on_chunk_or_null(user, samples_done, samples_total, out_packed_rgb)` sees the image so far
 * (what the reference's preview window, src/main.rs:60-75, would show); a non-zero return stops early and
 * leaves the image of `samples_done` samples in the outputs.  The final image equals mi355rt_render's.   */
typedef int (*mi355rt_progress_fn)(void* user, uint32_t samples_done, uint32_t samples_total, const uint32_t* packed_rgb);
int  mi355rt_render_progressive(const mi355rt_scene* scene, const mi355rt_camera* camera,
                                const mi355rt_settings* settings, const mi355rt_options* options_or_null,
                                uint32_t chunk_spp, mi355rt_progress_fn on_chunk_or_null, void* user,
                                uint32_t* out_packed_rgb, float* out_linear_rgb_or_null, mi355rt_stats* stats_or_null);

/* Kernel timing without extra synchronisation: while enabled, every mi355rt_context_render call that
 * passes stats == NULL records HIP events around its kernels on the caller's stream.  After the
 * caller has synchronised that stream, read_timing returns the summed kernel durations and the
 * number of (path tracing + resolve) launch pairs since the last read, and resets the pool.        */
int  mi355rt_context_set_timing(mi355rt_context* ctx, int enable);
int  mi355rt_context_read_timing(mi355rt_context* ctx, double* render_kernel_ms, double* resolve_kernel_ms,
                                 uint32_t* launches);

const char* mi355rt_last_error(void);
uint32_t    mi355rt_abi_version(void);

/* =================================================================================================
 * Host-side helpers (libmi355rt_host.so, pure CPU).  They stand in for the parts of the Rust host
 * that cannot be built here: the producers of the arrays above.
 * ================================================================================================= */

/* BVHNode::new (src/acceleration/bvh.rs:15-76) over object-space triangles: median split on the
 * largest-extent axis, leaf when <= 4 triangles or depth >= 25.  The centroid sort restates Rust's
 * slice::sort_unstable_by (ipnsort, Rust 1.81+) including its order of equal keys, so the arrays are the tree
 * the reference's own BVHNode::new builds (csrc/host/rust_sort_unstable.hpp).
 * Two-call pattern: pass NULL arrays to get the counts.                                           */
int mi355rt_bvh_build(const mi355rt_triangle* triangles, uint32_t n_triangles,
                      mi355rt_bvh_node* out_nodes, uint32_t* inout_n_nodes,
                      uint32_t* out_indices, uint32_t* inout_n_indices,
                      uint32_t* out_max_depth);

/* load_scene_from_json (src/tungsten/parser.rs:245-815) + Camera::new (src/camera.rs:14-31) +
 * Mesh::from_obj / from_wo3 (src/mesh/mesh_object.rs:59-259).  Overrides replace the values parsed
 * at parser.rs:260-285 (0 = keep the file's value).                                               */
typedef struct mi355rt_loaded_scene mi355rt_loaded_scene;

typedef struct mi355rt_load_overrides {
    uint32_t width, height, samples_per_pixel, max_depth;
    uint32_t skip_unknown_primitives;  /* 0 = hard error like serde (parser.rs:135-165), 1 = skip   */
    uint32_t wo3_four_index_stride;    /* 0 = the reference's reader, which steps 3 u32 per triangle through a file that stores 4
                                        *     (mesh_object.rs:190-192: ~1/4 of the triangles survive);
                                        * 1 = opt-in fix: read (v0, v1, v2, material) per triangle.  Changes the image.               */
} mi355rt_load_overrides;

int  mi355rt_scene_load_json(const char* json_path, const mi355rt_load_overrides* overrides_or_null,
                             mi355rt_loaded_scene** out_scene);
void mi355rt_scene_free(mi355rt_loaded_scene* s);
const mi355rt_scene*    mi355rt_loaded_scene_get(const mi355rt_loaded_scene* s);
const mi355rt_camera*   mi355rt_loaded_scene_camera(const mi355rt_loaded_scene* s);
const mi355rt_settings* mi355rt_loaded_scene_settings(const mi355rt_loaded_scene* s);

/* save_image's pixel conversion (src/renderer.rs:125-143) into an 8-bit RGB PNG.                  */
int mi355rt_write_png(const char* path, const uint32_t* packed_rgb, uint32_t width, uint32_t height);

/* The pre-gamma f32 image (out_linear_rgb) as a little-endian Portable FloatMap, for parity tooling.  */
int mi355rt_write_pfm(const char* path, const float* linear_rgb, uint32_t width, uint32_t height);
/* The same image as an OpenEXR file: scan-line, three 32-bit FLOAT channels (B, G, R), uncompressed, rows top-down. */
int mi355rt_write_exr(const char* path, const float* linear_rgb, uint32_t width, uint32_t height);

const char* mi355rt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
