// rt_api.cpp -- implementation of the device half of include/mi355rt.h (libmi355rt.so).
//
// Replaces `render_scene(&scene, &camera, &render_settings) -> Vec<u32>` (src/renderer.rs:67,
// called from src/main.rs:57).  There is NO CPU fallback in this library: without a HIP device every
// render entry point returns MI355RT_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/mi355rt.h"
#include "rt_device.h"

using namespace mi355rt;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
int fail_noexcept(int code, const char* msg) noexcept { try { g_err.assign(msg); } catch (...) { g_err.clear(); } return code; }
// The exception barrier of every extern "C" entry point below (mi355rt.h: "nothing aborts, nothing throws across the ABI"; the caller may
// be a Rust frame -- src/renderer.rs:67 is called from src/main.rs:57 -- into which a C++ exception must not unwind):
// std::bad_alloc / std::length_error -> MI355RT_ERR_OOM, anything else -> MI355RT_ERR_HIP with what() in mi355rt_last_error().
template <class F> int guard(F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail_noexcept(MI355RT_ERR_OOM, "host allocation failed (std::bad_alloc)"); }
    catch (const std::length_error&) { return fail_noexcept(MI355RT_ERR_OOM, "host allocation failed (std::length_error)"); }
    catch (const std::exception& e) {
        try { return fail(MI355RT_ERR_HIP, std::string("unexpected C++ exception: ") + e.what()); } catch (...) { return fail_noexcept(MI355RT_ERR_HIP, "unexpected C++ exception"); }
    }
    catch (...) { return fail_noexcept(MI355RT_ERR_HIP, "unexpected C++ exception"); }
}
#define HIP_TRY(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? MI355RT_ERR_OOM : MI355RT_ERR_HIP, \
         std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct RowSel { std::vector<uint32_t> rows; };

int select_rows(const mi355rt_settings& st, const mi355rt_options* o, RowSel& sel) {
    uint32_t rb = 0, re = st.height, strip = 1, parts = 1, part = 0;
    if (o) {
        if (o->abi_version != MI355RT_ABI_VERSION) return fail(MI355RT_ERR_INVALID, "options.abi_version mismatch");
        rb = o->row_begin; re = o->row_end ? o->row_end : st.height;
        strip = o->strip_rows ? o->strip_rows : 1; parts = o->n_parts ? o->n_parts : 1; part = o->part;
        if (o->rng_mode != MI355RT_RNG_CTR && o->rng_mode != MI355RT_RNG_REF) return fail(MI355RT_ERR_INVALID, "options.rng_mode");
    }
    if (re > st.height || rb > re || part >= parts) return fail(MI355RT_ERR_INVALID, "row selection out of range");
    sel.rows.clear();
    for (uint32_t y = rb; y < re; ++y) if ((y / strip) % parts == part) sel.rows.push_back(y);
    return MI355RT_OK;
}

// The three row tables of one selection, n entries each: [0, n) natural = absolute y of local output row j; [n, 2n) processing = absolute y
// of the row processed jp-th; [2n, 3n) out_row = the local output row that processing row jp is.  `cost` (per absolute image row, may be
// empty) orders the processing: rows sorted by decreasing cost (stable: equal costs keep image order) and dealt round-robin over `groups`
// consecutive ranges -- the launch's work shards (x bands) -- so that every range runs from its dearest rows to its cheapest.
void row_tables(const std::vector<uint32_t>& rows, const std::vector<float>& cost, uint32_t groups, std::vector<uint32_t>& out) {
    const size_t n = rows.size();
    out.resize(3 * n);
    std::vector<uint32_t> sorted(n);
    for (size_t j = 0; j < n; ++j) sorted[j] = (uint32_t)j;
    if (!cost.empty())
        std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) {
            const float ca = rows[a] < cost.size() ? cost[rows[a]] : 0.f, cb = rows[b] < cost.size() ? cost[rows[b]] : 0.f;
            return ca > cb; });
    groups = std::max(1u, std::min<uint32_t>(groups, (uint32_t)std::max<size_t>(n, 1)));
    size_t jp = 0;
    for (uint32_t g = 0; g < groups && !cost.empty(); ++g)
        for (size_t k = g; k < n; k += groups, ++jp) { out[n + jp] = rows[sorted[k]]; out[2 * n + jp] = sorted[k]; }
    for (size_t j = 0; j < n; ++j) {
        out[j] = rows[j];
        if (cost.empty()) { out[n + j] = rows[j]; out[2 * n + j] = (uint32_t)j; }
    }
}

int check_settings(const mi355rt_settings* st) {
    if (!st) return fail(MI355RT_ERR_INVALID, "settings is null");
    if (st->width == 0 || st->height == 0 || st->samples_per_pixel == 0) return fail(MI355RT_ERR_INVALID, "width/height/spp must be > 0");
    if ((uint64_t)st->width * st->height >= (1ull << 31)) return fail(MI355RT_ERR_INVALID, "image too large");
    if (st->width >= (1u << 24) || st->height >= (1u << 24)) return fail(MI355RT_ERR_INVALID, "width/height must be < 2^24 (x as f32 is exact, renderer.rs:96)");
    if (st->samples_per_pixel >= (1u << 30)) return fail(MI355RT_ERR_INVALID, "samples_per_pixel too large");
    return MI355RT_OK;
}

// q = n / d for every n < 2^31 as umulhi(n, mul) >> shift (mul == 0 encodes d == 1).  s = ceil(log2 d),
// mul = ceil(2^(31+s) / d) < 2^32, shift = s - 1.
void magic_div(uint32_t d, uint32_t& mul, uint32_t& shift) {
    if (d <= 1) { mul = 0; shift = 0; return; }
    uint32_t s = 0; while ((1ull << s) < d) ++s;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + s);
    mul = (uint32_t)((num + d - 1) / d); shift = s - 1;
}

template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int ensure(size_t count) {
        if (count <= n && p) return MI355RT_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (count == 0) count = 1;
        HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return MI355RT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct mi355rt_context;
static int report_device_error(mi355rt_context* ctx, bool this_render = false);

struct mi355rt_context {
    int device = 0;
    int cu_count = 0;
    int blocks_per_cu[KERNEL_VARIANTS] = {}, vgprs[KERNEL_VARIANTS] = {}, sgprs = 0;
    uint32_t variant = KERNEL_LOCKSTEP;  // chosen per scene in set_scene
    bool has_mesh = false;
    uint32_t grid_div = 1;               // this context launches 1 / grid_div of the grid that fills the device: the caller keeps grid_div frames in flight, each on its own
                                         // context and stream, so that their persistent kernels are co-resident (see render_samples; mi355rt_context_set_share)
    uint32_t guided_mult = 16;           // run length = (left in shard) / (guided_mult * waves per shard); 16 measured best at 1/8-image launches
    uint32_t inline_steps = 0;           // 1 when several meshes share the list (many rays miss a mesh's root box: teapot +5..12 %), 0 for a single mesh (semesterbild -10 % otherwise)
    uint32_t trav_min = 24;              // measured optimum 24-32 on semesterbild / teapot (tools/ab_kernel.py)
    bool have_scene = false;
    mi355rt_settings settings{};
    DevCamera cam{};
    float miss[3] = {0.5f, 0.5f, 0.5f};
    uint32_t n_prims = 0, n_mats = 0; size_t n_nodes = 0;
    DevBuf<DevPrim> prims; DevBuf<DevMat> mats; DevBuf<DevNode> nodes; DevBuf<DevTri> tris;
    DevBuf<uint32_t> rows; DevBuf<float> radiance; DevBuf<uint32_t> counters; DevBuf<unsigned long long> stats;
    DevBuf<float> fold_stack;
    DevBuf<float> sky; uint32_t sky_w = 0, sky_h = 0;      // equirect HDR skybox (renderer.rs:40-54); sky_w == 0: none
    DevBuf<DevTexture> textures; DevBuf<uint32_t> texels;  // MI355RT_MAT_TEXTURE images: table + all texels in one buffer
    DevBuf<unsigned long long> wave_times; uint32_t wave_times_n = 0;   // diagnostics (MI355RT_WAVE_TIMES=1)
    std::vector<uint32_t> rows_host;     // the selected rows (absolute y, ascending = the order of the output buffer)
    std::vector<uint32_t> tables_host;   // source of the async upload of the three row tables (row_tables()); must outlive the copy
    bool rows_valid = false;             // ctx->rows already holds the tables of rows_host (same selection and grouping as the last call)
    // Processing order (an experiment of round 4, opt-in).  The persistent kernels hand out a band's samples front to back, and a
    // launch ends with the paths of the rows handed out LAST.  The idea: process the rows in order of DECREASING expected cost -- cheap rows
    // (sky) last -- dealt over the work shards so that every shard ends with cheap rows, and a launch does not end on its longest paths.  The image cannot change: draws are keyed by absolute row / x /
    // sample, the resolve kernel writes each pixel where it belongs (ResolveParams.out_row).  `row_cost` = rays per path of each image
    // row, measured by set_scene with a small probe render of the same view; empty = natural order.
    std::vector<float> row_cost;
    int knob_row_order = -1;             // diagnostic knob "row_order": 1 on; -1 / 0 off (NOT shipped as a default: no measured gain, see set_scene)
    uint32_t order_groups = 0;           // how many groups the cached tables were dealt over (work shards x bands)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // The workspaces (rows, radiance, counters, stats) belong to one render at a time.  `done` is recorded behind the last
    // operation of every render; a render enqueued on ANOTHER stream waits on it first, so two streams can never touch
    // the workspaces concurrently (same stream: in-order execution already guarantees it).
    hipEvent_t done = nullptr; hipStream_t last_stream = nullptr; bool have_last = false;
    bool want_wave_times = false;        // diagnostic knob "wave_times" (stamps builds)
    // Sticky error word.  A wave of the wavefront kernel that gives up a bounded wait (RenderParams.spin_limit_*) adds 1 to `errword`
    // and leaves paths unfinished.  The word is never reset by a render; its value is copied into the pinned host word `h_err`
    // behind the kernels of EVERY render (8 bytes, in stream order, before `done`), and whichever entry point looks next --
    // the same call when it synchronises for stats, mi355rt_context_check, mi355rt_context_read_timing, or the next render on
    // this context -- compares it with what has been reported and returns MI355RT_ERR_HIP once per failed render.
    DevBuf<unsigned long long> errword; unsigned long long* h_err = nullptr; unsigned long long err_reported = 0;
    uint32_t spin_limit_idle = SPIN_LIMIT_IDLE, spin_limit_entry = SPIN_LIMIT_ENTRY;   // diagnostic knobs "spin_idle" / "spin_entry"
    uint32_t launched_variants = 0;      // bit v: kernel variant v was launched since the last failure report (names the kernel in the message)
    int forced_variant = -1;             // diagnostic knob "kernel": applied by set_scene when the scene allows it
    int knob_inline_steps = -1;          // diagnostic knob "inline_steps" (reference build's state machine)
    // timing pool (mi355rt_context_set_timing): event triples recorded around every kernel pair without
    // synchronising; mi355rt_context_read_timing sums them after the caller's own stream sync.
    bool timing = false;
    std::vector<hipEvent_t> pool; size_t pool_used = 0;
    uint32_t timed_launches = 0;
    hipEvent_t pool_get() {
        if (pool_used == pool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[pool_used++];
    }
};

// Compares the host copy of the context's error word with what has already been reported.  The copy is refreshed in stream order
// behind every render, so after a wait on `done` it covers every render enqueued so far; without a wait it covers those that
// have finished.
// `this_render`: the caller has just waited for the render it enqueued itself (the stats path), so the failure is that render's; otherwise it
// belongs to an earlier, asynchronous render on this context.  The word names the kernel(s) and the wait(s) that gave up (rt_device.h, err_tag).
static int report_device_error(mi355rt_context* ctx, bool this_render) {
    if (!ctx->h_err) return MI355RT_OK;
    const unsigned long long now = *(volatile unsigned long long*)ctx->h_err;
    const unsigned long long count = now & 0xFFFFFFFFull;
    if (count == ctx->err_reported) return MI355RT_OK;
    const unsigned long long n = count - ctx->err_reported;
    ctx->err_reported = count;
    static const char* const kernel_names[KERNEL_VARIANTS] = {"k_render_ctr_nomesh", "k_render_ctr_mesh", "k_render_ctr_sm", "k_render_ctr_simple", "k_render_ctr_sm_fixaabb",
        "(retired)", "(retired)", "k_render_ctr_wf", "k_render_ctr_wf_fixaabb", "k_render_ctr_nospec", "k_render_ctr_wf_nometal", "k_render_ctr_wf_meshfree",
        "k_render_ctr_wf_nometal_ident", "k_render_ctr_wf_nometal_shallow", "k_render_ctr_simple_qc"};
    static const struct { uint32_t bit; const char* what; } waits[] = {
        {WAIT_WF_IDLE, "idle: no progress in the workgroup"}, {WAIT_WF_RING, "ring entry: a reserved ticket was never written, or an entry never emptied"},
        {WAIT_WF_FOLLOWED, "waves that followed their workgroup's error flag out"}};
    std::string kernels, which;
    for (uint32_t v = 0; v < KERNEL_VARIANTS; ++v) if ((ctx->launched_variants >> v) & 1u) kernels += std::string(kernels.empty() ? "" : ", ") + kernel_names[v] + " (variant " + std::to_string(v) + ")";
    ctx->launched_variants = 0;
    for (const auto& w : waits) if ((now >> 32) & w.bit) which += std::string(which.empty() ? "" : "; ") + w.what;
    return fail(MI355RT_ERR_HIP, "kernel watchdog: " + std::to_string(n) + " wave(s) gave up a bounded wait " + (this_render ? "in this render" : "in an earlier render on this context") +
                                 " -- that image is incomplete [kernel(s) launched on this context since the last report: " + (kernels.empty() ? "?" : kernels) + "; wait: " + (which.empty() ? "?" : which) +
                                 "; limits: " + std::to_string(ctx->spin_limit_idle) + " idle polls, " + std::to_string(ctx->spin_limit_entry) + " polls of a ring entry]");
}

namespace {

// Re-lay the meshes' BVHs (any node order, explicit child indices -- the shape of BVHNode, bvh.rs:7-12) into the
// two-link form the kernels walk (rt_device.h, DevNode): every node carries where the walk goes when its box is hit
// (inner: the left child, bvh.rs:142) and where it goes otherwise / afterwards (the "escape": the next node of the
// reference's left-then-right recursion that is not below this one).  The links make the visit order independent of
// the storage order, so nodes are stored LEVEL BY LEVEL (level 0 of every mesh, then level 1, ...): the levels every
// ray touches come first and are the part the state-machine kernel keeps in LDS.  Triangles go into leaf-visit order.
struct MeshFlat {
    std::vector<uint32_t> order;                 // input node ids in BFS order
    std::vector<uint32_t> level_begin;           // order[level_begin[L] .. level_begin[L+1]) = level L
    std::vector<uint32_t> escape;                // per input node: input id of its escape node, NODE_END if none
    std::vector<uint32_t> first_tri;             // per input leaf: index of its first triangle in out_tris
    std::vector<uint32_t> global_id;             // per input node: index in the device array
};

int flatten_mesh(const mi355rt_scene* sc, const mi355rt_mesh& m, MeshFlat& f, std::vector<DevTri>& out_tris) {
    if ((uint64_t)m.first_triangle + m.triangle_count > sc->n_triangles || m.triangle_count == 0) return fail(MI355RT_ERR_INVALID, "mesh triangle range");
    if ((uint64_t)m.first_node + m.node_count > sc->n_nodes || m.node_count == 0) return fail(MI355RT_ERR_INVALID, "mesh node range (is the BVH missing? see mi355rt_bvh_build)");
    if ((uint64_t)m.first_index + m.index_count > sc->n_tri_indices) return fail(MI355RT_ERR_INVALID, "mesh index range");
    const mi355rt_bvh_node* nodes = sc->nodes + m.first_node;
    const uint32_t* indices = sc->tri_indices + m.first_index;
    const mi355rt_triangle* tris = sc->triangles + m.first_triangle;
    f.escape.assign(m.node_count, NODE_END); f.first_tri.assign(m.node_count, 0u); f.global_id.assign(m.node_count, NODE_END);
    // pre-order with an explicit stack (input depth is not trusted): escapes, leaf-order triangles, cycle check
    std::vector<uint8_t> seen(m.node_count, 0);
    std::vector<std::pair<uint32_t, uint32_t>> stack;   // (node, its escape)
    stack.emplace_back(0u, NODE_END);
    uint32_t visited = 0;
    while (!stack.empty()) {
        const auto [ni, esc] = stack.back(); stack.pop_back();
        if (ni >= m.node_count) return fail(MI355RT_ERR_INVALID, "BVH child index out of range");
        if (seen[ni] || ++visited > m.node_count) return fail(MI355RT_ERR_INVALID, "BVH has a cycle or shared nodes");
        seen[ni] = 1;
        f.escape[ni] = esc;
        const mi355rt_bvh_node& n = nodes[ni];
        if (n.index_count > 0) {
            if ((uint64_t)n.first_index + n.index_count > m.index_count) return fail(MI355RT_ERR_INVALID, "BVH leaf index range");
            f.first_tri[ni] = (uint32_t)out_tris.size();
            for (uint32_t k = 0; k < n.index_count; ++k) {
                const uint32_t id = indices[n.first_index + k];
                if (id >= m.triangle_count) return fail(MI355RT_ERR_INVALID, "BVH leaf triangle id out of range");
                const mi355rt_triangle& t = tris[id];
                DevTri dt;
                for (int c = 0; c < 3; ++c) { dt.v0[c] = t.v0[c]; dt.e1[c] = t.v1[c] - t.v0[c]; dt.e2[c] = t.v2[c] - t.v0[c]; dt.n[c] = t.normal[c]; }
                out_tris.push_back(dt);
            }
        } else {
            stack.emplace_back(n.right, esc);        // visited after the whole left subtree; it inherits the parent's escape
            stack.emplace_back(n.left, n.right);     // a left child escapes to its sibling
        }
    }
    // breadth-first order
    f.order.clear(); f.level_begin.clear();
    f.order.push_back(0u); f.level_begin.push_back(0u);
    for (size_t lb = 0; lb < f.order.size();) {
        const size_t le = f.order.size();
        for (size_t i = lb; i < le; ++i) {
            const mi355rt_bvh_node& n = nodes[f.order[i]];
            if (n.index_count == 0) { f.order.push_back(n.left); f.order.push_back(n.right); }
        }
        lb = le;
        if (f.order.size() > le) f.level_begin.push_back((uint32_t)le);
    }
    f.level_begin.push_back((uint32_t)f.order.size());
    return MI355RT_OK;
}

int flatten_meshes(const mi355rt_scene* sc, std::vector<DevNode>& out_nodes, std::vector<DevTri>& out_tris, std::vector<uint32_t>& roots) {
    std::vector<MeshFlat> flat(sc->n_meshes);
    size_t max_levels = 0, total = 0;
    for (uint32_t m = 0; m < sc->n_meshes; ++m) {
        int rc = flatten_mesh(sc, sc->meshes[m], flat[m], out_tris);
        if (rc) return rc;
        max_levels = std::max(max_levels, flat[m].level_begin.size() - 1);
        total += flat[m].order.size();
    }
    // the walk packs node indices into 26 bits (and addresses nodes / triangles with 32-bit byte offsets)
    if (total >= NODE_END || out_tris.size() > (1u << 26)) return fail(MI355RT_ERR_INVALID, "more than 2^26 BVH nodes or triangles");
    // Storage order: breadth-first, level by level across all meshes, until the LDS copy is full (LDS_NODE_CAP nodes: the
    // levels every ray touches); every subtree hanging below that front then follows in depth-first pre-order, so that a
    // walk through the global-memory part finds a node's left child right behind it (same or next cache line).
    uint32_t next = 0;
    for (size_t L = 0; L < max_levels && next < LDS_NODE_CAP; ++L)
        for (uint32_t m = 0; m < sc->n_meshes && next < LDS_NODE_CAP; ++m) {
            MeshFlat& f = flat[m];
            if (L + 1 >= f.level_begin.size()) continue;
            for (uint32_t i = f.level_begin[L]; i < f.level_begin[L + 1] && next < LDS_NODE_CAP; ++i) f.global_id[f.order[i]] = next++;
        }
    for (uint32_t m = 0; m < sc->n_meshes; ++m) {
        MeshFlat& f = flat[m];
        const mi355rt_bvh_node* nodes = sc->nodes + sc->meshes[m].first_node;
        std::vector<uint32_t> stack;
        for (uint32_t ni : f.order) {                                   // BFS order: parents before children
            if (f.global_id[ni] != NODE_END) continue;
            // ni is the root of an unplaced subtree (its parent was placed, or it is a mesh root beyond the cap)
            stack.assign(1, ni);
            while (!stack.empty()) {
                const uint32_t x = stack.back(); stack.pop_back();
                f.global_id[x] = next++;
                if (nodes[x].index_count == 0) { stack.push_back(nodes[x].right); stack.push_back(nodes[x].left); }
            }
        }
    }
    out_nodes.assign(total, DevNode{});
    roots.assign(sc->n_meshes, 0u);
    for (uint32_t m = 0; m < sc->n_meshes; ++m) {
        const MeshFlat& f = flat[m];
        const mi355rt_bvh_node* nodes = sc->nodes + sc->meshes[m].first_node;
        roots[m] = f.global_id[0];
        for (uint32_t ni : f.order) {
            const mi355rt_bvh_node& n = nodes[ni];
            DevNode& d = out_nodes[f.global_id[ni]];
            std::memcpy(d.bmin, n.bmin, 12); std::memcpy(d.bmax, n.bmax, 12);
            const uint32_t esc = f.escape[ni] == NODE_END ? NODE_END : f.global_id[f.escape[ni]];
            if (n.index_count == 0) { d.a = f.global_id[n.left]; d.b = esc; }
            else if (n.index_count <= NODE_MAX_LEAF) { d.a = f.first_tri[ni]; d.b = esc | (n.index_count << NODE_LINK_BITS); }
            else {
                // A leaf with more triangles than the count field holds (BVHNode::new makes them only at depth 25, bvh.rs:31;
                // a caller-built tree may have them anywhere): its box test stays where it is, as an inner node whose "left
                // child" is a chain of chunk leaves with infinite bounds.  An infinite box is hit by every ray (the slab test
                // leaves t_min / t_max untouched), so the chain only adds box tests that change nothing; a miss of the real
                // box skips the whole chain.  The chunks live behind the level-ordered part of the array.
                d.a = (uint32_t)out_nodes.size(); d.b = esc;
                const float inf = std::numeric_limits<float>::infinity();
                for (uint32_t k = 0; k < n.index_count; k += NODE_MAX_LEAF) {
                    const uint32_t cnt = std::min(NODE_MAX_LEAF, n.index_count - k);
                    const bool last = k + cnt == n.index_count;
                    DevNode c;
                    for (int x = 0; x < 3; ++x) { c.bmin[x] = -inf; c.bmax[x] = inf; }
                    c.a = f.first_tri[ni] + k;
                    c.b = (last ? esc : (uint32_t)out_nodes.size() + 1u) | (cnt << NODE_LINK_BITS);
                    out_nodes.push_back(c);       // may reallocate: `d` is not used after this loop
                }
            }
        }
    }
    if (out_nodes.size() >= NODE_END) return fail(MI355RT_ERR_INVALID, "more than 2^26 BVH nodes");
    return MI355RT_OK;
}

// Is a mesh untransformed?  world_to_object (column-major, w2o[4 * column + row]) with a diagonal of exact ones, exact zeros (of either sign) off the
// diagonal of the upper 3 x 3 and a zero translation: the case rt_intersect.h's ray_nonzero_finite() reasons about.
bool xform_is_identity(const float* w2o) {
    for (int k : {4, 8, 1, 9, 2, 6, 12, 13, 14}) if (w2o[k] != 0.0f) return false;            // (NaN != 0 too)
    return w2o[0] == 1.0f && w2o[5] == 1.0f && w2o[10] == 1.0f;
}

// The 6 world normals a cube hit can produce (cube.rs:105-136): normalized(world_to_object^T * (+-e_k, 0)) with
// exactly the device's operation order (xform_normal + normalized in rt_intersect.h / rt_math.h; this file is compiled
// with -ffp-contract=off too), so the kernel can select instead of recomputing sqrt and divide per hit.
void cube_normal_table(float* d) {
    const float EPS = 1e-4f;
    for (int k = 0; k < 3; ++k) for (int sgn = 0; sgn < 2; ++sgn) {
        volatile float n[3] = {0.0f, 0.0f, 0.0f};
        n[k] = sgn ? -1.0f : 1.0f;
        float v[3];
        for (int r = 0; r < 3; ++r) {
            volatile float a = d[4 * r + 0] * n[0], b = d[4 * r + 1] * n[1], c = d[4 * r + 2] * n[2];
            volatile float s1 = a + b; volatile float s2 = s1 + c; volatile float s3 = s2 + d[31 + r];
            v[r] = s3;
        }
        volatile float xx = v[0] * v[0], yy = v[1] * v[1], zz = v[2] * v[2];
        volatile float l2a = xx + yy; volatile float l2 = l2a + zz;
        const float l = std::sqrt((float)l2);
        float* out = d + 34 + 3 * (2 * k + sgn);
        if (l < EPS) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; }
        else { volatile float inv = 1.0f / l; out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv; }
    }
}

int build_device_scene(mi355rt_context* ctx, const mi355rt_scene* sc) {
    if (!sc) return fail(MI355RT_ERR_INVALID, "scene is null");
    const bool has_sky = sc->sky_rgb != nullptr;
    if (has_sky != (sc->sky_width != 0 && sc->sky_height != 0) || (!has_sky && (sc->sky_width || sc->sky_height)))
        return fail(MI355RT_ERR_INVALID, "sky_rgb / sky_width / sky_height are inconsistent");
    if (has_sky && ((uint64_t)sc->sky_width * sc->sky_height > (1ull << 28) || sc->sky_width >= (1u << 24) || sc->sky_height >= (1u << 24)))
        return fail(MI355RT_ERR_INVALID, "skybox too large");
    if (sc->n_primitives && !sc->primitives) return fail(MI355RT_ERR_INVALID, "primitives is null");
    if (sc->n_materials && !sc->materials) return fail(MI355RT_ERR_INVALID, "materials is null");
    if (sc->n_textures && !sc->textures) return fail(MI355RT_ERR_INVALID, "textures is null");
    uint64_t n_texels = 0;
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        const mi355rt_texture& t = sc->textures[i];
        if (!t.rgba8 || t.width == 0 || t.height == 0 || t.width >= (1u << 24) || t.height >= (1u << 24)) return fail(MI355RT_ERR_INVALID, "texture: null image or bad size");
        n_texels += (uint64_t)t.width * t.height;
    }
    if (n_texels > (1ull << 30)) return fail(MI355RT_ERR_INVALID, "textures larger than 2^30 texels in total");
    for (uint32_t i = 0; i < sc->n_materials; ++i) {
        if (sc->materials[i].kind >= MI355RT_MAT_KIND_COUNT) return fail(MI355RT_ERR_INVALID, "material kind");
        if (sc->materials[i].kind == MI355RT_MAT_TEXTURE && sc->materials[i].texture >= sc->n_textures) return fail(MI355RT_ERR_INVALID, "material texture index");
    }

    std::vector<DevNode> nodes; std::vector<DevTri> tris; std::vector<uint32_t> mesh_roots;
    if (sc->n_meshes && (!sc->meshes || !sc->nodes || !sc->triangles || (!sc->tri_indices && sc->n_tri_indices))) return fail(MI355RT_ERR_INVALID, "mesh arrays are null");
    { int rc = flatten_meshes(sc, nodes, tris, mesh_roots); if (rc) return rc; }
    std::vector<DevPrim> prims(sc->n_primitives);
    bool all_meshes_identity = true, all_meshes_shallow = true;
    for (uint32_t i = 0; i < sc->n_primitives; ++i) {
        const mi355rt_primitive& p = sc->primitives[i];
        DevPrim& d = prims[i];
        std::memset(&d, 0, sizeof d);
        if (p.kind >= MI355RT_PRIM_KIND_COUNT) return fail(MI355RT_ERR_INVALID, "primitive kind");
        if (p.material >= sc->n_materials) return fail(MI355RT_ERR_INVALID, "primitive material index");
        d.kind = p.kind; d.material = p.material;
        std::memcpy(d.mat0, &sc->materials[p.material], 16);      // kind + albedo, beside the geometry (rt_device.h)
        // The reference cannot render a sphere of |radius| < 1e-4: sphere.rs:38 divides by the radius with `Vec3 / f32`, which panics
        // below EPSILON (vec3.rs:120-122) the first time the sphere is hit.  Refused here rather than rendered.
        if (p.kind == MI355RT_PRIM_SPHERE && std::fabs(p.data[3]) < 1e-4f)
            return fail(MI355RT_ERR_INVALID, "sphere radius |r| < 1e-4: the reference panics on it (Vec3 / f32, vec3.rs:120-122 via sphere.rs:38)");
        // The quad test divides by dot(normal, direction) with the short division of rt_math.h (div_bounded), proven equal to `/` for divisors of
        // magnitude <= 2^25.  The reference's constructor always stores a unit normal (quad.rs:26-79, n = normalize(e0 x e1)), so |divisor| <= ~1;
        // a caller that hands in a scaled normal would leave the proven range while the reference semantics (IEEE division) go on: refused.
        if (p.kind == MI355RT_PRIM_QUAD) {
            bool ok = true;
            for (int k = 9; k < 12; ++k) ok = ok && std::fabs(p.data[k]) <= 0x1p20f;                       // (false for NaN and infinities too)
            if (!ok) return fail(MI355RT_ERR_INVALID, "quad normal (data[9..11]) is not finite or larger than 2^20: the reference stores a unit normal (quad.rs:26-79)");
        }
        if (p.kind == MI355RT_PRIM_CUBE || p.kind == MI355RT_PRIM_MESH) {
            const float* o2w = p.data; const float* w2o = p.data + 16;
            float t[52] = {};                                 // matrix-shaped staging: w2o[16] column-major, o2w[12], zd[3], zn[3], the cube's normal table
            std::memcpy(t, w2o, 64);
            for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) t[16 + 3 * c + r] = o2w[4 * c + r];
            volatile float zero = 0.0f;                       // keep the IEEE product (sign of zero, NaN) exactly
            for (int r = 0; r < 3; ++r) t[28 + r] = w2o[12 + r] * zero;
            for (int r = 0; r < 3; ++r) t[31 + r] = w2o[4 * r + 3] * zero;
            if (p.kind == MI355RT_PRIM_CUBE) cube_normal_table(t);
            // The record (rt_device.h): what the hit test reads -- the 3 x 3 part of w2o, its translation, zd -- as ONE run of 15 words, so that the
            // wave-uniform walk fetches it with one scalar load instead of ten pieces picked out of a 4 x 4 matrix.
            for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) d.d[3 * c + r] = t[4 * c + r];
            for (int r = 0; r < 3; ++r) d.d[12 + r] = t[28 + r];
            for (int k = 16; k < 28; ++k) d.d[k] = t[k];
            for (int k = 31; k < 52; ++k) d.d[k] = t[k];
            if (p.kind == MI355RT_PRIM_MESH) {
                if (p.mesh >= sc->n_meshes) return fail(MI355RT_ERR_INVALID, "primitive mesh index");
                d.node_begin = mesh_roots[p.mesh];
                all_meshes_identity = all_meshes_identity && xform_is_identity(w2o);
                all_meshes_shallow = all_meshes_shallow && sc->meshes[p.mesh].node_count <= WF_SHALLOW_NODES;
            }
        } else if (p.kind == MI355RT_PRIM_QUAD) {               // normal and plane constant first (what every ray needs), then base, e0, e1, the two 1 / |e|^2
            for (int k = 0; k < 4; ++k) d.d[k] = p.data[9 + k];
            for (int k = 0; k < 9; ++k) d.d[4 + k] = p.data[k];
            d.d[13] = p.data[13]; d.d[14] = p.data[14];
        } else {
            std::memcpy(d.d, p.data, 32 * sizeof(float));
        }
    }
    for (uint32_t i = sc->n_primitives; i-- > 0;)               // runs of one kind: the list walk loops over a run without re-dispatching on the kind
        prims[i].run_end = (i + 1 < sc->n_primitives && prims[i + 1].kind == prims[i].kind) ? prims[i + 1].run_end : i + 1;
    int rc;
    if ((rc = ctx->prims.ensure(prims.size()))) return rc;
    if ((rc = ctx->mats.ensure(sc->n_materials))) return rc;
    if ((rc = ctx->nodes.ensure(nodes.size()))) return rc;
    if ((rc = ctx->tris.ensure(tris.size()))) return rc;
    static_assert(sizeof(DevMat) == sizeof(mi355rt_material), "material layout is shared with the ABI");
    if (!prims.empty()) HIP_TRY(hipMemcpy(ctx->prims.p, prims.data(), prims.size() * sizeof(DevPrim), hipMemcpyHostToDevice));
    if (sc->n_materials) HIP_TRY(hipMemcpy(ctx->mats.p, sc->materials, sc->n_materials * sizeof(DevMat), hipMemcpyHostToDevice));
    if (!nodes.empty()) HIP_TRY(hipMemcpy(ctx->nodes.p, nodes.data(), nodes.size() * sizeof(DevNode), hipMemcpyHostToDevice));
    if (!tris.empty()) HIP_TRY(hipMemcpy(ctx->tris.p, tris.data(), tris.size() * sizeof(DevTri), hipMemcpyHostToDevice));
    ctx->n_prims = sc->n_primitives; ctx->n_mats = sc->n_materials; ctx->n_nodes = nodes.size();
    std::memcpy(ctx->miss, sc->miss_color, 12);
    ctx->sky_w = ctx->sky_h = 0;
    if (has_sky) {
        const size_t nf = (size_t)sc->sky_width * sc->sky_height * 3;
        if ((rc = ctx->sky.ensure(nf))) return rc;
        HIP_TRY(hipMemcpy(ctx->sky.p, sc->sky_rgb, nf * sizeof(float), hipMemcpyHostToDevice));
        ctx->sky_w = sc->sky_width; ctx->sky_h = sc->sky_height;
    }
    if (sc->n_textures) {
        if ((rc = ctx->texels.ensure((size_t)n_texels))) return rc;
        if ((rc = ctx->textures.ensure(sc->n_textures))) return rc;
        std::vector<DevTexture> table(sc->n_textures);
        size_t off = 0;
        for (uint32_t i = 0; i < sc->n_textures; ++i) {
            const mi355rt_texture& t = sc->textures[i];
            const size_t n = (size_t)t.width * t.height;
            HIP_TRY(hipMemcpy(ctx->texels.p + off, t.rgba8, n * 4, hipMemcpyHostToDevice));
            table[i] = DevTexture{ctx->texels.p + off, t.width, t.height};
            off += n;
        }
        HIP_TRY(hipMemcpy(ctx->textures.p, table.data(), table.size() * sizeof(DevTexture), hipMemcpyHostToDevice));
    }
    uint32_t n_mesh_prims = 0;
    for (const auto& pr : prims) n_mesh_prims += pr.kind == MI355RT_PRIM_MESH;
    const bool has_mesh = n_mesh_prims != 0;
    uint32_t scene_mats = 0u;                                        // which material kinds a ray can meet (bit k = MI355RT_MAT_k): those the primitives refer to
    uint32_t scene_prim_kinds = 0u;                                  // ... and which primitive kinds the list holds (bit k = MI355RT_PRIM_k)
    for (uint32_t i = 0; i < sc->n_primitives; ++i) { scene_mats |= MATBIT(sc->materials[sc->primitives[i].material].kind); scene_prim_kinds |= 1u << sc->primitives[i].kind; }
    auto covers = [&](uint32_t variant) { return (scene_mats & ~mats_of_variant(variant)) == 0u; };
    auto kinds_covered = [&](uint32_t variant) { return (scene_prim_kinds & ~prims_of_variant(variant)) == 0u; };          // likewise for the primitive kinds of the list
    ctx->has_mesh = has_mesh;
    // Scenes with meshes: the wavefront kernel (path state in LDS, stage queues).  No mesh: a lockstep kernel.  In both
    // families the most pruned instantiation whose material set covers the scene's (rt_device.h, mats_of_variant): the branches of
    // the kinds a scene does not have are compiled out -- they set the register peak.  The library reads NO environment
    // variables; the diagnostic hook mi355rt_debug_set_knob("kernel", v) may name another variant this library was built with.
    // ... and, where the meshes are all untransformed (OBJ data in world space: teapot), the instantiation whose mesh_setup skips the matrix products.
    // ... and, for transformed meshes whose trees are all small (semesterbild), the instantiation with the shorter WALK rounds (rt_wavefront.h).
    if (has_mesh) ctx->variant = covers(KERNEL_WAVEFRONT_NOMETAL) ? (all_meshes_identity ? KERNEL_WAVEFRONT_NOMETAL_IDENT : all_meshes_shallow ? KERNEL_WAVEFRONT_NOMETAL_SHALLOW : KERNEL_WAVEFRONT_NOMETAL)
                                                                  : KERNEL_WAVEFRONT;
    else {
        // Mesh-free lists run on a lockstep kernel -- unless the shading step diverges EXPENSIVELY: a rough conductor (ln, atan, two
        // sin_cos, the conductor's Fresnel term: ~400 instructions) next to another scattering material.  In lockstep a wave pays that branch
        // whenever any lane takes it (veach-mis: in 71 % of its iterations, for 6.8 lanes); the wavefront kernel's material-sorted SHADE
        // passes run it at ~57 lanes: veach-mis 18.30 -> 16.75 ms at 256 spp.  Cheap mixtures (Lambert + metal + dielectric + plastic)
        // measured 3-7 % FASTER in lockstep (tools/ab_fuzz_scene.py), and so stay there.
        const bool rough = (scene_mats & MATS_ROUGH) != 0u, other_scatter = (scene_mats & ~(MATS_ROUGH | MATS_TERMINAL)) != 0u;
        ctx->variant = covers(KERNEL_LOCKSTEP_SIMPLE) ? (kinds_covered(KERNEL_LOCKSTEP_SIMPLE_QC) ? KERNEL_LOCKSTEP_SIMPLE_QC : KERNEL_LOCKSTEP_SIMPLE)   // (... pruned to quads and cubes where the list holds nothing else: cornell)
                     : (rough && other_scatter && covers(KERNEL_WAVEFRONT_MESHFREE)) ? KERNEL_WAVEFRONT_MESHFREE
                     : covers(KERNEL_LOCKSTEP_NOSPEC) ? KERNEL_LOCKSTEP_NOSPEC : KERNEL_LOCKSTEP;
    }
    if (ctx->forced_variant >= 0) {
        const uint32_t v = (uint32_t)ctx->forced_variant;
        const bool mesh_free_only = v == KERNEL_LOCKSTEP || v == KERNEL_LOCKSTEP_SIMPLE || v == KERNEL_LOCKSTEP_SIMPLE_QC || v == KERNEL_LOCKSTEP_NOSPEC || v == KERNEL_WAVEFRONT_MESHFREE;
        const bool selectable = v == KERNEL_LOCKSTEP || v == KERNEL_LOCKSTEP_MESH || v == KERNEL_STATE_MACHINE || v == KERNEL_WAVEFRONT ||
                                v == KERNEL_LOCKSTEP_SIMPLE || v == KERNEL_LOCKSTEP_SIMPLE_QC || v == KERNEL_LOCKSTEP_NOSPEC || v == KERNEL_WAVEFRONT_NOMETAL || v == KERNEL_WAVEFRONT_MESHFREE ||
                                v == KERNEL_WAVEFRONT_NOMETAL_IDENT || v == KERNEL_WAVEFRONT_NOMETAL_SHALLOW;   // (the _FIXAABB forms follow options.flags; _SHALLOW is only a tuning: any tree is walked correctly)
        const bool ok = render_ctr_variant_built(v) && selectable && covers(v) && kinds_covered(v) && !(mesh_free_only && has_mesh) &&
                        !(v == KERNEL_WAVEFRONT_NOMETAL_IDENT && !(has_mesh && all_meshes_identity));     // (that form ASSUMES untransformed meshes)
        if (ok) ctx->variant = v;
    }
    // Root-box test right at mesh set-up (reference build's state machine): when several meshes share the list (teapot +5..12 %; a single
    // mesh loses 5-10 %).
    ctx->inline_steps = ctx->knob_inline_steps >= 0 ? (uint32_t)ctx->knob_inline_steps : (n_mesh_prims >= 2 ? 1u : 0u);
    return MI355RT_OK;
}

// Diagnostic knobs (mi355rt_debug_set_knob; not part of the public header).  They replace the environment variables earlier
// rounds read at set_scene: a product library should not change behaviour with the caller's environment.
std::mutex g_knob_mutex;
std::map<std::string, int> g_default_knobs;
int apply_knob(mi355rt_context* ctx, const std::string& name, int v) {
    if (name == "kernel") { if (v < -1 || v >= (int)KERNEL_VARIANTS) return fail(MI355RT_ERR_INVALID, "knob kernel"); ctx->forced_variant = v; }
    else if (name == "inline_steps") { if (v < -1 || v > 8) return fail(MI355RT_ERR_INVALID, "knob inline_steps"); ctx->knob_inline_steps = v; }
    else if (name == "grid_div") { if (v < 1 || v > 16) return fail(MI355RT_ERR_INVALID, "knob grid_div"); ctx->grid_div = (uint32_t)v; }
    else if (name == "guided_mult") { if (v < 1 || v > 64) return fail(MI355RT_ERR_INVALID, "knob guided_mult"); ctx->guided_mult = (uint32_t)v; }
    else if (name == "trav_min") { if (v < 1 || v > 64) return fail(MI355RT_ERR_INVALID, "knob trav_min"); ctx->trav_min = (uint32_t)v; }
    else if (name == "spin_idle") { if (v < 1) return fail(MI355RT_ERR_INVALID, "knob spin_idle"); ctx->spin_limit_idle = (uint32_t)v; }
    else if (name == "spin_entry") { if (v < 1) return fail(MI355RT_ERR_INVALID, "knob spin_entry"); ctx->spin_limit_entry = (uint32_t)v; }
    else if (name == "wave_times") ctx->want_wave_times = v != 0;
    else if (name == "row_order") { if (v < -1 || v > 1) return fail(MI355RT_ERR_INVALID, "knob row_order"); ctx->knob_row_order = v; }
    else return fail(MI355RT_ERR_INVALID, "unknown knob " + name);
    return MI355RT_OK;
}

}  // namespace

extern "C" {

// Diagnostic hook (not part of the public header).  ctx != NULL: set one knob of that context (before set_scene).  ctx == NULL: a
// process-wide default applied to every context created afterwards -- also those the one-shot calls create; name == NULL clears
// all defaults.  Knobs: kernel (KERNEL_* of rt_device.h, -1 = automatic), guided_mult, spin_idle, spin_entry, wave_times, and for the
// reference build's state machine inline_steps, trav_min.
int mi355rt_debug_set_knob(mi355rt_context* ctx, const char* name, int value) {
    return guard([&]() -> int {
    if (ctx) return name ? apply_knob(ctx, name, value) : fail(MI355RT_ERR_INVALID, "knob name is null");
    std::lock_guard<std::mutex> g(g_knob_mutex);
    if (!name) { g_default_knobs.clear(); return MI355RT_OK; }
    mi355rt_context probe;                                            // validate name and range on a scratch context
    const int rc = apply_knob(&probe, name, value);
    if (rc == MI355RT_OK) g_default_knobs[name] = value;
    return rc;
    });
}
// 1 when this library holds the counter-mode kernel `variant` (the retired mesh kernels exist in the reference build only)
int mi355rt_debug_has_variant(uint32_t variant) { return render_ctr_variant_built(variant) ? 1 : 0; }

const char* mi355rt_last_error(void) { return g_err.c_str(); }
uint32_t mi355rt_abi_version(void) { return MI355RT_ABI_VERSION; }

int mi355rt_rows_selected(const mi355rt_settings* settings, const mi355rt_options* options, uint32_t* out_rows) {
    return guard([&]() -> int {
    int rc = check_settings(settings); if (rc) return rc;
    if (!out_rows) return fail(MI355RT_ERR_INVALID, "out_rows is null");
    RowSel sel; rc = select_rows(*settings, options, sel); if (rc) return rc;
    *out_rows = (uint32_t)sel.rows.size();
    return MI355RT_OK;
    });
}

int mi355rt_context_create(int hip_device, mi355rt_context** out_ctx) {
    return guard([&]() -> int {
    if (!out_ctx) return fail(MI355RT_ERR_INVALID, "out_ctx is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MI355RT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (hip_device < 0 || hip_device >= n) return fail(MI355RT_ERR_NO_DEVICE, "hip_device out of range");
    HIP_TRY(hipSetDevice(hip_device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, hip_device));
    mi355rt_context* ctx = new (std::nothrow) mi355rt_context();
    if (!ctx) return fail(MI355RT_ERR_OOM, "host allocation failed");
    ctx->device = hip_device;
    ctx->cu_count = prop.multiProcessorCount;
    for (uint32_t v = 0; v < KERNEL_VARIANTS; ++v) {
        if (!render_ctr_variant_built(v)) continue;                    // the retired mesh kernels exist in the tests' reference build only
        if (query_render_ctr_occupancy(v, &ctx->blocks_per_cu[v], &ctx->vgprs[v], &ctx->sgprs) != 0 || ctx->blocks_per_cu[v] <= 0) {
            delete ctx;
            return fail(MI355RT_ERR_HIP, std::string("kernel image not usable on this device (") + prop.gcnArchName + "); built for gfx950");
        }
    }
    for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { delete ctx; return fail(MI355RT_ERR_HIP, "hipEventCreate"); }
    if (hipEventCreateWithFlags(&ctx->done, hipEventDisableTiming) != hipSuccess) { delete ctx; return fail(MI355RT_ERR_HIP, "hipEventCreate"); }
    if (ctx->errword.ensure(1) != MI355RT_OK || hipMemset(ctx->errword.p, 0, sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_err, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { mi355rt_context_destroy(ctx); return fail(MI355RT_ERR_HIP, "error word allocation"); }
    *ctx->h_err = 0ull;
    {   std::lock_guard<std::mutex> g(g_knob_mutex);                    // process-wide diagnostic defaults (mi355rt_debug_set_knob(NULL, ...))
        for (const auto& kv : g_default_knobs) (void)apply_knob(ctx, kv.first, kv.second); }
    *out_ctx = ctx;
    return MI355RT_OK;
    });
}

void mi355rt_context_destroy(mi355rt_context* ctx) {                 // (nothing in here allocates or throws: HIP calls and destructors of PODs' containers)
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->prims.release(); ctx->mats.release(); ctx->nodes.release(); ctx->tris.release(); ctx->rows.release();
    ctx->radiance.release(); ctx->counters.release(); ctx->stats.release(); ctx->fold_stack.release(); ctx->wave_times.release(); ctx->sky.release();
    ctx->textures.release(); ctx->texels.release(); ctx->errword.release();
    if (ctx->h_err) (void)hipHostFree(ctx->h_err);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->done) (void)hipEventDestroy(ctx->done);
    for (auto& e : ctx->pool) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

static int render_samples(mi355rt_context* ctx, const mi355rt_options* opt, uint32_t s0, uint32_t s1, void* d_accum,
                          void* d_out_packed, void* d_out_linear, void* hip_stream, mi355rt_stats* stats, unsigned long long* d_row_counters = nullptr);

int mi355rt_context_set_scene(mi355rt_context* ctx, const mi355rt_scene* scene, const mi355rt_camera* camera,
                              const mi355rt_settings* settings) {
    return guard([&]() -> int {
    if (!ctx) return fail(MI355RT_ERR_INVALID, "ctx is null");
    if (!camera) return fail(MI355RT_ERR_INVALID, "camera is null");
    int rc = check_settings(settings); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->have_scene = false; ctx->rows_valid = false;
    rc = build_device_scene(ctx, scene); if (rc) return rc;
    static_assert(sizeof(DevCamera) == sizeof(mi355rt_camera), "camera layout is shared with the ABI");
    std::memcpy(&ctx->cam, camera, sizeof(DevCamera));
    ctx->settings = *settings;
    ctx->have_scene = true;
    // The probe renders below are set_scene's own business: they stay out of the caller's timing pool (mi355rt_context_set_timing).
    struct TimingOff { mi355rt_context* c; bool was; ~TimingOff() { c->timing = was; } } timing_off{ctx, ctx->timing};
    ctx->timing = false;
    // The materials only say that the mesh-free wavefront kernel MAY pay (a rough conductor next to another scattering material).  Whether
    // it does depends on how much the paths scatter: veach-mis bounces 1.5 times per path and gains 13 %; a scene of the same materials that is
    // mostly sky (1.1 rays per path) has nothing to sort and lost 14 % to the queues (profiles/r03_ab_meshfree_wavefront_fuzz_scenes.txt).
    // A probe decides: the same view at 64 pixels across, 4 samples per pixel, on the lockstep kernel -- deterministic (counter RNG), a
    // fraction of a millisecond -- and the wavefront form is kept when a path traces at least PROBE_RAYS_PER_PATH rays.
    // Both probes below render with the context's settings (and, the first, its variant) swapped for the probe's.  Whatever way they are left --
    // a return code, a HIP failure, a C++ exception out of render_samples' host allocations -- ProbeScope puts the full-size settings back, drops
    // the row tables of the probe size, frees the probe's device buffers and, unless the probe was committed, leaves the context WITHOUT a scene:
    // a context that kept have_scene with 64-pixel-wide settings would render a probe-sized image into the caller's full-size buffer.
    struct ProbeScope {
        mi355rt_context* c; mi355rt_settings full; uint32_t variant; uint32_t* d_tmp = nullptr; unsigned long long* d_cnt = nullptr; bool committed = false;
        ProbeScope(mi355rt_context* ctx) : c(ctx), full(ctx->settings), variant(ctx->variant) {}
        ~ProbeScope() {
            c->settings = full; c->rows_valid = false;
            if (d_tmp) (void)hipFree(d_tmp);
            if (d_cnt) (void)hipFree(d_cnt);
            if (!committed) { c->variant = variant; c->have_scene = false; c->row_cost.clear(); }
        }
    };
    if (ctx->variant == KERNEL_WAVEFRONT_MESHFREE && ctx->forced_variant < 0) {
        constexpr double PROBE_RAYS_PER_PATH = 1.6;       // veach-mis with max_bounces 1 / 2 / 3 / 16: 1.00 / 1.90 / 2.20 / 2.48 rays per path, wavefront +5.5 / -4.0 / -6.5 / -10.2 %
                                                          // against lockstep (profiles/r03_probe_calibration.txt): break-even near 1.5
        ProbeScope scope(ctx);
        mi355rt_settings probe = scope.full;
        probe.width = std::min(scope.full.width, 64u);
        probe.height = std::max(1u, std::min(scope.full.height, (uint32_t)((uint64_t)probe.width * scope.full.height / scope.full.width)));
        probe.samples_per_pixel = std::min(scope.full.samples_per_pixel, 4u);
        if (hipMalloc((void**)&scope.d_tmp, (size_t)probe.width * probe.height * 4) != hipSuccess) { scope.d_tmp = nullptr; return fail(MI355RT_ERR_OOM, "hipMalloc(probe)"); }
        ctx->settings = probe; ctx->variant = KERNEL_LOCKSTEP_NOSPEC;
        mi355rt_stats st{};
        rc = render_samples(ctx, nullptr, 0, probe.samples_per_pixel, nullptr, scope.d_tmp, nullptr, nullptr, &st);
        if (rc) return rc;
        ctx->variant = ((double)st.rays >= PROBE_RAYS_PER_PATH * (double)std::max<uint64_t>(st.samples, 1)) ? KERNEL_WAVEFRONT_MESHFREE : KERNEL_LOCKSTEP_NOSPEC;
        scope.committed = true;
    }
    // Row costs for the processing order (see mi355rt_context::row_cost): the same view at <= 64 x 96 pixels, 4 samples per pixel, one small
    // launch per probe row with its own {paths, rays} counters, all enqueued back to back and read after ONE wait -- deterministic (counter
    // RNG), a few milliseconds.  Only the diagnostic knob turns it on (measured, profiles/r04/ab_processing_order.txt: +-0.5 % on full frames --
    // the tail of a launch is old paths that waited in thin queues, not the rows handed out last).
    ctx->row_cost.clear();
    if (ctx->knob_row_order == 1) {
        ProbeScope scope(ctx);
        mi355rt_settings probe = scope.full;
        probe.width = std::min(scope.full.width, 64u);
        probe.height = std::min(scope.full.height, 96u);
        probe.samples_per_pixel = std::min(scope.full.samples_per_pixel, 4u);
        std::vector<unsigned long long> h_cnt(STATS_WORDS * (size_t)probe.height);
        if (hipMalloc((void**)&scope.d_tmp, (size_t)probe.width * probe.height * 4) != hipSuccess) { scope.d_tmp = nullptr; return fail(MI355RT_ERR_OOM, "hipMalloc(row probe)"); }
        if (hipMalloc((void**)&scope.d_cnt, h_cnt.size() * 8) != hipSuccess) { scope.d_cnt = nullptr; return fail(MI355RT_ERR_OOM, "hipMalloc(row probe)"); }
        HIP_TRY(hipMemset(scope.d_cnt, 0, h_cnt.size() * 8));
        ctx->settings = probe;
        rc = render_samples(ctx, nullptr, 0, probe.samples_per_pixel, nullptr, scope.d_tmp, nullptr, nullptr, nullptr, scope.d_cnt);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(h_cnt.data(), scope.d_cnt, h_cnt.size() * 8, hipMemcpyDeviceToHost));                    // (waits for the launches)
        if ((rc = report_device_error(ctx))) return rc;
        ctx->row_cost.resize(scope.full.height);
        for (uint32_t y = 0; y < scope.full.height; ++y) {
            const uint32_t i = (uint32_t)std::min<uint64_t>(probe.height - 1, (uint64_t)y * probe.height / scope.full.height);
            ctx->row_cost[y] = (float)((double)h_cnt[STATS_WORDS * (size_t)i + 1] / (double)std::max<unsigned long long>(h_cnt[STATS_WORDS * (size_t)i], 1ull));
        }
        scope.committed = true;
    }
    return MI355RT_OK;
    });
}

// Samples [s0, s1) of every selected pixel.  The classic call is (0, settings.spp, no accumulator).
// d_row_counters (set_scene's row-cost probe only): one launch per selected ROW, each with its own block of device counters ({paths, rays, ...})
// at d_row_counters + STATS_WORDS * row, nothing resolved, nothing waited for.
static int render_samples(mi355rt_context* ctx, const mi355rt_options* opt, uint32_t s0, uint32_t s1, void* d_accum,
                          void* d_out_packed, void* d_out_linear, void* hip_stream, mi355rt_stats* stats, unsigned long long* d_row_counters) {
    if (!ctx || !ctx->have_scene) return fail(MI355RT_ERR_INVALID, "context has no scene");
    if (!d_out_packed) return fail(MI355RT_ERR_INVALID, "d_out_packed is null");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int erc = report_device_error(ctx)) return erc;              // an earlier asynchronous render on this context failed (no wait: what has finished so far)
    hipStream_t stream = (hipStream_t)hip_stream;
    const mi355rt_settings& st = ctx->settings;
    RowSel sel; int rc = select_rows(st, opt, sel); if (rc) return rc;
    if (ctx->have_last && ctx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->done, 0));   // the previous render owned the workspaces
    const uint32_t rng_mode = opt ? opt->rng_mode : (uint32_t)MI355RT_RNG_CTR;
    // How many consecutive ranges the processing order is dealt over: the work shards of every band this launch will be cut into.
    uint32_t groups = WORK_SHARDS;
    {   const uint64_t spp_now = std::max<uint64_t>(1, (uint64_t)s1 - s0), pixels = (uint64_t)sel.rows.size() * st.width;
        const uint64_t ws = (opt && opt->workspace_bytes) ? opt->workspace_bytes : (32ull << 30);
        const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(ws / 12, (1ull << 31) - 16 * RUN_LIMIT) / spp_now);
        const uint64_t band_pixels = std::max<uint64_t>(1, std::min(cap, pixels));                       // (the band plan below arrives at the same figure)
        groups *= (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (pixels + band_pixels - 1) / band_pixels)); }
    if (ctx->row_cost.empty() || rng_mode != MI355RT_RNG_CTR) groups = 0;                  // natural order
    const bool same_rows = ctx->rows_valid && sel.rows == ctx->rows_host && groups == ctx->order_groups;
    if (!same_rows) {
        if (ctx->have_last) HIP_TRY(hipEventSynchronize(ctx->done));   // a previous call's row-table upload may still read tables_host
        ctx->rows_host.swap(sel.rows);
        static const std::vector<float> no_cost;
        row_tables(ctx->rows_host, groups ? ctx->row_cost : no_cost, groups, ctx->tables_host);
        ctx->order_groups = groups;
        ctx->rows_valid = false;
    }
    const bool fixed_aabb = opt && (opt->flags & MI355RT_FLAG_FIXED_AABB) != 0u;
    if (opt && (opt->flags & ~MI355RT_FLAG_FIXED_AABB) != 0u) return fail(MI355RT_ERR_INVALID, "options.flags has unknown bits");
    if (fixed_aabb && rng_mode != MI355RT_RNG_CTR) return fail(MI355RT_ERR_INVALID, "MI355RT_FLAG_FIXED_AABB needs MI355RT_RNG_CTR (the replay mode reproduces the reference as it is)");
    uint32_t variant = ctx->variant;
    if (fixed_aabb && ctx->has_mesh) {                                // without a mesh the flag changes nothing
        variant = ctx->variant == KERNEL_STATE_MACHINE ? (uint32_t)KERNEL_STATE_MACHINE_FIXAABB : (uint32_t)KERNEL_WAVEFRONT_FIXAABB;
        if (!render_ctr_variant_built(variant)) variant = KERNEL_WAVEFRONT_FIXAABB;
    }
    // The mesh-free lockstep kernels are compiled under the assumption that the list holds something and that a path may take a step (rt_kernels.hip,
    // render_ctr_lockstep); the two degenerate renders -- every sample is the miss colour / BLACK -- go to the plain per-lane loop, which assumes nothing.
    if ((ctx->n_prims == 0 || st.max_depth == 0) && (variant == KERNEL_LOCKSTEP || variant == KERNEL_LOCKSTEP_SIMPLE || variant == KERNEL_LOCKSTEP_SIMPLE_QC || variant == KERNEL_LOCKSTEP_NOSPEC))
        variant = KERNEL_LOCKSTEP_MESH;
    const uint64_t seed = opt ? opt->seed : 0;
    const uint32_t n_rows = (uint32_t)ctx->rows_host.size();
    if (stats) { std::memset(stats, 0, sizeof *stats); stats->rows_rendered = n_rows; stats->kernel_vgprs = (uint32_t)ctx->vgprs[variant]; stats->kernel_sgprs = (uint32_t)ctx->sgprs; }
    if (n_rows == 0) return MI355RT_OK;

    if (!same_rows) {
        if ((rc = ctx->rows.ensure(3 * (size_t)n_rows))) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->rows.p, ctx->tables_host.data(), 3 * (size_t)n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        ctx->rows_valid = true;
    }
    const uint32_t* d_rows_natural = ctx->rows.p;
    const uint32_t* d_rows_processing = ctx->rows.p + n_rows;
    const uint32_t* d_out_row = ctx->order_groups ? ctx->rows.p + 2 * (size_t)n_rows : nullptr;      // null: processing order == output order
    if ((rc = ctx->stats.ensure(STATS_WORDS))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->stats.p, 0, STATS_WORDS * sizeof(unsigned long long), stream));

    double render_ms = 0, resolve_ms = 0, total_ms = 0;
    uint32_t n_bands = 0, grid_blocks = 0, block_threads = 0;

    if (rng_mode == MI355RT_RNG_REF) {
        if (d_accum || s0 != 0 || s1 != st.samples_per_pixel)
            return fail(MI355RT_ERR_INVALID, "progressive rendering needs MI355RT_RNG_CTR (the reference stream of a row is sequential over its pixels)");
        if ((rc = ctx->fold_stack.ensure((size_t)n_rows * std::max(st.max_depth, 1u) * 3))) return rc;
        RefParams rp{};
        rp.prims = ctx->prims.p; rp.mats = ctx->mats.p; rp.nodes = ctx->nodes.p; rp.tris = ctx->tris.p; rp.rows = d_rows_natural;
        rp.sky = ctx->sky_w ? ctx->sky.p : nullptr; rp.sky_w = ctx->sky_w; rp.sky_h = ctx->sky_h; rp.textures = ctx->textures.p;
        rp.out_packed = (uint32_t*)d_out_packed; rp.out_linear = (float*)d_out_linear; rp.fold_stack = ctx->fold_stack.p; rp.stats = ctx->stats.p;
        rp.n_prims = ctx->n_prims; rp.n_mats = ctx->n_mats; rp.n_rows = n_rows;
        std::memcpy(rp.miss, ctx->miss, 12); rp.cam = ctx->cam;
        rp.width = st.width; rp.height = st.height; rp.spp = st.samples_per_pixel; rp.max_depth = st.max_depth;
        rp.seed_lo = (uint32_t)seed; rp.seed_hi = (uint32_t)(seed >> 32);
        if (stats) HIP_TRY(hipEventRecord(ctx->ev[0], stream));
        if (launch_render_ref(rp, stream) != 0) return fail(MI355RT_ERR_HIP, "k_render_ref launch failed");
        if (stats) {
            HIP_TRY(hipEventRecord(ctx->ev[1], stream));
            HIP_TRY(hipEventSynchronize(ctx->ev[1]));
            float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            render_ms = total_ms = ms;
        }
        n_bands = 1; grid_blocks = n_rows; block_threads = 64;
    } else {
        // ---- band plan: the radiance workspace holds band_pixels * spp float4 ----
        const uint64_t spp = s1 - s0;                                 // samples per pixel in THIS launch
        const uint64_t total_pixels = (uint64_t)n_rows * st.width;
        uint64_t ws_cap = (opt && opt->workspace_bytes) ? opt->workspace_bytes : (32ull << 30);   // 288 GB of HBM: default = the 2^31-sample band limit; only what a band needs is allocated
        uint64_t max_samples = std::min<uint64_t>(ws_cap / 12, (1ull << 31) - 16 * RUN_LIMIT);     // the shard counters overshoot by at most one run per claiming wave's last try; 32-bit headroom
        if (max_samples < spp) return fail(MI355RT_ERR_INVALID, "workspace_bytes too small for one pixel (needs spp * 12 bytes)");
        uint64_t band_pixels_max = std::min<uint64_t>(max_samples / spp, total_pixels);
        if (d_row_counters) band_pixels_max = st.width;                  // the probe: a band = a row
        // Only what a band needs is allocated.  When even that does not fit (another tenant on the GPU, a small device),
        // halve the band and try again: more, smaller bands give the same image (tiling invariance), just more launches.
        for (;;) {
            rc = ctx->radiance.ensure((size_t)(band_pixels_max * spp * 3));
            // (the row probe's counter blocks are laid out one per ROW = per band: halving would make more bands than blocks -- it returns the OOM)
            if (rc != MI355RT_ERR_OOM || band_pixels_max <= 1 || d_row_counters) break;
            (void)hipGetLastError();
            band_pixels_max = (band_pixels_max + 1) / 2;
        }
        if (rc) return rc;
        n_bands = (uint32_t)((total_pixels + band_pixels_max - 1) / band_pixels_max);
        const size_t ctr_words = (size_t)WORK_SHARDS * WORK_SHARD_STRIDE;            // per band
        if ((rc = ctx->counters.ensure((size_t)n_bands * ctr_words))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, (size_t)n_bands * ctr_words * sizeof(uint32_t), stream));

        RenderParams p{};
        p.prims = ctx->prims.p; p.mats = ctx->mats.p; p.nodes = ctx->nodes.p; p.tris = ctx->tris.p; p.rows = d_rows_processing;
        p.sky = ctx->sky_w ? ctx->sky.p : nullptr; p.sky_w = ctx->sky_w; p.sky_h = ctx->sky_h; p.textures = ctx->textures.p;
        p.radiance = ctx->radiance.p; p.stats = ctx->stats.p;
        p.n_prims = ctx->n_prims; p.n_mats = ctx->n_mats;
        std::memcpy(p.miss, ctx->miss, 12); p.cam = ctx->cam;
        p.width = st.width; p.height = st.height; p.spp = (uint32_t)spp; p.max_depth = st.max_depth;
        p.width_f = (float)st.width; p.height_f = (float)st.height;                       // exact: both below 2^24
        { volatile float one = 1.0f; p.inv_width_rn = one / p.width_f; p.inv_height_rn = one / p.height_f; }   // IEEE division on the host = RN(1/x), what recip_normal_range() returns on the device
        p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32); p.sample0 = s0;
        magic_div((uint32_t)spp, p.spp_mul, p.spp_shift); magic_div(st.width, p.width_mul, p.width_shift);
        p.trav_min = ctx->trav_min; p.inline_steps = ctx->inline_steps;
        p.lds_nodes = (uint32_t)std::min<size_t>(ctx->n_nodes, LDS_NODE_CAP);    // (the wavefront kernel clamps to its own WF_LDS_NODES)
        p.err = ctx->errword.p; p.spin_limit_idle = ctx->spin_limit_idle; p.spin_limit_entry = ctx->spin_limit_entry;
        ResolveParams r{};
        r.radiance = ctx->radiance.p; r.out_packed = (uint32_t*)d_out_packed; r.out_linear = (float*)d_out_linear;
        r.spp = (uint32_t)spp; r.inv_spp = 1.0f / (float)s1;                             // renderer.rs:85
        r.accum = (float*)d_accum; r.accum_load = s0 != 0 ? 1u : 0u;
        r.out_row = d_out_row; r.width = st.width; r.width_mul = p.width_mul; r.width_shift = p.width_shift;
        // What fills the device, or this context's share of it: with F frames in flight (F contexts, F streams) each launch takes 1 / F of the wave slots,
        // F launches are co-resident, and a frame whose last paths are draining shares every SIMD with frames in their steady state.  A full-size grid
        // leaves the next frame's workgroups waiting for the draining frame's to retire one by one ("tail").
        const uint32_t resident = std::max(1u, (uint32_t)(ctx->cu_count * ctx->blocks_per_cu[variant]) / ctx->grid_div);
        block_threads = block_threads_of(variant);
        std::vector<float> band_ms;
        for (uint32_t b = 0; b < n_bands; ++b) {
            const uint64_t p0 = (uint64_t)b * band_pixels_max;
            const uint64_t np = std::min<uint64_t>(band_pixels_max, total_pixels - p0);
            p.band_pixel0 = (uint32_t)p0; p.band_samples = (uint32_t)(np * spp);
            p.batch_counter = ctx->counters.p + (size_t)b * ctr_words;
            if (d_row_counters) p.stats = d_row_counters + STATS_WORDS * (size_t)b;   // (a whole counter block per row: diagnostic builds write all of it)
            const bool wf = is_wavefront(variant);
            const uint32_t run_min = wf ? RUN_WAVEFRONT_MIN : BATCH_MIN, run_max = wf ? RUN_WAVEFRONT : BATCH_MAX;   // what the kernel's WorkCursorT is compiled with
            p.shard_samples = (p.band_samples + WORK_SHARDS - 1) / WORK_SHARDS;
            p.shard_samples = (p.shard_samples + run_max - 1) / run_max * run_max;           // shards begin on run boundaries (fixed runs then stay aligned)
            const uint32_t waves_per_block = block_threads / 64;
            const uint32_t min_runs = (p.band_samples + run_min - 1) / run_min;               // never more waves than minimum-size runs
            const uint32_t grid = std::max(1u, std::min(resident, (min_runs + waves_per_block - 1) / waves_per_block));
            p.guided_div = std::max(1u, ctx->guided_mult * grid * waves_per_block / WORK_SHARDS);
            p.wave_times = nullptr;
            if (ctx->want_wave_times) {
                ctx->wave_times_n = grid * waves_per_block;
                if ((rc = ctx->wave_times.ensure((size_t)ctx->wave_times_n * WAVE_TIME_WORDS))) return rc;
                HIP_TRY(hipMemsetAsync(ctx->wave_times.p, 0, (size_t)ctx->wave_times_n * WAVE_TIME_WORDS * 8, stream));
                p.wave_times = ctx->wave_times.p;
            }
            grid_blocks = std::max(grid_blocks, grid);
            r.band_pixel0 = (uint32_t)p0; r.band_pixels = (uint32_t)np;
            hipEvent_t pe0 = nullptr, pe1 = nullptr, pe2 = nullptr;
            if (ctx->timing && !stats) { pe0 = ctx->pool_get(); pe1 = ctx->pool_get(); pe2 = ctx->pool_get(); if (!pe0 || !pe1 || !pe2) return fail(MI355RT_ERR_HIP, "event pool"); }
            if (stats) HIP_TRY(hipEventRecord(ctx->ev[0], stream));
            if (pe0) HIP_TRY(hipEventRecord(pe0, stream));
            if (launch_render_ctr(p, variant, grid, stream) != 0) return fail(MI355RT_ERR_HIP, "k_render_ctr launch failed");
            ctx->launched_variants |= 1u << variant;
            if (stats) HIP_TRY(hipEventRecord(ctx->ev[1], stream));
            if (pe1) HIP_TRY(hipEventRecord(pe1, stream));
            if (!d_row_counters && launch_resolve(r, stream) != 0) return fail(MI355RT_ERR_HIP, "k_resolve launch failed");
            if (pe2) { HIP_TRY(hipEventRecord(pe2, stream)); ++ctx->timed_launches; }
            if (stats) {
                HIP_TRY(hipEventRecord(ctx->ev[2], stream));
                HIP_TRY(hipEventSynchronize(ctx->ev[2]));
                float a = 0, c = 0;
                HIP_TRY(hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
                HIP_TRY(hipEventElapsedTime(&c, ctx->ev[1], ctx->ev[2]));
                render_ms += a; resolve_ms += c; total_ms += a + c;
            }
        }
    }
    // the context's error word, behind the kernels and in front of `done`: whoever waits on `done` (or finds it complete) sees it
    HIP_TRY(hipMemcpyAsync(ctx->h_err, ctx->errword.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ctx->done, stream));
    ctx->last_stream = stream; ctx->have_last = true;
    if (stats) {
        unsigned long long h[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h, ctx->stats.p, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (int erc = report_device_error(ctx, true)) return erc;    // a wave of THIS render gave up: the image is incomplete
        stats->render_kernel_ms = render_ms; stats->resolve_kernel_ms = resolve_ms; stats->total_ms = total_ms;
        stats->samples = h[0]; stats->rays = h[1];
        stats->bands = n_bands; stats->grid_blocks = grid_blocks; stats->block_threads = block_threads;
    }
    return MI355RT_OK;
}

int mi355rt_context_set_share(mi355rt_context* ctx, uint32_t share_of) {
    return guard([&]() -> int {
    if (!ctx) return fail(MI355RT_ERR_INVALID, "ctx is null");
    if (share_of < 1u || share_of > 16u) return fail(MI355RT_ERR_INVALID, "share_of must be 1 .. 16");
    ctx->grid_div = share_of;
    return MI355RT_OK;
    });
}

int mi355rt_context_render(mi355rt_context* ctx, const mi355rt_options* opt, void* d_out_packed, void* d_out_linear,
                           void* hip_stream, mi355rt_stats* stats) {
    return guard([&]() -> int {
    if (!ctx || !ctx->have_scene) return fail(MI355RT_ERR_INVALID, "context has no scene");
    return render_samples(ctx, opt, 0, ctx->settings.samples_per_pixel, nullptr, d_out_packed, d_out_linear, hip_stream, stats);
    });
}

int mi355rt_context_render_progressive(mi355rt_context* ctx, const mi355rt_options* opt, uint32_t sample_begin, uint32_t sample_end,
                                       void* d_accum, void* d_out_packed, void* d_out_linear, void* hip_stream, mi355rt_stats* stats) {
    return guard([&]() -> int {
    if (!d_accum) return fail(MI355RT_ERR_INVALID, "d_accum is null");
    if (sample_end <= sample_begin) return fail(MI355RT_ERR_INVALID, "sample_end must be greater than sample_begin");
    return render_samples(ctx, opt, sample_begin, sample_end, d_accum, d_out_packed, d_out_linear, hip_stream, stats);
    });
}

// Diagnostic hook (not part of the public header): which counter-mode kernel set_scene selected (KERNEL_* in rt_device.h).
int mi355rt_debug_kernel_variant(mi355rt_context* ctx, uint32_t* out) {
    return guard([&]() -> int {
    if (!ctx || !out || !ctx->have_scene) return fail(MI355RT_ERR_INVALID, "context has no scene");
    *out = ctx->variant;
    return MI355RT_OK;
    });
}

// Diagnostic hook (not part of the public header): the STATS_WORDS (40) raw device counters of the last render.
int mi355rt_debug_read_counters(mi355rt_context* ctx, unsigned long long* out40) {
    return guard([&]() -> int {
    if (!ctx || !out40 || !ctx->stats.p) return fail(MI355RT_ERR_INVALID, "no counters");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(out40, ctx->stats.p, STATS_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MI355RT_OK;
    });
}

// Diagnostic hook (not part of the public header): the row tables of the last render on this context -- 3 x n entries (natural, processing,
// out_row; see row_tables()) -- and the per-image-row cost the processing order was built from (`cost`, `height` entries; may be null).
int mi355rt_debug_read_row_tables(mi355rt_context* ctx, uint32_t* tables, uint32_t capacity_entries, uint32_t* n_rows, float* cost, uint32_t cost_capacity, uint32_t* n_cost) {
    return guard([&]() -> int {
    if (!ctx || !n_rows) return fail(MI355RT_ERR_INVALID, "null");
    const size_t n = ctx->rows_host.size();
    *n_rows = (uint32_t)n;
    if (tables) { if (capacity_entries < 3 * n || ctx->tables_host.size() != 3 * n) return fail(MI355RT_ERR_INVALID, "row tables: capacity"); std::memcpy(tables, ctx->tables_host.data(), 3 * n * sizeof(uint32_t)); }
    if (n_cost) *n_cost = (uint32_t)ctx->row_cost.size();
    if (cost) { if (cost_capacity < ctx->row_cost.size()) return fail(MI355RT_ERR_INVALID, "row cost: capacity"); std::memcpy(cost, ctx->row_cost.data(), ctx->row_cost.size() * sizeof(float)); }
    return MI355RT_OK;
    });
}

int mi355rt_debug_read_wave_times(mi355rt_context* ctx, unsigned long long* out, uint32_t capacity_waves, uint32_t* n_waves) {
    return guard([&]() -> int {
    if (!ctx || !out || !n_waves) return fail(MI355RT_ERR_INVALID, "null");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = std::min(capacity_waves, ctx->wave_times_n);
    if (n) HIP_TRY(hipMemcpy(out, ctx->wave_times.p, (size_t)n * WAVE_TIME_WORDS * 8, hipMemcpyDeviceToHost));
    *n_waves = n;
    return MI355RT_OK;
    });
}

// Diagnostic hooks (not part of the public header): one Material::scatter / one HittableList::hit per record through the
// device functions of the render kernels, on the context's resident scene.  Host pointers in and out; records are the
// 16- / 6- / 12-word PODs of rt_device.h.
int mi355rt_debug_scatter(mi355rt_context* ctx, const void* in_records, uint32_t n, void* out_records) {
    return guard([&]() -> int {
    if (!ctx || !ctx->have_scene || !in_records || !out_records) return fail(MI355RT_ERR_INVALID, "debug_scatter: null / no scene");
    HIP_TRY(hipSetDevice(ctx->device));
    const DebugScatterIn* in = static_cast<const DebugScatterIn*>(in_records);
    for (uint32_t i = 0; i < n; ++i) if (in[i].material >= ctx->n_mats) return fail(MI355RT_ERR_INVALID, "debug_scatter: material index");
    DevBuf<DebugScatterIn> d_in; DevBuf<DebugScatterOut> d_out;
    int rc = d_in.ensure(n); if (!rc) rc = d_out.ensure(n);
    if (!rc && n) {
        if (hipMemcpy(d_in.p, in, n * sizeof(DebugScatterIn), hipMemcpyHostToDevice) != hipSuccess || launch_debug_scatter(ctx->mats.p, ctx->textures.p, d_in.p, d_out.p, n, nullptr) != 0 ||
            hipMemcpy(out_records, d_out.p, n * sizeof(DebugScatterOut), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MI355RT_ERR_HIP, "debug_scatter");
    }
    d_in.release(); d_out.release();
    return rc;
    });
}

int mi355rt_debug_hit(mi355rt_context* ctx, const void* in_rays, uint32_t n, void* out_records) {
    return guard([&]() -> int {
    if (!ctx || !ctx->have_scene || !in_rays || !out_records) return fail(MI355RT_ERR_INVALID, "debug_hit: null / no scene");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<DebugHitIn> d_in; DevBuf<DebugHitOut> d_out;
    int rc = d_in.ensure(n); if (!rc) rc = d_out.ensure(n);
    if (!rc && n) {
        if (hipMemcpy(d_in.p, in_rays, n * sizeof(DebugHitIn), hipMemcpyHostToDevice) != hipSuccess ||
            launch_debug_hit(ctx->prims.p, ctx->n_prims, ctx->nodes.p, ctx->tris.p, d_in.p, d_out.p, n, nullptr) != 0 ||
            hipMemcpy(out_records, d_out.p, n * sizeof(DebugHitOut), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MI355RT_ERR_HIP, "debug_hit");
    }
    d_in.release(); d_out.release();
    return rc;
    });
}

int mi355rt_context_set_timing(mi355rt_context* ctx, int enable) {
    return guard([&]() -> int {
    if (!ctx) return fail(MI355RT_ERR_INVALID, "ctx is null");
    ctx->timing = enable != 0; ctx->pool_used = 0; ctx->timed_launches = 0;
    return MI355RT_OK;
    });
}

int mi355rt_context_read_timing(mi355rt_context* ctx, double* render_kernel_ms, double* resolve_kernel_ms, uint32_t* launches) {
    return guard([&]() -> int {
    if (!ctx) return fail(MI355RT_ERR_INVALID, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    double a = 0, c = 0;
    for (size_t i = 0; i + 2 < ctx->pool_used; i += 3) {
        HIP_TRY(hipEventSynchronize(ctx->pool[i + 2]));
        float x = 0, y = 0;
        HIP_TRY(hipEventElapsedTime(&x, ctx->pool[i], ctx->pool[i + 1]));
        HIP_TRY(hipEventElapsedTime(&y, ctx->pool[i + 1], ctx->pool[i + 2]));
        a += x; c += y;
    }
    if (render_kernel_ms) *render_kernel_ms = a;
    if (resolve_kernel_ms) *resolve_kernel_ms = c;
    if (launches) *launches = ctx->timed_launches;
    ctx->pool_used = 0; ctx->timed_launches = 0;
    return mi355rt_context_check(ctx);                               // the timed renders must also have been COMPLETE renders
    });
}

int mi355rt_context_check(mi355rt_context* ctx) {
    return guard([&]() -> int {
    if (!ctx) return fail(MI355RT_ERR_INVALID, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->have_last) HIP_TRY(hipEventSynchronize(ctx->done));     // every render enqueued so far has finished and left its error word
    return report_device_error(ctx);
    });
}

int mi355rt_render(const mi355rt_scene* scene, const mi355rt_camera* camera, const mi355rt_settings* settings,
                   const mi355rt_options* opt, uint32_t* out_packed, float* out_linear, mi355rt_stats* stats) {
    return guard([&]() -> int {
    if (!out_packed) return fail(MI355RT_ERR_INVALID, "out_packed_rgb is null");
    int rc = check_settings(settings); if (rc) return rc;
    uint32_t n_rows = 0;
    rc = mi355rt_rows_selected(settings, opt, &n_rows); if (rc) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MI355RT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    mi355rt_context* ctx = nullptr;
    rc = mi355rt_context_create(dev, &ctx); if (rc) return rc;
    rc = mi355rt_context_set_scene(ctx, scene, camera, settings);
    uint32_t* d_packed = nullptr; float* d_linear = nullptr;
    const size_t npix = (size_t)n_rows * settings->width;
    if (!rc && npix) {
        if (hipMalloc((void**)&d_packed, npix * 4) != hipSuccess) rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_packed)");
        if (!rc && out_linear && hipMalloc((void**)&d_linear, npix * 12) != hipSuccess) rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_linear)");
        mi355rt_stats local{};
        if (!rc) rc = mi355rt_context_render(ctx, opt, d_packed, d_linear, nullptr, stats ? stats : &local);
        if (!rc && hipMemcpy(out_packed, d_packed, npix * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MI355RT_ERR_HIP, "copy back packed");
        if (!rc && out_linear && hipMemcpy(out_linear, d_linear, npix * 12, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MI355RT_ERR_HIP, "copy back linear");
    }
    if (d_packed) (void)hipFree(d_packed);
    if (d_linear) (void)hipFree(d_linear);
    std::string keep; keep.swap(g_err);                               // (destroy may overwrite the message of the failure being reported; swap never throws)
    mi355rt_context_destroy(ctx);
    g_err.swap(keep);
    return rc;
    });
}

// One host process, several GPUs (the shape of the reference's own host: a single `main`, src/main.rs:22-89).
// Row strips are dealt round-robin over `hip_devices` exactly as the one-process-per-GPU path deals them over ranks
// (options.strip_rows; 0 -> 4); every device gets the full scene, renders its strips on its own host thread and
// copies them straight into the caller's row-major image -- the exchange step is the device-to-host copy, no
// collective.  The image is bit-identical to the one-device image (draws are keyed by absolute row / x / sample).
int mi355rt_render_multi(const mi355rt_scene* scene, const mi355rt_camera* camera, const mi355rt_settings* settings,
                         const mi355rt_options* opt, const int* hip_devices, uint32_t n_devices,
                         uint32_t* out_packed, float* out_linear, mi355rt_stats* stats) {
    return guard([&]() -> int {
    if (!out_packed) return fail(MI355RT_ERR_INVALID, "out_packed_rgb is null");
    if (!hip_devices || n_devices == 0) return fail(MI355RT_ERR_INVALID, "hip_devices is empty");
    int rc = check_settings(settings); if (rc) return rc;
    mi355rt_options base{};
    if (opt) base = *opt; else { base.abi_version = MI355RT_ABI_VERSION; base.rng_mode = MI355RT_RNG_CTR; }
    if (base.n_parts > 1) return fail(MI355RT_ERR_INVALID, "render_multi deals the strips itself: leave options.n_parts / part at 0");
    if (base.strip_rows == 0) base.strip_rows = 4;
    RowSel all; rc = select_rows(*settings, &base, all); if (rc) return rc;       // the rows the caller's buffer holds (row window)
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0) return fail(MI355RT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    for (uint32_t d = 0; d < n_devices; ++d)
        if (hip_devices[d] < 0 || hip_devices[d] >= visible) return fail(MI355RT_ERR_INVALID, "hip_devices entry out of range");
    const uint32_t W = settings->width, row0 = all.rows.empty() ? 0u : all.rows.front();

    struct Part { int rc = MI355RT_OK; std::string err; mi355rt_stats st{}; };
    std::vector<Part> parts(n_devices);
    auto work_body = [&](uint32_t d, mi355rt_context*& ctx, uint32_t*& d_packed, float*& d_linear) {
        Part& me = parts[d];
        mi355rt_options o = base; o.n_parts = n_devices; o.part = d;
        RowSel sel;
        if ((me.rc = select_rows(*settings, &o, sel))) { me.err = g_err; return; }
        if (sel.rows.empty()) return;
        const size_t npix = sel.rows.size() * (size_t)W;
        std::vector<uint32_t> h_packed(npix); std::vector<float> h_linear(out_linear ? npix * 3 : 0);
        me.rc = mi355rt_context_create(hip_devices[d], &ctx);
        if (!me.rc) me.rc = mi355rt_context_set_scene(ctx, scene, camera, settings);
        if (!me.rc && hipMalloc((void**)&d_packed, npix * 4) != hipSuccess) me.rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_packed)");
        if (!me.rc && out_linear && hipMalloc((void**)&d_linear, npix * 12) != hipSuccess) me.rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_linear)");
        if (!me.rc) me.rc = mi355rt_context_render(ctx, &o, d_packed, d_linear, nullptr, &me.st);
        if (!me.rc && hipMemcpy(h_packed.data(), d_packed, npix * 4, hipMemcpyDeviceToHost) != hipSuccess) me.rc = fail(MI355RT_ERR_HIP, "copy back packed");
        if (!me.rc && out_linear && hipMemcpy(h_linear.data(), d_linear, npix * 12, hipMemcpyDeviceToHost) != hipSuccess) me.rc = fail(MI355RT_ERR_HIP, "copy back linear");
        if (me.rc) me.err = g_err;
        else for (size_t j = 0; j < sel.rows.size(); ++j) {                        // de-interleave: local row j is image row sel.rows[j]
            const size_t dst = (size_t)(sel.rows[j] - row0) * W;
            std::memcpy(out_packed + dst, h_packed.data() + j * W, (size_t)W * 4);
            if (out_linear) std::memcpy(out_linear + dst * 3, h_linear.data() + j * W * 3, (size_t)W * 12);
        }
    };
    // One part, on whatever thread runs it.  Nothing may leave this function by exception -- on a worker thread that would be
    // std::terminate -- and the device buffers are released on every path.
    auto work = [&](uint32_t d) noexcept {
        mi355rt_context* ctx = nullptr; uint32_t* d_packed = nullptr; float* d_linear = nullptr;
        const int rc = guard([&]() -> int { work_body(d, ctx, d_packed, d_linear); return MI355RT_OK; });
        if (rc != MI355RT_OK && parts[d].rc == MI355RT_OK) { parts[d].rc = rc; try { parts[d].err = g_err; } catch (...) {} }
        if (d_packed) (void)hipFree(d_packed);
        if (d_linear) (void)hipFree(d_linear);
        if (ctx) mi355rt_context_destroy(ctx);
    };
    // One host thread per further device.  A thread that cannot be had (std::system_error: EAGAIN under a thread / process limit) is not
    // an error: that part is rendered on the calling thread instead, after the threads that did start have been joined -- a joinable
    // std::thread must never be destroyed (std::terminate).
    std::vector<std::thread> threads;
    std::vector<uint32_t> inline_parts;
    try { threads.reserve(n_devices); } catch (...) {}
    for (uint32_t d = 1; d < n_devices; ++d) {
        try { threads.emplace_back(work, d); }
        catch (...) { try { inline_parts.push_back(d); } catch (...) { for (auto& t : threads) t.join(); throw; } }
    }
    work(0);
    for (auto& t : threads) t.join();
    for (uint32_t d : inline_parts) work(d);

    mi355rt_stats total{};
    for (uint32_t d = 0; d < n_devices; ++d) {
        if (parts[d].rc) return fail(parts[d].rc, "device " + std::to_string(hip_devices[d]) + ": " + parts[d].err);
        const mi355rt_stats& s = parts[d].st;
        total.render_kernel_ms = std::max(total.render_kernel_ms, s.render_kernel_ms);     // the devices run side by side
        total.resolve_kernel_ms = std::max(total.resolve_kernel_ms, s.resolve_kernel_ms);
        total.total_ms = std::max(total.total_ms, s.total_ms);
        total.samples += s.samples; total.rays += s.rays; total.rows_rendered += s.rows_rendered; total.bands += s.bands;
        total.grid_blocks = std::max(total.grid_blocks, s.grid_blocks); total.block_threads = s.block_threads ? s.block_threads : total.block_threads;
        total.kernel_vgprs = s.kernel_vgprs ? s.kernel_vgprs : total.kernel_vgprs; total.kernel_sgprs = s.kernel_sgprs ? s.kernel_sgprs : total.kernel_sgprs;
    }
    if (stats) *stats = total;
    return MI355RT_OK;
    });
}

// Host-buffer progressive render: what a preview window (src/main.rs:60-75) would be fed from.
int mi355rt_render_progressive(const mi355rt_scene* scene, const mi355rt_camera* camera, const mi355rt_settings* settings,
                               const mi355rt_options* opt, uint32_t chunk_spp, mi355rt_progress_fn on_chunk, void* user,
                               uint32_t* out_packed, float* out_linear, mi355rt_stats* stats) {
    return guard([&]() -> int {
    if (!out_packed) return fail(MI355RT_ERR_INVALID, "out_packed_rgb is null");
    if (chunk_spp == 0) return fail(MI355RT_ERR_INVALID, "chunk_spp is 0");
    int rc = check_settings(settings); if (rc) return rc;
    uint32_t n_rows = 0;
    rc = mi355rt_rows_selected(settings, opt, &n_rows); if (rc) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MI355RT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    mi355rt_context* ctx = nullptr;
    rc = mi355rt_context_create(dev, &ctx); if (rc) return rc;
    rc = mi355rt_context_set_scene(ctx, scene, camera, settings);
    uint32_t* d_packed = nullptr; float* d_linear = nullptr; float* d_accum = nullptr;
    const size_t npix = (size_t)n_rows * settings->width;
    mi355rt_stats total{};
    if (!rc && npix) {
        if (hipMalloc((void**)&d_packed, npix * 4) != hipSuccess) rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_packed)");
        if (!rc && hipMalloc((void**)&d_accum, npix * 16) != hipSuccess) rc = fail(MI355RT_ERR_OOM, "hipMalloc(accum)");
        if (!rc && out_linear && hipMalloc((void**)&d_linear, npix * 12) != hipSuccess) rc = fail(MI355RT_ERR_OOM, "hipMalloc(out_linear)");
        const uint32_t spp = settings->samples_per_pixel;
        for (uint32_t s0 = 0; !rc && s0 < spp; ) {
            const uint32_t s1 = s0 + std::min(chunk_spp, spp - s0);
            mi355rt_stats st{};
            rc = mi355rt_context_render_progressive(ctx, opt, s0, s1, d_accum, d_packed, d_linear, nullptr, &st);
            if (rc) break;
            total.render_kernel_ms += st.render_kernel_ms; total.resolve_kernel_ms += st.resolve_kernel_ms; total.total_ms += st.total_ms;
            total.samples += st.samples; total.rays += st.rays; total.bands += st.bands;
            total.rows_rendered = st.rows_rendered; total.grid_blocks = st.grid_blocks; total.block_threads = st.block_threads;
            total.kernel_vgprs = st.kernel_vgprs; total.kernel_sgprs = st.kernel_sgprs;
            const bool last = s1 == spp;
            if (on_chunk || last) {
                if (hipMemcpy(out_packed, d_packed, npix * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MI355RT_ERR_HIP, "copy back packed"); break; }
                if (out_linear && hipMemcpy(out_linear, d_linear, npix * 12, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MI355RT_ERR_HIP, "copy back linear"); break; }
            }
            s0 = s1;
            if (on_chunk && on_chunk(user, s1, spp, out_packed) != 0) break;            // the caller stops early: outputs hold s1 samples
        }
    }
    if (stats) *stats = total;
    if (d_packed) (void)hipFree(d_packed);
    if (d_linear) (void)hipFree(d_linear);
    if (d_accum) (void)hipFree(d_accum);
    std::string keep; keep.swap(g_err);                               // (destroy may overwrite the message of the failure being reported; swap never throws)
    mi355rt_context_destroy(ctx);
    g_err.swap(keep);
    return rc;
    });
}

}  // extern "C"
