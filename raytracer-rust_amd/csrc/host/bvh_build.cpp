// bvh_build.cpp -- mi355rt_bvh_build: the producer of the BVH arrays the kernels consume.
//
// Restates BVHNode::new (src/acceleration/bvh.rs:15-76) over object-space triangles:
//   bounds over all vertices (Aabb::add_point, aabb.rs:18-25); leaf when n <= 4 or depth >= 25;
//   split axis = x only if its extent is strictly the largest, else y if > z, else z; sort the index
//   slice by centroid[axis] with centroid = (v0 + v1 + v2) * (1/3); split at n/2.
// The reference topology is load-bearing for parity (zero-thickness leaf boxes
// never hit), so this builder must not be "improved" -- down to the order of equal centroids, which Rust's
// sort_unstable_by leaves unspecified but deterministic: the sort itself is restated (rust_sort_unstable.hpp).
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <stdexcept>
#include <system_error>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"
#include "host_common.hpp"
#include "rust_sort_unstable.hpp"

namespace mi355rt_host {

namespace {

// The shape of the tree is a function of the triangle COUNT alone (leaf when n <= 4 or depth >= 25, else split at n / 2,
// bvh.rs:31-60), so every subtree's node count -- and with it every node's slot in the pre-order array and every leaf's
// range in the index array -- is known before a single triangle is looked at.  That makes the two halves of a split
// independent jobs that write into disjoint parts of preallocated arrays: the halves are built by different threads
// (while threads are free) and the result is, node for node and index for index, the array the serial recursion produces.
struct Builder {
    const mi355rt_triangle* tris = nullptr;
    mi355rt_bvh_node* nodes = nullptr;         // preallocated: subtree_nodes(n, 0)
    uint32_t* leaf_indices = nullptr;          // preallocated: n
    std::atomic<uint32_t> max_depth{0};
    std::atomic<int> spare_threads{0};         // threads that may still be started
    std::atomic<bool> worker_failed{false};    // a helper thread's build threw (allocation in the sort): reported by the caller, never thrown across a thread boundary
    // Node count of the subtree over n triangles at `depth`.  Halving yields at most two distinct sizes per level, so the table
    // filled by the first call (before any helper thread exists) has ~2 x depth entries and is read-only afterwards.
    std::unordered_map<uint64_t, uint32_t> sizes;

    static constexpr size_t MAX_DEPTH = 25, MIN_TRIANGLES_PER_LEAF = 4;
    uint32_t fill_sizes(size_t n, uint32_t depth) {
        if (n <= MIN_TRIANGLES_PER_LEAF || depth >= MAX_DEPTH) return 1u;
        const uint64_t key = ((uint64_t)n << 8) | depth;
        auto it = sizes.find(key);
        if (it != sizes.end()) return it->second;
        const size_t mid = n / 2;
        const uint32_t v = 1u + fill_sizes(mid, depth + 1) + fill_sizes(n - mid, depth + 1);
        sizes.emplace(key, v);
        return v;
    }
    uint32_t subtree_nodes(size_t n, uint32_t depth) const {
        if (n <= MIN_TRIANGLES_PER_LEAF || depth >= MAX_DEPTH) return 1u;
        return sizes.at(((uint64_t)n << 8) | depth);
    }
    static float centroid_axis(const mi355rt_triangle& t, int axis) {
        float s = (t.v0[axis] + t.v1[axis]) + t.v2[axis];
        return s * (1.0f / 3.0f);
    }

    // Builds the subtree over idx[0..n) into nodes[slot ...]; its leaves own leaf_indices[index_offset .. index_offset + n).
    void build(uint32_t* idx, size_t n, uint32_t depth, uint32_t slot, uint32_t index_offset) {
        uint32_t seen = max_depth.load(std::memory_order_relaxed);
        while (depth > seen && !max_depth.compare_exchange_weak(seen, depth, std::memory_order_relaxed)) {}
        const float inf = std::numeric_limits<float>::infinity();
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (size_t i = 0; i < n; ++i) {
            const mi355rt_triangle& t = tris[idx[i]];
            const float* vs[3] = {t.v0, t.v1, t.v2};
            for (const float* v : vs)
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], v[a]); mx[a] = std::fmax(mx[a], v[a]); }
        }
        mi355rt_bvh_node& node = nodes[slot];
        std::memcpy(node.bmin, mn, 12); std::memcpy(node.bmax, mx, 12);
        if (n <= MIN_TRIANGLES_PER_LEAF || depth >= MAX_DEPTH) {
            node.left = node.right = 0; node.first_index = index_offset; node.index_count = (uint32_t)n;
            std::memcpy(leaf_indices + index_offset, idx, n * sizeof(uint32_t));
            return;
        }
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
        rustsort::sort_unstable_by(idx, n, [&](uint32_t a, uint32_t b) { return centroid_axis(tris[a], axis) < centroid_axis(tris[b], axis); });   // bvh.rs:45-53
        const size_t mid = n / 2;                                              // > 0 and < n because n > 4
        const uint32_t l = slot + 1u, r = l + subtree_nodes(mid, depth + 1);
        node.left = l; node.right = r; node.first_index = 0; node.index_count = 0;
        // the right half on another thread while one is free and the job is worth it, the left half here
        bool reserved = n >= 2048 && spare_threads.fetch_sub(1, std::memory_order_relaxed) > 0;
        if (n >= 2048 && !reserved) spare_threads.fetch_add(1, std::memory_order_relaxed);   // undo the reservation that failed
        if (reserved) {
            std::thread other;
            try {
                // nothing may leave the helper's thread function (it would be std::terminate): a throw inside is recorded instead
                other = std::thread([=] { try { build(idx + mid, n - mid, depth + 1, r, index_offset + (uint32_t)mid); }
                                          catch (...) { worker_failed.store(true, std::memory_order_relaxed); } });
            } catch (const std::system_error&) {                  // no thread to be had (EAGAIN under a process / thread limit): build inline
                spare_threads.fetch_add(1, std::memory_order_relaxed);
                reserved = false;
            }
            if (reserved) {
                try { build(idx, mid, depth + 1, l, index_offset); }
                catch (...) { other.join(); spare_threads.fetch_add(1, std::memory_order_relaxed); throw; }
                other.join();
                spare_threads.fetch_add(1, std::memory_order_relaxed);
            }
        }
        if (!reserved) {
            build(idx, mid, depth + 1, l, index_offset);
            build(idx + mid, n - mid, depth + 1, r, index_offset + (uint32_t)mid);
        }
    }
};

}  // namespace

int bvh_build(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
              uint32_t& max_depth) {
    if (!tris || n == 0) return set_error(MI355RT_ERR_INVALID, "bvh_build: no triangles");
    return bvh_build_threads(tris, n, nodes, indices, max_depth, 0);
}

// n_threads: 0 = one per hardware thread (at most 16), 1 = the plain serial recursion.  Any value gives the same arrays.
int bvh_build_threads(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
                      uint32_t& max_depth, int n_threads) {
    if (!tris || n == 0) return set_error(MI355RT_ERR_INVALID, "bvh_build: no triangles");
    if (n_threads <= 0) { n_threads = (int)std::thread::hardware_concurrency(); if (n_threads <= 0) n_threads = 1; if (n_threads > 16) n_threads = 16; }
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;          // mesh_object.rs:44
    Builder b;
    nodes.assign(b.fill_sizes(n, 0), mi355rt_bvh_node{});
    indices.assign(n, 0u);
    b.tris = tris; b.nodes = nodes.data(); b.leaf_indices = indices.data(); b.spare_threads.store(n_threads - 1);
    b.build(idx.data(), n, 0, 0u, 0u);
    if (b.worker_failed.load()) return set_error(MI355RT_ERR_OOM, "bvh_build: a helper thread ran out of memory");
    max_depth = b.max_depth.load();
    return MI355RT_OK;
}

}  // namespace mi355rt_host

// Test / tuning hook (not in the public header): the same build with an explicit thread count (1 = serial recursion).
extern "C" int mi355rt_bvh_build_threads(const mi355rt_triangle* triangles, uint32_t n_triangles, int n_threads, mi355rt_bvh_node* out_nodes,
                                         uint32_t* out_indices, uint32_t* out_n_nodes, uint32_t* out_max_depth) {
    using namespace mi355rt_host;
    return mi355rt_host::guard("bvh_build", MI355RT_ERR_INVALID, [&]() -> int {     // nothing is thrown across the C ABI (host_common.hpp)
    std::vector<mi355rt_bvh_node> nodes; std::vector<uint32_t> indices; uint32_t md = 0;
    const int rc = bvh_build_threads(triangles, n_triangles, nodes, indices, md, n_threads);
    if (rc) return rc;
    if (out_nodes) std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(mi355rt_bvh_node));
    if (out_indices) std::memcpy(out_indices, indices.data(), indices.size() * sizeof(uint32_t));
    if (out_n_nodes) *out_n_nodes = (uint32_t)nodes.size();
    if (out_max_depth) *out_max_depth = md;
    return MI355RT_OK;
    });
}

extern "C" int mi355rt_bvh_build(const mi355rt_triangle* triangles, uint32_t n_triangles, mi355rt_bvh_node* out_nodes,
                                 uint32_t* inout_n_nodes, uint32_t* out_indices, uint32_t* inout_n_indices, uint32_t* out_max_depth) {
    using namespace mi355rt_host;
    return mi355rt_host::guard("bvh_build", MI355RT_ERR_INVALID, [&]() -> int {     // nothing is thrown across the C ABI (host_common.hpp)
    if (!inout_n_nodes || !inout_n_indices) return set_error(MI355RT_ERR_INVALID, "bvh_build: count pointers are null");
    std::vector<mi355rt_bvh_node> nodes; std::vector<uint32_t> indices; uint32_t md = 0;
    const int rc = bvh_build(triangles, n_triangles, nodes, indices, md);
    if (rc) return rc;
    if (out_nodes || out_indices) {
        if (!out_nodes || !out_indices || *inout_n_nodes < nodes.size() || *inout_n_indices < indices.size())
            return set_error(MI355RT_ERR_INVALID, "bvh_build: output arrays too small");
        std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(mi355rt_bvh_node));
        std::memcpy(out_indices, indices.data(), indices.size() * sizeof(uint32_t));
    }
    *inout_n_nodes = (uint32_t)nodes.size(); *inout_n_indices = (uint32_t)indices.size();
    if (out_max_depth) *out_max_depth = md;
    return MI355RT_OK;
    });
}
