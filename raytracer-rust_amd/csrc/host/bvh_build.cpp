// bvh_build.cpp -- mi355rt_bvh_build: the producer of the BVH arrays the kernels consume.
//
// Restates BVHNode::new (src/acceleration/bvh.rs:15-76) over object-space triangles:
//   bounds over all vertices (Aabb::add_point, aabb.rs:18-25); leaf when n <= 4 or depth >= 25;
//   split axis = x only if its extent is strictly the largest, else y if > z, else z; sort the index
//   slice by centroid[axis] with centroid = (v0 + v1 + v2) * (1/3); split at n/2.
// The reference topology is load-bearing for parity (zero-thickness leaf boxes
// never hit), so this builder must not be "improved".  Tie order of Rust's sort_unstable_by is
// unspecified; ties keep their current slice order here (std::stable_sort).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"
#include "host_common.hpp"

namespace mi355rt_host {

namespace {

struct Builder {
    const mi355rt_triangle* tris;
    std::vector<mi355rt_bvh_node> nodes;
    std::vector<uint32_t> leaf_indices;
    uint32_t max_depth = 0;

    static float centroid_axis(const mi355rt_triangle& t, int axis) {
        float s = (t.v0[axis] + t.v1[axis]) + t.v2[axis];
        return s * (1.0f / 3.0f);
    }

    uint32_t make_leaf(uint32_t slot, const uint32_t* idx, size_t n) {
        nodes[slot].left = nodes[slot].right = 0;
        nodes[slot].first_index = (uint32_t)leaf_indices.size();
        nodes[slot].index_count = (uint32_t)n;
        leaf_indices.insert(leaf_indices.end(), idx, idx + n);
        return slot;
    }

    uint32_t build(uint32_t* idx, size_t n, uint32_t depth) {
        const uint32_t slot = (uint32_t)nodes.size();
        nodes.emplace_back();
        if (depth > max_depth) max_depth = depth;
        const float inf = std::numeric_limits<float>::infinity();
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (size_t i = 0; i < n; ++i) {
            const mi355rt_triangle& t = tris[idx[i]];
            const float* vs[3] = {t.v0, t.v1, t.v2};
            for (const float* v : vs)
                for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], v[a]); mx[a] = std::fmax(mx[a], v[a]); }
        }
        std::memcpy(nodes[slot].bmin, mn, 12); std::memcpy(nodes[slot].bmax, mx, 12);
        const size_t MAX_DEPTH = 25, MIN_TRIANGLES_PER_LEAF = 4;
        if (n <= MIN_TRIANGLES_PER_LEAF || depth >= MAX_DEPTH) return make_leaf(slot, idx, n);
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
        std::stable_sort(idx, idx + n, [&](uint32_t a, uint32_t b) { return centroid_axis(tris[a], axis) < centroid_axis(tris[b], axis); });
        const size_t mid = n / 2;
        if (mid == 0 || mid == n) return make_leaf(slot, idx, n);
        const uint32_t l = build(idx, mid, depth + 1);
        const uint32_t r = build(idx + mid, n - mid, depth + 1);
        nodes[slot].left = l; nodes[slot].right = r; nodes[slot].first_index = 0; nodes[slot].index_count = 0;
        return slot;
    }
};

}  // namespace

int bvh_build(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
              uint32_t& max_depth) {
    if (!tris || n == 0) return set_error(MI355RT_ERR_INVALID, "bvh_build: no triangles");
    Builder b; b.tris = tris;
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;          // mesh_object.rs:44
    b.build(idx.data(), n, 0);
    nodes.swap(b.nodes); indices.swap(b.leaf_indices); max_depth = b.max_depth;
    return MI355RT_OK;
}

}  // namespace mi355rt_host

extern "C" int mi355rt_bvh_build(const mi355rt_triangle* triangles, uint32_t n_triangles, mi355rt_bvh_node* out_nodes,
                                 uint32_t* inout_n_nodes, uint32_t* out_indices, uint32_t* inout_n_indices, uint32_t* out_max_depth) {
    using namespace mi355rt_host;
    if (!inout_n_nodes || !inout_n_indices) return set_error(MI355RT_ERR_INVALID, "bvh_build: count pointers are null");
    std::vector<mi355rt_bvh_node> nodes; std::vector<uint32_t> indices; uint32_t md = 0;
    int rc;
    try { rc = bvh_build(triangles, n_triangles, nodes, indices, md); }
    catch (const std::exception& e) { return set_error(MI355RT_ERR_OOM, std::string("bvh_build: ") + e.what()); }   // nothing is thrown across the C ABI
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
}
