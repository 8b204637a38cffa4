// mesh_io.cpp -- Mesh::from_obj / Mesh::from_wo3 (src/mesh/mesh_object.rs:59-259) + Triangle::new
// (src/mesh/triangle.rs:14-25): file -> object-space triangle soup in file order.
//
// tobj 4.0.3 (Cargo.lock pin, not under /root/reference) is restated for what from_obj consumes:
// `v` positions and `f` faces, faces fan-triangulated from their first vertex (GPU_LOAD_OPTIONS =
// triangulate + single_index; single_index may renumber vertices but never changes a corner's position).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "mesh_io.hpp"
#include "xform.hpp"

namespace mi355rt_host {

// Triangle::new + the degenerate filter |cross|^2 < EPSILON^2 (mesh_object.rs:128-134, :223-235)
static bool make_triangle(V3 v0, V3 v1, V3 v2, mi355rt_triangle& out) {
    const V3 e1 = v1 - v0, e2 = v2 - v0;
    const V3 c = cross(e1, e2);
    const V3 n = normalized(c);
    out = {{v0.x, v0.y, v0.z}, {v1.x, v1.y, v1.z}, {v2.x, v2.y, v2.z}, {n.x, n.y, n.z}};
    return !(dot(c, c) < EPSILON * EPSILON);
}

// Malformed input fails the whole mesh, as `tobj::load_obj(..)?` does (mesh_object.rs:64): a `v` line without three
// floats (tobj PositionParseError), a face corner that is not an integer (FaceParseError) or that points outside the
// vertices read so far (FaceVertexOutOfBounds), and a file without any face (:67-69).  (mesh_object.rs:98-104's "index count
// not a multiple of 3" cannot occur: with triangulate + ignore_points + ignore_lines every face tobj keeps is triangles.)
int load_obj(const std::string& path, std::vector<mi355rt_triangle>& tris) {
    std::ifstream f(path);
    if (!f) return set_error(MI355RT_ERR_IO, "cannot open OBJ " + path);
    std::vector<V3> verts;
    std::vector<long> face;
    std::string line;
    size_t line_no = 0, n_faces = 0;
    auto at = [&](const char* what) { return set_error(MI355RT_ERR_IO, std::string("OBJ ") + path + ":" + std::to_string(line_no) + ": " + what); };
    while (std::getline(f, line)) {
        ++line_no;
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            double x = 0, y = 0, z = 0;
            if (std::sscanf(s + 1, "%lf %lf %lf", &x, &y, &z) != 3) return at("vertex position needs three numbers");
            verts.push_back({(float)x, (float)y, (float)z});
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            std::istringstream ss(s + 1);
            std::string tok;
            while (ss >> tok) {
                char* end = nullptr;
                const long i = std::strtol(tok.c_str(), &end, 10);         // "a", "a/b", "a/b/c", "a//c"
                if (end == tok.c_str() || (*end != 0 && *end != '/')) return at("face corner is not an index");
                const long v = i > 0 ? i - 1 : (long)verts.size() + i;
                if (i == 0 || v < 0 || v >= (long)verts.size()) return at("face corner index out of bounds");
                face.push_back(v);
            }
            if (face.empty()) return at("face without corners");
            // tobj 4.0.3 GPU_LOAD_OPTIONS = triangulate + single_index + ignore_points + ignore_lines (mesh_object.rs:64): an `f`
            // with one or two corners is validated like any other and then dropped -- it never reaches mesh.indices, so it can
            // neither add a triangle nor break the "multiple of 3" check of mesh_object.rs:98-104.  The model still exists.
            ++n_faces;
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                mi355rt_triangle t;
                if (make_triangle(verts[face[0]], verts[face[k]], verts[face[k + 1]], t)) tris.push_back(t);
            }
        }
    }
    if (n_faces == 0) return set_error(MI355RT_ERR_IO, "No models found in OBJ file: " + path);
    return MI355RT_OK;
}

// Mesh::from_wo3 INCLUDING its index-stride bug: the file stores 4 u32 per
// triangle (v0, v1, v2, material); the reference reads 3 u32 per iteration for num_tris iterations
// (mesh_object.rs:190-192), i.e. it consumes the first 3/4 of the index stream with a sliding phase.
// `four_index_stride` (opt-in, never the default: it changes the image) reads the file as Tungsten wrote it.
int load_wo3(const std::string& path, std::vector<mi355rt_triangle>& tris, bool four_index_stride) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return set_error(MI355RT_ERR_IO, "cannot open WO3 " + path);
    std::vector<unsigned char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto rd_u64 = [&](size_t off, uint64_t& v) { if (off + 8 > b.size()) return false; std::memcpy(&v, b.data() + off, 8); return true; };
    uint64_t nv = 0, nt = 0;
    if (!rd_u64(0, nv)) return set_error(MI355RT_ERR_IO, "WO3 truncated header");
    const size_t voff = 8;
    if (nv > (b.size() - voff) / 32) return set_error(MI355RT_ERR_IO, "WO3 truncated vertices");
    const size_t toff = voff + (size_t)nv * 32;
    if (!rd_u64(toff, nt)) return set_error(MI355RT_ERR_IO, "WO3 truncated triangle header");
    const size_t ioff = toff + 8;
    const size_t stride = four_index_stride ? 16 : 12;
    if (nt > (b.size() - ioff) / stride) return set_error(MI355RT_ERR_IO, "WO3 truncated indices");   // read_u32 would hit EOF -> Err
    std::vector<V3> verts((size_t)nv);
    for (size_t i = 0; i < (size_t)nv; ++i) {
        float p[3]; std::memcpy(p, b.data() + voff + i * 32, 12);      // position, then normal(3) + uv(2) skipped
        verts[i] = {p[0], p[1], p[2]};
    }
    for (size_t i = 0; i < (size_t)nt; ++i) {
        uint32_t id[3]; std::memcpy(id, b.data() + ioff + i * stride, 12);
        if (id[0] >= nv || id[1] >= nv || id[2] >= nv) continue;         // mesh_object.rs:202-214
        mi355rt_triangle t;
        if (make_triangle(verts[id[0]], verts[id[1]], verts[id[2]], t)) tris.push_back(t);
    }
    return MI355RT_OK;
}

}  // namespace mi355rt_host
