// scene_loader.cpp -- mi355rt_scene_load_json: the stand-in for the Rust host's loader.
//
// Restates load_scene_from_json (src/tungsten/parser.rs:245-815) with serde's observable behaviour
// (SURVEY.md App. C), Camera::new (src/camera.rs:14-31), Quad::new_transformed
// (src/tungsten/objects/quad.rs:26-79), Cube::new_transformed (src/objects/cube.rs:20-29), the sphere
// light rule (parser.rs:566-576), and flattens the result into the POD arrays of include/mi355rt.h.
// Material indices: parsed `bsdfs` in file order, then the materials primitives create for themselves
// (emitters, inline plane materials, magenta fallbacks) in primitive order.
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <memory>
#include "host_common.hpp"
#include "json.hpp"
#include "mesh_io.hpp"
#include "xform.hpp"

using namespace mi355rt_host;

struct mi355rt_loaded_scene {
    std::vector<mi355rt_primitive> prims;
    std::vector<mi355rt_material> mats;
    std::vector<mi355rt_mesh> meshes;
    std::vector<mi355rt_triangle> tris;
    std::vector<mi355rt_bvh_node> nodes;
    std::vector<uint32_t> indices;
    std::vector<float> sky; uint32_t sky_w = 0, sky_h = 0;     // scene.skybox_hdr_image (parser.rs:497-509)
    mi355rt_scene scene{};
    mi355rt_camera camera{};
    mi355rt_settings settings{};
};

namespace {

struct ParseError { std::string msg; };
[[noreturn]] void bad(const std::string& m) { throw ParseError{m}; }

float num_f32(const JsonValue& v) { if (!v.is_number()) bad("expected a number"); return (float)v.num; }
// serde's `usize` fields (parser.rs:64-74,169-177): a non-negative JSON integer token -- 1.5, 1.0, 1e3 and -1 are
// "invalid type" errors there, not truncations.  The C ABI carries these as u32, so larger values are refused too.
uint32_t num_usize(const JsonValue& v, const char* what) {
    if (!v.is_number() || !v.integral) bad(std::string("invalid type for `") + what + "`: expected usize");
    if (!(v.num >= 0.0 && v.num <= 4294967295.0)) bad(std::string("`") + what + "` is out of range (this build carries it as u32)");
    return (uint32_t)v.num;
}

// Vec3Config (parser.rs:23-28): derived struct -> serde accepts {"x","y","z"} or a 3-sequence
V3 vec3cfg(const JsonValue& v) {
    if (v.is_object()) {
        const JsonValue *x = v.get("x"), *y = v.get("y"), *z = v.get("z");
        if (!x || !y || !z) bad("Vec3Config needs x, y, z");
        return {num_f32(*x), num_f32(*y), num_f32(*z)};
    }
    if (v.is_array() && v.arr.size() == 3) return {num_f32(v.arr[0]), num_f32(v.arr[1]), num_f32(v.arr[2])};
    bad("bad Vec3Config");
}
// ColorConfig (parser.rs:36-37): exactly a 3-tuple of numbers
bool color3(const JsonValue& v, float out[3]) {
    if (!v.is_array() || v.arr.size() != 3) return false;
    for (int i = 0; i < 3; ++i) { if (!v.arr[i].is_number()) return false; out[i] = (float)v.arr[i].num; }
    return true;
}

mi355rt_material make_mat(uint32_t kind, const float albedo[3], float p0 = 0.0f) {
    mi355rt_material m; std::memset(&m, 0, sizeof m);
    m.kind = kind; std::memcpy(m.albedo, albedo, 12); m.p0 = p0;
    return m;
}
mi355rt_material make_mat(uint32_t kind, float r, float g, float b, float p0 = 0.0f) { const float a[3] = {r, g, b}; return make_mat(kind, a, p0); }

struct MetalEntry { const char* name; float eta[3], k[3]; };
const MetalEntry METALS[] = {                              // MetalType::ior_k, tungsten/materials.rs:115-152
    {"cu", {0.200f, 1.090f, 1.420f}, {3.910f, 2.570f, 2.300f}}, {"au", {0.170f, 0.350f, 1.500f}, {3.140f, 2.300f, 1.920f}},
    {"ag", {0.155f, 0.145f, 0.135f}, {3.910f, 2.610f, 2.370f}}, {"al", {1.360f, 0.965f, 0.620f}, {7.570f, 6.690f, 5.440f}},
    {"ni", {1.920f, 1.920f, 1.920f}, {3.670f, 3.670f, 3.670f}}, {"ti", {2.740f, 2.740f, 2.740f}, {3.170f, 3.170f, 3.170f}},
    {"fe", {2.870f, 2.870f, 2.870f}, {3.140f, 3.140f, 3.140f}}, {"pb", {1.910f, 1.910f, 1.910f}, {3.180f, 3.180f, 3.180f}},
};
std::string lower(std::string s) { for (auto& c : s) c = (char)std::tolower((unsigned char)c); return s; }
const MetalEntry& metal_by_name(const std::string& n) { for (const auto& m : METALS) if (n == m.name) return m; return METALS[0]; }

float checker_inv_scale(float scale) { return (std::fabs(scale) < 1e-6f) ? 1.0f : 1.0f / scale; }   // CheckerTexture::new, materials.rs:80-87

mi355rt_material rough_conductor(const float albedo[3], float roughness, const MetalEntry& me, bool ggx) {
    mi355rt_material m = make_mat(ggx ? MI355RT_MAT_ROUGH_GGX : MI355RT_MAT_ROUGH_BECKMANN, albedo, std::fmax(roughness, 0.01f));   // materials.rs:177
    std::memcpy(m.eta, me.eta, 12); std::memcpy(m.k, me.k, 12);
    return m;
}

// AlbedoConfig (untagged: Solid | GrayscaleSolid | Checker), parser.rs:82-88
bool lambert_from_albedo(const JsonValue& alb, mi355rt_material& out) {
    float c[3];
    if (color3(alb, c)) { out = make_mat(MI355RT_MAT_LAMBERT_SOLID, c); return true; }
    if (alb.is_number()) { float g = (float)alb.num; out = make_mat(MI355RT_MAT_LAMBERT_SOLID, g, g, g); return true; }
    if (alb.is_object()) {
        const JsonValue *on = alb.get("on_color"), *off = alb.get("off_color");
        float con[3], coff[3];
        if (!on || !off || !color3(*on, con) || !color3(*off, coff)) return false;
        const JsonValue* ru = alb.opt("res_u"); const JsonValue* rv = alb.opt("res_v");
        if ((ru && !ru->is_number()) || (rv && !rv->is_number())) return false;
        float scale = ru ? (float)ru->num : (rv ? (float)rv->num : 10.0f);        // parser.rs:435
        out = make_mat(MI355RT_MAT_LAMBERT_CHECKER, con, checker_inv_scale(scale));
        std::memcpy(out.aux, coff, 12);
        return true;
    }
    return false;
}

// one `bsdfs[]` entry, parser.rs:310-495.  false = skipped with a warning.
bool parse_bsdf(const JsonValue& b, mi355rt_material& out) {
    const JsonValue* tv = b.get("type");
    if (!tv || !tv->is_string()) bad("bsdf without type");
    const std::string& t = tv->str;
    const JsonValue* alb = b.opt("albedo");
    const JsonValue* iorv = b.opt("ior");
    if (iorv && !iorv->is_number()) bad("bsdf.ior must be a number");
    float c[3];
    if (t == "lambert") return alb && lambert_from_albedo(*alb, out);
    if (t == "plastic") {
        float a[3] = {0.8f, 0.8f, 0.8f};
        if (alb) { if (color3(*alb, c)) std::memcpy(a, c, 12); else if (alb->is_number()) a[0] = a[1] = a[2] = (float)alb->num; }
        out = make_mat(MI355RT_MAT_PLASTIC, a, iorv ? (float)iorv->num : 1.5f);
        return true;
    }
    if (t == "null") { out = make_mat(MI355RT_MAT_LAMBERT_SOLID, 0.f, 0.f, 0.f); return true; }      // parser.rs:357-359
    if (t == "glass" || t == "dielectric") { out = make_mat(MI355RT_MAT_DIELECTRIC, 0.f, 0.f, 0.f, iorv ? (float)iorv->num : 1.5f); return true; }
    if (t == "rough_conductor") {
        float a[3] = {1.0f, 1.0f, 1.0f};
        if (alb) { if (color3(*alb, c)) std::memcpy(a, c, 12); else if (alb->is_number()) a[0] = a[1] = a[2] = (float)alb->num; }
        const JsonValue* r = b.opt("roughness");
        if (r && !r->is_number()) bad("bsdf.roughness must be a number");
        const JsonValue* mt = b.opt("material"); const JsonValue* dist = b.opt("distribution");
        if ((mt && !mt->is_string()) || (dist && !dist->is_string())) bad("bsdf.material/distribution must be strings");
        const MetalEntry& me = mt ? metal_by_name(lower(mt->str)) : METALS[0];
        bool ggx = dist ? (lower(dist->str) != "beckmann") : true;
        out = rough_conductor(a, r ? (float)r->num : 0.1f, me, ggx);
        return true;
    }
    return false;                                          // unsupported type (parser.rs:418-424)
}

// `plane.material`: MaterialTypeConfig, externally tagged PascalCase (parser.rs:90-119, :590-630)
mi355rt_material inline_plane_material(const JsonValue& mc) {
    if (!mc.is_object() || mc.obj.size() != 1) bad("plane.material must be a single-key map");
    const std::string& tag = mc.obj[0].first; const JsonValue& body = mc.obj[0].second;
    auto need = [&](const char* k) -> const JsonValue& { const JsonValue* v = body.get(k); if (!v) bad(std::string("plane.material missing ") + k); return *v; };
    float c[3];
    if (tag == "Lambertian") { mi355rt_material m; if (!lambert_from_albedo(need("albedo"), m)) bad("plane Lambertian albedo"); return m; }
    if (tag == "Metal") {
        if (!color3(need("albedo"), c)) bad("plane Metal albedo");
        float fuzz = num_f32(need("fuzz")); fuzz = fuzz < 0.0f ? 0.0f : (fuzz > 1.0f ? 1.0f : fuzz);   // Metal::new, material.rs:79-84
        return make_mat(MI355RT_MAT_METAL, c, fuzz);
    }
    if (tag == "Glass") return make_mat(MI355RT_MAT_DIELECTRIC, 0.f, 0.f, 0.f, num_f32(need("index_of_refraction")));
    if (tag == "Plastic") { if (!color3(need("albedo"), c)) bad("plane Plastic albedo"); return make_mat(MI355RT_MAT_PLASTIC, c, num_f32(need("ior"))); }
    if (tag == "RoughConductor") {
        if (!color3(need("albedo"), c)) bad("plane RoughConductor albedo");
        const JsonValue& mt = need("metal_type");
        MetalEntry me = METALS[0];
        if (mt.is_string()) me = metal_by_name(lower(mt.str));
        else if (mt.is_object() && mt.get("Custom")) {                          // MetalType::Custom(Color) -> (c, 1)
            const JsonValue& cc = *mt.get("Custom");
            const JsonValue *r = cc.get("r"), *g = cc.get("g"), *bb = cc.get("b");
            if (!r || !g || !bb) bad("MetalType::Custom needs r, g, b");
            me.eta[0] = num_f32(*r); me.eta[1] = num_f32(*g); me.eta[2] = num_f32(*bb); me.k[0] = me.k[1] = me.k[2] = 1.0f;
        } else bad("plane RoughConductor metal_type");
        const JsonValue& d = need("distribution");
        return rough_conductor(c, num_f32(need("roughness")), me, d.is_string() && d.str == "Ggx");
    }
    return make_mat(MI355RT_MAT_LAMBERT_SOLID, 1.f, 1.f, 1.f);                   // Texture / Light -> white Lambertian (parser.rs:626-629)
}

// transform -> Mat4, parser.rs:647-674 / :736-763 / :777-804
Mat4 object_matrix(const JsonValue& tr) {
    if (!tr.is_object()) bad("transform must be a map");
    V3 pos{0, 0, 0}, scale{1, 1, 1}, rot{0, 0, 0};
    if (const JsonValue* p = tr.opt("position")) pos = vec3cfg(*p);
    if (const JsonValue* s = tr.opt("scale")) { if (s->is_number()) { float u = (float)s->num; scale = {u, u, u}; } else scale = vec3cfg(*s); }
    if (const JsonValue* r = tr.opt("rotation")) rot = vec3cfg(*r);
    Quat q = quat_from_euler_yxz(to_radians(rot.y), to_radians(rot.x), to_radians(rot.z));
    return mat4_from_scale_rotation_translation(scale, q, pos);
}

mi355rt_camera camera_new(V3 position, V3 look_at, V3 world_up, float fov, float aspect) {   // camera.rs:14-31
    const V3 forward = normalized(look_at - position);
    const V3 right = normalized(cross(forward, normalized(world_up)));
    const V3 true_up = normalized(cross(right, forward));
    const float fov_rad = fov * PI_F / 180.0f;
    // f32::tan, correctly rounded (through double) rather than the platform's tanf: glibc's tanf returns the upper neighbour of
    // tan(30 deg) where the correctly rounded value -- and the reference's committed render -- has the lower one
    const float half_height = (float)std::tan((double)(fov_rad / 2.0f));
    const float half_width = half_height * aspect;
    mi355rt_camera c;
    c.position[0] = position.x; c.position[1] = position.y; c.position[2] = position.z;
    c.forward[0] = forward.x; c.forward[1] = forward.y; c.forward[2] = forward.z;
    c.right[0] = right.x; c.right[1] = right.y; c.right[2] = right.z;
    c.true_up[0] = true_up.x; c.true_up[1] = true_up.y; c.true_up[2] = true_up.z;
    c.half_width = half_width; c.half_height = half_height;
    return c;
}

void load_impl(const std::string& json_path, const mi355rt_load_overrides* ov, mi355rt_loaded_scene& out) {
    std::ifstream f(json_path);
    if (!f) throw ParseError{"cannot open " + json_path};
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    JsonValue cfg;
    try { cfg = JsonParser(text).parse(); } catch (const std::exception& e) { throw ParseError{e.what()}; }
    if (!cfg.is_object()) bad("scene root must be a map");
    const size_t slash = json_path.find_last_of('/');
    const std::string scene_dir = slash == std::string::npos ? std::string(".") : json_path.substr(0, slash);

    uint32_t width = 800, height = 600, spp = 16, max_depth = 10;                // parser.rs:255-258
    const JsonValue* cam = cfg.get("camera");
    if (!cam || !cam->is_object()) bad("missing field `camera`");
    if (const JsonValue* res = cam->opt("resolution")) {
        if (res->is_number()) width = height = num_usize(*res, "camera.resolution");                    // ResolutionConfig::Square
        else if (res->is_array() && res->arr.size() == 2) {                                              // ::Explicit([usize; 2]), parser.rs:69-72: serde accepts
            width = num_usize(res->arr[0], "camera.resolution[]");                                        // exactly two elements -- any other length
            height = num_usize(res->arr[1], "camera.resolution[]");                                       // matches no variant and fails the whole load
        } else bad("camera.resolution: data did not match any variant of untagged enum ResolutionConfig");
    }
    if (const JsonValue* r = cfg.opt("renderer")) if (const JsonValue* s = r->opt("spp")) spp = num_usize(*s, "renderer.spp");
    if (const JsonValue* i = cfg.opt("integrator")) if (const JsonValue* m = i->opt("max_bounces")) max_depth = num_usize(*m, "integrator.max_bounces");
    if (ov) {
        if (ov->width) width = ov->width;
        if (ov->height) height = ov->height;
        if (ov->samples_per_pixel) spp = ov->samples_per_pixel;
        if (ov->max_depth) max_depth = ov->max_depth;
    }
    out.settings = {width, height, spp, max_depth};

    const JsonValue* ctr = cam->get("transform");
    if (!ctr || !ctr->get("position") || !ctr->get("look_at") || !ctr->get("up") || !cam->get("fov")) bad("camera.transform/fov incomplete");
    const JsonValue* asp = cam->opt("aspect");
    const float aspect = asp ? num_f32(*asp) : (float)width / (float)height;     // parser.rs:294-297
    out.camera = camera_new(vec3cfg(*ctr->get("position")), vec3cfg(*ctr->get("look_at")), vec3cfg(*ctr->get("up")), num_f32(*cam->get("fov")), aspect);

    std::map<std::string, uint32_t> bsdf_index;
    if (const JsonValue* bl = cfg.opt("bsdfs")) {
        if (!bl->is_array()) bad("bsdfs must be a sequence");
        for (const JsonValue& b : bl->arr) {
            const JsonValue* name = b.get("name");
            if (!name || !name->is_string()) bad("bsdf without name");
            mi355rt_material m;
            if (parse_bsdf(b, m)) { bsdf_index[name->str] = (uint32_t)out.mats.size(); out.mats.push_back(m); }
        }
    }
    auto add_material = [&](const mi355rt_material& m) { out.mats.push_back(m); return (uint32_t)(out.mats.size() - 1); };
    auto material_for = [&](const JsonValue& p) {
        const JsonValue* b = p.get("bsdf");
        if (!b || !b->is_string()) bad("primitive missing field `bsdf`");
        auto it = bsdf_index.find(b->str);
        if (it != bsdf_index.end()) return it->second;
        return add_material(make_mat(MI355RT_MAT_LAMBERT_SOLID, 1.f, 0.f, 1.f));  // Color::MAGENTA fallback (parser.rs:541-543)
    };

    // `sky` (parser.rs:497-521): only a texture ending in ".hdr" feeds trace_ray (renderer.rs:40); an LDR image is
    // loaded into scene.skybox_image, which nothing reads.  A load error keeps the default background.
    if (const JsonValue* sky = cfg.opt("sky")) {
        if (!sky->is_object()) bad("sky must be a map");
        if (const JsonValue* tex = sky->opt("texture")) {
            if (!tex->is_string()) bad("sky.texture must be a string");
            const std::string& rel = tex->str;
            if (rel.size() >= 4 && rel.compare(rel.size() - 4, 4, ".hdr") == 0) {
                if (load_radiance_hdr(scene_dir + "/" + rel, out.sky_w, out.sky_h, out.sky) != MI355RT_OK) { out.sky.clear(); out.sky_w = out.sky_h = 0; }
            }
        }
    }

    const JsonValue* prims = cfg.get("primitives");
    if (!prims || !prims->is_array()) bad("missing field `primitives`");
    for (const JsonValue& p : prims->arr) {
        const JsonValue* tv = p.get("type");
        if (!tv || !tv->is_string()) bad("primitive without `type`");
        const std::string& t = tv->str;
        mi355rt_primitive prim; std::memset(&prim, 0, sizeof prim);
        auto transform = [&]() -> const JsonValue& { const JsonValue* tr = p.get("transform"); if (!tr) bad("primitive missing field `transform`"); return *tr; };
        if (t == "sphere") {                                                     // parser.rs:525-584
            const JsonValue& tr = transform();
            V3 center{0, 0, 0};
            if (const JsonValue* pos = tr.opt("position")) center = vec3cfg(*pos);
            float radius = 1.0f;
            if (const JsonValue* r = p.opt("radius")) radius = num_f32(*r);
            else if (const JsonValue* s = tr.opt("scale")) radius = s->is_number() ? (float)s->num : vec3cfg(*s).x;
            const JsonValue* power = p.opt("power");
            if (!p.get("bsdf") || !p.get("bsdf")->is_string()) bad("sphere missing field `bsdf`");
            uint32_t mat;
            if (power) {
                const float pv = num_f32(*power);
                const float rad = (radius > 1e-6f) ? pv / (4.0f * PI_F * PI_F * radius * radius) : 0.0f;   // parser.rs:567-575
                mat = add_material(make_mat(MI355RT_MAT_EMISSIVE, rad, rad, rad));
            } else mat = material_for(p);
            prim.kind = MI355RT_PRIM_SPHERE; prim.material = mat;
            prim.data[0] = center.x; prim.data[1] = center.y; prim.data[2] = center.z; prim.data[3] = radius;
        } else if (t == "plane") {                                               // parser.rs:585-633
            const JsonValue *pt = p.get("point"), *nn = p.get("normal"), *mc = p.get("material");
            if (!pt || !nn || !mc) bad("plane needs point, normal, material");
            const uint32_t mat = add_material(inline_plane_material(*mc));
            const V3 point = vec3cfg(*pt), n = normalized(vec3cfg(*nn));          // Plane::new, plane.rs:16-22
            prim.kind = MI355RT_PRIM_PLANE; prim.material = mat;
            prim.data[0] = point.x; prim.data[1] = point.y; prim.data[2] = point.z; prim.data[3] = n.x; prim.data[4] = n.y; prim.data[5] = n.z;
        } else if (t == "quad") {                                                // parser.rs:702-767 + quad.rs:26-79
            const JsonValue& tr = transform();
            if (!p.get("bsdf") || !p.get("bsdf")->is_string()) bad("quad missing field `bsdf`");
            uint32_t mat;
            float c[3];
            const JsonValue* em = p.opt("emission");
            if (em && color3(*em, c)) mat = add_material(make_mat(MI355RT_MAT_EMISSIVE, c));
            else if (em && em->is_string()) mat = add_material(make_mat(MI355RT_MAT_EMISSIVE, 5.f, 5.f, 5.f));
            else mat = material_for(p);
            const Mat4 m = object_matrix(tr);
            const V3 base = mat4_mul_point(m, {-0.5f, 0.0f, -0.5f}), pb = mat4_mul_point(m, {0.5f, 0.0f, -0.5f}), pd = mat4_mul_point(m, {-0.5f, 0.0f, 0.5f});
            const V3 e0 = pb - base, e1 = pd - base;
            const V3 n = normalized(cross(e0, e1));
            const float d = dot(n, base), l0 = dot(e0, e0), l1 = dot(e1, e1);
            prim.kind = MI355RT_PRIM_QUAD; prim.material = mat;
            const float vals[15] = {base.x, base.y, base.z, e0.x, e0.y, e0.z, e1.x, e1.y, e1.z, n.x, n.y, n.z, d,
                                    l0 > EPSILON ? 1.0f / l0 : 0.0f, l1 > EPSILON ? 1.0f / l1 : 0.0f};
            std::memcpy(prim.data, vals, sizeof vals);
        } else if (t == "cube") {                                                // parser.rs:768-810 + cube.rs:20-29
            const uint32_t mat = material_for(p);
            const Mat4 m = object_matrix(transform()), inv = mat4_inverse(m);
            prim.kind = MI355RT_PRIM_CUBE; prim.material = mat;
            std::memcpy(prim.data, m.m, 64); std::memcpy(prim.data + 16, inv.m, 64);
        } else if (t == "mesh") {                                                // parser.rs:634-701 + mesh_object.rs:25-57
            const JsonValue* file = p.get("file");
            if (!file || !file->is_string()) bad("mesh missing field `file`");
            const uint32_t mat = material_for(p);
            const Mat4 m = object_matrix(transform()), inv = mat4_inverse(m);
            const std::string path = scene_dir + "/" + file->str;
            std::vector<mi355rt_triangle> tris;
            const bool wo3 = file->str.size() >= 4 && file->str.compare(file->str.size() - 4, 4, ".wo3") == 0;
            const int rc = wo3 ? load_wo3(path, tris, ov && ov->wo3_four_index_stride) : load_obj(path, tris);
            if (rc != MI355RT_OK || tris.empty()) continue;                       // "Error loading ... mesh" -> object dropped (parser.rs:685-698)
            mi355rt_mesh mesh; std::memset(&mesh, 0, sizeof mesh);
            std::vector<mi355rt_bvh_node> nodes; std::vector<uint32_t> idx; uint32_t md = 0;
            if (bvh_build(tris.data(), (uint32_t)tris.size(), nodes, idx, md) != MI355RT_OK) bad("BVH build failed");
            mesh.first_triangle = (uint32_t)out.tris.size(); mesh.triangle_count = (uint32_t)tris.size();
            mesh.first_node = (uint32_t)out.nodes.size(); mesh.node_count = (uint32_t)nodes.size();
            mesh.first_index = (uint32_t)out.indices.size(); mesh.index_count = (uint32_t)idx.size();
            mesh.max_depth = md;
            out.tris.insert(out.tris.end(), tris.begin(), tris.end());
            out.nodes.insert(out.nodes.end(), nodes.begin(), nodes.end());
            out.indices.insert(out.indices.end(), idx.begin(), idx.end());
            out.meshes.push_back(mesh);
            prim.kind = MI355RT_PRIM_MESH; prim.material = mat; prim.mesh = (uint32_t)(out.meshes.size() - 1);
            std::memcpy(prim.data, m.m, 64); std::memcpy(prim.data + 16, inv.m, 64);
        } else {
            if (ov && ov->skip_unknown_primitives) continue;
            bad("unknown variant `" + t + "`, expected one of `sphere`, `plane`, `mesh`, `quad`, `cube`");   // serde, parser.rs:135-165
        }
        out.prims.push_back(prim);
    }

    mi355rt_scene& s = out.scene;
    s.primitives = out.prims.data(); s.n_primitives = (uint32_t)out.prims.size();
    s.materials = out.mats.data(); s.n_materials = (uint32_t)out.mats.size();
    s.meshes = out.meshes.data(); s.n_meshes = (uint32_t)out.meshes.size();
    s.triangles = out.tris.data(); s.n_triangles = (uint32_t)out.tris.size();
    s.nodes = out.nodes.data(); s.n_nodes = (uint32_t)out.nodes.size();
    s.tri_indices = out.indices.data(); s.n_tri_indices = (uint32_t)out.indices.size();
    s.miss_color[0] = s.miss_color[1] = s.miss_color[2] = 0.5f;                  // Color::GRAY, renderer.rs:61
    s.sky_width = out.sky_w; s.sky_height = out.sky_h; s.sky_rgb = out.sky.empty() ? nullptr : out.sky.data();   // renderer.rs:40-54
}

}  // namespace

extern "C" {

int mi355rt_scene_load_json(const char* json_path, const mi355rt_load_overrides* overrides, mi355rt_loaded_scene** out_scene) {
    // The barrier of host_common.hpp around everything: bad_alloc -> MI355RT_ERR_OOM without allocating a message, anything else -> _IO.  The
    // handlers inside build their messages with std::string, which may throw again under the same shortage -- that, too, ends in the barrier.
    return mi355rt_host::guard("scene_load_json", MI355RT_ERR_IO, [&]() -> int {
    if (!json_path || !out_scene) return set_error(MI355RT_ERR_INVALID, "scene_load_json: null argument");
    std::unique_ptr<mi355rt_loaded_scene> s(new (std::nothrow) mi355rt_loaded_scene());      // (freed on every path out, exceptions included)
    if (!s) return set_error(MI355RT_ERR_OOM, "host allocation failed");
    try {
        load_impl(json_path, overrides, *s);
    } catch (const ParseError& e) {
        return set_error(MI355RT_ERR_IO, std::string("Failed to load scene '") + json_path + "': " + e.msg);
    } catch (const std::bad_alloc&) {
        throw;                                                                               // -> MI355RT_ERR_OOM in the barrier
    } catch (const std::length_error&) {
        throw;
    } catch (const std::exception& e) {
        return set_error(MI355RT_ERR_IO, std::string("Failed to load scene '") + json_path + "': " + e.what());
    }
    *out_scene = s.release();
    return MI355RT_OK;
    });
}
void mi355rt_scene_free(mi355rt_loaded_scene* s) { delete s; }
const mi355rt_scene* mi355rt_loaded_scene_get(const mi355rt_loaded_scene* s) { return s ? &s->scene : nullptr; }
const mi355rt_camera* mi355rt_loaded_scene_camera(const mi355rt_loaded_scene* s) { return s ? &s->camera : nullptr; }
const mi355rt_settings* mi355rt_loaded_scene_settings(const mi355rt_loaded_scene* s) { return s ? &s->settings : nullptr; }

}  // extern "C"
