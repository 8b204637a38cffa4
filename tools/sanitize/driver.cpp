// Sanitizer driver for the CPU-side producers (libmi355rt_host: scene loader, JSON, OBJ / WO3 / HDR readers, BVH build,
// PNG / PFM writers) and for the CPU oracle's entry points.  Built with -fsanitize=address,undefined by
// tools/sanitize_host.py; every case must return (OK or an error code) without a sanitizer report.
//   usage: driver <repo root> <scratch dir>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

extern "C" int oracle_render(const mi355rt_scene*, const mi355rt_camera*, const mi355rt_settings*, const mi355rt_options*, int, int, uint32_t*, float*, void*);

static int g_fail = 0;
static void expect(bool ok, const std::string& what) { if (!ok) { std::printf("UNEXPECTED: %s\n", what.c_str()); ++g_fail; } }
static void put(const std::string& path, const std::string& bytes) { std::ofstream f(path, std::ios::binary); f.write(bytes.data(), (std::streamsize)bytes.size()); }
static std::string u64(uint64_t v) { return std::string(reinterpret_cast<const char*>(&v), 8); }
static std::string u32(uint32_t v) { return std::string(reinterpret_cast<const char*>(&v), 4); }
static std::string f32s(float v) { return std::string(reinterpret_cast<const char*>(&v), 4); }

static uint32_t g_last_prims = 0, g_last_sky = 0;
static int load(const std::string& path, bool want_ok, uint32_t w = 16, uint32_t h = 12, bool skip_unknown = false) {
    g_last_prims = g_last_sky = 0;
    mi355rt_load_overrides ov{}; ov.width = w; ov.height = h; ov.samples_per_pixel = 1; ov.max_depth = 3; ov.skip_unknown_primitives = skip_unknown;
    mi355rt_loaded_scene* s = nullptr;
    const int rc = mi355rt_scene_load_json(path.c_str(), &ov, &s);
    if (want_ok) expect(rc == MI355RT_OK, path + " should load: " + mi355rt_host_last_error());
    else expect(rc != MI355RT_OK, path + " should be refused");
    if (rc == MI355RT_OK) {
        g_last_prims = mi355rt_loaded_scene_get(s)->n_primitives; g_last_sky = mi355rt_loaded_scene_get(s)->sky_width;
        // push the loaded scene through the oracle too (tiny render): exercises its scene import and BVH build under ASan
        std::vector<uint32_t> px((size_t)w * h);
        mi355rt_options o{}; o.abi_version = MI355RT_ABI_VERSION; o.rng_mode = MI355RT_RNG_CTR;
        const int orc = oracle_render(mi355rt_loaded_scene_get(s), mi355rt_loaded_scene_camera(s), mi355rt_loaded_scene_settings(s), &o, 2, -1, px.data(), nullptr, nullptr);
        expect(orc == 0, path + ": oracle_render");
        mi355rt_scene_free(s);
    }
    return rc;
}

static std::string scene_with(const std::string& prim) {
    return std::string("{\"bsdfs\":[{\"name\":\"m\",\"type\":\"lambert\",\"albedo\":[0.5,0.5,0.5]}],\"primitives\":[") + prim +
           "],\"camera\":{\"transform\":{\"position\":[0,0,5],\"look_at\":[0,0,0],\"up\":[0,1,0]},\"fov\":40,\"resolution\":[8,8]},\"renderer\":{\"spp\":1},\"integrator\":{\"max_bounces\":2}}";
}
static std::string mesh_prim(const std::string& file) { return "{\"type\":\"mesh\",\"file\":\"" + file + "\",\"bsdf\":\"m\",\"transform\":{}}"; }

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: driver <repo root> <scratch dir>\n"); return 2; }
    const std::string root = argv[1], tmp = argv[2];

    // ---- the shipped scenes ----
    load(root + "/data/scenes/tungsten/cornell-box/scene.json", true);
    load(root + "/data/scenes/tungsten/veach-mis/scene.json", true);
    load(root + "/data/scenes/semesterbild.json", true);
    load(root + "/data/scenes/tungsten/teapot/scene.json", true, 16, 12, true);
    load(root + "/data/scenes/tungsten/teapot/scene.json", false);            // infinite_sphere: an unknown primitive type is a load error
    load(root + "/no/such/file.json", false);

    // ---- malformed JSON ----
    const char* bad_json[] = {"", "{", "[", "{\"a\":}", "{\"a\":1,}", "nul", "{\"a\":1} x", "\"\\u12", "\"abc", "{\"a\":01}", "{\"a\":1.}", "{\"a\":.5}",
                              "{\"a\":1e}", "{\"a\":0x10}", "{\"a\":-}", "{\"a\":-inf}", "{\"a\":nan}", "{\"a\":\"\\q\"}", "{1:2}", "[1 2]"};
    for (size_t i = 0; i < sizeof bad_json / sizeof bad_json[0]; ++i) { const std::string p = tmp + "/bad" + std::to_string(i) + ".json"; put(p, bad_json[i]); load(p, false); }
    { std::string deep(200000, '['); put(tmp + "/deep.json", deep); load(tmp + "/deep.json", false); }
    { std::string deep; for (int i = 0; i < 100000; ++i) deep += "{\"a\":"; put(tmp + "/deep2.json", deep); load(tmp + "/deep2.json", false); }
    // well-formed JSON, wrong content
    put(tmp + "/empty_obj.json", "{}"); load(tmp + "/empty_obj.json", false);
    put(tmp + "/arr.json", "[1,2,3]"); load(tmp + "/arr.json", false);
    const char* bad_res[] = {"-1", "1.5", "1e3", "\"800\"", "[800,\"a\"]", "4294967296", "[1.5,2]", "{\"w\":1}"};
    for (size_t i = 0; i < sizeof bad_res / sizeof bad_res[0]; ++i) {
        std::string s = scene_with("");
        const std::string from = "\"resolution\":[8,8]";
        s.replace(s.find(from), from.size(), std::string("\"resolution\":") + bad_res[i]);
        const std::string p = tmp + "/res" + std::to_string(i) + ".json"; put(p, s);
        mi355rt_loaded_scene* ls = nullptr;
        expect(mi355rt_scene_load_json(p.c_str(), nullptr, &ls) != MI355RT_OK, std::string("resolution ") + bad_res[i] + " should be refused");
        if (ls) mi355rt_scene_free(ls);
    }
    { std::string s = scene_with(""); const std::string from = "\"spp\":1"; s.replace(s.find(from), from.size(), "\"spp\":2.5"); put(tmp + "/spp.json", s);
      mi355rt_loaded_scene* ls = nullptr; expect(mi355rt_scene_load_json((tmp + "/spp.json").c_str(), nullptr, &ls) != MI355RT_OK, "spp 2.5 should be refused"); if (ls) mi355rt_scene_free(ls); }
    put(tmp + "/ok_empty.json", scene_with("")); load(tmp + "/ok_empty.json", true);
    put(tmp + "/res3.json", [&] { std::string s = scene_with(""); const std::string from = "\"resolution\":[8,8]"; s.replace(s.find(from), from.size(), "\"resolution\":[1,2,3]"); return s; }());
    load(tmp + "/res3.json", false);                                           // Explicit([usize; 2]), parser.rs:69-72: three elements match no variant of the untagged enum

    // ---- OBJ ----
    struct Case { const char* name; std::string body; bool ok; };
    const std::string tri = "v 0 0 0\nv 1 0 0\nv 0 1 0\n";
    const Case objs[] = {
        {"good", tri + "f 1 2 3\n", true}, {"neg", tri + "f -3 -2 -1\n", true}, {"slashes", tri + "vt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3//1\n", true},
        {"quad", tri + "v 1 1 0\nf 1 2 4 3\n", true}, {"crlf", "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n", true},
        {"novert", "f 1 2 3\n", false}, {"zero", tri + "f 0 1 2\n", false}, {"oob", tri + "f 1 2 4\n", false}, {"negoob", tri + "f -4 -2 -1\n", false},
        {"huge", tri + "f 1 2 99999999999999999999\n", false}, {"word", tri + "f 1 x 3\n", false}, {"shortv", "v 1 2\nf 1 1 1\n", false},
        {"line", tri + "f 1 2\n", false}, {"nofaces", tri, false}, {"empty", "", false}, {"degenerate_only", tri + "f 1 1 1\n", false},
    };
    // A mesh that fails to load is reported and DROPPED, the scene itself loads (parser.rs:685-698): `ok` = the mesh survives.
    for (const Case& c : objs) {
        put(tmp + "/" + c.name + ".obj", c.body);
        put(tmp + "/obj_" + c.name + ".json", scene_with(mesh_prim(std::string(c.name) + ".obj")));
        load(tmp + "/obj_" + std::string(c.name) + ".json", true);
        expect(g_last_prims == (c.ok ? 1u : 0u), std::string("OBJ case ") + c.name + (c.ok ? " should yield a mesh" : " should drop the mesh"));
    }
    put(tmp + "/obj_missing.json", scene_with(mesh_prim("does_not_exist.obj"))); load(tmp + "/obj_missing.json", true);
    expect(g_last_prims == 0u, "missing OBJ should drop the mesh");

    // ---- WO3 ----
    auto vert = [&](float x, float y, float z) { return f32s(x) + f32s(y) + f32s(z) + f32s(0) + f32s(0) + f32s(1) + f32s(0) + f32s(0); };
    const std::string v3 = vert(0, 0, 0) + vert(1, 0, 0) + vert(0, 1, 0);
    const Case wo3s[] = {
        {"good", u64(3) + v3 + u64(1) + u32(0) + u32(1) + u32(2) + u32(0), true},
        {"empty", "", false}, {"short", "abcd", false}, {"hugenv", u64(1ull << 60), false}, {"truncv", u64(3) + vert(0, 0, 0), false},
        {"nohdr2", u64(3) + v3, false}, {"hugent", u64(3) + v3 + u64(1ull << 61), false}, {"trunci", u64(3) + v3 + u64(2) + u32(0) + u32(1) + u32(2), false},
        {"oobidx", u64(3) + v3 + u64(1) + u32(0) + u32(1) + u32(7) + u32(0), false},        // every triangle skipped -> empty mesh -> error
        {"nvwrap", u64(0x0800000000000001ull) + v3, false},                                     // nv * 32 wraps around 2^64
    };
    for (const Case& c : wo3s) {
        put(tmp + "/" + c.name + ".wo3", c.body);
        put(tmp + "/wo3_" + c.name + ".json", scene_with(mesh_prim(std::string(c.name) + ".wo3")));
        load(tmp + "/wo3_" + std::string(c.name) + ".json", true);
        expect(g_last_prims == (c.ok ? 1u : 0u), std::string("WO3 case ") + c.name + (c.ok ? " should yield a mesh" : " should drop the mesh"));
    }

    // ---- Radiance HDR (sky.texture): a load error keeps the default background, so every scene loads; the reader must not crash ----
    auto sky_scene = [&](const std::string& file) { std::string s = scene_with(""); s.insert(1, "\"sky\":{\"texture\":\"" + file + "\"},"); return s; };
    const std::string hdr_head = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n";
    const std::string px4 = std::string("\x80\x40\x20\x81", 4);
    const Case hdrs[] = {
        {"flat", hdr_head + "-Y 2 +X 2\n" + px4 + px4 + px4 + px4, true}, {"notrad", "P6\n1 1\n255\n", false}, {"nores", hdr_head, false},
        {"negdim", hdr_head + "-Y -2 +X 2\n", false}, {"hugedim", hdr_head + "-Y 2147483647 +X 2147483647\n", false}, {"bigdim", hdr_head + "-Y 16000 +X 16000\n", false},
        {"truncflat", hdr_head + "-Y 2 +X 2\n" + px4, false},
        {"rle", hdr_head + "-Y 1 +X 8\n" + std::string("\x02\x02\x00\x08", 4) + std::string("\x88\x10", 2) + std::string("\x88\x20", 2) + std::string("\x88\x30", 2) + std::string("\x88\x80", 2), true},
        {"rlelong", hdr_head + "-Y 1 +X 8\n" + std::string("\x02\x02\x00\x08", 4) + std::string("\xff\x10", 2) + std::string(40, '\x01'), false},  // run of 127 > width
        {"rlezero", hdr_head + "-Y 1 +X 8\n" + std::string("\x02\x02\x00\x08", 4) + std::string(40, '\x00'), false},                              // literal of length 0
        {"rletrunc", hdr_head + "-Y 1 +X 8\n" + std::string("\x02\x02\x00\x08", 4) + std::string("\x05\x01\x02", 3), false},                     // literal longer than the file
        {"orient", hdr_head + "+Y 2 +X 2\n" + px4 + px4 + px4 + px4, false},
    };
    for (const Case& c : hdrs) {
        put(tmp + "/" + c.name + ".hdr", c.body);
        put(tmp + "/hdr_" + c.name + ".json", sky_scene(std::string(c.name) + ".hdr"));
        load(tmp + "/hdr_" + std::string(c.name) + ".json", true);                    // `ok` = the skybox survives; the scene always loads
        expect((g_last_sky != 0u) == c.ok, std::string("HDR case ") + c.name + (c.ok ? " should yield a skybox" : " should fall back to the default background"));
    }

    // ---- BVH build: ties, NaNs, one triangle, many equal centroids ----
    {
        std::vector<mi355rt_triangle> t;
        for (int i = 0; i < 300; ++i) {
            mi355rt_triangle x{}; const float o = (float)(i % 7);
            x.v0[0] = o; x.v1[0] = o + 1; x.v2[1] = 1; x.normal[2] = 1;
            if (i % 41 == 0) x.v0[1] = std::nanf("");
            t.push_back(x);
        }
        for (uint32_t n : {1u, 4u, 5u, 300u}) {
            uint32_t nn = 0, ni = 0, md = 0;
            expect(mi355rt_bvh_build(t.data(), n, nullptr, &nn, nullptr, &ni, &md) == MI355RT_OK, "bvh_build count");
            std::vector<mi355rt_bvh_node> nodes(nn); std::vector<uint32_t> idx(ni);
            expect(mi355rt_bvh_build(t.data(), n, nodes.data(), &nn, idx.data(), &ni, &md) == MI355RT_OK, "bvh_build fill");
            uint32_t small_n = nn ? nn - 1 : 0;
            expect(nn == 0 || mi355rt_bvh_build(t.data(), n, nodes.data(), &small_n, idx.data(), &ni, &md) != MI355RT_OK, "bvh_build must refuse short arrays");
        }
        uint32_t nn = 0, ni = 0;
        expect(mi355rt_bvh_build(nullptr, 3, nullptr, &nn, nullptr, &ni, nullptr) != MI355RT_OK, "bvh_build null");
        expect(mi355rt_bvh_build(t.data(), 0, nullptr, &nn, nullptr, &ni, nullptr) != MI355RT_OK, "bvh_build empty");
    }

    // ---- writers ----
    {
        std::vector<uint32_t> px(7 * 5, 0x00FF8040u); std::vector<float> lin(7 * 5 * 3, 0.25f);
        expect(mi355rt_write_png((tmp + "/a.png").c_str(), px.data(), 7, 5) == MI355RT_OK, "write_png");
        expect(mi355rt_write_pfm((tmp + "/a.pfm").c_str(), lin.data(), 7, 5) == MI355RT_OK, "write_pfm");
        expect(mi355rt_write_exr((tmp + "/a.exr").c_str(), lin.data(), 7, 5) == MI355RT_OK, "write_exr");
        expect(mi355rt_write_exr((tmp + "/a.exr").c_str(), lin.data(), 0, 5) != MI355RT_OK, "write_exr of an empty image");
        expect(mi355rt_write_png((tmp + "/no/dir/a.png").c_str(), px.data(), 7, 5) != MI355RT_OK, "write_png to a missing directory");
        expect(mi355rt_write_png((tmp + "/z.png").c_str(), px.data(), 0, 5) != MI355RT_OK, "write_png of an empty image");
        expect(mi355rt_write_png(nullptr, px.data(), 7, 5) != MI355RT_OK, "write_png null path");
    }
    std::printf(g_fail ? "sanitize driver: %d unexpected result(s)\n" : "sanitize driver: all cases behaved (%d unexpected)\n", g_fail);
    return g_fail ? 1 : 0;
}
