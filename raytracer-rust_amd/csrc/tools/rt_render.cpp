// rt_render.cpp -- CLI stand-in for the reference's `main` (src/main.rs:22-89): load a Tungsten JSON
// scene, call the MI355X render loop through the C ABI exactly where main.rs:57 calls render_scene,
// save the PNG (main.rs:58).  The minifb preview window (main.rs:60-75) becomes --chunk: a PNG that refines.
//
//   rt_render <scene.json> [-o out.png] [--width W] [--height H] [--spp N] [--max-depth D]
//             [--rng ctr|ref] [--seed S] [--skip-unknown] [--chunk N] [--pfm out.pfm] [--exr out.exr] [--gpus N | --devices a,b,..]
// --fix-aabb / --fix-wo3 switch on the two opt-in fixes (MI355RT_FLAG_FIXED_AABB, wo3_four_index_stride): not the reference's image.
// --gpus N deals row strips over HIP devices 0..N-1 from this one process (mi355rt_render_multi); --devices a,b,... names them
// (a device may repeat: the strip plan of N GPUs on a one-GPU machine).
// --chunk N renders N samples per pixel at a time and rewrites the PNG after every chunk (a preview that refines).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <scene.json> [-o out.png] [--width W] [--height H] [--spp N] [--max-depth D] [--rng ctr|ref] [--seed S] [--skip-unknown] [--chunk N] [--pfm out.pfm] [--exr out.exr] [--gpus N | --devices a,b,..] [--fix-aabb] [--fix-wo3]\n", argv[0]); return 2; }
    std::string scene_path = argv[1], out_path = "render_pt.png", pfm_path, exr_path;
    mi355rt_load_overrides ov{}; mi355rt_options opt{}; uint32_t chunk = 0, gpus = 1; std::vector<int> device_list;
    opt.abi_version = MI355RT_ABI_VERSION; opt.rng_mode = MI355RT_RNG_CTR; opt.strip_rows = 1; opt.n_parts = 1;
    for (int i = 2; i < argc; ++i) {
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "-o")) out_path = next();
        else if (!std::strcmp(argv[i], "--width")) ov.width = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) ov.height = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--spp")) ov.samples_per_pixel = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--max-depth")) ov.max_depth = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--rng")) opt.rng_mode = std::strcmp(next(), "ref") ? MI355RT_RNG_CTR : MI355RT_RNG_REF;
        else if (!std::strcmp(argv[i], "--seed")) opt.seed = std::strtoull(next(), nullptr, 0);
        else if (!std::strcmp(argv[i], "--skip-unknown")) ov.skip_unknown_primitives = 1;
        else if (!std::strcmp(argv[i], "--chunk")) chunk = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--pfm")) pfm_path = next();
        else if (!std::strcmp(argv[i], "--exr")) exr_path = next();
        else if (!std::strcmp(argv[i], "--gpus")) gpus = (uint32_t)std::atoi(next());
        else if (!std::strcmp(argv[i], "--devices")) {
            for (const char* q = next(); *q;) { char* e = nullptr; const long d = std::strtol(q, &e, 10); if (e == q) { std::fprintf(stderr, "--devices wants a comma-separated list of device numbers\n"); return 2; }
                                                device_list.push_back((int)d); q = (*e == ',') ? e + 1 : e; if (*e && *e != ',') { std::fprintf(stderr, "--devices wants a comma-separated list of device numbers\n"); return 2; } }
            gpus = (uint32_t)device_list.size();
        }
        else if (!std::strcmp(argv[i], "--fix-aabb")) opt.flags |= MI355RT_FLAG_FIXED_AABB;
        else if (!std::strcmp(argv[i], "--fix-wo3")) ov.wo3_four_index_stride = 1;
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (gpus == 0) { std::fprintf(stderr, "--gpus / --devices: at least one device\n"); return 2; }
    const bool multi = gpus > 1 || !device_list.empty();
    if (multi && chunk) { std::fprintf(stderr, "--chunk (progressive preview) renders on one GPU: it cannot be combined with --gpus %u\n", gpus); return 2; }
    std::printf("Attempting to load scene from: %s\n", scene_path.c_str());
    auto t0 = std::chrono::steady_clock::now();
    mi355rt_loaded_scene* ls = nullptr;
    if (mi355rt_scene_load_json(scene_path.c_str(), &ov, &ls) != MI355RT_OK) { std::fprintf(stderr, "%s\n", mi355rt_host_last_error()); return 1; }
    const mi355rt_scene* sc = mi355rt_loaded_scene_get(ls);
    const mi355rt_settings* st = mi355rt_loaded_scene_settings(ls);
    std::printf("Scene loaded. Objects: %u. Image: %ux%u, Samples: %u, Max Depth: %u\n", sc->n_primitives, st->width, st->height,
                st->samples_per_pixel, st->max_depth);
    std::vector<uint32_t> buffer((size_t)st->width * st->height);
    std::vector<float> linear(pfm_path.empty() && exr_path.empty() ? 0 : (size_t)st->width * st->height * 3);
    float* lin = linear.empty() ? nullptr : linear.data();
    mi355rt_stats stats{};
    std::printf("Rendering frame (%ux%u) with %u AA samples...\n", st->width, st->height, st->samples_per_pixel);
    struct Preview { const char* path; uint32_t w, h; } pv{out_path.c_str(), st->width, st->height};
    auto on_chunk = [](void* user, uint32_t done, uint32_t total, const uint32_t* packed) -> int {
        const Preview* p = static_cast<const Preview*>(user);
        std::printf("  %u / %u samples per pixel\n", done, total);
        if (done < total) (void)mi355rt_write_png(p->path, packed, p->w, p->h);       // the final image is written below
        return 0;
    };
    std::vector<int> devices = device_list;
    if (devices.empty()) for (uint32_t d = 0; d < gpus; ++d) devices.push_back((int)d);
    if (multi) { opt.strip_rows = 4; opt.n_parts = 0; }
    const auto t_render = std::chrono::steady_clock::now();
    int rc = multi ? mi355rt_render_multi(sc, mi355rt_loaded_scene_camera(ls), st, &opt, devices.data(), gpus, buffer.data(), lin, &stats)
           : chunk ? mi355rt_render_progressive(sc, mi355rt_loaded_scene_camera(ls), st, &opt, chunk, on_chunk, &pv, buffer.data(), lin, &stats)
                   : mi355rt_render(sc, mi355rt_loaded_scene_camera(ls), st, &opt, buffer.data(), lin, &stats);   // <- src/main.rs:57
    if (rc != MI355RT_OK) { std::fprintf(stderr, "render failed (%d): %s\n", rc, mi355rt_last_error()); mi355rt_scene_free(ls); return 1; }
    // The call's wall time includes context creation, scene upload, workspace allocation and the copy back; the kernel
    // figures are device time (with --gpus: the maximum over the devices, which run side by side).
    const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_render).count();
    std::printf("Rendered in %.3f seconds (%.1f Msamples/s end to end; kernels %.3f ms path tracing + %.3f ms resolve%s = %.1f Msamples/s; %.2f rays/sample)\n",
                wall_s, wall_s > 0 ? (double)stats.samples / wall_s / 1e6 : 0.0, stats.render_kernel_ms, stats.resolve_kernel_ms,
                multi ? " (max over devices)" : "",
                stats.total_ms > 0 ? (double)stats.samples / stats.total_ms / 1e3 : 0.0, stats.samples ? (double)stats.rays / (double)stats.samples : 0.0);
    if (mi355rt_write_png(out_path.c_str(), buffer.data(), st->width, st->height) != MI355RT_OK) { std::fprintf(stderr, "%s\n", mi355rt_host_last_error()); mi355rt_scene_free(ls); return 1; }
    std::printf("Image saved as '%s'\n", out_path.c_str());
    if (lin && !exr_path.empty() && mi355rt_write_exr(exr_path.c_str(), lin, st->width, st->height) != MI355RT_OK) { std::fprintf(stderr, "%s\n", mi355rt_host_last_error()); mi355rt_scene_free(ls); return 1; }
    if (lin && !pfm_path.empty() && mi355rt_write_pfm(pfm_path.c_str(), lin, st->width, st->height) != MI355RT_OK) { std::fprintf(stderr, "%s\n", mi355rt_host_last_error()); mi355rt_scene_free(ls); return 1; }
    mi355rt_scene_free(ls);
    std::printf("Total %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
