// host_api.cpp -- error channel and ABI self-description of libmi355rt_host.so.
#include "host_common.hpp"

namespace mi355rt_host {
namespace { thread_local std::string g_err; }
int set_error(int code, const std::string& msg) { g_err = msg; return code; }
int set_error_noexcept(int code, const char* msg) noexcept {
    try { g_err.assign(msg); g_err += code == MI355RT_ERR_OOM ? ": host allocation failed" : ": unexpected C++ exception"; } catch (...) { g_err.clear(); }
    return code;
}
const char* last_error() { return g_err.c_str(); }
}  // namespace mi355rt_host

extern "C" {
const char* mi355rt_host_last_error(void) { return mi355rt_host::last_error(); }

// sizeof() of every ABI struct as the C++ compiler sees it, for the ctypes mirror check
// (order: camera, settings, material, primitive, triangle, bvh_node, mesh, scene, options, stats, load_overrides).
uint32_t mi355rt_host_struct_sizes(uint32_t* out, uint32_t capacity) {
    const uint32_t v[] = {(uint32_t)sizeof(mi355rt_camera), (uint32_t)sizeof(mi355rt_settings), (uint32_t)sizeof(mi355rt_material),
                          (uint32_t)sizeof(mi355rt_primitive), (uint32_t)sizeof(mi355rt_triangle), (uint32_t)sizeof(mi355rt_bvh_node),
                          (uint32_t)sizeof(mi355rt_mesh), (uint32_t)sizeof(mi355rt_scene), (uint32_t)sizeof(mi355rt_options),
                          (uint32_t)sizeof(mi355rt_stats), (uint32_t)sizeof(mi355rt_load_overrides)};
    const uint32_t n = (uint32_t)(sizeof v / sizeof v[0]);
    for (uint32_t i = 0; i < n && i < capacity; ++i) out[i] = v[i];
    return n;
}
}
