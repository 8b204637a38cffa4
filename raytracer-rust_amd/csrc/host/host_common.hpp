// host_common.hpp -- shared bits of libmi355rt_host.so (pure CPU; no HIP, no oracle).
#pragma once
#include <stdint.h>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"

namespace mi355rt_host {

int set_error(int code, const std::string& msg);      // records the thread-local message, returns code
int set_error_noexcept(int code, const char* msg) noexcept;   // the same from a catch handler: never throws (drops the text if even that allocation fails)

// The exception barrier of every extern "C" entry point (mi355rt.h: "nothing aborts, nothing throws across the ABI"; the caller may be
// a Rust frame, src/renderer.rs:67, into which a C++ exception must not unwind): std::bad_alloc / std::length_error -> MI355RT_ERR_OOM,
// anything else -> `other` with what() in mi355rt_host_last_error().
template <class F> int guard(const char* where, int other, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return set_error_noexcept(MI355RT_ERR_OOM, where); }
    catch (const std::length_error&) { return set_error_noexcept(MI355RT_ERR_OOM, where); }
    catch (const std::exception& e) {
        try { return set_error(other, std::string(where) + ": " + e.what()); } catch (...) { return set_error_noexcept(other, where); }
    }
    catch (...) { return set_error_noexcept(other, where); }
}

// bvh_build.cpp
int bvh_build(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
              uint32_t& max_depth);
int bvh_build_threads(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
                      uint32_t& max_depth, int n_threads);   // 0 = all hardware threads (<= 16), 1 = serial; identical arrays either way

// hdr_io.cpp
int load_radiance_hdr(const std::string& path, uint32_t& width, uint32_t& height, std::vector<float>& rgb);

}  // namespace mi355rt_host
