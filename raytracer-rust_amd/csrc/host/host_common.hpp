// host_common.hpp -- shared bits of libmi355rt_host.so (pure CPU; no HIP, no oracle).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../../include/mi355rt.h"

namespace mi355rt_host {

int set_error(int code, const std::string& msg);      // records the thread-local message, returns code

// bvh_build.cpp
int bvh_build(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
              uint32_t& max_depth);
int bvh_build_threads(const mi355rt_triangle* tris, uint32_t n, std::vector<mi355rt_bvh_node>& nodes, std::vector<uint32_t>& indices,
                      uint32_t& max_depth, int n_threads);   // 0 = all hardware threads (<= 16), 1 = serial; identical arrays either way

// hdr_io.cpp
int load_radiance_hdr(const std::string& path, uint32_t& width, uint32_t& height, std::vector<float>& rgb);

}  // namespace mi355rt_host
