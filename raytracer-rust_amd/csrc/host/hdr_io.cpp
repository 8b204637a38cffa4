// hdr_io.cpp -- Radiance .hdr (RGBE) reader for `sky.texture` (src/tungsten/parser.rs:497-509:
// image::open(path).into_rgb32f()).  The `image` crate (Cargo.lock pin, not under /root/reference) is
// restated for what that call yields: header lines up to a blank line, "-Y H +X W", scanlines either flat
// RGBE or new-style run-length encoded (marker 2 2 hi lo, four channel planes), and the RGBE -> f32
// conversion  c * 2^(e - 136)  (zero when e == 0).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "host_common.hpp"

namespace mi355rt_host {

int load_radiance_hdr(const std::string& path, uint32_t& width, uint32_t& height, std::vector<float>& rgb) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return set_error(MI355RT_ERR_IO, "cannot open HDR " + path);
    std::vector<unsigned char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto line = [&](std::string& out) { out.clear(); while (p < b.size() && b[p] != '\n') out.push_back((char)b[p++]); if (p < b.size()) ++p; return true; };
    std::string ln;
    line(ln);
    if (ln.rfind("#?", 0) != 0) return set_error(MI355RT_ERR_IO, "not a Radiance HDR file: " + path);
    while (p < b.size()) { line(ln); if (ln.empty()) break; }
    line(ln);
    int H = 0, W = 0;
    if (std::sscanf(ln.c_str(), "-Y %d +X %d", &H, &W) != 2 || H <= 0 || W <= 0) return set_error(MI355RT_ERR_IO, "unsupported HDR orientation: " + ln);
    // Refuse before allocating: the device caps a skybox at 2^28 texels, and a file cannot hold more scanlines than its
    // bytes allow (flat: 4 W bytes per line; run-length: at least the 4-byte marker + 2 bytes per 127-pixel run and channel).
    if ((uint64_t)W * (uint64_t)H > (1ull << 28)) return set_error(MI355RT_ERR_IO, "HDR larger than 2^28 pixels: " + ln);
    const uint64_t min_line = std::min<uint64_t>((uint64_t)W * 4u, 4u + 8u * (((uint64_t)W + 126u) / 127u));
    if ((uint64_t)H * min_line > b.size() - std::min(p, b.size())) return set_error(MI355RT_ERR_IO, "HDR truncated");
    width = (uint32_t)W; height = (uint32_t)H;
    rgb.assign((size_t)W * H * 3, 0.0f);
    std::vector<unsigned char> scan((size_t)W * 4);
    for (int y = 0; y < H; ++y) {
        bool rle = false;
        if (W >= 8 && W <= 32767 && p + 4 <= b.size() && b[p] == 2 && b[p + 1] == 2 && (((int)b[p + 2] << 8) | b[p + 3]) == W) rle = true;
        if (rle) {
            p += 4;
            for (int c = 0; c < 4; ++c) {
                int x = 0;
                while (x < W) {
                    if (p >= b.size()) return set_error(MI355RT_ERR_IO, "HDR truncated");
                    int n = b[p++];
                    if (n > 128) {
                        n -= 128;
                        if (p >= b.size() || x + n > W) return set_error(MI355RT_ERR_IO, "HDR bad run");
                        const unsigned char v = b[p++];
                        for (int k = 0; k < n; ++k) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (n == 0 || p + (size_t)n > b.size() || x + n > W) return set_error(MI355RT_ERR_IO, "HDR bad literal");
                        for (int k = 0; k < n; ++k) scan[(size_t)(x++) * 4 + c] = b[p++];
                    }
                }
            }
        } else {
            if (p + (size_t)W * 4 > b.size()) return set_error(MI355RT_ERR_IO, "HDR truncated");
            std::memcpy(scan.data(), b.data() + p, (size_t)W * 4); p += (size_t)W * 4;
        }
        for (int x = 0; x < W; ++x) {
            const unsigned char* q = &scan[(size_t)x * 4];
            float* o = &rgb[((size_t)y * W + x) * 3];
            if (q[3] == 0) { o[0] = o[1] = o[2] = 0.0f; }
            else { const float e = std::exp2((float)q[3] - (128.0f + 8.0f)); o[0] = e * (float)q[0]; o[1] = e * (float)q[1]; o[2] = e * (float)q[2]; }
        }
    }
    return MI355RT_OK;
}

}  // namespace mi355rt_host
