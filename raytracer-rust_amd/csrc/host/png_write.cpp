// png_write.cpp -- mi355rt_write_png: save_image's pixel conversion (src/renderer.rs:131-138:
// r = (c >> 16) & 0xFF, g = (c >> 8) & 0xFF, b = c & 0xFF) into an 8-bit RGB PNG.  The `image` crate's
// encoder is replaced by a minimal zlib-based writer (one IDAT, filter 0); decoded pixels are identical.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_common.hpp"

namespace {
void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<unsigned char>& out, const char type[4], const std::vector<unsigned char>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}
}  // namespace

extern "C" int mi355rt_write_png(const char* path, const uint32_t* packed, uint32_t width, uint32_t height) {
    using mi355rt_host::set_error;
    if (!path || !packed || width == 0 || height == 0) return set_error(MI355RT_ERR_INVALID, "write_png: bad argument");
    std::vector<unsigned char> raw((size_t)height * (1 + (size_t)width * 3));
    size_t o = 0;
    for (uint32_t y = 0; y < height; ++y) {
        raw[o++] = 0;
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t c = packed[(size_t)y * width + x];
            raw[o++] = (unsigned char)((c >> 16) & 0xFF); raw[o++] = (unsigned char)((c >> 8) & 0xFF); raw[o++] = (unsigned char)(c & 0xFF);
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return set_error(MI355RT_ERR_IO, "write_png: zlib failed");
    z.resize(zlen);
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return set_error(MI355RT_ERR_IO, std::string("write_png: cannot open ") + path);
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? MI355RT_OK : set_error(MI355RT_ERR_IO, "write_png: short write");
}

// Linear f32 dump for parity tooling: Portable FloatMap, "PF", little-endian, rows BOTTOM-UP as
// the format prescribes; `linear_rgb` is the pre-gamma mean image (row 0 = top) that mi355rt_render returns.
extern "C" int mi355rt_write_pfm(const char* path, const float* linear_rgb, uint32_t width, uint32_t height) {
    using mi355rt_host::set_error;
    if (!path || !linear_rgb || width == 0 || height == 0) return set_error(MI355RT_ERR_INVALID, "write_pfm: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return set_error(MI355RT_ERR_IO, std::string("write_pfm: cannot open ") + path);
    bool ok = std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
    for (uint32_t y = height; ok && y-- > 0; )
        ok = std::fwrite(linear_rgb + (size_t)y * width * 3, sizeof(float), (size_t)width * 3, f) == (size_t)width * 3;
    std::fclose(f);
    return ok ? MI355RT_OK : set_error(MI355RT_ERR_IO, "write_pfm: short write");
}
