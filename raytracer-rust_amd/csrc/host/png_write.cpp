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
    return mi355rt_host::guard("write_png", MI355RT_ERR_IO, [&]() -> int {
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
    });
}

// Linear f32 dump for parity tooling: Portable FloatMap, "PF", little-endian, rows BOTTOM-UP as
// the format prescribes; `linear_rgb` is the pre-gamma mean image (row 0 = top) that mi355rt_render returns.
extern "C" int mi355rt_write_pfm(const char* path, const float* linear_rgb, uint32_t width, uint32_t height) {
    using mi355rt_host::set_error;
    return mi355rt_host::guard("write_pfm", MI355RT_ERR_IO, [&]() -> int {
    if (!path || !linear_rgb || width == 0 || height == 0) return set_error(MI355RT_ERR_INVALID, "write_pfm: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return set_error(MI355RT_ERR_IO, std::string("write_pfm: cannot open ") + path);
    bool ok = std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
    for (uint32_t y = height; ok && y-- > 0; )
        ok = std::fwrite(linear_rgb + (size_t)y * width * 3, sizeof(float), (size_t)width * 3, f) == (size_t)width * 3;
    std::fclose(f);
    return ok ? MI355RT_OK : set_error(MI355RT_ERR_IO, "write_pfm: short write");
    });
}

// OpenEXR dump of the same image (SURVEY.md 8f-3 "f32/EXR dump"): single-part scan-line file, version 2, three FLOAT
// channels B, G, R (alphabetical, as the format requires), no compression, one scan line per chunk, rows top-down
// (lineOrder INCREASING_Y).  Layout per the OpenEXR file-layout document: magic, version, attributes (name\0 type\0 size
// value) closed by \0, one u64 offset per scan line, then per line: y, byte count, B row, G row, R row.
extern "C" int mi355rt_write_exr(const char* path, const float* linear_rgb, uint32_t width, uint32_t height) {
    using mi355rt_host::set_error;
    return mi355rt_host::guard("write_exr", MI355RT_ERR_IO, [&]() -> int {
    if (!path || !linear_rgb || width == 0 || height == 0 || width > (1u << 24) || height > (1u << 24)) return set_error(MI355RT_ERR_INVALID, "write_exr: bad argument");
    std::vector<unsigned char> h;
    auto u8 = [&](unsigned v) { h.push_back((unsigned char)v); };
    auto i32 = [&](int32_t v) { for (int k = 0; k < 4; ++k) u8(((uint32_t)v >> (8 * k)) & 0xFFu); };
    auto f32 = [&](float v) { uint32_t b; std::memcpy(&b, &v, 4); i32((int32_t)b); };
    auto str = [&](const char* s) { while (*s) u8((unsigned char)*s++); u8(0); };
    auto attr = [&](const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); };
    i32(20000630); i32(2);                                            // magic 0x76 0x2f 0x31 0x01, version 2, no flags
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char* c : {"B", "G", "R"}) { str(c); i32(2 /* FLOAT */); u8(0); u8(0); u8(0); u8(0); i32(1); i32(1); }
    u8(0);
    attr("compression", "compression", 1); u8(0);                     // NO_COMPRESSION
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("lineOrder", "lineOrder", 1); u8(0);                         // INCREASING_Y
    attr("pixelAspectRatio", "float", 4); f32(1.0f);
    attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4); f32(1.0f);
    u8(0);                                                            // end of header
    const uint64_t line_bytes = 8ull + 12ull * width, table = 8ull * height;
    std::vector<float> row((size_t)3 * width);                        // (every allocation before the file is opened: nothing can throw past an open FILE*)
    FILE* f = std::fopen(path, "wb");
    if (!f) return set_error(MI355RT_ERR_IO, std::string("write_exr: cannot open ") + path);
    bool ok = std::fwrite(h.data(), 1, h.size(), f) == h.size();
    for (uint32_t y = 0; ok && y < height; ++y) {
        const uint64_t off = (uint64_t)h.size() + table + (uint64_t)y * line_bytes;
        unsigned char b[8]; for (int k = 0; k < 8; ++k) b[k] = (unsigned char)(off >> (8 * k));
        ok = std::fwrite(b, 1, 8, f) == 8;
    }
    for (uint32_t y = 0; ok && y < height; ++y) {
        const float* src = linear_rgb + (size_t)y * width * 3;
        for (uint32_t x = 0; x < width; ++x) { row[x] = src[3 * x + 2]; row[(size_t)width + x] = src[3 * x + 1]; row[(size_t)2 * width + x] = src[3 * x]; }
        const int32_t head[2] = {(int32_t)y, (int32_t)(12u * width)};
        ok = std::fwrite(head, 4, 2, f) == 2 && std::fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    std::fclose(f);
    return ok ? MI355RT_OK : set_error(MI355RT_ERR_IO, "write_exr: short write");
    });
}
