// Host-side output of the render path: image files and the accumulation checkpoint.
//
// The reference interleaves its (u8,u8,u8) pixels and hands them to the `image` crate to save "./test.png"
// (main.rs:486-497); an all-or-nothing 10 000-spp render has no way to stop and go on.  Here:
//   rt_image_write   8-bit RGB as PNG (one zlib stream, filter 0 rows) or binary PPM,
//   rt_accum_write / rt_accum_read   the f32 sample sums plus how many samples they hold, so that a render can
//                    be continued with rt_render(sample0 = samples_done, ...) and still match an uninterrupted
//                    one bit for bit (the sums are added in sample order either way).
// No GPU is involved; these calls work without a device.
#include <unistd.h>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace {

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t>& png, const char type[4], const uint8_t* data, size_t n) {
    put_be32(png, (uint32_t)n);
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_be32(png, (uint32_t)crc32(0L, png.data() + at, (uInt)(n + 4)));
}
// Writes to "<path>.tmp", flushes it to the disk and renames it over the target: an interruption leaves either the
// old file or the new one, never a torn one (a checkpoint is the only resume point of a long render).
bool write_all(const char* path, const void* a, size_t na, const void* b, size_t nb) {
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = (na == 0 || fwrite(a, 1, na, f) == na) && (nb == 0 || fwrite(b, 1, nb, f) == nb);
    ok = (fflush(f) == 0) && ok;
    ok = (fsync(fileno(f)) == 0) && ok;
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path) == 0;
    if (!ok) (void)remove(tmp.c_str());
    return ok;
}

// Version 1 files end after `seed`; version 2 adds the caller's tag.
struct AccumHeader {
    char magic[8];  // "RTACCUM1" / "RTACCUM2"
    uint32_t width, height, samples_done, seed;
};
struct AccumHeader2 {
    AccumHeader h;
    uint64_t tag;
};

}  // namespace

extern "C" {

int rt_image_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb, int format) {
    if (!path || !rgb || !width || !height) return RT_ERR_INVALID;
    const size_t row = 3 * (size_t)width;
    if (format == RT_IMAGE_PPM) {
        char head[64];
        int n = snprintf(head, sizeof(head), "P6\n%u %u\n255\n", width, height);
        return write_all(path, head, (size_t)n, rgb, row * height) ? RT_OK : RT_ERR_IO;
    }
    if (format != RT_IMAGE_PNG) return RT_ERR_INVALID;
    if ((uint64_t)(row + 1) * height > 0x7FFFFFFFull) return RT_ERR_UNSUPPORTED;  // one IDAT chunk, 31-bit length
    std::vector<uint8_t> raw((row + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw[(row + 1) * y] = 0;  // filter type 0 (None)
        memcpy(&raw[(row + 1) * y + 1], rgb + row * y, row);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RT_ERR_IO;
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(png, "IDAT", z.data(), (size_t)zn);
    put_chunk(png, "IEND", nullptr, 0);
    return write_all(path, png.data(), png.size(), nullptr, 0) ? RT_OK : RT_ERR_IO;
}

int rt_accum_write_tagged(const char* path, uint32_t width, uint32_t height, uint32_t samples_done, uint32_t seed, uint64_t tag,
                          const float* accum_rgb) {
    if (!path || !accum_rgb || !width || !height) return RT_ERR_INVALID;
    AccumHeader2 h;
    memset(&h, 0, sizeof(h));
    memcpy(h.h.magic, "RTACCUM2", 8);
    h.h.width = width;
    h.h.height = height;
    h.h.samples_done = samples_done;
    h.h.seed = seed;
    h.tag = tag;
    return write_all(path, &h, sizeof(h), accum_rgb, (size_t)width * height * 3 * sizeof(float)) ? RT_OK : RT_ERR_IO;
}
int rt_accum_write(const char* path, uint32_t width, uint32_t height, uint32_t samples_done, uint32_t seed, const float* accum_rgb) {
    return rt_accum_write_tagged(path, width, height, samples_done, seed, 0, accum_rgb);
}

int rt_accum_read_tagged(const char* path, uint32_t* width, uint32_t* height, uint32_t* samples_done, uint32_t* seed, uint64_t* tag,
                         float* accum_rgb, uint64_t cap_floats) {
    if (!path) return RT_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return RT_ERR_IO;
    AccumHeader h;
    uint64_t file_tag = 0;
    int rc = RT_OK;
    bool head_ok = fread(&h, sizeof(h), 1, f) == 1 && h.width && h.height;
    if (head_ok && memcmp(h.magic, "RTACCUM2", 8) == 0)
        head_ok = fread(&file_tag, sizeof(file_tag), 1, f) == 1;
    else if (head_ok)
        head_ok = memcmp(h.magic, "RTACCUM1", 8) == 0;  // files written before the tag existed: tag 0
    if (!head_ok) {
        rc = RT_ERR_IO;
    } else {
        if (tag) *tag = file_tag;
        if (width) *width = h.width;
        if (height) *height = h.height;
        if (samples_done) *samples_done = h.samples_done;
        if (seed) *seed = h.seed;
        if (accum_rgb) {
            const uint64_t n = (uint64_t)h.width * h.height * 3;
            if (cap_floats < n)
                rc = RT_ERR_INVALID;
            else if (fread(accum_rgb, sizeof(float), n, f) != n)
                rc = RT_ERR_IO;
        }
    }
    fclose(f);
    return rc;
}
int rt_accum_read(const char* path, uint32_t* width, uint32_t* height, uint32_t* samples_done, uint32_t* seed, float* accum_rgb,
                  uint64_t cap_floats) {
    return rt_accum_read_tagged(path, width, height, samples_done, seed, nullptr, accum_rgb, cap_floats);
}

}  // extern "C"
