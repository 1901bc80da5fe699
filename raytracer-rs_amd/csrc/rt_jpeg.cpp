// Host-side image input for ImageTexture::new(path) (texture.rs:89-93: image::open(path).into_rgb8()).
//
// The reference decodes whatever the `image` crate knows; its one texture file is a baseline JPEG (earthmap.jpg:
// 1024x512, three components, no subsampling).  rt_image_read decodes
//   * baseline (SOF0, Huffman, 8 bit) JPEG, greyscale or YCbCr with chroma sampled 1x1, 2x1 or 2x2, restart markers, and
//   * binary PPM (P6, maxval 255)
// into 8-bit RGB rows, row 0 = top.  The JPEG path follows the arithmetic of the IJG library's default decoder so
// that its output can be checked against a libjpeg decode byte for byte: the slow-but-accurate integer inverse DCT
// (13-bit constants, 2 extra bits after the column pass), "fancy" triangle-filter upsampling of subsampled chroma, and
// the 16-bit fixed-point YCbCr -> RGB tables.  (The `image` crate's own pure-Rust decoder may differ from that by a
// least significant bit on some texels; no source of it is at hand.)  Progressive and arithmetic-coded files are refused.
// No GPU is involved.
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace {

struct Huff {
    // canonical code tables (ITU T.81 annex C / F.2.2.3): per code length the smallest code, the largest code and the
    // index of its first value
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};
struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // blocks per row / column of the padded plane
    int pred = 0;
    std::vector<uint8_t> plane;      // blocks_w * 8 by blocks_h * 8 samples
};
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t bits = 0;
    int nbits = 0;
    bool bad = false;
    int marker = 0;  // a marker met inside the entropy-coded data (restart or end of image)
    // next bit of the entropy-coded segment; 0xFF00 stands for a data byte 0xFF, any other 0xFFxx ends the segment
    int bit() {
        if (nbits == 0) {
            if (marker || p >= end) {
                bits = 0;  // (past the data: zeros, as the IJG decoder feeds)
            } else {
                uint32_t b = *p++;
                if (b == 0xFF) {
                    while (p < end && *p == 0xFF) p++;
                    const int m = p < end ? *p++ : 0xD9;
                    if (m != 0) {
                        marker = m;
                        b = 0;
                    }
                }
                bits = b;
            }
            nbits = 8;
        }
        nbits--;
        return (int)((bits >> nbits) & 1u);
    }
    int receive(int n) {
        int v = 0;
        for (int i = 0; i < n; i++) v = (v << 1) | bit();
        return v;
    }
    int decode(const Huff& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        bad = true;
        return 0;
    }
};
int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const int ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
long descale(long x, int n) { return (x + (1L << (n - 1))) >> n; }

// The accurate integer inverse DCT (Loeffler, Ligtenberg, Moschytz with 13-bit constants; columns first, two extra
// bits kept between the passes), on dequantised coefficients; output samples are level-shifted by 128 and clamped.
void idct_block(const int coef[64], const uint16_t q[64], uint8_t* out, int stride) {
    const long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
               F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    const int CB = 13, P1 = 2;
    long ws[64];
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < 8; k++) {
            long in[8];
            for (int j = 0; j < 8; j++) in[j] = pass == 0 ? (long)coef[8 * j + k] * q[8 * j + k] : ws[8 * k + j];
            long z2 = in[2], z3 = in[6];
            long z1 = (z2 + z3) * F0_541;
            long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            z2 = in[0];
            z3 = in[4];
            long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = in[7];
            tmp1 = in[5];
            tmp2 = in[3];
            tmp3 = in[1];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298;
            tmp1 *= F2_053;
            tmp2 *= F3_072;
            tmp3 *= F1_501;
            z1 *= -F0_899;
            z2 *= -F2_562;
            z3 *= -F1_961;
            z4 *= -F0_390;
            z3 += z5;
            z4 += z5;
            tmp0 += z1 + z3;
            tmp1 += z2 + z4;
            tmp2 += z2 + z3;
            tmp3 += z1 + z4;
            const long r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
            if (pass == 0)
                for (int j = 0; j < 8; j++) ws[8 * j + k] = descale(r[j], CB - P1);
            else
                for (int j = 0; j < 8; j++) out[k * stride + j] = clamp8((int)descale(r[j], CB + P1 + 3) + 128);
        }
    }
}

// Chroma planes of a subsampled image brought to full size with the triangle filter of the IJG decoder ("fancy
// upsampling"): horizontally 3/4 of the nearer sample + 1/4 of the farther one (rounding alternates), for 2x2 the
// same vertically first, on sums of 16ths.  Edge samples have no farther neighbour and are taken as they are.
// (Planes of one or two samples per row do not come here: those are replicated.)
void upsample_h2v1(const uint8_t* in, int w_in, uint8_t* out) {
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w_in - 1; i++) {
        const int v = in[i] * 3;
        out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
    }
    out[2 * w_in - 2] = (uint8_t)((in[w_in - 1] * 3 + in[w_in - 2] + 1) >> 2);
    out[2 * w_in - 1] = in[w_in - 1];
}
void upsample_h2v2_row(const uint8_t* near_row, const uint8_t* far_row, int w_in, uint8_t* out) {
    auto colsum = [&](int i) { return near_row[i] * 3 + far_row[i]; };
    int last = colsum(0), cur = colsum(0), next = colsum(1);
    out[0] = (uint8_t)((cur * 4 + 8) >> 4);
    out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
    for (int i = 1; i < w_in - 1; i++) {
        last = cur;
        cur = next;
        next = colsum(i + 1);
        out[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4);
        out[2 * i + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
    }
    last = cur;
    cur = next;
    out[2 * w_in - 2] = (uint8_t)((cur * 3 + last + 8) >> 4);
    out[2 * w_in - 1] = (uint8_t)((cur * 4 + 7) >> 4);
}

int fail_msg(std::string* err, const char* msg) {
    if (err) *err = msg;
    return RT_ERR_UNSUPPORTED;
}

// header_only: stops at the frame header (width and height known).
int decode_jpeg(const std::vector<uint8_t>& d, bool header_only, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgb, std::string* err) {
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Component> comp;
    int restart_interval = 0, hmax = 1, vmax = 1;
    bool have_frame = false;
    size_t i = 2;
    auto be16 = [&](size_t at) { return (int)((d[at] << 8) | d[at + 1]); };
    while (i + 4 <= d.size()) {
        if (d[i] != 0xFF) return fail_msg(err, "JPEG: marker expected");
        const int m = d[i + 1];
        if (m == 0xFF) {
            i++;
            continue;
        }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) {
            i += 2;
            continue;
        }
        if (m == 0xD9) break;
        const size_t len = (size_t)be16(i + 2);
        if (len < 2 || i + 2 + len > d.size()) return fail_msg(err, "JPEG: truncated segment");
        const uint8_t* s = &d[i + 4];
        const size_t n = len - 2;
        if (m == 0xDB) {  // quantisation tables (zigzag order in the file)
            size_t at = 0;
            while (at < n) {
                const int pq = s[at] >> 4, tq = s[at] & 15;
                at++;
                if (tq > 3 || at + (pq ? 128 : 64) > n) return fail_msg(err, "JPEG: bad quantisation table");
                for (int k = 0; k < 64; k++) {
                    qt[tq][ZIGZAG[k]] = pq ? (uint16_t)((s[at] << 8) | s[at + 1]) : s[at];
                    at += pq ? 2 : 1;
                }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {  // Huffman tables
            size_t at = 0;
            while (at + 17 <= n) {
                const int tc = s[at] >> 4, th = s[at] & 15;
                if (tc > 1 || th > 3) return fail_msg(err, "JPEG: bad Huffman table");
                Huff& h = tc ? hac[th] : hdc[th];
                int counts[17], total = 0;
                for (int l = 1; l <= 16; l++) total += counts[l] = s[at + l];
                at += 17;
                if (total > 256 || at + (size_t)total > n) return fail_msg(err, "JPEG: bad Huffman table");
                memcpy(h.vals, s + at, (size_t)total);
                at += (size_t)total;
                int code = 0, k = 0;
                for (int l = 1; l <= 16; l++) {
                    h.valptr[l] = k;
                    h.mincode[l] = code;
                    code += counts[l];
                    k += counts[l];
                    h.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {  // baseline / extended sequential, Huffman
            if (n < 6 || s[0] != 8) return fail_msg(err, "JPEG: only 8-bit samples");
            height = (uint32_t)((s[1] << 8) | s[2]);
            width = (uint32_t)((s[3] << 8) | s[4]);
            const int nc = s[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc || !width || !height) return fail_msg(err, "JPEG: one or three components expected");
            comp.assign((size_t)nc, Component());
            for (int c = 0; c < nc; c++) {
                comp[c].id = s[6 + 3 * c];
                comp[c].h = s[7 + 3 * c] >> 4;
                comp[c].v = s[7 + 3 * c] & 15;
                comp[c].tq = s[8 + 3 * c];
                if (comp[c].tq > 3) return fail_msg(err, "JPEG: bad quantisation table index");
                hmax = comp[c].h > hmax ? comp[c].h : hmax;
                vmax = comp[c].v > vmax ? comp[c].v : vmax;
            }
            const bool luma_full = comp[0].h == hmax && comp[0].v == vmax;
            bool chroma_ok = true;
            for (int c = 1; c < nc; c++) chroma_ok = chroma_ok && comp[c].h == 1 && comp[c].v == 1;
            if (hmax > 2 || vmax > 2 || !luma_full || (nc == 3 && !chroma_ok) || (hmax == 1 && vmax == 2))
                return fail_msg(err, "JPEG: chroma sampling other than 1x1, 2x1 or 2x2");
            // (65535 x 65535 would be 12 GB of planes: a texture of more than 2^28 texels is refused, not attempted)
            if ((uint64_t)width * height > (1ull << 28)) return fail_msg(err, "JPEG: more than 2^28 pixels");
            have_frame = true;
            if (header_only) return RT_OK;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return fail_msg(err, "JPEG: only baseline Huffman files (this one is progressive, lossless or arithmetic-coded)");
        } else if (m == 0xDD) {
            if (n < 2) return fail_msg(err, "JPEG: bad restart interval");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {  // start of scan: one interleaved scan of all components
            if (!have_frame) return fail_msg(err, "JPEG: scan before frame header");
            const int ns = s[0];
            if (ns != (int)comp.size() || n < 1 + 2 * (size_t)ns + 3) return fail_msg(err, "JPEG: a single interleaved scan expected");
            for (int k = 0; k < ns; k++) {
                Component* c = nullptr;
                for (auto& cc : comp)
                    if (cc.id == s[1 + 2 * k]) c = &cc;
                if (!c) return fail_msg(err, "JPEG: scan names an unknown component");
                c->td = s[2 + 2 * k] >> 4;
                c->ta = s[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3 || !hdc[c->td].present || !hac[c->ta].present || !qt_present[c->tq])
                    return fail_msg(err, "JPEG: scan refers to a table the file does not define");
            }
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            const int mcus_x = (int)((width + (uint32_t)mcu_w - 1) / (uint32_t)mcu_w), mcus_y = (int)((height + (uint32_t)mcu_h - 1) / (uint32_t)mcu_h);
            for (auto& c : comp) {
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
                c.pred = 0;
            }
            Reader r{&d[i + 2 + len], d.data() + d.size()};
            int until_restart = restart_interval;
            int coef[64];
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && until_restart == 0) {
                        // the segment ended with RSTn (met by the bit reader, or still ahead): byte-align, reset predictions
                        if (!r.marker) {
                            r.nbits = 0;
                            while (r.p + 1 < r.end && !(r.p[0] == 0xFF && r.p[1] >= 0xD0 && r.p[1] <= 0xD7)) r.p++;
                            if (r.p + 1 < r.end) r.p += 2;
                        }
                        r.marker = 0;
                        r.nbits = 0;
                        for (auto& c : comp) c.pred = 0;
                        until_restart = restart_interval;
                    }
                    for (auto& c : comp)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                memset(coef, 0, sizeof(coef));
                                const int t = r.decode(hdc[c.td]);  // F.2.2.1: DC difference
                                if (t > 11) return fail_msg(err, "JPEG: corrupt DC coefficient");
                                c.pred += extend(r.receive(t), t);
                                coef[0] = c.pred;
                                for (int k = 1; k < 64;) {  // F.2.2.2: AC run / size pairs
                                    const int rs = r.decode(hac[c.ta]), run = rs >> 4, size = rs & 15;
                                    if (size == 0) {
                                        if (run != 15) break;  // end of block
                                        k += 16;
                                        continue;
                                    }
                                    k += run;
                                    if (k > 63) return fail_msg(err, "JPEG: corrupt AC coefficients");
                                    coef[ZIGZAG[k]] = extend(r.receive(size), size);
                                    k++;
                                }
                                if (r.bad) return fail_msg(err, "JPEG: corrupt entropy-coded data");
                                const int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                idct_block(coef, qt[c.tq], &c.plane[(size_t)py * c.blocks_w * 8 + px], c.blocks_w * 8);
                            }
                    if (restart_interval) until_restart--;
                }
            // planes -> RGB
            rgb.assign((size_t)width * height * 3, 0);
            const int W = (int)width, H = (int)height;
            if (comp.size() == 1) {
                const int stride = comp[0].blocks_w * 8;
                for (int y = 0; y < H; y++)
                    for (int x = 0; x < W; x++) {
                        const uint8_t v = comp[0].plane[(size_t)y * stride + x];
                        uint8_t* o = &rgb[((size_t)y * width + x) * 3];
                        o[0] = o[1] = o[2] = v;
                    }
                return RT_OK;
            }
            // chroma at full size (rows of the width the luma plane is padded to)
            const int ys = comp[0].blocks_w * 8;
            std::vector<uint8_t> cb((size_t)ys * H), cr((size_t)ys * H);
            for (int ci = 1; ci <= 2; ci++) {
                const Component& c = comp[ci];
                std::vector<uint8_t>& full = ci == 1 ? cb : cr;
                const int cs = c.blocks_w * 8;
                const int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;  // true size of the subsampled plane
                for (int y = 0; y < H; y++) {
                    uint8_t* out = &full[(size_t)y * ys];
                    if (hmax == 1) {
                        memcpy(out, &c.plane[(size_t)y * cs], (size_t)W);
                    } else if (cw <= 2) {
                        // planes one or two samples wide are replicated, not filtered (in both directions)
                        const uint8_t* in = &c.plane[(size_t)(y / vmax) * cs];
                        for (int x = 0; x < W; x++) out[x] = in[x / 2];
                    } else if (vmax == 1) {
                        upsample_h2v1(&c.plane[(size_t)y * cs], cw, out);
                    } else {
                        const int yc = y / 2;
                        int yn = (y & 1) ? yc + 1 : yc - 1;  // the farther row: above for an even output row, below for an odd one
                        yn = yn < 0 ? 0 : (yn >= ch ? ch - 1 : yn);
                        upsample_h2v2_row(&c.plane[(size_t)yc * cs], &c.plane[(size_t)yn * cs], cw, out);
                    }
                }
            }
            // YCbCr -> RGB in 16-bit fixed point: R = Y + 1.402 Cr', G = Y - 0.34414 Cb' - 0.71414 Cr', B = Y + 1.772 Cb'
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const int Y = comp[0].plane[(size_t)y * ys + x], b = cb[(size_t)y * ys + x] - 128, rr = cr[(size_t)y * ys + x] - 128;
                    uint8_t* o = &rgb[((size_t)y * width + x) * 3];
                    o[0] = clamp8(Y + (int)((91881L * rr + 32768L) >> 16));
                    o[1] = clamp8(Y + (int)((-22554L * b + 32768L - 46802L * rr) >> 16));
                    o[2] = clamp8(Y + (int)((116130L * b + 32768L) >> 16));
                }
            return RT_OK;
        }
        i += 2 + len;
    }
    return fail_msg(err, "JPEG: no scan found");
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long n = ok ? ftell(f) : -1;
    ok = ok && n >= 0 && n <= (1L << 30) && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out.resize((size_t)n);
        ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    return ok;
}

}  // namespace

namespace {
int image_read(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgb, uint64_t cap_bytes) {
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return RT_ERR_IO;
    std::vector<uint8_t> pix;
    uint32_t w = 0, h = 0;
    if (d.size() >= 4 && d[0] == 0xFF && d[1] == 0xD8) {
        const int rc = decode_jpeg(d, rgb == nullptr, w, h, pix, nullptr);
        if (rc) return rc;
        if (!rgb) {
            *width = w;
            *height = h;
            return RT_OK;
        }
    } else if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') {
        // "P6" width height maxval, separated by white space (comments start with #), one white-space byte, then the rows
        size_t at = 2;
        unsigned long v[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            for (;;) {
                while (at < d.size() && (d[at] == ' ' || d[at] == '\t' || d[at] == '\n' || d[at] == '\r')) at++;
                if (at < d.size() && d[at] == '#') {
                    while (at < d.size() && d[at] != '\n') at++;
                    continue;
                }
                break;
            }
            if (at >= d.size() || d[at] < '0' || d[at] > '9') return RT_ERR_UNSUPPORTED;
            while (at < d.size() && d[at] >= '0' && d[at] <= '9' && v[k] < (1ul << 24)) v[k] = v[k] * 10 + (unsigned long)(d[at++] - '0');
        }
        at++;
        if (v[2] != 255 || !v[0] || !v[1] || v[0] > 65535 || v[1] > 65535 || at + v[0] * v[1] * 3 > d.size()) return RT_ERR_UNSUPPORTED;
        w = (uint32_t)v[0];
        h = (uint32_t)v[1];
        pix.assign(d.begin() + (long)at, d.begin() + (long)(at + (size_t)w * h * 3));
    } else {
        return RT_ERR_UNSUPPORTED;
    }
    *width = w;
    *height = h;
    if (rgb) {
        if (cap_bytes < pix.size()) return RT_ERR_INVALID;
        memcpy(rgb, pix.data(), pix.size());
    }
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_image_read(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgb, uint64_t cap_bytes) {
    if (!path || !width || !height) return RT_ERR_INVALID;
    try {
        return image_read(path, width, height, rgb, cap_bytes);
    } catch (const std::exception&) {  // (no memory for what the file's header claims): nothing is thrown across the boundary
        return RT_ERR_UNSUPPORTED;
    }
}

}  // extern "C"
