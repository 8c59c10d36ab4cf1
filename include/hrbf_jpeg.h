/* hrbf_jpeg.h — baseline JPEG decoder for the colour frames of .klg logs, header-only, no dependencies.
 *
 * The reference decodes them with libjpeg at its defaults (GUI/src/Tools/JPEGLoader.h:46-97: jpeg_read_header,
 * jpeg_start_decompress, jpeg_read_scanlines — i.e. JDCT_ISLOW, fancy upsampling, JCS_RGB). The image this library is
 * built in has no libjpeg headers, so the same arithmetic is written out here: the 13-bit fixed-point "islow" inverse DCT,
 * the triangle-filter ("fancy") chroma upsampling for 2x1 and 2x2 subsampling with libjpeg's edge rules, and the 16-bit
 * fixed-point YCbCr -> RGB tables — bit for bit what libjpeg / libjpeg-turbo produce (tests/test_cpp_io.py compares with
 * Pillow = libjpeg-turbo on photographs-like and noise images, every subsampling, odd sizes, restart markers).
 *
 * Supported: SOF0 / SOF1 (Huffman, 8-bit, sequential), 1 or 3 components, sampling 1x1, 2x1, 2x2 (luma) with 1x1 chroma,
 * restart intervals. Progressive / arithmetic / 12-bit / CMYK files are rejected with an exception that says so. */
#ifndef HRBF_JPEG_H
#define HRBF_JPEG_H
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hrbf_mi355 {

class JpegDecoder {
public:
    /* decodes to interleaved RGB, 8 bit; returns width / height through the arguments */
    static std::vector<uint8_t> decodeRGB(const uint8_t *data, size_t size, int &width, int &height)
    {
        JpegDecoder d(data, size);
        d.parse();
        width = d.W_; height = d.H_;
        return d.toRGB();
    }

private:
    struct Huff { uint8_t bits[17]; uint8_t vals[256]; int maxcode[18]; int valptr[17]; int mincode[17]; uint16_t look[512]; bool set = false; };
    struct Comp { int id, h, v, tq, td, ta; int bw, bh; int dw, dh; int pred; std::vector<uint8_t> plane; int stride; };

    const uint8_t *p_, *end_;
    int W_ = 0, H_ = 0, nc_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0;
    uint16_t qt_[4][64]; bool qset_[4] = {false, false, false, false};
    Huff dc_[4], ac_[4];
    Comp c_[3];
    uint32_t bitbuf_ = 0; int bitcnt_ = 0; bool hitMarker_ = false;

    JpegDecoder(const uint8_t *d, size_t n) : p_(d), end_(d + n) {}
    [[noreturn]] static void fail(const std::string &m) { throw std::runtime_error("jpeg: " + m); }

    int u8() { if (p_ >= end_) fail("truncated"); return *p_++; }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    static const int *zigzag()
    {
        static const int z[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return z;
    }

    void parse()
    {
        if (u8() != 0xFF || u8() != 0xD8) fail("no SOI");
        bool haveFrame = false;
        for (;;) {
            int m = u8();
            if (m != 0xFF) continue;              /* tolerate fill bytes between segments, as libjpeg does */
            while ((m = u8()) == 0xFF) {}
            if (m == 0) continue;
            if (m == 0xD9) fail("EOI before any scan");
            if (m >= 0xD0 && m <= 0xD7) continue;
            const int len = u16();
            if (len < 2 || p_ + (len - 2) > end_) fail("bad segment length");
            const uint8_t *seg = p_, *segEnd = p_ + len - 2;
            switch (m) {
            case 0xDB:   /* DQT */
                while (p_ < segEnd) {
                    const int pq = u8(); const int t = pq & 15, prec = pq >> 4;
                    if (t > 3) fail("bad quantisation table id");
                    for (int i = 0; i < 64; ++i) qt_[t][zigzag()[i]] = (uint16_t)(prec ? u16() : u8());
                    qset_[t] = true;
                }
                break;
            case 0xC4:   /* DHT */
                while (p_ < segEnd) {
                    const int tc = u8(); const int t = tc & 15, cls = tc >> 4;
                    if (t > 3 || cls > 1) fail("bad Huffman table id");
                    Huff &h = cls ? ac_[t] : dc_[t];
                    int total = 0;
                    h.bits[0] = 0;
                    for (int i = 1; i <= 16; ++i) { h.bits[i] = (uint8_t)u8(); total += h.bits[i]; }
                    if (total > 256) fail("bad Huffman table");
                    for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8();
                    buildHuff(h, total);
                }
                break;
            case 0xC0: case 0xC1: {   /* SOF0 / SOF1 */
                if (u8() != 8) fail("only 8-bit samples are supported");
                H_ = u16(); W_ = u16(); nc_ = u8();
                if (W_ <= 0 || H_ <= 0) fail("empty image");
                if ((long long)W_ * H_ > (1ll << 26)) fail("image larger than 64 M pixels");
                if (nc_ != 1 && nc_ != 3) fail("only greyscale and YCbCr files are supported");
                for (int i = 0; i < nc_; ++i) {
                    c_[i].id = u8(); const int hv = u8(); c_[i].h = hv >> 4; c_[i].v = hv & 15; c_[i].tq = u8();
                    if (c_[i].h < 1 || c_[i].v < 1 || c_[i].tq > 3) fail("bad component");
                    if (c_[i].h > hmax_) hmax_ = c_[i].h;
                    if (c_[i].v > vmax_) vmax_ = c_[i].v;
                }
                if (nc_ == 1) { c_[0].h = c_[0].v = 1; hmax_ = vmax_ = 1; }   /* a single component is never subsampled */
                else {
                    const bool ok = c_[1].h == 1 && c_[1].v == 1 && c_[2].h == 1 && c_[2].v == 1 &&
                                    ((c_[0].h == 1 && c_[0].v == 1) || (c_[0].h == 2 && c_[0].v == 1) || (c_[0].h == 2 && c_[0].v == 2));
                    if (!ok) fail("chroma subsampling other than 4:4:4, 4:2:2 and 4:2:0 is not supported");
                }
                haveFrame = true;
                break;
            }
            case 0xC2: case 0xC6: case 0xCA: case 0xCE: fail("progressive files are not supported (baseline only)");
            case 0xC3: case 0xC5: case 0xC7: case 0xC9: case 0xCB: case 0xCD: case 0xCF: fail("lossless / hierarchical / arithmetic-coded files are not supported");
            case 0xDD: restart_ = u16(); break;
            case 0xDA: {   /* SOS: one interleaved scan with every component */
                if (!haveFrame) fail("SOS before SOF");
                const int ns = u8();
                if (ns != nc_) fail("multi-scan files are not supported");
                for (int i = 0; i < ns; ++i) {
                    const int id = u8(), t = u8();
                    int k = -1;
                    for (int j = 0; j < nc_; ++j) if (c_[j].id == id) k = j;
                    if (k != i) fail("unexpected component order in the scan");
                    c_[k].td = t >> 4; c_[k].ta = t & 15;
                    if (c_[k].td > 3 || c_[k].ta > 3 || !dc_[c_[k].td].set || !ac_[c_[k].ta].set || !qset_[c_[k].tq]) fail("scan refers to a missing table");
                }
                p_ = segEnd;
                decodeScan();
                return;
            }
            default: break;   /* APPn, COM, ... */
            }
            p_ = segEnd; (void)seg;
        }
    }

    static void buildHuff(Huff &h, int total)
    {
        int code = 0, k = 0;
        uint16_t codes[256]; uint8_t sizes[256];
        for (int l = 1; l <= 16; ++l) {
            h.valptr[l] = k; h.mincode[l] = code;
            for (int i = 0; i < h.bits[l]; ++i) { codes[k] = (uint16_t)code; sizes[k] = (uint8_t)l; ++k; ++code; }
            if (code > (1 << l)) fail("bad Huffman table");     /* more codes of this length than the length can hold */
            h.maxcode[l] = h.bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7FFFFFFF;
        std::memset(h.look, 0, sizeof(h.look));
        for (int i = 0; i < total; ++i)
            if (sizes[i] <= 9) {
                const int shift = 9 - sizes[i];
                for (int j = 0; j < (1 << shift); ++j) h.look[(codes[i] << shift) | j] = (uint16_t)((sizes[i] << 8) | h.vals[i]);
            }
        h.set = true;
    }

    /* entropy-coded segment: 0xFF00 is a stuffed 0xFF; any other marker ends the data (zeros are fed from there on) */
    void fill()
    {
        while (bitcnt_ <= 24) {
            int b = 0;
            if (!hitMarker_ && p_ < end_) {
                b = *p_++;
                if (b == 0xFF) {
                    int m = p_ < end_ ? *p_ : 0xD9;
                    if (m == 0) ++p_;
                    else { hitMarker_ = true; --p_; b = 0; }
                }
            }
            bitbuf_ |= (uint32_t)b << (24 - bitcnt_);
            bitcnt_ += 8;
        }
    }
    int getBits(int n)
    {
        if (n == 0) return 0;
        if (bitcnt_ < n) fill();
        const int v = (int)(bitbuf_ >> (32 - n));
        bitbuf_ <<= n; bitcnt_ -= n;
        return v;
    }
    int decodeHuff(const Huff &h)
    {
        if (bitcnt_ < 16) fill();
        const int lk = h.look[bitbuf_ >> 23];
        if (lk) { const int n = lk >> 8; bitbuf_ <<= n; bitcnt_ -= n; return lk & 255; }
        int code = (int)(bitbuf_ >> 22), l = 10;
        while (l <= 16 && code > h.maxcode[l]) { ++l; code = (int)(bitbuf_ >> (32 - l)); }
        if (l > 16) fail("bad Huffman code");
        bitbuf_ <<= l; bitcnt_ -= l;
        return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

    void decodeScan()
    {
        const int mcuW = 8 * hmax_, mcuH = 8 * vmax_;
        const int mcux = (W_ + mcuW - 1) / mcuW, mcuy = (H_ + mcuH - 1) / mcuH;
        for (int i = 0; i < nc_; ++i) {
            Comp &c = c_[i];
            c.bw = mcux * c.h; c.bh = mcuy * c.v;                                  /* blocks, padded to whole MCUs */
            c.dw = (W_ * c.h + hmax_ - 1) / hmax_; c.dh = (H_ * c.v + vmax_ - 1) / vmax_;   /* downsampled_width / height */
            c.stride = c.bw * 8; c.plane.assign((size_t)c.stride * c.bh * 8, 0); c.pred = 0;
        }
        int untilRestart = restart_, nextRst = 0;
        int coef[64];
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                if (restart_ && untilRestart == 0) {
                    /* byte-align, expect RSTn */
                    bitbuf_ = 0; bitcnt_ = 0;
                    if (!hitMarker_) { while (p_ < end_ && *p_ != 0xFF) ++p_; }
                    while (p_ + 1 < end_ && p_[0] == 0xFF && p_[1] == 0xFF) ++p_;
                    if (p_ + 1 < end_ && p_[0] == 0xFF && p_[1] == (0xD0 | nextRst)) p_ += 2; else fail("missing restart marker");
                    hitMarker_ = false; nextRst = (nextRst + 1) & 7; untilRestart = restart_;
                    for (int i = 0; i < nc_; ++i) c_[i].pred = 0;
                }
                for (int i = 0; i < nc_; ++i) {
                    Comp &c = c_[i];
                    for (int by = 0; by < c.v; ++by)
                        for (int bx = 0; bx < c.h; ++bx) {
                            std::memset(coef, 0, sizeof(coef));
                            int s = decodeHuff(dc_[c.td]);
                            if (s > 11) fail("bad DC size");
                            c.pred = (int)((unsigned)c.pred + (unsigned)(s ? extend(getBits(s), s) : 0));
                            coef[0] = (int)((long long)c.pred * qt_[c.tq][0]);
                            for (int k = 1; k < 64;) {
                                const int rs = decodeHuff(ac_[c.ta]);
                                const int r = rs >> 4; s = rs & 15;
                                if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                                k += r;
                                if (k > 63) fail("bad AC run");
                                const int zz = zigzag()[k];
                                coef[zz] = extend(getBits(s), s) * qt_[c.tq][zz];
                                ++k;
                            }
                            idctIslow(coef, c.plane.data() + (size_t)((my * c.v + by) * 8) * c.stride + (size_t)(mx * c.h + bx) * 8, c.stride);
                        }
                }
                if (restart_) --untilRestart;
            }
    }

    typedef long long wide;   /* libjpeg's JLONG: the products of a damaged file must not overflow (undefined behaviour in 32 bits) */
    static inline wide descale(wide x, int n) { return (x + ((wide)1 << (n - 1))) >> n; }
    static inline uint8_t clampSample(wide x) { x += 128; return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

    /* jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2, on already dequantised coefficients */
    static void idctIslow(const int *in, uint8_t *out, int stride)
    {
        enum { CB = 13, P1 = 2 };
        const wide F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                   F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int ws[64];
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < 8; ++k) {
                wide i0, i1, i2, i3, i4, i5, i6, i7;
                if (pass == 0) { i0 = in[k]; i1 = in[8 + k]; i2 = in[16 + k]; i3 = in[24 + k]; i4 = in[32 + k]; i5 = in[40 + k]; i6 = in[48 + k]; i7 = in[56 + k]; }
                else { const int *w = ws + 8 * k; i0 = w[0]; i1 = w[1]; i2 = w[2]; i3 = w[3]; i4 = w[4]; i5 = w[5]; i6 = w[6]; i7 = w[7]; }
                wide z1 = (i2 + i6) * F0_541;
                const wide t2 = z1 + i6 * (-F1_847), t3 = z1 + i2 * F0_765;
                const wide t0 = (i0 + i4) * ((wide)1 << CB), t1 = (i0 - i4) * ((wide)1 << CB);
                const wide t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
                wide o0 = i7, o1 = i5, o2 = i3, o3 = i1;
                z1 = o0 + o3; wide z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
                const wide z5 = (z3 + z4) * F1_175;
                o0 *= F0_298; o1 *= F2_053; o2 *= F3_072; o3 *= F1_501;
                z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
                z3 += z5; z4 += z5;
                o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
                if (pass == 0) {
                    const int n = CB - P1;
                    ws[k] = (int)descale(t10 + o3, n); ws[56 + k] = (int)descale(t10 - o3, n); ws[8 + k] = (int)descale(t11 + o2, n); ws[48 + k] = (int)descale(t11 - o2, n);
                    ws[16 + k] = (int)descale(t12 + o1, n); ws[40 + k] = (int)descale(t12 - o1, n); ws[24 + k] = (int)descale(t13 + o0, n); ws[32 + k] = (int)descale(t13 - o0, n);
                } else {
                    const int n = CB + P1 + 3;
                    uint8_t *o = out + (size_t)k * stride;
                    o[0] = clampSample(descale(t10 + o3, n)); o[7] = clampSample(descale(t10 - o3, n)); o[1] = clampSample(descale(t11 + o2, n));
                    o[6] = clampSample(descale(t11 - o2, n)); o[2] = clampSample(descale(t12 + o1, n)); o[5] = clampSample(descale(t12 - o1, n));
                    o[3] = clampSample(descale(t13 + o0, n)); o[4] = clampSample(descale(t13 - o0, n));
                }
            }
    }

    /* jdsample.c h2v1_upsample / h2v2_upsample: plain replication — what jinit_upsampler picks when downsampled_width <= 2 */
    static void boxH(const uint8_t *in, int n, uint8_t *out) { for (int i = 0; i < n; ++i) out[2 * i] = out[2 * i + 1] = in[i]; }
    /* jdsample.c h2v1_fancy_upsample on one row of n = downsampled_width > 2 samples -> 2n samples */
    static void fancyH(const uint8_t *in, int n, uint8_t *out)
    {
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < n - 1; ++i) {
            const int v = in[i] * 3;
            out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2); out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
        }
        out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2); out[2 * n - 1] = in[n - 1];
    }
    /* jdsample.c h2v2_fancy_upsample: `near` is the row of the output line's own chroma sample, `far` its vertical neighbour */
    static void fancyHV(const uint8_t *near, const uint8_t *far, int n, uint8_t *out)
    {
        int thiscol = near[0] * 3 + far[0], nextcol = near[1] * 3 + far[1], lastcol;
        out[0] = (uint8_t)((thiscol * 4 + 8) >> 4); out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
        for (int i = 1; i < n - 1; ++i) {
            nextcol = near[i + 1] * 3 + far[i + 1];
            out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
        }
        out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
    }

    std::vector<uint8_t> toRGB()
    {
        std::vector<uint8_t> rgb((size_t)W_ * H_ * 3);
        if (nc_ == 1) {
            for (int y = 0; y < H_; ++y)
                for (int x = 0; x < W_; ++x) { const uint8_t v = c_[0].plane[(size_t)y * c_[0].stride + x]; uint8_t *o = &rgb[((size_t)y * W_ + x) * 3]; o[0] = o[1] = o[2] = v; }
            return rgb;
        }
        /* jdcolor.c build_ycc_rgb_table: SCALEBITS 16 */
        int crR[256], cbB[256], crG[256], cbG[256];
        for (int i = 0; i < 256; ++i) {
            const int x = i - 128;
            crR[i] = (91881 * x + 32768) >> 16; cbB[i] = (116130 * x + 32768) >> 16; crG[i] = -46802 * x; cbG[i] = -22554 * x + 32768;
        }
        const bool h2 = c_[0].h == 2, v2 = c_[0].v == 2;
        const int cw = c_[1].dw, ch = c_[1].dh;
        std::vector<uint8_t> cbRow((size_t)2 * cw + 2), crRow((size_t)2 * cw + 2);
        for (int y = 0; y < H_; ++y) {
            const uint8_t *cb, *cr;
            if (!h2) { cb = &c_[1].plane[(size_t)y * c_[1].stride]; cr = &c_[2].plane[(size_t)y * c_[2].stride]; }
            else {
                for (int k = 1; k <= 2; ++k) {
                    const Comp &c = c_[k];
                    uint8_t *dst = k == 1 ? cbRow.data() : crRow.data();
                    if (cw <= 2) boxH(&c.plane[(size_t)(v2 ? y >> 1 : y) * c.stride], cw, dst);      /* jinit_upsampler: fancy only if downsampled_width > 2 */
                    else if (!v2) fancyH(&c.plane[(size_t)y * c.stride], cw, dst);
                    else {
                        const int cy = y >> 1;
                        int ny = (y & 1) ? cy + 1 : cy - 1;          /* the nearer vertical neighbour; the image's first / last row stands in for itself */
                        if (ny < 0) ny = 0;
                        if (ny > ch - 1) ny = ch - 1;
                        fancyHV(&c.plane[(size_t)cy * c.stride], &c.plane[(size_t)ny * c.stride], cw, dst);
                    }
                }
                cb = cbRow.data(); cr = crRow.data();
            }
            const uint8_t *yy = &c_[0].plane[(size_t)y * c_[0].stride];
            uint8_t *o = &rgb[(size_t)y * W_ * 3];
            for (int x = 0; x < W_; ++x, o += 3) {
                const int Y = yy[x], b = cb[x], r = cr[x];
                const int R = Y + crR[r], G = Y + ((cbG[b] + crG[r]) >> 16), B = Y + cbB[b];
                o[0] = (uint8_t)(R < 0 ? 0 : R > 255 ? 255 : R); o[1] = (uint8_t)(G < 0 ? 0 : G > 255 ? 255 : G); o[2] = (uint8_t)(B < 0 ? 0 : B > 255 ? 255 : B);
            }
        }
        return rgb;
    }
};

}  // namespace hrbf_mi355
#endif
