// Baseline JPEG -> 8-bit BGR frame in HBM: the image decode at the HTTP edge (reference
// backend_server.py:139-145, cv2.imdecode(IMREAD_COLOR) = libjpeg with its defaults: islow IDCT, fancy upsampling).
//
//   host   markers, Huffman tables and the entropy-coded segment (inherently serial: one bit stream with DC
//          prediction) -> quantised coefficients, int16, block-major per component;
//   device dequantisation + jidctint.c (one thread per 8x8 block), then per output pixel h2v2 / h2v1 "fancy"
//          (triangle) chroma upsampling + YCbCr -> RGB in libjpeg's 16-bit fixed point - the kernels of the ELA
//          round trip (forensic_kernels.hip) generalised to any image size - written as packed BGR where
//          dfd_analyze_frame would have uploaded the frame.
//
// Supported: what browsers and cv2.imencode write - 8-bit baseline (SOF0) or extended-sequential Huffman (SOF1),
// gray or YCbCr with 4:4:4 / 4:2:2 (h2v1) / 4:2:0 (h2v2) sampling, one interleaved scan, restart intervals.
// Anything else (progressive, arithmetic, CMYK, 12-bit, multi-scan) returns DFD_ERR_UNSUPPORTED and the host
// keeps its own decoder for it (backend_server.decode_image).
#include <cstring>
#include <new>
#include <vector>

#include "dfd_common.h"
#include "jpeg_dct.h"

using namespace dfd;

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr size_t kMaxJpegPixels = (size_t)1 << 26;       // 8192 x 8192: 0.4 GB of coefficients at 4:4:4

struct HuffTable {
    bool present = false;
    uint8_t vals[256];
    int maxcode[18], valptr[17], mincode[17];
    uint16_t look[512];                       // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    // false: the code lengths over-subscribe the code space (jdhuff.c's "code >= 1 << si" check, the Kraft
    // inequality) - such a table would index past look[] here and below vals[0] in huff_decode
    bool build(const uint8_t* bits, const uint8_t* v, int nvals) {
        present = false;
        if (nvals < 0 || nvals > 256) return false;
        memset(vals, 0, sizeof vals);
        memcpy(vals, v, (size_t)nvals);
        int code = 0, k = 0;
        memset(look, 0, sizeof look);
        for (int l = 1; l <= 16; ++l) {
            if (code + (int)bits[l] > (1 << l)) return false;
            valptr[l] = k;
            mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                if (l <= 9) {
                    const int base = code << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); ++f) look[base + f] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, bw = 0, bh = 0; };   // bw/bh: blocks incl. MCU padding

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill() {
        while (nbits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0) ++p;                   // stuffed zero
                    else { hit_marker = true; --p; b = 0; }        // a marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)(acc >> (32 - n)); }
    void skip(int n) { acc <<= n; nbits -= n; }
    int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const HuffTable& t) {
    const int look = br.peek(9);
    const uint16_t e = t.look[look];
    if (e) { br.skip(e >> 8); return e & 0xff; }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (c <= t.maxcode[l]) {
            const int idx = t.valptr[l] + c - t.mincode[l];
            br.skip(l);
            return (idx >= 0 && idx < 256) ? t.vals[idx] : -1;
        }
    }
    br.skip(16);
    return -1;                                                     // corrupt stream
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Parsed {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    Component comp[3];
    uint16_t q[4][64];                         // natural order
    bool qpresent[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    const uint8_t* scan = nullptr;
    const uint8_t* end = nullptr;
};

int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// -> DFD_OK, DFD_ERR_ARG (not a JPEG / truncated / corrupt) or DFD_ERR_UNSUPPORTED
int parse_headers(dfd_handle* h, const uint8_t* d, size_t len, Parsed* P) {
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(h, DFD_ERR_ARG, "decode_jpeg: not a JPEG (no SOI)");
    size_t pos = 2;
    P->end = d + len;
    bool have_sof = false;
    while (pos + 4 <= len) {
        if (d[pos] != 0xFF) return fail(h, DFD_ERR_ARG, "decode_jpeg: marker expected at byte %zu", pos);
        while (pos < len && d[pos] == 0xFF) ++pos;                 // fill bytes
        if (pos >= len) break;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > len) break;
        const int seg = be16(d + pos);
        if (seg < 2 || pos + seg > len) return fail(h, DFD_ERR_ARG, "decode_jpeg: truncated segment");
        const uint8_t* s = d + pos + 2;
        const int n = seg - 2;
        if (m == 0xDB) {                                           // DQT
            int i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DQT");
                for (int k = 0; k < 64; ++k) {
                    P->q[tq][kZigzag[k]] = (uint16_t)(pq ? be16(s + i + 2 * k) : s[i + k]);
                }
                P->qpresent[tq] = true;
                i += pq ? 128 : 64;
            }
        } else if (m == 0xC4) {                                    // DHT
            int i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                uint8_t bits[17];
                bits[0] = 0;
                int total = 0;
                for (int l = 1; l <= 16; ++l) { bits[l] = s[i + l]; total += bits[l]; }
                if (tc > 1 || th > 3 || total > 256 || i + 17 + total > n) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DHT");
                if (!(tc ? P->ac[th] : P->dc[th]).build(bits, s + i + 17, total))
                    return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DHT (code lengths over-subscribed)");
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                       // SOF0 / SOF1: sequential Huffman
            if (n < 6 || s[0] != 8) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d-bit samples", n >= 1 ? s[0] : 0);
            P->height = be16(s + 1);
            P->width = be16(s + 3);
            P->ncomp = s[5];
            if (P->ncomp != 1 && P->ncomp != 3) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d components", P->ncomp);
            if (n < 6 + 3 * P->ncomp || P->width <= 0 || P->height <= 0) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad SOF");
            // cv2.imdecode is capped by CV_IO_MAX_IMAGE_PIXELS (2^30) and Pillow by MAX_IMAGE_PIXELS; here a header
            // alone would make decode_scan allocate width x height coefficients, so the cap comes first
            if ((size_t)P->width * (size_t)P->height > kMaxJpegPixels)
                return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d x %d exceeds the %zu-pixel limit of the GPU path", P->width,
                            P->height, kMaxJpegPixels);
            for (int c = 0; c < P->ncomp; ++c) {
                Component& C = P->comp[c];
                C.id = s[6 + 3 * c];
                C.h = s[7 + 3 * c] >> 4;
                C.v = s[7 + 3 * c] & 15;
                C.tq = s[8 + 3 * c];
                if (C.tq > 3) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad quantisation table index");
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: SOF%d (progressive / lossless / arithmetic) is not decoded on the GPU path", m - 0xC0);
        } else if (m == 0xDD) {
            if (n >= 2) P->restart = be16(s);
        } else if (m == 0xDA) {                                    // SOS
            if (!have_sof) return fail(h, DFD_ERR_ARG, "decode_jpeg: SOS before SOF");
            if (n < 1 || s[0] != P->ncomp || n < 1 + 2 * P->ncomp + 3)
                return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: non-interleaved (multi-scan) file");
            for (int c = 0; c < P->ncomp; ++c) {
                if (s[1 + 2 * c] != P->comp[c].id) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: scan component order");
                P->comp[c].td = s[2 + 2 * c] >> 4;
                P->comp[c].ta = s[2 + 2 * c] & 15;
                if (P->comp[c].td > 3 || P->comp[c].ta > 3) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad table selector");
            }
            P->scan = d + pos + seg;
            break;
        }
        pos += seg;
    }
    if (!have_sof || !P->scan) return fail(h, DFD_ERR_ARG, "decode_jpeg: no frame or scan found");
    // sampling: luma h x v in {1x1, 2x1, 2x2}, chroma 1x1
    if (P->ncomp == 1) { P->comp[0].h = P->comp[0].v = 1; }
    else {
        const Component &Y = P->comp[0], &B = P->comp[1], &R = P->comp[2];
        const bool ok = B.h == 1 && B.v == 1 && R.h == 1 && R.v == 1 &&
                        ((Y.h == 1 && Y.v == 1) || (Y.h == 2 && Y.v == 1) || (Y.h == 2 && Y.v == 2));
        if (!ok) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: sampling %dx%d,%dx%d,%dx%d", Y.h, Y.v, B.h, B.v, R.h, R.v);
    }
    P->hmax = P->comp[0].h;
    P->vmax = P->comp[0].v;
    for (int c = 0; c < P->ncomp; ++c) {
        if (!P->qpresent[P->comp[c].tq]) return fail(h, DFD_ERR_ARG, "decode_jpeg: quantisation table %d missing", P->comp[c].tq);
        if (!P->dc[P->comp[c].td].present || !P->ac[P->comp[c].ta].present) return fail(h, DFD_ERR_ARG, "decode_jpeg: Huffman table missing");
    }
    return DFD_OK;
}

// entropy-coded segment -> coefficients (natural order) per component, blocks row-major with MCU padding
int decode_scan(dfd_handle* h, Parsed* P, std::vector<int16_t>* coef, size_t* comp_off) {
    const int mcu_w = 8 * P->hmax, mcu_h = 8 * P->vmax;
    const int mcux = (P->width + mcu_w - 1) / mcu_w, mcuy = (P->height + mcu_h - 1) / mcu_h;
    size_t total = 0;
    for (int c = 0; c < P->ncomp; ++c) {
        Component& C = P->comp[c];
        C.bw = mcux * C.h;
        C.bh = mcuy * C.v;
        comp_off[c] = total;
        total += (size_t)C.bw * C.bh * 64;
    }
    try {
        coef->assign(total, 0);
    } catch (const std::bad_alloc&) {                              // never unwinds through the C ABI
        return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: %zu coefficients do not fit in host memory", total);
    }
    BitReader br{P->scan, P->end};
    int pred[3] = {0, 0, 0};
    int until_restart = P->restart;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (P->restart && until_restart == 0) {
                // byte-align, expect RSTn
                const uint8_t* p = br.p;
                while (p + 1 < P->end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
                if (p + 1 >= P->end) return fail(h, DFD_ERR_ARG, "decode_jpeg: restart marker missing");
                br.p = p + 2;
                br.reset();
                pred[0] = pred[1] = pred[2] = 0;
                until_restart = P->restart;
            }
            for (int c = 0; c < P->ncomp; ++c) {
                const Component& C = P->comp[c];
                const HuffTable &D = P->dc[C.td], &A = P->ac[C.ta];
                for (int by = 0; by < C.v; ++by)
                    for (int bx = 0; bx < C.h; ++bx) {
                        int16_t* blk = coef->data() + comp_off[c] + ((size_t)(my * C.v + by) * C.bw + (mx * C.h + bx)) * 64;
                        int s = huff_decode(br, D);
                        if (s < 0 || s > 11) return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt DC code");
                        if (s) pred[c] += extend(br.get(s), s);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = huff_decode(br, A);
                            if (rs < 0) return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt AC code");
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return fail(h, DFD_ERR_ARG, "decode_jpeg: coefficient index out of range");
                            blk[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                            ++k;
                        }
                    }
            }
            if (P->restart) --until_restart;
        }
    return DFD_OK;
}

// ------------------------------------------------------------------------------------------------ device
struct JpegPlanes {
    const int16_t* coef[3];
    uint8_t* plane[3];
    int bw[3], bh[3];                          // blocks
    int qoff[3];                               // offset of the component's table in q (64 entries each)
};

__device__ __forceinline__ int clampu8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per 8x8 block: dequantise, jidctint.c (columns, then rows), +128, clamp
__global__ __launch_bounds__(64) void jpeg_idct_kernel(JpegPlanes J, const uint16_t* __restrict__ q, int nb0, int nb1, int nb2) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    int c = 0, b = g;
    if (b >= nb0) { b -= nb0; c = 1; if (b >= nb1) { b -= nb1; c = 2; if (b >= nb2) return; } }
    const int16_t* src = J.coef[c] + (size_t)b * 64;
    const uint16_t* qt = q + J.qoff[c];
    int d[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = (int)src[i] * (int)qt[i];
#pragma unroll
    for (int col = 0; col < 8; ++col) idct8<true>(d + col, 8);
#pragma unroll
    for (int r = 0; r < 8; ++r) idct8<false>(d + 8 * r, 1);
    const int by = b / J.bw[c], bx = b - by * J.bw[c];
    uint8_t* dst = J.plane[c] + ((size_t)by * 8) * (J.bw[c] * 8) + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            lo |= (uint32_t)clampu8(d[8 * r + x] + 128) << (8 * x);
            hi |= (uint32_t)clampu8(d[8 * r + 4 + x] + 128) << (8 * x);
        }
        uint32_t* o = reinterpret_cast<uint32_t*>(dst + (size_t)r * (J.bw[c] * 8));
        o[0] = lo;
        o[1] = hi;
    }
}

// jdsample.c h2v2_fancy_upsample on a chroma plane with `cw` x `ch` real samples (row stride `cs`)
__device__ __forceinline__ int fancy_h2v2(const uint8_t* p, int cs, int cw, int ch, int Y, int X) {
    const int i = Y >> 1, c = X >> 1;
    const int nb = (Y & 1) ? (i + 1 < ch ? i + 1 : ch - 1) : (i > 0 ? i - 1 : 0);
    const uint8_t *r0 = p + (size_t)i * cs, *r1 = p + (size_t)nb * cs;
    const int cur = 3 * r0[c] + r1[c];
    if (cw == 1) return (4 * cur + ((X & 1) ? 7 : 8)) >> 4;
    if ((X & 1) == 0) {
        if (c == 0) return (4 * cur + 8) >> 4;
        return (3 * cur + (3 * r0[c - 1] + r1[c - 1]) + 8) >> 4;
    }
    if (c == cw - 1) return (4 * cur + 7) >> 4;
    return (3 * cur + (3 * r0[c + 1] + r1[c + 1]) + 7) >> 4;
}

// jdsample.c h2v1_fancy_upsample
__device__ __forceinline__ int fancy_h2v1(const uint8_t* p, int cs, int cw, int Y, int X) {
    const uint8_t* r = p + (size_t)Y * cs;
    const int c = X >> 1;
    if (cw == 1) return r[0];
    if ((X & 1) == 0) return c == 0 ? r[0] : (3 * r[c] + r[c - 1] + 1) >> 2;
    return c == cw - 1 ? r[c] : (3 * r[c] + r[c + 1] + 2) >> 2;
}

// mode 0 gray, 1 4:4:4, 2 h2v1, 3 h2v2.  Output: packed BGR rows of `out_stride` bytes (cv2.imdecode IMREAD_COLOR)
__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegPlanes J, int mode, int width, int height, uint8_t* __restrict__ out,
                                                         int out_stride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const int ys = J.bw[0] * 8;
    const int Yv = J.plane[0][(size_t)y * ys + x];
    uint8_t* o = out + (size_t)y * out_stride + 3 * x;
    if (mode == 0) { o[0] = o[1] = o[2] = (uint8_t)Yv; return; }
    const int cs = J.bw[1] * 8;
    int cb, cr;
    if (mode == 1) {
        cb = J.plane[1][(size_t)y * cs + x];
        cr = J.plane[2][(size_t)y * cs + x];
    } else if (mode == 2) {
        const int cw = (width + 1) >> 1;
        cb = fancy_h2v1(J.plane[1], cs, cw, y, x);
        cr = fancy_h2v1(J.plane[2], cs, cw, y, x);
    } else {
        const int cw = (width + 1) >> 1, ch = (height + 1) >> 1;
        cb = fancy_h2v2(J.plane[1], cs, cw, ch, y, x);
        cr = fancy_h2v2(J.plane[2], cs, cw, ch, y, x);
    }
    cb -= 128;
    cr -= 128;
    o[2] = (uint8_t)clampu8(Yv + ((JFIX(1.40200) * cr + 32768) >> 16));
    o[1] = (uint8_t)clampu8(Yv + ((-JFIX(0.34414) * cb + 32768 - JFIX(0.71414) * cr) >> 16));
    o[0] = (uint8_t)clampu8(Yv + ((JFIX(1.77200) * cb + 32768) >> 16));
}

}  // namespace

namespace dfd {

// JPEG bytes -> packed BGR frame in h->frame_buf (row stride width * 3); *hh / *ww receive the size
int jpeg_decode_to_frame(dfd_handle* h, const uint8_t* jpeg, size_t len, int* hh, int* ww) {
    Parsed P;
    int rc = parse_headers(h, jpeg, len, &P);
    if (rc) return rc;
    std::vector<int16_t> coef;
    size_t off[3] = {0, 0, 0};
    if ((rc = decode_scan(h, &P, &coef, off))) return rc;
    const size_t coef_bytes = coef.size() * 2;
    size_t plane_off[3], plane_total = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        plane_off[c] = plane_total;
        plane_total += ((size_t)P.comp[c].bw * 8 * P.comp[c].bh * 8 + 255) & ~(size_t)255;
    }
    const size_t qbytes = 3 * 64 * 2;
    const size_t coef_al = (coef_bytes + 255) & ~(size_t)255;
    if ((rc = ensure(h, &h->jpeg_work, coef_al + plane_total + qbytes))) return rc;      // [coefficients][planes][tables]
    if ((rc = ensure(h, &h->frame_buf, (size_t)P.height * P.width * 3))) return rc;
    uint8_t* base = static_cast<uint8_t*>(h->jpeg_work.p);
    uint16_t qhost[3 * 64];
    JpegPlanes J{};
    int nb[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c) {
        const int cc = c < P.ncomp ? c : 0;
        J.coef[c] = reinterpret_cast<const int16_t*>(base) + off[cc];
        J.plane[c] = base + coef_al + plane_off[cc];
        J.bw[c] = P.comp[cc].bw;
        J.bh[c] = P.comp[cc].bh;
        J.qoff[c] = 64 * c;
        memcpy(qhost + 64 * c, P.q[P.comp[cc].tq], 128);
        if (c < P.ncomp) nb[c] = P.comp[c].bw * P.comp[c].bh;
    }
    uint16_t* qdev = reinterpret_cast<uint16_t*>(base + coef_al + plane_total);
    DFD_HIP_TRY(h, hipMemcpyAsync(base, coef.data(), coef_bytes, hipMemcpyHostToDevice, h->stream));
    DFD_HIP_TRY(h, hipMemcpyAsync(qdev, qhost, qbytes, hipMemcpyHostToDevice, h->stream));
    const int total_blocks = nb[0] + nb[1] + nb[2];
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total_blocks + 63) / 64), dim3(64), 0, h->stream, J, qdev, nb[0], nb[1], nb[2]);
    const int mode = P.ncomp == 1 ? 0 : (P.hmax == 1 ? 1 : (P.vmax == 1 ? 2 : 3));
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((P.width + 255) / 256, P.height), dim3(256), 0, h->stream, J, mode, P.width,
                       P.height, static_cast<uint8_t*>(h->frame_buf.p), P.width * 3);
    DFD_HIP_TRY(h, hipGetLastError());
    // the host vectors (coef, qhost) must outlive the asynchronous copies
    DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
    *hh = P.height;
    *ww = P.width;
    return DFD_OK;
}

}  // namespace dfd

extern "C" {

// Host half only (no GPU): headers + entropy decoding.  info[16] = width, height, components, hmax, vmax,
// then per component blocks_w, blocks_h, table index; qtables_out: 4 x 64 uint16 (natural order);
// coef_out: int16 coefficients (natural order, block-major per component, components concatenated).
int dfd_jpeg_coefficients(const uint8_t* jpeg, size_t len, int* info, uint16_t* qtables_out, int16_t* coef_out,
                          size_t capacity, size_t* count) {
    if (!jpeg || !info || !count) return fail(nullptr, DFD_ERR_ARG, "jpeg_coefficients: null pointer");
    Parsed P;
    int rc = parse_headers(nullptr, jpeg, len, &P);
    if (rc) return rc;
    std::vector<int16_t> coef;
    size_t off[3] = {0, 0, 0};
    if ((rc = decode_scan(nullptr, &P, &coef, off))) return rc;
    info[0] = P.width; info[1] = P.height; info[2] = P.ncomp; info[3] = P.hmax; info[4] = P.vmax;
    for (int c = 0; c < 3; ++c) {
        info[5 + 3 * c] = c < P.ncomp ? P.comp[c].bw : 0;
        info[6 + 3 * c] = c < P.ncomp ? P.comp[c].bh : 0;
        info[7 + 3 * c] = c < P.ncomp ? P.comp[c].tq : 0;
    }
    *count = coef.size();
    if (qtables_out) memcpy(qtables_out, P.q, sizeof P.q);
    if (coef_out) {
        if (coef.size() > capacity) return fail(nullptr, DFD_ERR_ARG, "jpeg_coefficients: capacity");
        memcpy(coef_out, coef.data(), coef.size() * 2);
    }
    return DFD_OK;
}

int dfd_decode_jpeg(dfd_handle* h, const uint8_t* jpeg, size_t len, uint8_t* bgr_out, size_t capacity, int* height, int* width) {
    if (!h) return DFD_ERR_ARG;
    if (!jpeg || !height || !width) return fail(h, DFD_ERR_ARG, "decode_jpeg: null pointer");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int hh = 0, ww = 0;
    const int rc = jpeg_decode_to_frame(h, jpeg, len, &hh, &ww);
    if (rc) return rc;
    *height = hh;
    *width = ww;
    if (bgr_out) {
        const size_t need = (size_t)hh * ww * 3;
        if (need > capacity) return fail(h, DFD_ERR_ARG, "decode_jpeg: %zu bytes needed, capacity %zu", need, capacity);
        DFD_HIP_TRY(h, hipMemcpyAsync(bgr_out, h->frame_buf.p, need, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return DFD_OK;
}

}  // extern "C"
