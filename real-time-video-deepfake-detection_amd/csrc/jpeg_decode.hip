// Baseline JPEG -> 8-bit BGR frame in HBM: the image decode at the HTTP edge (reference
// backend_server.py:139-145, cv2.imdecode(IMREAD_COLOR) = libjpeg with its defaults: islow IDCT, fancy upsampling).
//
//   host   markers, Huffman tables and the entropy-coded segment -> quantised coefficients, int16, block-major per
//          component (jpeg_entropy.h: speculative chunks on the host pool).  Round 4: for a BATCH of restart-less files
//          the scan is decoded on the device instead (jpeg_gpu_entropy.h: a lane per 512-byte chunk) - the JPEG bytes
//          cross PCIe, not 6.2 MB of coefficients per 1080p frame; the host decoder remains the path of single files,
//          of restart-interval files and of any frame the device decoder's own checks reject;
//   device dequantisation + jidctint.c (one thread per 8x8 block), then per output pixel h2v2 / h2v1 "fancy"
//          (triangle) chroma upsampling + YCbCr -> RGB in libjpeg's 16-bit fixed point - the kernels of the ELA
//          round trip (forensic_kernels.hip) generalised to any image size - written as packed BGR where
//          dfd_analyze_frame would have uploaded the frame.
//
// Supported: what browsers and cv2.imencode write - 8-bit baseline (SOF0) or extended-sequential Huffman (SOF1),
// gray or YCbCr with 4:4:4 / 4:2:2 (h2v1) / 4:2:0 (h2v2) sampling, one interleaved scan, restart intervals.
// Anything else (progressive, arithmetic, CMYK, 12-bit, multi-scan) returns DFD_ERR_UNSUPPORTED and the host
// keeps its own decoder for it (backend_server.decode_image).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "dfd_common.h"
#include "jpeg_dct.h"

using namespace dfd;

#include "jpeg_entropy.h"
#include "jpeg_gpu_entropy.h"

using namespace dfd_jpeg;
using namespace dfd_jpeg_gpu;

namespace {

// ------------------------------------------------------------------------------------------------ device
struct JpegPlanes {
    const int16_t* coef[3];
    uint8_t* plane[3];
    int bw[3], bh[3];                          // blocks
    int qoff[3];                               // offset of the component's table in q (64 entries each)
};

__device__ __forceinline__ int clampu8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per 8x8 block: dequantise, jidctint.c (columns, then rows), +128, clamp
// (blockIdx.y = frame of a batch: coefficients / planes / tables of frame f lie f * stride further on)
__global__ __launch_bounds__(64) void jpeg_idct_kernel(JpegPlanes J, const uint16_t* __restrict__ q, int nb0, int nb1, int nb2,
                                                       size_t coef_stride, size_t plane_stride) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    int c = 0, b = g;
    if (b >= nb0) { b -= nb0; c = 1; if (b >= nb1) { b -= nb1; c = 2; if (b >= nb2) return; } }
    const int16_t* src = J.coef[c] + blockIdx.y * coef_stride + (size_t)b * 64;
    const uint16_t* qt = q + blockIdx.y * 192 + J.qoff[c];
    int d[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = (int)src[i] * (int)qt[i];
#pragma unroll
    for (int col = 0; col < 8; ++col) idct8<true>(d + col, 8);
#pragma unroll
    for (int r = 0; r < 8; ++r) idct8<false>(d + 8 * r, 1);
    const int by = b / J.bw[c], bx = b - by * J.bw[c];
    uint8_t* dst = J.plane[c] + blockIdx.y * plane_stride + ((size_t)by * 8) * (J.bw[c] * 8) + bx * 8;
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
                                                         int out_stride, size_t plane_stride, size_t out_frame_stride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    J.plane[0] += blockIdx.z * plane_stride;
    J.plane[1] += blockIdx.z * plane_stride;
    J.plane[2] += blockIdx.z * plane_stride;
    out += blockIdx.z * out_frame_stride;
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

// the same arithmetic, four pixels per thread: 12 bytes leave as three dword stores (the one-pixel kernel writes three
// single bytes per thread: 0.87 ms per 64 frames of 1080p against 0.12 ms for their IDCT).  Rows must start dword-aligned
// (out_stride % 4 == 0); the last width % 4 pixels of a row are written byte by byte.
__global__ __launch_bounds__(256) void jpeg_color4_kernel(JpegPlanes J, int mode, int width, int height, uint8_t* __restrict__ out,
                                                          int out_stride, size_t plane_stride, size_t out_frame_stride) {
    const int x0 = 4 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y;
    if (x0 >= width) return;
    J.plane[0] += blockIdx.z * plane_stride;
    J.plane[1] += blockIdx.z * plane_stride;
    J.plane[2] += blockIdx.z * plane_stride;
    out += blockIdx.z * out_frame_stride;
    const int ys = J.bw[0] * 8, cs = J.bw[1] * 8;
    const int cw = (width + 1) >> 1, ch = (height + 1) >> 1;
    uint8_t px[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + i < width ? x0 + i : width - 1;
        const int Yv = J.plane[0][(size_t)y * ys + x];
        if (mode == 0) { px[3 * i] = px[3 * i + 1] = px[3 * i + 2] = (uint8_t)Yv; continue; }
        int cb, cr;
        if (mode == 1) {
            cb = J.plane[1][(size_t)y * cs + x];
            cr = J.plane[2][(size_t)y * cs + x];
        } else if (mode == 2) {
            cb = fancy_h2v1(J.plane[1], cs, cw, y, x);
            cr = fancy_h2v1(J.plane[2], cs, cw, y, x);
        } else {
            cb = fancy_h2v2(J.plane[1], cs, cw, ch, y, x);
            cr = fancy_h2v2(J.plane[2], cs, cw, ch, y, x);
        }
        cb -= 128;
        cr -= 128;
        px[3 * i + 2] = (uint8_t)clampu8(Yv + ((JFIX(1.40200) * cr + 32768) >> 16));
        px[3 * i + 1] = (uint8_t)clampu8(Yv + ((-JFIX(0.34414) * cb + 32768 - JFIX(0.71414) * cr) >> 16));
        px[3 * i] = (uint8_t)clampu8(Yv + ((JFIX(1.77200) * cb + 32768) >> 16));
    }
    uint8_t* o = out + (size_t)y * out_stride + 3 * x0;
    if (x0 + 4 <= width) {
        uint32_t w[3];
        memcpy(w, px, 12);
        uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
        o4[0] = w[0];
        o4[1] = w[1];
        o4[2] = w[2];
    } else {
        for (int i = 0; i < 3 * (width - x0); ++i) o[i] = px[i];
    }
}

// 4:2:0 (h2v2), width % 8 == 0: a thread owns 8 x 2 output pixels - the pixel rows 2 i and 2 i + 1 that share chroma row i.
// Per plane it loads chroma rows i - 1, i, i + 1 (clamped) at columns c0 - 1 .. c0 + 4 (one dword + two bytes per row) and
// blends vertically once per column; clamping the neighbour indices reproduces fancy_h2v2's edge rules exactly (at the
// first / last column 3 * cur + cur = 4 * cur).  22 loads and 12 dword stores for 16 pixels; the four-pixel kernel issues
// 36 byte loads per 4 pixels.
__global__ __launch_bounds__(256) void jpeg_color420_kernel(JpegPlanes J, int width, int height, uint8_t* __restrict__ out, int out_stride,
                                                            size_t plane_stride, size_t out_frame_stride) {
    const int x0 = 8 * (blockIdx.x * 256 + threadIdx.x), i = blockIdx.y;
    if (x0 >= width) return;
    const uint8_t* py = J.plane[0] + blockIdx.z * plane_stride;
    out += blockIdx.z * out_frame_stride;
    const int ys = J.bw[0] * 8, cs = J.bw[1] * 8;
    const int cw = (width + 1) >> 1, ch = (height + 1) >> 1;
    const int c0 = x0 >> 1;
    const int iu = i > 0 ? i - 1 : 0, id = i + 1 < ch ? i + 1 : ch - 1;
    const int cl = c0 > 0 ? c0 - 1 : 0, cr4 = c0 + 4 < cw ? c0 + 4 : cw - 1;
    int ve[2][6], vo[2][6];                                         // [plane][column c0 - 1 + k]: vertical blends for rows 2 i / 2 i + 1
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const uint8_t* p = J.plane[1 + pl] + blockIdx.z * plane_stride;
        int r[3][6];
        const int rows[3] = {iu, i, id};
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const uint8_t* q = p + (size_t)rows[rr] * cs;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(q + c0);      // c0 % 4 == 0; columns past cw - 1 are MCU padding (unused: clamped below)
            r[rr][0] = q[cl];
            r[rr][5] = q[cr4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[rr][1 + k] = (int)((w >> (8 * k)) & 255u);
        }
#pragma unroll
        for (int k = 1; k <= 4; ++k) {                               // a column past the plane's last real sample takes the last one
            const int c = c0 + k - 1;
            if (c > cw - 1) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) r[rr][k] = r[rr][5];
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ve[pl][k] = 3 * r[1][k] + r[0][k];
            vo[pl][k] = 3 * r[1][k] + r[2][k];
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int y = 2 * i + half;
        if (y >= height) break;
        const uint2 yw = *reinterpret_cast<const uint2*>(py + (size_t)y * ys + x0);
        uint8_t px[24];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int Yv = (int)(((k < 4 ? yw.x : yw.y) >> (8 * (k & 3))) & 255u);
            const int c = 1 + (k >> 1);                              // index of column x >> 1 in the six
            int cbv, crv;
            if (half == 0) {
                cbv = (k & 1) ? (3 * ve[0][c] + ve[0][c + 1] + 7) >> 4 : (3 * ve[0][c] + ve[0][c - 1] + 8) >> 4;
                crv = (k & 1) ? (3 * ve[1][c] + ve[1][c + 1] + 7) >> 4 : (3 * ve[1][c] + ve[1][c - 1] + 8) >> 4;
            } else {
                cbv = (k & 1) ? (3 * vo[0][c] + vo[0][c + 1] + 7) >> 4 : (3 * vo[0][c] + vo[0][c - 1] + 8) >> 4;
                crv = (k & 1) ? (3 * vo[1][c] + vo[1][c + 1] + 7) >> 4 : (3 * vo[1][c] + vo[1][c - 1] + 8) >> 4;
            }
            cbv -= 128;
            crv -= 128;
            px[3 * k + 2] = (uint8_t)clampu8(Yv + ((JFIX(1.40200) * crv + 32768) >> 16));
            px[3 * k + 1] = (uint8_t)clampu8(Yv + ((-JFIX(0.34414) * cbv + 32768 - JFIX(0.71414) * crv) >> 16));
            px[3 * k] = (uint8_t)clampu8(Yv + ((JFIX(1.77200) * cbv + 32768) >> 16));
        }
        uint2 w[3];
        memcpy(w, px, 24);
        uint2* o = reinterpret_cast<uint2*>(out + (size_t)y * out_stride + 3 * x0);      // 24 x0 / 8: 8-byte aligned
        o[0] = w[0];
        o[1] = w[1];
        o[2] = w[2];
    }
}

}  // namespace

namespace dfd {

// device half for one parsed file whose coefficients sit at coef_host (pinned): upload + IDCT + upsample + colour ->
// packed BGR rows at out_dev (stride width * 3).  work_dev: [coefficients][planes][tables] scratch of work_bytes.
static int jpeg_device_half(dfd_handle* h, const Parsed& P, const ScanLayout& L, const int16_t* coef_host, uint8_t* work_dev,
                            uint8_t* out_dev) {
    const size_t coef_bytes = L.total * 64 * 2;
    size_t plane_off[3], plane_total = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        plane_off[c] = plane_total;
        plane_total += ((size_t)P.comp[c].bw * 8 * P.comp[c].bh * 8 + 255) & ~(size_t)255;
    }
    const size_t qbytes = 3 * 64 * 2;
    const size_t coef_al = (coef_bytes + 255) & ~(size_t)255;
    uint8_t* base = work_dev;
    uint16_t qhost[3 * 64];
    JpegPlanes J{};
    int nb[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c) {
        const int cc = c < P.ncomp ? c : 0;
        J.coef[c] = reinterpret_cast<const int16_t*>(base) + L.comp_off[cc];
        J.plane[c] = base + coef_al + plane_off[cc];
        J.bw[c] = P.comp[cc].bw;
        J.bh[c] = P.comp[cc].bh;
        J.qoff[c] = 64 * c;
        memcpy(qhost + 64 * c, P.q[P.comp[cc].tq], 128);
        if (c < P.ncomp) nb[c] = P.comp[c].bw * P.comp[c].bh;
    }
    uint16_t* qdev = reinterpret_cast<uint16_t*>(base + coef_al + plane_total);
    DFD_HIP_TRY(h, hipMemcpyAsync(base, coef_host, coef_bytes, hipMemcpyHostToDevice, h->stream));
    int rc = mailbox_h2d(h, qdev, qhost, qbytes);                  // copied before the call returns
    if (rc) return rc;
    const int total_blocks = nb[0] + nb[1] + nb[2];
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total_blocks + 63) / 64), dim3(64), 0, h->stream, J, qdev, nb[0], nb[1], nb[2],
                       (size_t)0, (size_t)0);
    const int mode = P.ncomp == 1 ? 0 : (P.hmax == 1 ? 1 : (P.vmax == 1 ? 2 : 3));
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((P.width + 255) / 256, P.height), dim3(256), 0, h->stream, J, mode, P.width,
                       P.height, out_dev, P.width * 3, (size_t)0, (size_t)0);
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}

static size_t jpeg_work_bytes(const Parsed& P, const ScanLayout& L) {
    size_t plane_total = 0;
    for (int c = 0; c < P.ncomp; ++c) plane_total += ((size_t)P.comp[c].bw * 8 * P.comp[c].bh * 8 + 255) & ~(size_t)255;
    return ((L.total * 64 * 2 + 255) & ~(size_t)255) + plane_total + 512;
}

static int jpeg_pinned(dfd_handle* h, size_t bytes) {
    if (bytes <= h->jpeg_host_cap) return DFD_OK;
    DFD_HIP_TRY(h, stream_sync(h));
    if (h->jpeg_host) DFD_HIP_TRY(h, hipHostFree(h->jpeg_host));
    h->jpeg_host = nullptr;
    h->jpeg_host_cap = 0;
    const size_t want = (bytes + (1 << 20) - 1) & ~((size_t)(1 << 20) - 1);
    DFD_HIP_TRY(h, hipHostMalloc(&h->jpeg_host, want, hipHostMallocDefault));
    h->jpeg_host_cap = want;
    return DFD_OK;
}

// JPEG bytes -> packed BGR frame in h->frame_buf (row stride width * 3); *hh / *ww receive the size.
// The coefficients are decoded straight into pinned host memory (one DMA at link speed; from a pageable vector the
// runtime staged 6 MB through its own bounce buffer first).
int jpeg_decode_to_frame(dfd_handle* h, const uint8_t* jpeg, size_t len, int* hh, int* ww) {
    Parsed P;
    int rc = parse_headers(h, jpeg, len, &P);
    if (rc) return rc;
    ScanLayout L;
    scan_layout(&P, &L);
    if ((rc = jpeg_pinned(h, L.total * 64 * 2))) return rc;
    int16_t* coef_host = static_cast<int16_t*>(h->jpeg_host);
    if ((rc = entropy_decode(h, &P, coef_host, L))) return rc;
    if ((rc = ensure(h, &h->jpeg_work, jpeg_work_bytes(P, L)))) return rc;
    if ((rc = ensure(h, &h->frame_buf, (size_t)P.height * P.width * 3))) return rc;
    if ((rc = jpeg_device_half(h, P, L, coef_host, static_cast<uint8_t*>(h->jpeg_work.p), static_cast<uint8_t*>(h->frame_buf.p)))) return rc;
    // the pinned coefficients are overwritten by the next call: the upload must have left them
    DFD_HIP_TRY(h, stream_sync(h));
    *hh = P.height;
    *ww = P.width;
    return DFD_OK;
}

// ---- a batch whose scans are decoded on the device (jpeg_gpu_entropy.h) -------------------------------------------
// Frames of one size AND one layout (components, sampling): n scans -> frames_dev [n][H][W][3] on stream h->stream.
//   raw_dev       the files' scans on the device, scan i at raw_off[i] (multiple of 16), raw_len[i] bytes, at least 16
//                 readable bytes behind each
//   work          device scratch of jpeg_gpu_work_bytes(...)
// The call waits once (after the scan kernel) to read the per-frame verdicts: a frame the device decoder does not vouch
// for is decoded by the host path and its coefficients uploaded before the batched IDCT / colour launches.
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct JpegGpuLayout {
    size_t ds, frames, tabs, blkmap, cblkmap, removed, st, en0, en1, cnt, dcs, gfirst, dcb, err, redone, coef, planes, q, total;
    size_t ds_bytes, nchunks, ndsblk, ncblk, coef_stride, plane_stride;
};

static JpegGpuLayout jpeg_gpu_layout(const std::vector<Parsed>& P, const std::vector<ScanLayout>& L, const uint32_t* raw_len, int n,
                                     int chunk_bytes) {
    JpegGpuLayout W{};
    size_t ds = 0, chunks = 0, dsblk = 0, cblk = 0;
    for (int i = 0; i < n; ++i) {
        const size_t nc = ((size_t)raw_len[i] + chunk_bytes - 1) / chunk_bytes;
        ds += ((nc + JG_CB - 1) / JG_CB * JG_CB + 64) * (size_t)chunk_bytes;      // chunk-interleaved image: whole groups of 64 chunks + one
        cblk += (nc + JG_CB - 1) / JG_CB;
        chunks += (nc + JG_CB - 1) / JG_CB * JG_CB;
        dsblk += ((size_t)raw_len[i] + 15 + JG_DS_BLOCK - 1) / JG_DS_BLOCK;      // (+ up to 15 bytes in front of an unaligned scan)
    }
    W.ds_bytes = ds; W.nchunks = chunks; W.ndsblk = dsblk; W.ncblk = cblk;
    size_t plane_total = 0;
    for (int c = 0; c < P[0].ncomp; ++c) plane_total += al256((size_t)P[0].comp[c].bw * 8 * P[0].comp[c].bh * 8);
    W.coef_stride = al256(L[0].total * 64 * 2) / 2;               // int16 elements
    W.plane_stride = plane_total;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
    W.ds = take(ds + 256);
    W.frames = take(sizeof(JgFrame) * (size_t)n);
    W.tabs = take(sizeof(JgTableSet) * (size_t)n);
    W.blkmap = take(2 * dsblk);
    W.cblkmap = take(2 * cblk);
    W.removed = take(4 * dsblk);
    W.st = take(8 * chunks); W.en0 = take(8 * chunks); W.en1 = take(8 * chunks);
    W.cnt = take(4 * chunks); W.dcs = take(12 * chunks); W.gfirst = take(4 * chunks); W.dcb = take(12 * chunks); W.err = take(chunks);
    W.redone = take(4 * JG_MAX_ROUNDS);
    W.coef = take(W.coef_stride * 2 * (size_t)n);
    W.planes = take(plane_total * (size_t)n);
    W.q = take((size_t)n * 192 * 2);
    W.total = o;
    return W;
}

static bool jpeg_gpu_batch_ok(const dfd_handle* h, const std::vector<Parsed>& P, const std::vector<ScanLayout>& L, int n) {
    static const bool off = getenv("DFD_JPEG_DEVICE_ENTROPY") && atoi(getenv("DFD_JPEG_DEVICE_ENTROPY")) == 0;
    if (off || n < 1 || n > 65535) return false;
    // the device descriptors address scans, the de-stuffed image and coefficients with 32-bit offsets
    size_t scan_total = 0;
    for (int i = 0; i < n; ++i) scan_total += (size_t)(P[i].end - P[i].scan) + 65536 + 64 * (size_t)h->jpeg_chunk_bytes;
    if (scan_total >= ((size_t)1 << 31) || (size_t)n * (L[0].total * 64 + 256) >= ((size_t)1 << 31)) return false;
    for (int i = 0; i < n; ++i) {
        if (!jg_supported(P[i], L[i])) return false;
        if (P[i].ncomp != P[0].ncomp || P[i].hmax != P[0].hmax || P[i].vmax != P[0].vmax || P[i].width != P[0].width ||
            P[i].height != P[0].height)
            return false;
    }
    return true;
}

// stage: pinned host memory for the descriptors going up and the verdicts coming back (>= jpeg_gpu_stage_bytes)
static size_t jpeg_gpu_stage_bytes(const JpegGpuLayout& W, int n) {
    return al256(sizeof(JgFrame) * (size_t)n) * 2 + al256(sizeof(JgTableSet) * (size_t)n) + al256(2 * W.ndsblk) + al256(2 * W.ncblk) +
           al256((size_t)n * 192 * 2) + al256(4 * JG_MAX_ROUNDS);
}

// do two parsed files use the same Huffman tables in every MCU slot? (then the second shares the first's device image)
static bool jpeg_same_tables(const Parsed& A, const ScanLayout& LA, const Parsed& B, const ScanLayout& LB) {
    if (LA.bpm != LB.bpm) return false;
    for (int s = 0; s < LA.bpm; ++s) {
        if (LA.slot_comp[s] != LB.slot_comp[s]) return false;
        const Component &ca = A.comp[LA.slot_comp[s]], &cb = B.comp[LB.slot_comp[s]];
        if (ca.td != cb.td || ca.ta != cb.ta) return false;
        if (memcmp(&A.dc[ca.td], &B.dc[cb.td], sizeof(HuffTable)) || memcmp(&A.ac[ca.ta], &B.ac[cb.ta], sizeof(HuffTable))) return false;
    }
    return true;
}

// One batch on the device decoder, in two halves so that the caller decides where the wait goes:
//   enqueue()  descriptors up, de-stuffing, rounds, scan, emit, IDCT, colour, verdicts down - nothing waits;
//   finish()   AFTER the stream has been waited for: reads the verdicts; frames the device decoder does not vouch for
//              go through the host decoder, their coefficients are uploaded and IDCT / colour run again (then it waits).
struct JpegGpuJob {
    dfd_handle* h = nullptr;
    hipStream_t s = nullptr;                     // the stream everything is queued on (null: the handle's)
    std::vector<Parsed>* P = nullptr;
    const std::vector<ScanLayout>* L = nullptr;
    int n = 0, chunk_bytes = 512, rounds = 16;
    uint8_t* work = nullptr;
    JpegGpuLayout W{};
    char* stage = nullptr;
    uint8_t* frames_dev = nullptr;
    JgFrame *Fh = nullptr, *Fback = nullptr;
    uint32_t* redone_h = nullptr;
    uint16_t* qd = nullptr;
    int16_t* coef = nullptr;
    size_t chunks = 0;
    std::vector<char> on_host;

    int idct_colour() {
        const Parsed& P0 = (*P)[0];
        const ScanLayout& L0 = (*L)[0];
        hipStream_t s = this->s ? this->s : h->stream;
        JpegPlanes J{};
        int nb[3] = {0, 0, 0};
        size_t plane_off = 0;
        for (int c = 0; c < 3; ++c) {
            const int cc = c < P0.ncomp ? c : 0;
            if (c < P0.ncomp) {
                J.plane[c] = work + W.planes + plane_off;
                plane_off += al256((size_t)P0.comp[c].bw * 8 * P0.comp[c].bh * 8);
                nb[c] = P0.comp[c].bw * P0.comp[c].bh;
            } else {
                J.plane[c] = J.plane[0];
            }
            J.coef[c] = coef + L0.comp_off[cc];
            J.bw[c] = P0.comp[cc].bw;
            J.bh[c] = P0.comp[cc].bh;
            J.qoff[c] = 64 * c;
        }
        const int total_blocks = nb[0] + nb[1] + nb[2];
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total_blocks + 63) / 64, n), dim3(64), 0, s, J, qd, nb[0], nb[1], nb[2], W.coef_stride,
                           W.plane_stride);
        const int mode = P0.ncomp == 1 ? 0 : (P0.hmax == 1 ? 1 : (P0.vmax == 1 ? 2 : 3));
        const size_t fstride = (size_t)P0.height * P0.width * 3;
        const bool aligned = (P0.width * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(frames_dev) & 7) == 0;
        if (aligned && mode == 3 && P0.width % 8 == 0)
            hipLaunchKernelGGL(jpeg_color420_kernel, dim3((P0.width / 8 + 255) / 256, (P0.height + 1) / 2, n), dim3(256), 0, s, J, P0.width,
                               P0.height, frames_dev, P0.width * 3, W.plane_stride, fstride);
        else if (aligned)
            hipLaunchKernelGGL(jpeg_color4_kernel, dim3((P0.width + 1023) / 1024, P0.height, n), dim3(256), 0, s, J, mode, P0.width,
                               P0.height, frames_dev, P0.width * 3, W.plane_stride, fstride);
        else
            hipLaunchKernelGGL(jpeg_color_kernel, dim3((P0.width + 255) / 256, P0.height, n), dim3(256), 0, s, J, mode, P0.width,
                               P0.height, frames_dev, P0.width * 3, W.plane_stride, fstride);
        DFD_HIP_TRY(h, hipGetLastError());
        return DFD_OK;
    }

    int enqueue(const uint8_t* raw_dev, const uint32_t* raw_off, const uint32_t* raw_len) {
        hipStream_t s = this->s ? this->s : h->stream;
        const std::vector<Parsed>& PP = *P;
        const std::vector<ScanLayout>& LL = *L;
        Fh = reinterpret_cast<JgFrame*>(stage);
        char* sp = stage + al256(sizeof(JgFrame) * (size_t)n);
        Fback = reinterpret_cast<JgFrame*>(sp);
        sp += al256(sizeof(JgFrame) * (size_t)n);
        JgTableSet* Th = reinterpret_cast<JgTableSet*>(sp);
        sp += al256(sizeof(JgTableSet) * (size_t)n);
        uint16_t* bm = reinterpret_cast<uint16_t*>(sp);
        sp += al256(2 * W.ndsblk);
        uint16_t* cm = reinterpret_cast<uint16_t*>(sp);
        sp += al256(2 * W.ncblk);
        uint16_t* qh = reinterpret_cast<uint16_t*>(sp);
        sp += al256((size_t)n * 192 * 2);
        redone_h = reinterpret_cast<uint32_t*>(sp);
        int nsets = 0;
        size_t ds = 0, dsblk = 0, cblk = 0;
        chunks = 0;
        on_host.assign(n, 0);
        for (int i = 0; i < n; ++i) {
            JgFrame& F = Fh[i];
            memset(&F, 0, sizeof F);
            F.raw_off = raw_off[i];
            F.raw_len = raw_len[i];
            F.ds_off = (uint32_t)ds;
            const size_t nc = ((size_t)raw_len[i] + chunk_bytes - 1) / chunk_bytes;
            ds += ((nc + JG_CB - 1) / JG_CB * JG_CB + 64) * (size_t)chunk_bytes;
            F.chunk0 = (uint32_t)chunks; F.nchunks = (uint32_t)nc; F.cblk0 = (uint32_t)cblk;
            F.dsblk0 = (uint32_t)dsblk;
            F.ndsblk = (uint32_t)(((size_t)raw_len[i] + (raw_off[i] & 15u) + JG_DS_BLOCK - 1) / JG_DS_BLOCK);
            F.coef_off = (uint32_t)((size_t)i * W.coef_stride);
            F.bpm = LL[i].bpm; F.total_blocks = (int32_t)LL[i].total; F.mcux = LL[i].mcux; F.chunk_bytes = chunk_bytes;
            F.cw_shift = 0;
            while ((4 << F.cw_shift) < chunk_bytes) ++F.cw_shift;
            for (int k = 0; k < LL[i].bpm; ++k) {
                F.slot_comp[k] = (uint8_t)LL[i].slot_comp[k]; F.slot_bx[k] = (uint8_t)LL[i].slot_bx[k]; F.slot_by[k] = (uint8_t)LL[i].slot_by[k];
            }
            for (int c = 0; c < PP[i].ncomp; ++c) {
                F.comp_h[c] = PP[i].comp[c].h; F.comp_v[c] = PP[i].comp[c].v; F.comp_bw[c] = LL[i].bw[c]; F.comp_off[c] = (uint32_t)LL[i].comp_off[c];
            }
            F.marker_pos = 0xffffffffu;
            // the device image of the tables is built once per run of frames that share them (a stream's frames usually do)
            if (i > 0 && !on_host[i - 1] && jpeg_same_tables(PP[i], LL[i], PP[i - 1], LL[i - 1])) {
                F.tabset = Fh[i - 1].tabset;
                memcpy(F.slot_dc, Fh[i - 1].slot_dc, 8);
                memcpy(F.slot_ac, Fh[i - 1].slot_ac, 8);
            } else if (jg_build_tables(PP[i], &Th[nsets], F.slot_dc, F.slot_ac, LL[i])) {
                F.tabset = (uint32_t)nsets++;
            } else {
                on_host[i] = 1;                                      // more distinct tables than the device image holds
                F.raw_len = 0; F.nchunks = 0; F.ndsblk = 0;
            }
            for (size_t b = 0; b < (F.nchunks + JG_CB - 1) / JG_CB; ++b) cm[cblk++] = (uint16_t)i;
            chunks += ((size_t)F.nchunks + JG_CB - 1) / JG_CB * JG_CB;
            for (uint32_t b = 0; b < F.ndsblk; ++b) bm[dsblk++] = (uint16_t)i;
            for (int c = 0; c < 3; ++c) memcpy(qh + (size_t)i * 192 + 64 * c, PP[i].q[PP[i].comp[c < PP[i].ncomp ? c : 0].tq], 128);
        }
        JgFrame* Fd = reinterpret_cast<JgFrame*>(work + W.frames);
        JgTableSet* Td = reinterpret_cast<JgTableSet*>(work + W.tabs);
        uint16_t* bmd = reinterpret_cast<uint16_t*>(work + W.blkmap);
        uint16_t* cmd = reinterpret_cast<uint16_t*>(work + W.cblkmap);
        qd = reinterpret_cast<uint16_t*>(work + W.q);
        DFD_HIP_TRY(h, hipMemcpyAsync(Fd, Fh, sizeof(JgFrame) * (size_t)n, hipMemcpyHostToDevice, s));
        if (nsets) DFD_HIP_TRY(h, hipMemcpyAsync(Td, Th, sizeof(JgTableSet) * (size_t)nsets, hipMemcpyHostToDevice, s));
        if (dsblk) DFD_HIP_TRY(h, hipMemcpyAsync(bmd, bm, 2 * dsblk, hipMemcpyHostToDevice, s));
        if (cblk) DFD_HIP_TRY(h, hipMemcpyAsync(cmd, cm, 2 * cblk, hipMemcpyHostToDevice, s));
        DFD_HIP_TRY(h, hipMemcpyAsync(qd, qh, (size_t)n * 192 * 2, hipMemcpyHostToDevice, s));
        coef = reinterpret_cast<int16_t*>(work + W.coef);
        DFD_HIP_TRY(h, hipMemsetAsync(coef, 0, W.coef_stride * 2 * (size_t)n, s));
        JgChunks S;
        S.st = reinterpret_cast<uint2*>(work + W.st);
        S.en[0] = reinterpret_cast<uint2*>(work + W.en0);
        S.en[1] = reinterpret_cast<uint2*>(work + W.en1);
        S.cnt = reinterpret_cast<uint32_t*>(work + W.cnt);
        S.dcs = reinterpret_cast<int32_t*>(work + W.dcs);
        S.gfirst = reinterpret_cast<uint32_t*>(work + W.gfirst);
        S.dcb = reinterpret_cast<int32_t*>(work + W.dcb);
        S.err = work + W.err;
        S.redone = reinterpret_cast<uint32_t*>(work + W.redone);
        DFD_HIP_TRY(h, hipMemsetAsync(S.redone, 0, 4 * JG_MAX_ROUNDS, s));
        rounds = h->jpeg_rounds < 2 ? 2 : (h->jpeg_rounds > JG_MAX_ROUNDS ? JG_MAX_ROUNDS : h->jpeg_rounds);
        uint8_t* dsd = work + W.ds;
        uint32_t* removed = reinterpret_cast<uint32_t*>(work + W.removed);
        if (dsblk) {
            hipLaunchKernelGGL(jg_count_kernel, dim3((unsigned)dsblk), dim3(JG_DS_THREADS), 0, s, raw_dev, Fd, bmd, removed);
            hipLaunchKernelGGL(jg_compact_kernel, dim3((unsigned)dsblk), dim3(JG_DS_THREADS), 0, s, raw_dev, dsd, Fd, bmd, removed);
        }
        if (cblk) {
            for (int r = 0; r < rounds; ++r)
                hipLaunchKernelGGL(jg_round_kernel, dim3((unsigned)cblk), dim3(JG_CB), 0, s, dsd, Fd, Td, cmd, S, r);
            hipLaunchKernelGGL(jg_scan_kernel, dim3(n), dim3(1024), 0, s, Fd, S, rounds - 1);
            hipLaunchKernelGGL(jg_emit_kernel, dim3((unsigned)cblk), dim3(JG_CB), 0, s, dsd, Fd, Td, cmd, S, coef);
        }
        DFD_HIP_TRY(h, hipMemcpyAsync(Fback, Fd, sizeof(JgFrame) * (size_t)n, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, hipMemcpyAsync(redone_h, S.redone, 4 * JG_MAX_ROUNDS, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, hipGetLastError());
        return idct_colour();
    }

    int finish(int* host_decoded_out) {
        std::vector<Parsed>& PP = *P;
        const std::vector<ScanLayout>& LL = *L;
        if (getenv("DFD_JPEG_VERBOSE")) {
            fprintf(stderr, "[dfd] jpeg device entropy: %d frames, %zu chunks of %d bytes, lanes decoding per round:", n, chunks, chunk_bytes);
            for (int r = 0; r < rounds; ++r) fprintf(stderr, " %u", redone_h[r]);
            fprintf(stderr, "\n");
        }
        int host_decoded = 0;
        for (int i = 0; i < n; ++i) {
            if (!on_host[i] && Fback[i].status == JG_OK) continue;
            if (getenv("DFD_JPEG_VERBOSE"))
                fprintf(stderr, "[dfd] jpeg device entropy: frame %d status %d (%u of %d blocks, %u payload bits) -> host decoder\n", i,
                        on_host[i] ? -1 : Fback[i].status, Fback[i].blocks_found, Fh[i].total_blocks, Fback[i].nbits);
            std::vector<int16_t> tmp;
            try {
                tmp.resize(LL[i].total * 64);
            } catch (const std::bad_alloc&) { return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory"); }
            const int rc = entropy_decode(h, &PP[i], tmp.data(), LL[i], true);
            if (rc) return rc;
            DFD_HIP_TRY(h, hipMemcpy(coef + (size_t)i * W.coef_stride, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
            ++host_decoded;
        }
        if (host_decoded_out) *host_decoded_out = host_decoded;
        if (host_decoded) {
            const int rc = idct_colour();
            if (rc) return rc;
            DFD_HIP_TRY(h, hipStreamSynchronize(this->s ? this->s : h->stream));
        }
        return DFD_OK;
    }
};

// n JPEGs of ONE size -> frames [n][H][W][3] at frames_dev (null: only parse, report the size).  The scans are entropy-
// decoded one per pool thread (each sequentially), the device halves are queued frame by frame.
int jpeg_decode_batch_to(dfd_handle* h, const uint8_t* const* jpegs, const size_t* lens, int n, uint8_t* frames_dev, int* hh, int* ww) {
    std::vector<Parsed> P(n);
    std::vector<ScanLayout> L(n);
    int rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = parse_headers(h, jpegs[i], lens[i], &P[i]))) return rc;
        scan_layout(&P[i], &L[i]);
        if (P[i].width != P[0].width || P[i].height != P[0].height)
            return fail(h, DFD_ERR_ARG, "analyze_stream_batch: frame %d is %d x %d, frame 0 is %d x %d - the frames of a batch share one size",
                        i, P[i].width, P[i].height, P[0].width, P[0].height);
    }
    if ((*hh && *hh != P[0].height) || (*ww && *ww != P[0].width))
        return fail(h, DFD_ERR_ARG, "analyze_stream_batch: JPEG frames are %d x %d, raw frames %d x %d", P[0].width, P[0].height, *ww, *hh);
    *hh = P[0].height;
    *ww = P[0].width;
    if ((size_t)n * (size_t)P[0].width * (size_t)P[0].height > kMaxBatchPixels)
        return fail(h, DFD_ERR_UNSUPPORTED, "analyze_stream_batch: %d frames of %d x %d exceed the %zu-pixel budget of one request", n,
                    P[0].width, P[0].height, kMaxBatchPixels);
    if (!frames_dev) return DFD_OK;
    // Which decoder: the device path costs ~1.6 ms whatever the batch holds (three latency-bound passes) + 0.13 ms per MB,
    // the host pool ~0.45 ms + 1.0-1.5 ms per MB (profiles/jpeg_batch_latency_probe.py: 8 x 480p 0.8 vs 1.7 ms, 8 x 1080p
    // 4.1 vs 2.4 ms, 32 x 1080p 10.5 vs 2.9 ms): option "jpeg_device_entropy" = 2 (default) takes the device from 1 MiB of
    // entropy-coded data per call, 1 always, 0 never
    size_t scan_bytes = 0;
    for (int i = 0; i < n; ++i) scan_bytes += (size_t)(P[i].end - P[i].scan);
    const bool want_device = h->jpeg_device_entropy == 1 || (h->jpeg_device_entropy >= 2 && scan_bytes >= ((size_t)1 << 20));
    if (want_device && jpeg_gpu_batch_ok(h, P, L, n)) {
        // the scans go up as bytes (16-byte aligned starts, 16 bytes of slack each) and are decoded there
        std::vector<uint32_t> roff(n), rlen(n);
        size_t up = 0;
        for (int i = 0; i < n; ++i) {
            roff[i] = (uint32_t)up;
            rlen[i] = (uint32_t)(P[i].end - P[i].scan);
            up += al256((size_t)rlen[i] + 16);
        }
        // a pass costs one lane's sequential decode whatever the batch holds: a small batch (few waves anyway) takes half-size
        // chunks - half the latency per pass; a large one keeps the configured size (fewer lanes, less speculation overhead)
        const int chunk_bytes = scan_bytes < ((size_t)8 << 20) && h->jpeg_chunk_bytes > 256 ? h->jpeg_chunk_bytes / 2 : h->jpeg_chunk_bytes;
        const JpegGpuLayout W = jpeg_gpu_layout(P, L, rlen.data(), n, chunk_bytes);
        const size_t stage_bytes = jpeg_gpu_stage_bytes(W, n);
        if ((rc = jpeg_pinned(h, up + stage_bytes))) return rc;
        if ((rc = ensure(h, &h->jpeg_work, al256(up) + W.total))) return rc;
        char* pinned = static_cast<char*>(h->jpeg_host);
        for (int i = 0; i < n; ++i) memcpy(pinned + roff[i], P[i].scan, rlen[i]);
        uint8_t* raw_dev = static_cast<uint8_t*>(h->jpeg_work.p);
        DFD_HIP_TRY(h, hipMemcpyAsync(raw_dev, pinned, up, hipMemcpyHostToDevice, h->stream));
        int on_host = 0;
        JpegGpuJob job;
        job.h = h; job.P = &P; job.L = &L; job.n = n; job.chunk_bytes = chunk_bytes; job.work = raw_dev + al256(up); job.W = W;
        job.stage = pinned + up; job.frames_dev = frames_dev;
        if ((rc = job.enqueue(raw_dev, roff.data(), rlen.data()))) return rc;
        DFD_HIP_TRY(h, stream_sync(h));                              // verdicts are in; the pinned bytes are reused by the next call
        if ((rc = job.finish(&on_host))) return rc;
        h->jpeg_frames_device += (unsigned long long)(n - on_host);
        h->jpeg_frames_host += (unsigned long long)on_host;
        return DFD_OK;
    }
    h->jpeg_frames_host += (unsigned long long)n;
    std::vector<size_t> coff(n + 1, 0), woff(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        coff[i + 1] = coff[i] + ((L[i].total * 64 * 2 + 255) & ~(size_t)255);
        woff[i + 1] = woff[i] + ((jpeg_work_bytes(P[i], L[i]) + 255) & ~(size_t)255);
    }
    if ((rc = jpeg_pinned(h, coff[n]))) return rc;
    if ((rc = ensure(h, &h->jpeg_work, woff[n]))) return rc;
    char* pinned = static_cast<char*>(h->jpeg_host);
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    // one scan per pool thread; a single file (or a tiny batch) still splits its own scan over the pool
    if (n < 3) {
        for (int i = 0; i < n; ++i)
            if ((rc = entropy_decode(h, &P[i], reinterpret_cast<int16_t*>(pinned + coff[i]), L[i], true))) return rc;
    } else {
        HostPool::get().run(n, [&](int i) {
            rcs[i] = entropy_decode(nullptr, &P[i], reinterpret_cast<int16_t*>(pinned + coff[i]), L[i], false);
        });
        for (int i = 0; i < n; ++i)
            if (rcs[i]) return fail(h, rcs[i], "analyze_stream_batch: frame %d: corrupt or truncated JPEG scan", i);
    }
    const size_t frame_bytes = (size_t)P[0].height * P[0].width * 3;
    for (int i = 0; i < n; ++i)
        if ((rc = jpeg_device_half(h, P[i], L[i], reinterpret_cast<const int16_t*>(pinned + coff[i]),
                                   static_cast<uint8_t*>(h->jpeg_work.p) + woff[i], frames_dev + (size_t)i * frame_bytes)))
            return rc;
    DFD_HIP_TRY(h, stream_sync(h));                                  // the pinned coefficients are reused by the next call
    return DFD_OK;
}

}  // namespace dfd

extern "C" {

// Host half only (no GPU): headers + entropy decoding.  info[16] = width, height, components, hmax, vmax,
// then per component blocks_w, blocks_h, table index; qtables_out: 4 x 64 uint16 (natural order);
// coef_out: int16 coefficients (natural order, block-major per component, components concatenated).
int dfd_jpeg_coefficients(const uint8_t* jpeg, size_t len, int* info, uint16_t* qtables_out, int16_t* coef_out,
                          size_t capacity, size_t* count) {
    return dfd_jpeg::coefficients(jpeg, len, info, qtables_out, coef_out, capacity, count);
}

// n JPEGs of one size -> n packed BGR frames (the batch path of dfd_analyze_stream_batch / dfd_analyze_jpegs_host on its
// own: with the default options the scans of restart-less files are entropy-decoded on the device)
int dfd_decode_jpeg_batch(dfd_handle* h, int n, const uint8_t* const* jpegs, const size_t* lens, uint8_t* bgr_out, size_t capacity,
                          int* height, int* width) {
    if (!h) return DFD_ERR_ARG;
    if (n <= 0 || !jpegs || !lens || !height || !width) return fail(h, DFD_ERR_ARG, "decode_jpeg_batch: bad pointer or count");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int hh = 0, ww = 0, rc;
    if ((rc = jpeg_decode_batch_to(h, jpegs, lens, n, nullptr, &hh, &ww))) return rc;
    *height = hh;
    *width = ww;
    const size_t need = (size_t)n * hh * ww * 3;
    if (bgr_out && need > capacity) return fail(h, DFD_ERR_ARG, "decode_jpeg_batch: %zu bytes needed, capacity %zu", need, capacity);
    if ((rc = ensure(h, &h->frame_buf, need))) return rc;            // dfd_frame_ptr: the n frames, packed
    if ((rc = jpeg_decode_batch_to(h, jpegs, lens, n, static_cast<uint8_t*>(h->frame_buf.p), &hh, &ww))) return rc;
    if (bgr_out) {
        DFD_HIP_TRY(h, hipMemcpyAsync(bgr_out, h->frame_buf.p, need, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, stream_sync(h));
    }
    return DFD_OK;
}

// ---- JPEG bytes in (pinned) host memory -> detect + classify (+ forensics): the PCIe-inclusive path with the BYTES of
// the files crossing the link instead of raw frames (dfd_analyze_frames_host moves 6.2 MB per 1080p frame and is bound by
// the upload: 9.2 k frames/s).  n_total files of ONE size and sampling, `batch` at a time: the scans of chunk k + 1 are
// copied to the device on the copy stream while chunk k is entropy-decoded (jpeg_gpu_entropy.h), turned into frames and
// analysed on the compute stream.  Results as dfd_analyze_batch_device.  Files the device decoder cannot take (restart
// intervals, mixed layouts) make the call fail with DFD_ERR_UNSUPPORTED - decode those with dfd_decode_jpeg_batch.
int dfd_analyze_jpegs_host(dfd_handle* h, const uint8_t* const* jpegs, const size_t* lens, int n_total, int batch,
                           const int32_t* forced_xywh, int forced_k, float conf_thr, int max_faces, int apply_clahe, int with_forensics,
                           int32_t* xywh_out, int* n_faces_out, float* logits_out, double* forensic_prob_out, int* height_out,
                           int* width_out) {
    if (!h) return DFD_ERR_ARG;
    if (!jpegs || !lens || n_total <= 0 || batch <= 0 || max_faces <= 0 || !xywh_out || !n_faces_out || !logits_out)
        return fail(h, DFD_ERR_ARG, "analyze_jpegs_host: bad pointer or count");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if (!h->copy_stream) {
        DFD_HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->copy_done[i], hipEventDisableTiming));
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->slot_free[i], hipEventDisableTiming));
        }
    }
    if (!h->aux_stream) {
        // The decode runs on the handle's SECOND compute stream (the one the forensic launch set of a batch call uses):
        // lowest priority - it has a whole analysis period to finish in and takes what the analysis leaves idle.  One side
        // stream, not one per purpose: a process gets a handful of hardware queues, and with main / forensics / copy /
        // decode streams two of them shared a queue - the decode then ran IN LINE with the analysis it was meant to
        // run beside (8.7 k frames/s inside the long bench process against 10.2 k in a process that had made no forensic
        // call).  Forensics of chunk k and the decode of chunk k + 1 share the stream in order; both are fill-in work.
        int least = 0, greatest = 0;
        DFD_HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        DFD_HIP_TRY(h, hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, least));
        DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->aux_go, hipEventDisableTiming));
        DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->aux_done, hipEventDisableTiming));
    }
    if (!h->jpeg_done[0]) {
        h->jpeg_stream = h->aux_stream;
        for (int i = 0; i < 2; ++i) {
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->jpeg_done[i], hipEventDisableTiming));
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->frames_free[i], hipEventDisableTiming));
        }
    }
    // ---- every chunk's headers first (host only, microseconds per file): sizes of all buffers are known before anything runs
    struct Chunk {
        int first = 0, cnt = 0;
        std::vector<Parsed> P;
        std::vector<ScanLayout> L;
        std::vector<uint32_t> roff, rlen;
        size_t up = 0;
        bool one_copy = false;                   // the files lie back to back in host memory: ONE DMA for the chunk
        const uint8_t* span = nullptr;
        size_t span_bytes = 0;
        JpegGpuLayout W{};
    };
    std::vector<Chunk> chunks;
    const int chunk_bytes = h->jpeg_chunk_bytes;
    int hh = 0, ww = 0;
    size_t max_up = 0, max_work = 0, max_stage = 0;
    for (int first = 0; first < n_total; first += batch) {
        chunks.emplace_back();
        Chunk& c = chunks.back();
        c.first = first;
        c.cnt = std::min(batch, n_total - first);
        c.P.resize(c.cnt);
        c.L.resize(c.cnt);
        c.roff.resize(c.cnt);
        c.rlen.resize(c.cnt);
        size_t sum_len = 0;
        bool ascending = true;
        for (int i = 0; i < c.cnt; ++i) {
            if ((rc = parse_headers(h, jpegs[first + i], lens[first + i], &c.P[i]))) return rc;
            scan_layout(&c.P[i], &c.L[i]);
            if (hh == 0) { hh = c.P[i].height; ww = c.P[i].width; }
            if (c.P[i].height != hh || c.P[i].width != ww)
                return fail(h, DFD_ERR_ARG, "analyze_jpegs_host: file %d is %d x %d, file 0 is %d x %d", first + i, c.P[i].width,
                            c.P[i].height, ww, hh);
            c.roff[i] = (uint32_t)c.up;
            c.rlen[i] = (uint32_t)(c.P[i].end - c.P[i].scan);
            c.up += al256((size_t)c.rlen[i] + 16);
            sum_len += lens[first + i];
            if (i > 0 && jpegs[first + i] < jpegs[first + i - 1] + lens[first + i - 1]) ascending = false;
        }
        // files packed into one (pinned) buffer - Handle.pack_jpegs, a receive buffer - go up as ONE transfer: 64 small DMAs per
        // chunk cost the host ~10 us each on the path that feeds the decoder (the de-stuffing kernels take any alignment)
        const size_t span = (size_t)(jpegs[first + c.cnt - 1] + lens[first + c.cnt - 1] - jpegs[first]);
        if (ascending && span <= sum_len + sum_len / 4 + 4096 && span < ((size_t)1 << 31)) {
            c.one_copy = true;
            c.span = jpegs[first];
            c.span_bytes = span;
            for (int i = 0; i < c.cnt; ++i) c.roff[i] = (uint32_t)(c.P[i].scan - c.span);
            c.up = al256(span + 32);
        }
        if (!jpeg_gpu_batch_ok(h, c.P, c.L, c.cnt))
            return fail(h, DFD_ERR_UNSUPPORTED, "analyze_jpegs_host: files %d.. need the host decoder (restart intervals or mixed layouts)", first);
        c.W = jpeg_gpu_layout(c.P, c.L, c.rlen.data(), c.cnt, chunk_bytes);
        max_up = std::max(max_up, c.up);
        max_work = std::max(max_work, c.W.total);
        max_stage = std::max(max_stage, al256(jpeg_gpu_stage_bytes(c.W, c.cnt)));
    }
    if (height_out) *height_out = hh;
    if (width_out) *width_out = ww;
    const size_t frame_bytes = (size_t)hh * ww * 3;
    if ((size_t)batch * (size_t)hh * (size_t)ww > kMaxBatchPixels)
        return fail(h, DFD_ERR_UNSUPPORTED, "analyze_jpegs_host: %d frames of %d x %d per chunk exceed the %zu-pixel budget", batch, ww, hh,
                    kMaxBatchPixels);
    // two of everything a chunk touches: chunk k + 1 is uploaded and DECODED (third stream) while chunk k is analysed - the decode
    // kernels are latency-bound at one or two waves per SIMD and run in the shadow of the analysis
    if ((rc = jpeg_pinned(h, 2 * max_stage))) return rc;
    for (int i = 0; i < 2; ++i) {
        if ((rc = ensure(h, &h->jpeg_raw[i], max_up))) return rc;
        if ((rc = ensure(h, &h->jpeg_work2[i], max_work))) return rc;
        if ((rc = ensure(h, &h->stage[i], (size_t)batch * frame_bytes))) return rc;
    }
    const int nb = (int)chunks.size();
    std::vector<JpegGpuJob> jobs(nb);
    auto start = [&](int k) -> int {                                // chunk k: scans up (copy stream), decode queued (jpeg stream)
        Chunk& c = chunks[k];
        const int slot = k & 1;
        if (k >= 2) DFD_HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->slot_free[slot], 0));      // decode k - 2 has read its scans
        uint8_t* dst = static_cast<uint8_t*>(h->jpeg_raw[slot].p);
        if (c.one_copy) {
            DFD_HIP_TRY(h, hipMemcpyAsync(dst, c.span, c.span_bytes, hipMemcpyHostToDevice, h->copy_stream));
        } else {
            for (int i = 0; i < c.cnt; ++i)
                DFD_HIP_TRY(h, hipMemcpyAsync(dst + c.roff[i], c.P[i].scan, c.rlen[i], hipMemcpyHostToDevice, h->copy_stream));
        }
        DFD_HIP_TRY(h, hipEventRecord(h->copy_done[slot], h->copy_stream));
        DFD_HIP_TRY(h, hipStreamWaitEvent(h->jpeg_stream, h->copy_done[slot], 0));
        if (k >= 2) DFD_HIP_TRY(h, hipStreamWaitEvent(h->jpeg_stream, h->frames_free[slot], 0));    // analysis k - 2 has read its frames
        JpegGpuJob& job = jobs[k];
        job.h = h; job.s = h->jpeg_stream; job.P = &c.P; job.L = &c.L; job.n = c.cnt; job.chunk_bytes = chunk_bytes;
        job.work = static_cast<uint8_t*>(h->jpeg_work2[slot].p); job.W = c.W;
        job.stage = static_cast<char*>(h->jpeg_host) + (size_t)slot * max_stage;
        job.frames_dev = static_cast<uint8_t*>(h->stage[slot].p);
        const int erc = job.enqueue(dst, c.roff.data(), c.rlen.data());   // (no shared scratch: start() also runs on the producer thread)
        if (erc) return erc;
        DFD_HIP_TRY(h, hipEventRecord(h->slot_free[slot], h->jpeg_stream));
        DFD_HIP_TRY(h, hipEventRecord(h->jpeg_done[slot], h->jpeg_stream));
        return DFD_OK;
    };
    struct Drain {                                                   // no exit leaves work behind on the side streams
        dfd_handle* h;
        ~Drain() { hipStreamSynchronize(h->copy_stream); hipStreamSynchronize(h->jpeg_stream); }
    } drain{h};
    // The host side of start(k + 1) - one or a few dozen transfers, the descriptors, ~25 launches - runs on a second thread
    // while this one sits in the (blocking) analysis of chunk k: done in line it kept the main stream idle for that long
    // in front of every chunk.  The producer starts chunk k only after the analysis of chunk k - 2 has been queued (its
    // frames_free event recorded: a stream wait picks up the LAST record of an event, so the record must exist first).
    std::mutex mu;
    std::condition_variable cv;
    int started = 0, analysed = -1, prod_rc = DFD_OK;                // chunks started / chunks whose analysis is queued
    bool stop = false;
    if ((rc = start(0))) return rc;
    auto produce = [&] {                                            // (no exception leaves the thread: it would end the process)
        if (hipSetDevice(h->device) != hipSuccess) {
            { std::lock_guard<std::mutex> lk(mu); prod_rc = DFD_ERR_HIP; }
            cv.notify_all();
            return;
        }
        for (int k = 1; k < nb; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || analysed >= k - 2; });
                if (stop) return;
            }
            int r;
            try {
                r = start(k);
            } catch (...) {
                r = fail(h, DFD_ERR_CAPACITY, "analyze_jpegs_host: out of host memory while starting chunk %d", k);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (r) prod_rc = r; else started = k;
            }
            cv.notify_all();
            if (r) return;
        }
    };
    std::thread producer;
    try {
        producer = std::thread(produce);
    } catch (...) {
        return fail(h, DFD_ERR_STATE, "analyze_jpegs_host: cannot start the producer thread");
    }
    struct Join {
        std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& stop;
        ~Join() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
    } join{producer, mu, cv, stop};
    for (int k = 0; k < nb; ++k) {
        Chunk& c = chunks[k];
        const int slot = k & 1;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return started >= k || prod_rc != DFD_OK; });
            if (prod_rc != DFD_OK) return prod_rc;
        }
        DFD_HIP_TRY(h, hipEventSynchronize(h->jpeg_done[slot]));      // chunk k is decoded: its verdicts are on the host
        int on_host = 0;
        if ((rc = jobs[k].finish(&on_host))) return rc;              // (a frame the device decoder did not vouch for: host decoder, rare)
        h->jpeg_frames_device += (unsigned long long)(c.cnt - on_host);
        h->jpeg_frames_host += (unsigned long long)on_host;
        rc = dfd_analyze_batch_device(h, static_cast<const uint8_t*>(h->stage[slot].p), c.cnt, hh, ww,
                                      forced_xywh ? forced_xywh + (size_t)c.first * forced_k * 4 : nullptr, forced_k, conf_thr, max_faces,
                                      apply_clahe, with_forensics, xywh_out + (size_t)c.first * max_faces * 4, n_faces_out + c.first,
                                      logits_out + (size_t)c.first * max_faces, forensic_prob_out ? forensic_prob_out + c.first : nullptr);
        if (rc) return rc;
        DFD_HIP_TRY(h, hipEventRecord(h->frames_free[slot], h->stream));
        {
            std::lock_guard<std::mutex> lk(mu);
            analysed = k;
        }
        cv.notify_all();
    }
    DFD_HIP_TRY(h, stream_sync(h));
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
        DFD_HIP_TRY(h, stream_sync(h));
    }
    return DFD_OK;
}

}  // extern "C"
