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

using namespace dfd_jpeg;

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
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total_blocks + 63) / 64), dim3(64), 0, h->stream, J, qdev, nb[0], nb[1], nb[2]);
    const int mode = P.ncomp == 1 ? 0 : (P.hmax == 1 ? 1 : (P.vmax == 1 ? 2 : 3));
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((P.width + 255) / 256, P.height), dim3(256), 0, h->stream, J, mode, P.width,
                       P.height, out_dev, P.width * 3);
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
