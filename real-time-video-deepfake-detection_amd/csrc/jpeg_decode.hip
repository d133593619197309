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
        dsblk += ((size_t)raw_len[i] + JG_DS_BLOCK - 1) / JG_DS_BLOCK;
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

static int jpeg_gpu_decode(dfd_handle* h, std::vector<Parsed>& P, const std::vector<ScanLayout>& L, const uint8_t* raw_dev,
                           const uint32_t* raw_off, const uint32_t* raw_len, int n, uint8_t* work, const JpegGpuLayout& W, char* stage,
                           uint8_t* frames_dev, int chunk_bytes, int* host_decoded_out) {
    hipStream_t s = h->stream;
    // ---- descriptors
    JgFrame* Fh = reinterpret_cast<JgFrame*>(stage);
    char* sp = stage + al256(sizeof(JgFrame) * (size_t)n);
    JgFrame* Fback = reinterpret_cast<JgFrame*>(sp);
    sp += al256(sizeof(JgFrame) * (size_t)n);
    JgTableSet* Th = reinterpret_cast<JgTableSet*>(sp);
    sp += al256(sizeof(JgTableSet) * (size_t)n);
    uint16_t* bm = reinterpret_cast<uint16_t*>(sp);
    sp += al256(2 * W.ndsblk);
    uint16_t* cm = reinterpret_cast<uint16_t*>(sp);
    sp += al256(2 * W.ncblk);
    uint16_t* qh = reinterpret_cast<uint16_t*>(sp);
    sp += al256((size_t)n * 192 * 2);
    uint32_t* redone_h = reinterpret_cast<uint32_t*>(sp);
    int nsets = 0;
    size_t ds = 0, chunks = 0, dsblk = 0, cblk = 0;
    std::vector<char> on_host(n, 0);
    for (int i = 0; i < n; ++i) {
        JgFrame& F = Fh[i];
        memset(&F, 0, sizeof F);
        F.raw_off = raw_off[i];
        F.raw_len = raw_len[i];
        F.ds_off = (uint32_t)ds;
        const size_t nc = ((size_t)raw_len[i] + chunk_bytes - 1) / chunk_bytes;
        ds += ((nc + JG_CB - 1) / JG_CB * JG_CB + 64) * (size_t)chunk_bytes;
        F.chunk0 = (uint32_t)chunks; F.nchunks = (uint32_t)nc; F.cblk0 = (uint32_t)cblk;
        for (size_t b = 0; b < (nc + JG_CB - 1) / JG_CB; ++b) cm[cblk++] = (uint16_t)i;
        chunks += (nc + JG_CB - 1) / JG_CB * JG_CB;
        F.dsblk0 = (uint32_t)dsblk;
        F.ndsblk = (uint32_t)(((size_t)raw_len[i] + JG_DS_BLOCK - 1) / JG_DS_BLOCK);
        for (uint32_t b = 0; b < F.ndsblk; ++b) bm[dsblk++] = (uint16_t)i;
        F.coef_off = (uint32_t)((size_t)i * W.coef_stride);
        F.bpm = L[i].bpm; F.total_blocks = (int32_t)L[i].total; F.mcux = L[i].mcux; F.chunk_bytes = chunk_bytes;
        F.cw_shift = 0;
        while ((4 << F.cw_shift) < chunk_bytes) ++F.cw_shift;
        for (int k = 0; k < L[i].bpm; ++k) {
            F.slot_comp[k] = (uint8_t)L[i].slot_comp[k]; F.slot_bx[k] = (uint8_t)L[i].slot_bx[k]; F.slot_by[k] = (uint8_t)L[i].slot_by[k];
        }
        for (int c = 0; c < P[i].ncomp; ++c) {
            F.comp_h[c] = P[i].comp[c].h; F.comp_v[c] = P[i].comp[c].v; F.comp_bw[c] = L[i].bw[c]; F.comp_off[c] = (uint32_t)L[i].comp_off[c];
        }
        F.marker_pos = 0xffffffffu;
        JgTableSet* ts = &Th[nsets];
        if (!jg_build_tables(P[i], ts, F.slot_dc, F.slot_ac, L[i])) { on_host[i] = 1; F.raw_len = 0; F.nchunks = 0; F.ndsblk = 0; }
        else if (nsets > 0 && memcmp(ts, &Th[nsets - 1], sizeof *ts) == 0) F.tabset = (uint32_t)(nsets - 1);
        else F.tabset = (uint32_t)nsets++;
        for (int c = 0; c < 3; ++c) memcpy(qh + (size_t)i * 192 + 64 * c, P[i].q[P[i].comp[c < P[i].ncomp ? c : 0].tq], 128);
    }
    JgFrame* Fd = reinterpret_cast<JgFrame*>(work + W.frames);
    JgTableSet* Td = reinterpret_cast<JgTableSet*>(work + W.tabs);
    uint16_t* bmd = reinterpret_cast<uint16_t*>(work + W.blkmap);
    uint16_t* cmd = reinterpret_cast<uint16_t*>(work + W.cblkmap);
    uint16_t* qd = reinterpret_cast<uint16_t*>(work + W.q);
    DFD_HIP_TRY(h, hipMemcpyAsync(Fd, Fh, sizeof(JgFrame) * (size_t)n, hipMemcpyHostToDevice, s));
    if (nsets) DFD_HIP_TRY(h, hipMemcpyAsync(Td, Th, sizeof(JgTableSet) * (size_t)nsets, hipMemcpyHostToDevice, s));
    if (dsblk) DFD_HIP_TRY(h, hipMemcpyAsync(bmd, bm, 2 * dsblk, hipMemcpyHostToDevice, s));
    if (cblk) DFD_HIP_TRY(h, hipMemcpyAsync(cmd, cm, 2 * cblk, hipMemcpyHostToDevice, s));
    DFD_HIP_TRY(h, hipMemcpyAsync(qd, qh, (size_t)n * 192 * 2, hipMemcpyHostToDevice, s));
    int16_t* coef = reinterpret_cast<int16_t*>(work + W.coef);
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
    const int rounds = h->jpeg_rounds < 2 ? 2 : (h->jpeg_rounds > JG_MAX_ROUNDS ? JG_MAX_ROUNDS : h->jpeg_rounds);
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
    DFD_HIP_TRY(h, stream_sync(h));
    if (getenv("DFD_JPEG_VERBOSE")) {
        fprintf(stderr, "[dfd] jpeg device entropy: %d frames, %zu chunks of %d bytes, lanes decoding per round:", n, chunks, chunk_bytes);
        for (int r = 0; r < rounds; ++r) fprintf(stderr, " %u", redone_h[r]);
        fprintf(stderr, "\n");
    }
    // ---- frames the device decoder does not vouch for: the host decoder says what they are
    int host_decoded = 0;
    for (int i = 0; i < n; ++i) {
        if (!on_host[i] && Fback[i].status == JG_OK) continue;
        if (getenv("DFD_JPEG_VERBOSE"))
            fprintf(stderr, "[dfd] jpeg device entropy: frame %d status %d (%u of %d blocks, %u payload bits) -> host decoder\n", i,
                    on_host[i] ? -1 : Fback[i].status, Fback[i].blocks_found, Fh[i].total_blocks, Fback[i].nbits);
        std::vector<int16_t> tmp;
        try {
            tmp.resize(L[i].total * 64);
        } catch (const std::bad_alloc&) { return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory"); }
        const int rc = entropy_decode(h, &P[i], tmp.data(), L[i], true);
        if (rc) return rc;
        DFD_HIP_TRY(h, hipMemcpy(coef + (size_t)i * W.coef_stride, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
        ++host_decoded;
    }
    if (host_decoded_out) *host_decoded_out = host_decoded;
    // ---- IDCT + upsampling + colour of all frames: two launches
    JpegPlanes J{};
    int nb[3] = {0, 0, 0};
    size_t plane_off = 0;
    for (int c = 0; c < 3; ++c) {
        const int cc = c < P[0].ncomp ? c : 0;
        if (c < P[0].ncomp) {
            J.plane[c] = work + W.planes + plane_off;
            plane_off += al256((size_t)P[0].comp[c].bw * 8 * P[0].comp[c].bh * 8);
            nb[c] = P[0].comp[c].bw * P[0].comp[c].bh;
        } else {
            J.plane[c] = J.plane[0];
        }
        J.coef[c] = coef + L[0].comp_off[cc];
        J.bw[c] = P[0].comp[cc].bw;
        J.bh[c] = P[0].comp[cc].bh;
        J.qoff[c] = 64 * c;
    }
    const int total_blocks = nb[0] + nb[1] + nb[2];
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total_blocks + 63) / 64, n), dim3(64), 0, s, J, qd, nb[0], nb[1], nb[2], W.coef_stride,
                       W.plane_stride);
    const int mode = P[0].ncomp == 1 ? 0 : (P[0].hmax == 1 ? 1 : (P[0].vmax == 1 ? 2 : 3));
    if ((P[0].width * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(frames_dev) & 3) == 0)
        hipLaunchKernelGGL(jpeg_color4_kernel, dim3((P[0].width + 1023) / 1024, P[0].height, n), dim3(256), 0, s, J, mode, P[0].width,
                           P[0].height, frames_dev, P[0].width * 3, W.plane_stride, (size_t)P[0].height * P[0].width * 3);
    else
        hipLaunchKernelGGL(jpeg_color_kernel, dim3((P[0].width + 255) / 256, P[0].height, n), dim3(256), 0, s, J, mode, P[0].width,
                           P[0].height, frames_dev, P[0].width * 3, W.plane_stride, (size_t)P[0].height * P[0].width * 3);
    DFD_HIP_TRY(h, hipGetLastError());
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
    if (h->jpeg_device_entropy && n >= h->jpeg_device_entropy && jpeg_gpu_batch_ok(h, P, L, n)) {
        // the scans go up as bytes (16-byte aligned starts, 16 bytes of slack each) and are decoded there
        std::vector<uint32_t> roff(n), rlen(n);
        size_t up = 0;
        for (int i = 0; i < n; ++i) {
            roff[i] = (uint32_t)up;
            rlen[i] = (uint32_t)(P[i].end - P[i].scan);
            up += al256((size_t)rlen[i] + 16);
        }
        const int chunk_bytes = h->jpeg_chunk_bytes;
        const JpegGpuLayout W = jpeg_gpu_layout(P, L, rlen.data(), n, chunk_bytes);
        const size_t stage_bytes = jpeg_gpu_stage_bytes(W, n);
        if ((rc = jpeg_pinned(h, up + stage_bytes))) return rc;
        if ((rc = ensure(h, &h->jpeg_work, al256(up) + W.total))) return rc;
        char* pinned = static_cast<char*>(h->jpeg_host);
        for (int i = 0; i < n; ++i) memcpy(pinned + roff[i], P[i].scan, rlen[i]);
        uint8_t* raw_dev = static_cast<uint8_t*>(h->jpeg_work.p);
        DFD_HIP_TRY(h, hipMemcpyAsync(raw_dev, pinned, up, hipMemcpyHostToDevice, h->stream));
        int on_host = 0;
        if ((rc = jpeg_gpu_decode(h, P, L, raw_dev, roff.data(), rlen.data(), n, raw_dev + al256(up), W, pinned + up, frames_dev, chunk_bytes,
                                  &on_host)))
            return rc;
        h->jpeg_frames_device += (unsigned long long)(n - on_host);
        h->jpeg_frames_host += (unsigned long long)on_host;
        DFD_HIP_TRY(h, stream_sync(h));                              // the pinned bytes are reused by the next call
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
    struct Chunk {
        int first, cnt;
        std::vector<Parsed> P;
        std::vector<ScanLayout> L;
        std::vector<uint32_t> roff, rlen;
        size_t up = 0;
    };
    std::vector<Chunk> chunks;
    for (int first = 0; first < n_total; first += batch) {
        Chunk c;
        c.first = first;
        c.cnt = std::min(batch, n_total - first);
        chunks.push_back(std::move(c));
    }
    const int chunk_bytes = h->jpeg_chunk_bytes;
    int hh = 0, ww = 0;
    auto prepare = [&](Chunk& c) -> int {                           // headers + layout of a chunk (host only)
        c.P.resize(c.cnt);
        c.L.resize(c.cnt);
        c.roff.resize(c.cnt);
        c.rlen.resize(c.cnt);
        for (int i = 0; i < c.cnt; ++i) {
            if ((rc = parse_headers(h, jpegs[c.first + i], lens[c.first + i], &c.P[i]))) return rc;
            scan_layout(&c.P[i], &c.L[i]);
            if (hh == 0) { hh = c.P[i].height; ww = c.P[i].width; }
            if (c.P[i].height != hh || c.P[i].width != ww)
                return fail(h, DFD_ERR_ARG, "analyze_jpegs_host: file %d is %d x %d, file 0 is %d x %d", c.first + i, c.P[i].width,
                            c.P[i].height, ww, hh);
            c.roff[i] = (uint32_t)c.up;
            c.rlen[i] = (uint32_t)(c.P[i].end - c.P[i].scan);
            c.up += al256((size_t)c.rlen[i] + 16);
        }
        if (!jpeg_gpu_batch_ok(h, c.P, c.L, c.cnt))
            return fail(h, DFD_ERR_UNSUPPORTED, "analyze_jpegs_host: files %d.. need the host decoder (restart intervals or mixed layouts)", c.first);
        return DFD_OK;
    };
    auto upload = [&](int k) -> int {                               // chunk k's scans -> raw slot k & 1, on the copy stream
        Chunk& c = chunks[k];
        const int slot = k & 1;
        if ((rc = prepare(c))) return rc;
        if (k >= 2) DFD_HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->slot_free[slot], 0));
        if (c.up > h->jpeg_raw[slot].cap) {
            // (growing a slot another chunk may still read would be a race: chunk k - 2 has finished only on the device)
            if (k >= 2) DFD_HIP_TRY(h, hipEventSynchronize(h->slot_free[slot]));
            if ((rc = ensure(h, &h->jpeg_raw[slot], c.up + (c.up >> 2)))) return rc;
        }
        uint8_t* dst = static_cast<uint8_t*>(h->jpeg_raw[slot].p);
        for (int i = 0; i < c.cnt; ++i)
            DFD_HIP_TRY(h, hipMemcpyAsync(dst + c.roff[i], c.P[i].scan, c.rlen[i], hipMemcpyHostToDevice, h->copy_stream));
        DFD_HIP_TRY(h, hipEventRecord(h->copy_done[slot], h->copy_stream));
        return DFD_OK;
    };
    const int nb = (int)chunks.size();
    if ((rc = upload(0))) return rc;
    if (height_out) *height_out = hh;
    if (width_out) *width_out = ww;
    const size_t frame_bytes = (size_t)hh * ww * 3;
    if ((size_t)batch * (size_t)hh * (size_t)ww > kMaxBatchPixels)
        return fail(h, DFD_ERR_UNSUPPORTED, "analyze_jpegs_host: %d frames of %d x %d per chunk exceed the %zu-pixel budget", batch, ww, hh,
                    kMaxBatchPixels);
    if ((rc = ensure(h, &h->stage[0], (size_t)batch * frame_bytes))) return rc;
    for (int k = 0; k < nb; ++k) {
        Chunk& c = chunks[k];
        const int slot = k & 1;
        if (k + 1 < nb && (rc = upload(k + 1))) return rc;          // in flight while chunk k is decoded and analysed
        const JpegGpuLayout W = jpeg_gpu_layout(c.P, c.L, c.rlen.data(), c.cnt, chunk_bytes);
        if ((rc = jpeg_pinned(h, jpeg_gpu_stage_bytes(W, c.cnt)))) return rc;
        if ((rc = ensure(h, &h->jpeg_work, W.total))) return rc;
        DFD_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->copy_done[slot], 0));
        int on_host = 0;
        if ((rc = jpeg_gpu_decode(h, c.P, c.L, static_cast<const uint8_t*>(h->jpeg_raw[slot].p), c.roff.data(), c.rlen.data(), c.cnt,
                                  static_cast<uint8_t*>(h->jpeg_work.p), W, static_cast<char*>(h->jpeg_host),
                                  static_cast<uint8_t*>(h->stage[0].p), chunk_bytes, &on_host)))
            return rc;
        DFD_HIP_TRY(h, hipEventRecord(h->slot_free[slot], h->stream));   // the scans have been read (the decoder waited for its verdicts)
        h->jpeg_frames_device += (unsigned long long)(c.cnt - on_host);
        h->jpeg_frames_host += (unsigned long long)on_host;
        rc = dfd_analyze_batch_device(h, static_cast<const uint8_t*>(h->stage[0].p), c.cnt, hh, ww,
                                      forced_xywh ? forced_xywh + (size_t)c.first * forced_k * 4 : nullptr, forced_k, conf_thr, max_faces,
                                      apply_clahe, with_forensics, xywh_out + (size_t)c.first * max_faces * 4, n_faces_out + c.first,
                                      logits_out + (size_t)c.first * max_faces, forensic_prob_out ? forensic_prob_out + c.first : nullptr);
        if (rc) return rc;
        c.P.clear(); c.L.clear();
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
