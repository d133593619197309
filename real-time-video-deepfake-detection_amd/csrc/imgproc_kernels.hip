// 8-bit image kernels of the per-face pre-processing path (gfx950).  HBM-bound byte work:
// no MFMA, coalesced u8 rows, LDS only for the per-tile CLAHE histogram.
//
//   resize_linear_u8    cv2.resize(INTER_LINEAR) fixed-point bilinear    frame_analysis.py:71, face_detection.py:77
//   clahe_hist_kernel   BGR->Lab + per-tile histogram/clip/LUT            deepfake_detection.py:363-366
//   clahe_apply_kernel  4-LUT bilinear blend + Lab->BGR                   deepfake_detection.py:366-368
//   crop_norm_kernel    bilinear 224x224 + /255 + ImageNet normalise      deepfake_detection.py:382-389
//
// Compiled with -ffp-contract=off: the float expressions below are written in the exact
// operation order of the algorithms they restate and must not be fused.
#include "imgproc_kernels.h"

namespace dfd {

__device__ __forceinline__ int descale(long long x, int n) { return (int)((x + (1ll << (n - 1))) >> n); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------ resize
struct Tap { int i0, i1, a0, a1; };

// x direction: coefficient forced to 0 at the borders; y direction: indices clipped (OpenCV)
__device__ __forceinline__ Tap linear_tap(int d, int src_n, int dst_n, bool is_x) {
    const double scale = (double)src_n / (double)dst_n;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    if (is_x) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src_n - 1) { f = 0.f; s = src_n - 1; }
        t.i0 = s;
        t.i1 = s + 1 < src_n ? s + 1 : src_n - 1;
    } else {
        t.i0 = clampi(s, 0, src_n - 1);
        t.i1 = clampi(s + 1, 0, src_n - 1);
    }
    t.a0 = clampi((int)rintf((1.f - f) * 2048.f), -32768, 32767);
    t.a1 = clampi((int)rintf(f * 2048.f), -32768, 32767);
    return t;
}

template <int C>
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sh,
                                                               int sw, size_t sstride, size_t simg,
                                                               uint8_t* __restrict__ dst, int dh, int dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dh * dw) return;
    const int dy = idx / dw, dx = idx % dw;
    const uint8_t* s = src + (size_t)blockIdx.y * simg;
    uint8_t* d = dst + ((size_t)blockIdx.y * dh * dw + idx) * C;
    if (sh == dh && sw == dw) {
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] = s[(size_t)dy * sstride + dx * C + c];
        return;
    }
    const Tap tx = linear_tap(dx, sw, dw, true), ty = linear_tap(dy, sh, dh, false);
    const uint8_t* r0 = s + (size_t)ty.i0 * sstride;
    const uint8_t* r1 = s + (size_t)ty.i1 * sstride;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int h0 = r0[tx.i0 * C + c] * tx.a0 + r0[tx.i1 * C + c] * tx.a1;
        const int h1 = r1[tx.i0 * C + c] * tx.a0 + r1[tx.i1 * C + c] * tx.a1;
        const int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)clampi(v, 0, 255);
    }
}

void launch_resize_bgr(const uint8_t* src, int n, int sh, int sw, size_t sstride, size_t simg,
                       uint8_t* dst, int dh, int dw, hipStream_t s) {
    hipLaunchKernelGGL(resize_linear_u8_kernel<3>, dim3((dh * dw + 255) / 256, n), dim3(256), 0, s, src, sh,
                       sw, sstride, simg, dst, dh, dw);
}

void launch_resize_gray(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, hipStream_t s) {
    hipLaunchKernelGGL(resize_linear_u8_kernel<1>, dim3((dh * dw + 255) / 256, 1), dim3(256), 0, s, src, sh, sw,
                       (size_t)sw, 0, dst, dh, dw);
}

__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int n,
                                                          float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i] * scale;
}

void u8_to_float(const uint8_t* src, float* dst, int n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(u8_to_float_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n, scale);
}

// ------------------------------------------------------------------------------ Lab
// ------------------------------------------------------------------------------ CLAHE
// One block per (tile, crop).  The tile grid is 8x8 over the crop extended by reflect-101 to
// a multiple of 8 (OpenCV pads BOTH dimensions by a full 8-rem when either is ragged).
__device__ __forceinline__ void clahe_geometry(const CropDesc& cd, int& ew, int& eh) {
    if (cd.w % 8 == 0 && cd.h % 8 == 0) { ew = cd.w; eh = cd.h; }
    else { ew = cd.w + (8 - cd.w % 8); eh = cd.h + (8 - cd.h % 8); }
}

// Round 3: the first version spent its time in the texture-address unit, not in HBM - per pixel 3 byte loads, 6 table
// gathers from global memory and 3 byte stores, each a full 64-address instruction (206 us per 256 crops of ~150k px =
// 0.56 TB/s).  Now: gamma + cbrt tables in LDS (6.5 KB as u16), four pixels per thread = one 12-byte load and one
// 12-byte store (the conversions in 32-bit integers: every intermediate stays below 2^31, see the table ranges in
// luts.py), one sub-histogram per wave.  Pixels of the reflected border (the last tile column / row of a ragged crop)
// take the one-pixel path.
struct LabLds { unsigned short gamma[256]; unsigned short cbrt[3072]; };

__device__ __forceinline__ void lab_tables_to_lds(const ColorTables& T, LabLds& S, int tid) {
    S.gamma[tid] = (unsigned short)T.gamma[tid];
    for (int i = tid; i < 3072; i += 256) S.cbrt[i] = (unsigned short)T.cbrt[i];
}

__device__ __forceinline__ void bgr2lab_lds(const ColorTables& T, const LabLds& S, int b, int g, int r, int& L, int& A, int& B) {
    const int R_ = S.gamma[r], G_ = S.gamma[g], B_ = S.gamma[b];
    const int fX = S.cbrt[(R_ * T.fwd[0] + G_ * T.fwd[1] + B_ * T.fwd[2] + (1 << 11)) >> 12];
    const int fY = S.cbrt[(R_ * T.fwd[3] + G_ * T.fwd[4] + B_ * T.fwd[5] + (1 << 11)) >> 12];
    const int fZ = S.cbrt[(R_ * T.fwd[6] + G_ * T.fwd[7] + B_ * T.fwd[8] + (1 << 11)) >> 12];
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    L = clampi((Lscale * fY + Lshift + (1 << 14)) >> 15, 0, 255);
    A = clampi((500 * (fX - fY) + 128 * (1 << 15) + (1 << 14)) >> 15, 0, 255);
    B = clampi((200 * (fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15, 0, 255);
}

__global__ __launch_bounds__(256) void clahe_hist_kernel(const uint8_t* __restrict__ frame, size_t fstride,
                                                         const CropDesc* __restrict__ crops,
                                                         uint8_t* __restrict__ lab_out,
                                                         uint8_t* __restrict__ luts, ColorTables T,
                                                         float clip_limit) {
    __shared__ LabLds S;
    __shared__ int hist[4][256];
    __shared__ int scan[256];
    __shared__ int clipped_total;
    const int tid = threadIdx.x;
    const CropDesc cd = crops[blockIdx.y];
    int ew, eh;
    clahe_geometry(cd, ew, eh);
    const int tw = ew / 8, th = eh / 8;
    const int tx = blockIdx.x % 8, ty = blockIdx.x / 8;
    lab_tables_to_lds(T, S, tid);
    for (int w = 0; w < 4; ++w) hist[w][tid] = 0;
    if (tid == 0) clipped_total = 0;
    __syncthreads();
    uint8_t* lab = lab_out + cd.offset;
    int* myhist = hist[tid >> 6];
    const uint8_t* src = frame + cd.src_off + (size_t)cd.y * fstride + (size_t)cd.x * 3;
    const unsigned gpr = (unsigned)(tw + 3) / 4, ngroups = gpr * (unsigned)th;
    for (unsigned gi = tid; gi < ngroups; gi += 256) {
        const unsigned row = gi / gpr, gx = gi - row * gpr;
        const int ex0 = tx * tw + (int)gx * 4, ey = ty * th + (int)row;
        const int cnt = tw - (int)gx * 4 < 4 ? tw - (int)gx * 4 : 4;
        if (cnt == 4 && ex0 + 3 < cd.w && ey < cd.h) {
            unsigned char px[12], res[12];
            __builtin_memcpy(px, src + (size_t)ey * fstride + (size_t)ex0 * 3, 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int L, A, B;
                bgr2lab_lds(T, S, px[3 * k], px[3 * k + 1], px[3 * k + 2], L, A, B);
                atomicAdd(&myhist[L], 1);
                res[3 * k] = (unsigned char)L; res[3 * k + 1] = (unsigned char)A; res[3 * k + 2] = (unsigned char)B;
            }
            __builtin_memcpy(lab + ((size_t)ey * cd.w + ex0) * 3, res, 12);
        } else {
            for (int k = 0; k < cnt; ++k) {
                const int ex = ex0 + k;
                const int sx = reflect101(ex, cd.w), sy = reflect101(ey, cd.h);
                const uint8_t* p = src + (size_t)sy * fstride + (size_t)sx * 3;
                int L, A, B;
                bgr2lab_lds(T, S, p[0], p[1], p[2], L, A, B);
                atomicAdd(&myhist[L], 1);
                if (ex < cd.w && ey < cd.h) {
                    uint8_t* o = lab + ((size_t)ey * cd.w + ex) * 3;
                    o[0] = (uint8_t)L; o[1] = (uint8_t)A; o[2] = (uint8_t)B;
                }
            }
        }
    }
    __syncthreads();
    const int area = tw * th;
    int clip = (int)((double)clip_limit * area / 256);
    if (clip < 1) clip = 1;
    int hv = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
    if (hv > clip) { atomicAdd(&clipped_total, hv - clip); hv = clip; }
    __syncthreads();
    const int clipped = clipped_total;
    const int batch = clipped / 256;
    int residual = clipped - batch * 256;
    hv += batch;
    if (residual != 0) {
        int step = 256 / residual;
        if (step < 1) step = 1;
        if (tid % step == 0 && tid / step < residual) hv += 1;
    }
    // inclusive scan of the 256 bins
    scan[tid] = hv;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const float lut_scale = 255.0f / (float)area;
    const float v = rintf((float)scan[tid] * lut_scale);
    luts[((size_t)blockIdx.y * 64 + blockIdx.x) * 256 + tid] = (uint8_t)clampi((int)v, 0, 255);
}

// Lab -> BGR with the two linear offset tables and abToXZ_b evaluated instead of gathered (the same integers: the
// tables are defined by these formulas, luts.py / oracle lab_tables; C division truncates toward zero like the
// Python restatement): a_div / b_div are L1 hits but abToXZ_b is 147 KB - two L2 gathers per pixel.
__device__ __forceinline__ int ab_to_xz(int v) {
    const int lin = (v * 108) / 841 - 290;                                   // BASE*16/116*108/841 = 290
    const int cube = (int)((((unsigned)(v * v) >> 14) * (unsigned)v) >> 14);  // v > 3390 here: all factors positive
    return v <= 3390 ? lin : cube;
}

// Round 3: a block owns a contiguous run of a crop's pixels and keeps what it gathers from in LDS - the crop's 64 tile
// LUTs (16 KB), sRGBInvGammaTab_b as bytes (4 KB), LabToYF_b (2 KB): the 9 global gathers per pixel of the first version
// (222 us per 256 crops, address-unit bound) become LDS reads; the matrix in 32-bit integers (|terms| < 1.2e9, luts.py).
struct ApplyLds { unsigned char lut[64 * 256]; unsigned char inv_gamma[4096]; int L_fy[256]; int L_y[256]; };

__device__ __forceinline__ void lab2bgr_lds(const ColorTables& T, const ApplyLds& S, int L, int A, int B, int& b, int& g, int& r) {
    const int TAB = 1 << 12;
    const int fy = S.L_fy[L], y = S.L_y[L];
    const int adiv = ((5 * A * 53687 + 128) >> 13) - 4194, bdiv = ((B * 41943 + 16) >> 9) - 10485 + 1;
    const int x = ab_to_xz(fy + adiv), z = ab_to_xz(fy - bdiv);
    const int i0 = (int)T.inv[0], i1 = (int)T.inv[1], i2 = (int)T.inv[2], i3 = (int)T.inv[3], i4 = (int)T.inv[4],
              i5 = (int)T.inv[5], i6 = (int)T.inv[6], i7 = (int)T.inv[7], i8 = (int)T.inv[8];
    r = S.inv_gamma[clampi((i0 * x + i1 * y + i2 * z + (1 << 13)) >> 14, 0, TAB - 1)];
    g = S.inv_gamma[clampi((i3 * x + i4 * y + i5 * z + (1 << 13)) >> 14, 0, TAB - 1)];
    b = S.inv_gamma[clampi((i6 * x + i7 * y + i8 * z + (1 << 13)) >> 14, 0, TAB - 1)];
}

// four pixels (12 bytes of the packed crop: three aligned dwords in, three out) per thread and iteration
__global__ __launch_bounds__(256) void clahe_apply_kernel(const CropDesc* __restrict__ crops,
                                                          const uint8_t* __restrict__ lab_in,
                                                          const uint8_t* __restrict__ luts,
                                                          uint8_t* __restrict__ bgr_out, ColorTables T) {
    __shared__ ApplyLds S;
    const CropDesc cd = crops[blockIdx.y];
    int ew, eh;
    clahe_geometry(cd, ew, eh);
    const int npix = cd.w * cd.h, ngroups = (npix + 3) / 4;
    const int per_block = (ngroups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g_begin = (int)blockIdx.x * per_block, g_end = g_begin + per_block < ngroups ? g_begin + per_block : ngroups;
    if (g_begin >= g_end) return;                                           // uniform per block
    {
        const uint4* src = reinterpret_cast<const uint4*>(luts + (size_t)blockIdx.y * 64 * 256);
        uint4* dst = reinterpret_cast<uint4*>(S.lut);
        for (int i = threadIdx.x; i < 64 * 256 / 16; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < 4096; i += 256) S.inv_gamma[i] = (unsigned char)T.inv_gamma[i];
        S.L_fy[threadIdx.x] = T.L_fy[threadIdx.x];
        S.L_y[threadIdx.x] = T.L_y[threadIdx.x];
    }
    __syncthreads();
    const float inv_tw = 1.0f / (float)(ew / 8), inv_th = 1.0f / (float)(eh / 8);
    const uint8_t* lab = lab_in + cd.offset;
    uint8_t* out = bgr_out + cd.offset;
    for (int gi = g_begin + (int)threadIdx.x; gi < g_end; gi += 256) {
        const int i0 = gi * 4;
        const int cnt = npix - i0 < 4 ? npix - i0 : 4;                     // the crop's last group may be short
        unsigned char px[12], res8[12];
        if (cnt == 4) *reinterpret_cast<uint3*>(px) = *reinterpret_cast<const uint3*>(lab + (size_t)i0 * 3);
        else
            for (int k = 0; k < 12; ++k) px[k] = k < cnt * 3 ? lab[(size_t)i0 * 3 + k] : 0;
        int y = (int)((unsigned)i0 / (unsigned)cd.w), x = i0 - y * cd.w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
            int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
            const float xa = txf - (float)tx1, ya = tyf - (float)ty1;
            const float xa1 = 1.0f - xa, ya1 = 1.0f - ya;
            int tx2 = tx1 + 1 < 7 ? tx1 + 1 : 7, ty2 = ty1 + 1 < 7 ? ty1 + 1 : 7;
            tx1 = tx1 > 0 ? tx1 : 0;
            ty1 = ty1 > 0 ? ty1 : 0;
            const int L = px[3 * k];
            const float l11 = S.lut[(ty1 * 8 + tx1) * 256 + L], l12 = S.lut[(ty1 * 8 + tx2) * 256 + L];
            const float l21 = S.lut[(ty2 * 8 + tx1) * 256 + L], l22 = S.lut[(ty2 * 8 + tx2) * 256 + L];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            const int Ln = clampi((int)rintf(res), 0, 255);
            int b, g, r;
            lab2bgr_lds(T, S, Ln, px[3 * k + 1], px[3 * k + 2], b, g, r);
            res8[3 * k] = (unsigned char)b; res8[3 * k + 1] = (unsigned char)g; res8[3 * k + 2] = (unsigned char)r;
            if (++x == cd.w) {                                             // a pixel past the crop's end (short last group)
                x = 0;                                                      // is clamped below
                if (y + 1 < cd.h) ++y; else x = cd.w - 1;
            }
        }
        if (cnt == 4) *reinterpret_cast<uint3*>(out + (size_t)i0 * 3) = *reinterpret_cast<const uint3*>(res8);
        else
            for (int k = 0; k < cnt * 3; ++k) out[(size_t)i0 * 3 + k] = res8[k];
    }
}

// ------------------------------------------------------------------ crop -> network input
// torch F.interpolate(bilinear, align_corners=False) on the RGB float crop, /255, normalise.
// from_scratch: read the CLAHE'd packed crop; else read the box straight from the frame.
__global__ __launch_bounds__(256) void crop_norm_kernel(const uint8_t* __restrict__ frame, size_t fstride,
                                                        const uint8_t* __restrict__ scratch,
                                                        const CropDesc* __restrict__ crops,
                                                        float* __restrict__ out_nchw, int from_scratch) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 224 * 224) return;
    const CropDesc cd = crops[blockIdx.y];
    const int oy = idx / 224, ox = idx % 224;
    const float sy = (float)cd.h / 224.0f, sx = (float)cd.w / 224.0f;
    float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
    if (fy < 0.f) fy = 0.f;
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < cd.h - 1 ? 1 : 0), x1 = x0 + (x0 < cd.w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const uint8_t *p00, *p01, *p10, *p11;
    if (from_scratch) {
        const uint8_t* b = scratch + cd.offset;
        p00 = b + ((size_t)y0 * cd.w + x0) * 3; p01 = b + ((size_t)y0 * cd.w + x1) * 3;
        p10 = b + ((size_t)y1 * cd.w + x0) * 3; p11 = b + ((size_t)y1 * cd.w + x1) * 3;
    } else {
        const uint8_t* b = frame + cd.src_off + (size_t)cd.y * fstride + (size_t)cd.x * 3;
        p00 = b + (size_t)y0 * fstride + x0 * 3; p01 = b + (size_t)y0 * fstride + x1 * 3;
        p10 = b + (size_t)y1 * fstride + x0 * 3; p11 = b + (size_t)y1 * fstride + x1 * 3;
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float* o = out_nchw + (size_t)blockIdx.y * 3 * 224 * 224 + idx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                 // output channel c = R,G,B = source byte 2-c
        const int sc = 2 - c;
        const float v = hy * (hx * (float)p00[sc] + lx * (float)p01[sc]) + ly * (hx * (float)p10[sc] + lx * (float)p11[sc]);
        o[(size_t)c * 224 * 224] = (v / 255.0f - mean[c]) / stdv[c];
    }
}

void launch_clahe(const uint8_t* frame, size_t fstride, const CropDesc* crops_dev, int n, uint8_t* lab,
                  uint8_t* luts, uint8_t* bgr_out, const ColorTables& T, int max_pixels, hipStream_t s) {
    hipLaunchKernelGGL(clahe_hist_kernel, dim3(64, n), dim3(256), 0, s, frame, fstride, crops_dev, lab, luts, T, 2.0f);
    // blocks per crop: each stages 22 KB of tables, so a block gets >= 1024 pixel groups (one crop alone still fills
    // the chip's eighth: the request path) and a large batch 8 blocks per crop
    int gx = (max_pixels / 4 + 1023) / 1024;
    const int cap = n >= 256 ? 8 : n >= 64 ? 16 : 64;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(clahe_apply_kernel, dim3(gx, n), dim3(256), 0, s, crops_dev, lab, luts, bgr_out, T);
}

void launch_crop_norm(const uint8_t* frame, size_t fstride, const uint8_t* scratch, const CropDesc* crops_dev,
                      int n, float* out_nchw, bool from_scratch, hipStream_t s) {
    hipLaunchKernelGGL(crop_norm_kernel, dim3((224 * 224 + 255) / 256, n), dim3(256), 0, s, frame, fstride,
                       scratch, crops_dev, out_nchw, from_scratch ? 1 : 0);
}

// ------------------------------------------------------------------- test-time augmentation
// One augmented copy of a face crop as reference deepfake_detection.py:419-433 builds it with cv2:
//   cv2.flip(img, 1) (optional) -> cv2.convertScaleAbs(img, alpha=brightness, beta=0) -> cv2.warpAffine(img,
//   getRotationMatrix2D((w/2, h/2), angle, 1.0), (w, h))   [INTER_LINEAR, BORDER_CONSTANT 0]
// composed per destination pixel.  warpAffine as OpenCV computes it: the INVERSE matrix Mi in double, fixed-point
// source coordinates (10 fractional bits, 5 kept for interpolation: X = (round((Mi[1]*y + Mi[2]) * 1024) + 16 +
// round(Mi[0]*x * 1024)) >> 5), bilinear weights (32 - fx)(32 - fy) * 32 of 32768, result (sum + 16384) >> 15,
// samples outside the image = 0.  convertScaleAbs on 8-bit data: saturate(rint(|v * (float)alpha|)).
__global__ __launch_bounds__(256) void tta_augment_kernel(const uint8_t* __restrict__ src, int h, int w, int stride, int flip, float alpha,
                                                          double m0, double m1, double m2, double m3, double m4, double m5,
                                                          uint8_t* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int X0 = (int)rint((m1 * y + m2) * 1024.0) + 16, Y0 = (int)rint((m4 * y + m5) * 1024.0) + 16;
    const int X = (X0 + (int)rint(m0 * x * 1024.0)) >> 5, Y = (Y0 + (int)rint(m3 * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int wgt[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = sy + (k >> 1), xx = sx + (k & 1);
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
            const uint8_t* p = src + (size_t)yy * stride + 3 * (flip ? w - 1 - xx : xx);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = fabsf((float)p[c] * alpha);
                int q = (int)rintf(v);
                q = q > 255 ? 255 : q;
                acc[c] += q * wgt[k];
            }
        }
    }
    uint8_t* o = dst + ((size_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((acc[c] + (1 << 14)) >> 15);
}

void launch_tta_augment(const uint8_t* src, int h, int w, int stride, int flip, float alpha, const double mi[6], uint8_t* dst,
                        hipStream_t s) {
    hipLaunchKernelGGL(tta_augment_kernel, dim3((w + 255) / 256, h), dim3(256), 0, s, src, h, w, stride, flip, alpha, mi[0], mi[1],
                       mi[2], mi[3], mi[4], mi[5], dst);
}

}  // namespace dfd
