// Forensic-signal kernels on the 256x256 analysis image (gfx950).  All HBM/latency-bound
// integer or fp32 byte work: no MFMA.  One launch handles a batch of frames (blockIdx.y or
// blockIdx.z = frame); every reduction is written as per-block partials and summed in a fixed
// order by stats_finalize_kernel, so results are run-to-run identical (no float atomics).
//
//   gray_kernel            BGR->GRAY fixed point                      frame_analysis.py:136,188,...
//   fft256_kernel          256-point complex FFT rows (LDS radix-2)   frame_analysis.py:139-141
//   fft_band_kernel        second FFT pass + log1p|X| band sums       frame_analysis.py:141-165
//   noise_block_kernel     gray - GaussianBlur5 -> 32x32 block std    frame_analysis.py:188-202
//   jpeg_block_kernel      q90 4:2:0 islow DCT round trip per block   frame_analysis.py:233-236
//   ela_block_kernel       fancy upsample + YCC->RGB + absdiff stats  frame_analysis.py:242-253
//   sobel_lap_kernel       Sobel dx/dy + Laplacian sums               frame_analysis.py:289-294
//   canny_nms_kernel       fixed-point non-maximum suppression        frame_analysis.py:289
//   canny_hyst_kernel      8-connected hysteresis in LDS + edge count frame_analysis.py:289-290
//   hsv_stats_kernel       BGR->HSV integer + S/V moments + hue set   frame_analysis.py:318-338
//   absdiff_kernel         sum |gray - prev gray|                     frame_analysis.py:363-364
//
// Compiled with -ffp-contract=off (operation orders restate OpenCV's float filters).
#include "forensic_kernels.h"
#include "jpeg_dct.h"

namespace dfd {

constexpr int FS = 256;            // analysis edge
constexpr int FPIX = FS * FS;

__device__ __forceinline__ int r101(int i) { i = i < 0 ? -i : i; return i >= FS ? 2 * (FS - 1) - i : i; }
__device__ __forceinline__ int clampi2(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// block-wide sum of doubles (256 or 1024 threads); result valid in thread 0
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0)
        for (int i = 0; i < NT / 64; ++i) r += sh[i];
    __syncthreads();
    return r;
}

// N block-wide sums at once: the same shuffle tree and the same wave-order fold per value as block_sum (identical bits),
// one barrier pair for all of them instead of one per value (fft_band: 7, hsv_stats: 4, sobel_lap: 2 - the barriers were
// most of what these 256-pixel blocks did after their loads).  Results valid in thread 0.
template <int NT, int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off);
    if ((tid & 63) == 0)
#pragma unroll
        for (int j = 0; j < N; ++j) sh[(tid >> 6) * N + j] = v[j];
    __syncthreads();
    if (tid == 0)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double r = 0.0;
            for (int i = 0; i < NT / 64; ++i) r += sh[i * N + j];
            v[j] = r;
        }
    __syncthreads();
}

// ---------------------------------------------------------------------------------- gray
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray) {
    const size_t i = (size_t)blockIdx.y * FPIX + blockIdx.x * 256 + threadIdx.x;
    const uint8_t* p = bgr + i * 3;
    gray[i] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
}

// ---------------------------------------------------------------------------------- FFT
// 256-point radix-2 DIT in LDS, 128 threads = one butterfly each per stage.
__device__ __forceinline__ void fft256_lds(float2* x, const float2* __restrict__ tw, int tid) {
    for (int half = 1; half < 256; half <<= 1) {
        const int pos = tid & (half - 1);
        const int i0 = ((tid - pos) << 1) + pos, i1 = i0 + half;
        const float2 w = tw[pos * (128 / half)];
        const float2 a = x[i0], b = x[i1];
        const float2 t = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
        x[i0] = make_float2(a.x + t.x, a.y + t.y);
        x[i1] = make_float2(a.x - t.x, a.y - t.y);
        __syncthreads();
    }
}

// pass 1: FFT of each image row (real input); output stored transposed [k][row]
__global__ __launch_bounds__(128) void fft256_kernel(const uint8_t* __restrict__ gray, float2* __restrict__ out,
                                                     const float2* __restrict__ tw) {
    __shared__ float2 x[256];
    const int tid = threadIdx.x, row = blockIdx.x;
    const uint8_t* g = gray + (size_t)blockIdx.y * FPIX + row * FS;
    for (int i = tid; i < 256; i += 128) x[__brev((unsigned)i) >> 24] = make_float2((float)g[i], 0.f);
    __syncthreads();
    fft256_lds(x, tw, tid);
    float2* o = out + (size_t)blockIdx.y * FPIX;
    for (int k = tid; k < 256; k += 128) o[(size_t)k * FS + row] = x[k];
}

// pass 2: FFT along the other axis, then log1p|X| accumulated into the three radial bands.
// The band masks depend on k1^2+k2^2 only, so working on the transposed array changes nothing.
__global__ __launch_bounds__(128) void fft_band_kernel(const float2* __restrict__ in, double* __restrict__ part,
                                                       const float2* __restrict__ tw) {
    __shared__ float2 x[256];
    __shared__ double red[2 * 7];
    const int tid = threadIdx.x, k1 = blockIdx.x;
    const float2* src = in + (size_t)blockIdx.y * FPIX + (size_t)k1 * FS;
    for (int i = tid; i < 256; i += 128) x[__brev((unsigned)i) >> 24] = src[i];
    __syncthreads();
    fft256_lds(x, tw, tid);
    const int s1 = k1 < 128 ? k1 : k1 - 256;
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};      // low sum,cnt | mid sum,sumsq,cnt | high sum,cnt
    for (int k2 = tid; k2 < 256; k2 += 128) {
        const int s2 = k2 < 128 ? k2 : k2 - 256;
        const int d2 = s1 * s1 + s2 * s2;
        const float m = log1pf(hypotf(x[k2].x, x[k2].y));
        if (d2 <= 32 * 32) { acc[0] += m; acc[1] += 1.0; }
        else if (d2 <= 64 * 64) { acc[2] += m; acc[3] += (double)m * m; acc[4] += 1.0; }
        else if (d2 <= 128 * 128) { acc[5] += m; acc[6] += 1.0; }
    }
    double* p = part + ((size_t)blockIdx.y * FS + k1) * 7;
    block_sum_n<128, 7>(acc, red);
    if (tid == 0)
#pragma unroll
        for (int j = 0; j < 7; ++j) p[j] = acc[j];
}

// --------------------------------------------------------------------------------- noise
// residual = gray - blur5(gray) (separable [1,4,6,4,1]/16, reflect-101, fp32 in OpenCV's
// symmetric-filter order); population std of each 32x32 block.
__device__ __forceinline__ float blur_row(const uint8_t* g, int y, int x) {
    const uint8_t* r = g + y * FS;
    float s = 0.375f * (float)r[x];
    s = s + 0.25f * ((float)r[r101(x - 1)] + (float)r[r101(x + 1)]);
    s = s + 0.0625f * ((float)r[r101(x - 2)] + (float)r[r101(x + 2)]);
    return s;
}

__global__ __launch_bounds__(256) void noise_block_kernel(const uint8_t* __restrict__ gray, double* __restrict__ stds) {
    __shared__ float res[1024];
    __shared__ double red[4];
    __shared__ double mean_sh;
    const int tid = threadIdx.x, blk = blockIdx.x;           // 64 blocks of 32x32
    const uint8_t* g = gray + (size_t)blockIdx.y * FPIX;
    const int by = (blk >> 3) * 32, bx = (blk & 7) * 32;
    double s = 0.0;
    for (int i = tid; i < 1024; i += 256) {
        const int y = by + (i >> 5), x = bx + (i & 31);
        float o = 0.375f * blur_row(g, y, x);
        o = o + 0.25f * (blur_row(g, r101(y - 1), x) + blur_row(g, r101(y + 1), x));
        o = o + 0.0625f * (blur_row(g, r101(y - 2), x) + blur_row(g, r101(y + 2), x));
        const float r = (float)g[y * FS + x] - o;
        res[i] = r;
        s += r;
    }
    const double tot = block_sum<256>(s, red);
    if (tid == 0) mean_sh = tot / 1024.0;
    __syncthreads();
    const double mean = mean_sh;
    double q = 0.0;
    for (int i = tid; i < 1024; i += 256) { const double d = (double)res[i] - mean; q += d * d; }
    const double ss = block_sum<256>(q, red);
    if (tid == 0) stds[(size_t)blockIdx.y * 64 + blk] = sqrt(ss / 1024.0);
}

// ---------------------------------------------------------------------------------- JPEG
// libjpeg integer pipeline per 8x8 block, one thread per block (64 coefficients in registers).
// quantise + dequantise at quality 90 with the divisors as compile-time constants (the tables again as constexpr: after
// unrolling every `/ dv` is a multiply-shift; with the divisor read from __constant__ memory each of the 64 divisions
// per block was a ~25-instruction sequence - a quarter of this kernel's instructions)
template <bool CHROMA>
__device__ __forceinline__ void jpeg_quant_q90(int* d) {
    constexpr int L[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                           69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
                           81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    constexpr int C[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                           99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
#pragma unroll
    for (int i = 0; i < 64; ++i) {                                // quality 90: scale = 200 - 2*90 = 20
        const int q0 = CHROMA ? C[i] : L[i];
        int qv = (q0 * 20 + 50) / 100;
        qv = qv < 1 ? 1 : (qv > 255 ? 255 : qv);
        const int dv = qv << 3, a = d[i] < 0 ? -d[i] : d[i];
        const int lev = (a + (dv >> 1)) / dv;
        d[i] = (d[i] < 0 ? -lev : lev) * qv;                       // quantise, then dequantise
    }
}

// (libjpeg fixed-point colour conversion and jfdctint / jidctint passes: jpeg_dct.h, shared with jpeg_decode.hip)
// blocks 0..1023: Y (32x32 blocks); 1024..1279: Cb (16x16); 1280..1535: Cr
__global__ __launch_bounds__(64) void jpeg_block_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ yp,
                                                        uint8_t* __restrict__ cbp, uint8_t* __restrict__ crp) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= 1536) return;
    const uint8_t* img = bgr + (size_t)blockIdx.y * FPIX * 3;
    int d[64];
    uint8_t* dst;
    int dstride;
    if (b < 1024) {
        const int by = (b >> 5) * 8, bx = (b & 31) * 8;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const uint8_t* p = img + ((by + (i >> 3)) * FS + bx + (i & 7)) * 3;
            d[i] = ycc_y(p[2], p[1], p[0]) - 128;
        }
        dst = yp + (size_t)blockIdx.y * FPIX + by * FS + bx;
        dstride = FS;
    } else {
        const bool is_cr = b >= 1280;
        const int c = b - (is_cr ? 1280 : 1024);
        const int by = (c >> 4) * 8, bx = (c & 15) * 8;          // in the 128x128 chroma plane
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int cy = by + (i >> 3), cx = bx + (i & 7);
            int s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint8_t* p = img + ((2 * cy + (k >> 1)) * FS + 2 * cx + (k & 1)) * 3;
                s += is_cr ? ycc_cr(p[2], p[1], p[0]) : ycc_cb(p[2], p[1], p[0]);
            }
            d[i] = ((s + ((cx & 1) ? 2 : 1)) >> 2) - 128;          // h2v2_downsample, bias 1,2,1,2,...
        }
        dst = (is_cr ? crp : cbp) + (size_t)blockIdx.y * (FPIX / 4) + by * (FS / 2) + bx;
        dstride = FS / 2;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) fdct8<true>(d + 8 * r, 1);
#pragma unroll
    for (int c = 0; c < 8; ++c) fdct8<false>(d + c, 8);
    if (b < 1024) jpeg_quant_q90<false>(d);                     // (wave-uniform: 64-thread blocks, 1024 = 16 x 64)
    else jpeg_quant_q90<true>(d);
#pragma unroll
    for (int c = 0; c < 8; ++c) idct8<true>(d + c, 8);
#pragma unroll
    for (int r = 0; r < 8; ++r) idct8<false>(d + 8 * r, 1);
#pragma unroll
    for (int i = 0; i < 64; ++i) dst[(i >> 3) * dstride + (i & 7)] = (uint8_t)clampi2(d[i] + 128, 0, 255);
}

__device__ __forceinline__ int fancy_up(const uint8_t* p, int Y, int X) {      // p: 128x128 plane
    const int i = Y >> 1, c = X >> 1;
    const int nb = (Y & 1) ? (i + 1 < 128 ? i + 1 : 127) : (i > 0 ? i - 1 : 0);
    const uint8_t *r0 = p + i * 128, *r1 = p + nb * 128;
    const int cur = 3 * r0[c] + r1[c];
    if ((X & 1) == 0) {
        if (c == 0) return (4 * cur + 8) >> 4;
        return (3 * cur + (3 * r0[c - 1] + r1[c - 1]) + 8) >> 4;
    }
    if (c == 127) return (4 * cur + 7) >> 4;
    return (3 * cur + (3 * r0[c + 1] + r1[c + 1]) + 7) >> 4;
}

// per 32x32 block: sum of gray(|frame - decoded|); exact integers
__global__ __launch_bounds__(256) void ela_block_kernel(const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ yp,
                                                        const uint8_t* __restrict__ cbp, const uint8_t* __restrict__ crp,
                                                        double* __restrict__ means) {
    __shared__ double red[4];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const size_t f = blockIdx.y;
    const int by = (blk >> 3) * 32, bx = (blk & 7) * 32;
    long long s = 0;
    for (int i = tid; i < 1024; i += 256) {
        const int y = by + (i >> 5), x = bx + (i & 31);
        const int Yv = yp[f * FPIX + y * FS + x];
        const int cb = fancy_up(cbp + f * (FPIX / 4), y, x) - 128, cr = fancy_up(crp + f * (FPIX / 4), y, x) - 128;
        const int r = clampi2(Yv + ((JFIX(1.40200) * cr + 32768) >> 16), 0, 255);
        const int g = clampi2(Yv + ((-JFIX(0.34414) * cb + 32768 - JFIX(0.71414) * cr) >> 16), 0, 255);
        const int b = clampi2(Yv + ((JFIX(1.77200) * cb + 32768) >> 16), 0, 255);
        const uint8_t* p = bgr + (f * FPIX + y * FS + x) * 3;
        const int db = abs((int)p[0] - b), dg = abs((int)p[1] - g), dr = abs((int)p[2] - r);
        s += (db * 1868 + dg * 9617 + dr * 4899 + (1 << 13)) >> 14;
    }
    const double tot = block_sum<256>((double)s, red);
    if (tid == 0) means[f * 64 + blk] = tot / 1024.0;
}

// --------------------------------------------------------------------------------- edges
// Sobel (BORDER_REPLICATE) dx,dy as int16 pairs + Laplacian ([0 1 0;1 -4 1;0 1 0], reflect-101)
// partial sums (sum, sum of squares as exact integers).
__global__ __launch_bounds__(256) void sobel_lap_kernel(const uint8_t* __restrict__ gray, short2* __restrict__ grad,
                                                        double* __restrict__ part) {
    __shared__ double red[4 * 2];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const uint8_t* g = gray + (size_t)blockIdx.y * FPIX;
    const int y = i >> 8, x = i & 255;
    const int ym = y > 0 ? y - 1 : 0, yp = y < FS - 1 ? y + 1 : FS - 1, xm = x > 0 ? x - 1 : 0, xp = x < FS - 1 ? x + 1 : FS - 1;
    const int a = g[ym * FS + xm], b = g[ym * FS + x], c = g[ym * FS + xp];
    const int d = g[y * FS + xm], e = g[y * FS + x], f = g[y * FS + xp];
    const int h = g[yp * FS + xm], k = g[yp * FS + x], l = g[yp * FS + xp];
    const int dx = (c + 2 * f + l) - (a + 2 * d + h), dy = (h + 2 * k + l) - (a + 2 * b + c);
    grad[(size_t)blockIdx.y * FPIX + i] = make_short2((short)dx, (short)dy);
    const int lap = g[r101(y - 1) * FS + x] + g[r101(y + 1) * FS + x] + g[y * FS + r101(x - 1)] + g[y * FS + r101(x + 1)] - 4 * e;
    double ss[2] = {(double)lap, (double)lap * (double)lap};
    block_sum_n<256, 2>(ss, red);
    if (tid == 0) {
        part[((size_t)blockIdx.y * 256 + blockIdx.x) * 2] = ss[0];
        part[((size_t)blockIdx.y * 256 + blockIdx.x) * 2 + 1] = ss[1];
    }
}

__device__ __forceinline__ int mag_at(const short2* g, int y, int x) {
    if ((unsigned)y >= (unsigned)FS || (unsigned)x >= (unsigned)FS) return 0;    // OpenCV's zero mag border
    const short2 v = g[y * FS + x];
    return abs((int)v.x) + abs((int)v.y);
}

// map: 1 = not an edge, 0 = candidate (passed NMS, above low), 2 = strong (above high)
__global__ __launch_bounds__(256) void canny_nms_kernel(const short2* __restrict__ grad, uint8_t* __restrict__ map,
                                                        int low, int high) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const short2* g = grad + (size_t)blockIdx.y * FPIX;
    const int y = i >> 8, x = i & 255;
    const int xs = g[i].x, ys = g[i].y;
    const int m = abs(xs) + abs(ys);
    uint8_t lab = 1;
    if (m > low) {
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22 = ax * 13573;
        bool keep;
        if (ay < tg22) keep = m > mag_at(g, y, x - 1) && m >= mag_at(g, y, x + 1);
        else {
            const int tg67 = tg22 + (ax << 16);
            if (ay > tg67) keep = m > mag_at(g, y - 1, x) && m >= mag_at(g, y + 1, x);
            else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > mag_at(g, y - 1, x - s) && m > mag_at(g, y + 1, x + s);
            }
        }
        if (keep) lab = m > high ? 2 : 0;
    }
    map[(size_t)blockIdx.y * FPIX + i] = lab;
}

// One 1024-thread block per frame, one 64-pixel word of a row per thread: strong pixels S and weak candidates W as
// bitboards (8 KiB of LDS for S).  A sweep ORs the three rows around a word, dilates by one column (with the edge bits
// of the neighbouring words), ANDs with W and then floods along the row inside the word (Kogge-Stone occluded fill,
// both directions); sweeps repeat until no word changes.  The fixpoint - weak pixels 8-connected to a strong one - is
// the set OpenCV's stack-based flood fill reaches, whatever the visiting order.  (The byte-map version of this kernel
// scanned 64 pixels x 9 LDS reads per thread and sweep: 480 us per 64 frames, more than the other nine forensic
// kernels together.)
__device__ __forceinline__ unsigned long long fill_row(unsigned long long gen, unsigned long long pro) {
    unsigned long long g = gen, p = pro;                    // towards higher columns
    g |= p & (g << 1);  p &= p << 1;
    g |= p & (g << 2);  p &= p << 2;
    g |= p & (g << 4);  p &= p << 4;
    g |= p & (g << 8);  p &= p << 8;
    g |= p & (g << 16); p &= p << 16;
    g |= p & (g << 32);
    unsigned long long h = gen;                             // towards lower columns
    p = pro;
    h |= p & (h >> 1);  p &= p >> 1;
    h |= p & (h >> 2);  p &= p >> 2;
    h |= p & (h >> 4);  p &= p >> 4;
    h |= p & (h >> 8);  p &= p >> 8;
    h |= p & (h >> 16); p &= p >> 16;
    h |= p & (h >> 32);
    return g | h;
}

__global__ __launch_bounds__(1024) void canny_hyst_kernel(const uint8_t* __restrict__ map, double* __restrict__ count) {
    __shared__ unsigned long long S[FS * 4 + 8];            // [row][word], one guard word each side
    __shared__ double red[16];
    const int tid = threadIdx.x, row = tid >> 2, wd = tid & 3;
    const uint8_t* src = map + (size_t)blockIdx.x * FPIX + row * FS + wd * 64;
    unsigned long long s = 0ull, w = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[k];
        const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned lab = (wv[q] >> (8 * b)) & 0xFFu;
                const int bit = k * 16 + q * 4 + b;
                s |= (unsigned long long)(lab == 2u) << bit;
                w |= (unsigned long long)(lab == 0u) << bit;
            }
    }
    const unsigned long long pass = s | w;
    unsigned long long* Sw = S + 4;                         // word index row * 4 + wd, guards at -4..-1 and FS*4..FS*4+3
    if (tid < 4) { S[tid] = 0ull; S[FS * 4 + 4 + tid] = 0ull; }
    Sw[tid] = s;
    __syncthreads();
    for (int iter = 0; iter < FPIX; ++iter) {               // bounded: each productive sweep adds >= 1 edge
        const unsigned long long up = Sw[tid - 4], dn = Sw[tid + 4];
        const unsigned long long v = up | s | dn;
        // edge bits of the horizontal neighbours (rows above / below included); none beyond the image border
        const unsigned long long vl = wd > 0 ? (Sw[tid - 5] | Sw[tid - 1] | Sw[tid + 3]) : 0ull;
        const unsigned long long vr = wd < 3 ? (Sw[tid - 3] | Sw[tid + 1] | Sw[tid + 5]) : 0ull;
        const unsigned long long dil = v | (v << 1) | (v >> 1) | (vl >> 63) | (vr << 63);
        const unsigned long long nw = fill_row(s | (w & dil), pass);
        const int any = __syncthreads_or(nw != s);          // also: every read of this sweep is done
        s = nw;
        Sw[tid] = s;
        __syncthreads();
        if (!any) break;
    }
    const double tot = block_sum<1024>((double)__popcll(s), red);
    if (tid == 0) count[blockIdx.x] = tot;
}

// --------------------------------------------------------------------------------- colour
__global__ __launch_bounds__(256) void hsv_stats_kernel(const uint8_t* __restrict__ bgr, double* __restrict__ part,
                                                        unsigned* __restrict__ hue_bits, ColorTables T) {
    __shared__ double red[4 * 4];
    __shared__ unsigned bits[6];
    const int tid = threadIdx.x;
    if (tid < 6) bits[tid] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.y * FPIX + blockIdx.x * 256 + tid;
    const uint8_t* p = bgr + i * 3;
    const int b = p[0], g = p[1], r = p[2];
    const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
    const int s = (diff * T.hsv_sdiv[v] + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * T.hsv_hdiv[diff] + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    atomicOr(&bits[h >> 5], 1u << (h & 31));
    double acc[4] = {(double)s, (double)s * s, (double)v, (double)v * v};
    double* o = part + ((size_t)blockIdx.y * 256 + blockIdx.x) * 4;
    block_sum_n<256, 4>(acc, red);
    if (tid == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = acc[j];
    __syncthreads();
    if (tid < 6 && bits[tid]) atomicOr(&hue_bits[(size_t)blockIdx.y * 6 + tid], bits[tid]);   // integer OR: order-free
}

// ------------------------------------------------------------------------------- temporal
__global__ __launch_bounds__(256) void absdiff_kernel(const uint8_t* __restrict__ gray, const uint8_t* __restrict__ prev,
                                                      double* __restrict__ part) {
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int d = abs((int)gray[i] - (int)prev[i]);
    const double t = block_sum<256>((double)d, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ------------------------------------------------------------------------------- finalize
// stats layout per frame (doubles), see forensic_kernels.h
// One wave per frame: lane l folds partial rows l, l + 64, l + 128, l + 192 (in that order), then a butterfly over the
// lanes - a fixed order, so the sums are run-to-run and batch-size invariant.  (One thread per frame walking all 256
// rows was 3,300 dependent L2 round trips: 69 us, the longest forensic kernel once the hysteresis was fixed.)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64) void stats_finalize_kernel(ForensicBuffers B, int full, int nframes) {
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= nframes) return;
    double* st = B.stats + (size_t)f * FORENSIC_STATS;
    double a[7] = {0, 0, 0, 0, 0, 0, 0}, l1 = 0, l2 = 0, s1 = 0, s2 = 0, v1 = 0, v2 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t r = (size_t)f * 256 + lane + 64 * k;
#pragma unroll
        for (int j = 0; j < 7; ++j) a[j] += B.fft_part[r * 7 + j];
        l1 += B.lap_part[r * 2];
        l2 += B.lap_part[r * 2 + 1];
        if (full) {
            const double* p = B.hsv_part + r * 4;
            s1 += p[0]; s2 += p[1]; v1 += p[2]; v2 += p[3];
        }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) a[j] = wave_sum(a[j]);
    l1 = wave_sum(l1); l2 = wave_sum(l2);
    s1 = wave_sum(s1); s2 = wave_sum(s2); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if (lane != 0) return;
    const double mid_mean = a[2] / a[4];
    st[ST_FREQ_LOW] = a[0] / a[1];
    st[ST_FREQ_MID] = mid_mean;
    st[ST_FREQ_HIGH] = a[5] / a[6];
    const double var = a[3] / a[4] - mid_mean * mid_mean;
    st[ST_FREQ_MID_STD] = sqrt(var > 0 ? var : 0);
    const double lm = l1 / FPIX;
    st[ST_LAP_VAR] = l2 / FPIX - lm * lm;
    st[ST_EDGE_COUNT] = B.edge_count[f];
    if (full) {
        const double sm = s1 / FPIX, vm = v1 / FPIX;
        const double sv = s2 / FPIX - sm * sm, vv = v2 / FPIX - vm * vm;
        st[ST_SAT_STD] = sqrt(sv > 0 ? sv : 0);
        st[ST_VAL_STD] = sqrt(vv > 0 ? vv : 0);
        int hues = 0;
        for (int w = 0; w < 6; ++w) hues += __popc(B.hue_bits[(size_t)f * 6 + w]);
        st[ST_HUES] = hues;
    }
}

// ------------------------------------------------------------------------------- launchers
// n frames get every signal; `gray_only` further frames (behind them in the buffers) only their gray plane
void launch_forensics(const ForensicBuffers& B, int n, bool full, const ColorTables& T, const float2* tw, hipStream_t s,
                      int gray_only) {
    hipLaunchKernelGGL(gray_kernel, dim3(256, n + gray_only), dim3(256), 0, s, B.rs, B.gray);
    if (n <= 0) return;
    hipLaunchKernelGGL(fft256_kernel, dim3(256, n), dim3(128), 0, s, B.gray, B.fft_tmp, tw);
    hipLaunchKernelGGL(fft_band_kernel, dim3(256, n), dim3(128), 0, s, B.fft_tmp, B.fft_part, tw);
    hipLaunchKernelGGL(sobel_lap_kernel, dim3(256, n), dim3(256), 0, s, B.gray, B.grad, B.lap_part);
    hipLaunchKernelGGL(canny_nms_kernel, dim3(256, n), dim3(256), 0, s, B.grad, B.map, 50, 150);
    hipLaunchKernelGGL(canny_hyst_kernel, dim3(n), dim3(1024), 0, s, B.map, B.edge_count);
    if (full) {
        hipLaunchKernelGGL(noise_block_kernel, dim3(64, n), dim3(256), 0, s, B.gray, B.stats_noise);
        hipLaunchKernelGGL(jpeg_block_kernel, dim3(24, n), dim3(64), 0, s, B.rs, B.jy, B.jcb, B.jcr);
        hipLaunchKernelGGL(ela_block_kernel, dim3(64, n), dim3(256), 0, s, B.rs, B.jy, B.jcb, B.jcr, B.stats_ela);
        hipMemsetAsync(B.hue_bits, 0, (size_t)n * 6 * sizeof(unsigned), s);
        hipLaunchKernelGGL(hsv_stats_kernel, dim3(256, n), dim3(256), 0, s, B.rs, B.hsv_part, B.hue_bits, T);
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(n), dim3(64), 0, s, B, full ? 1 : 0, n);
}

// frame-sharded streams: frame f against frame prev_index[f] of the same batch (gray planes [n][65536]);
// prev_index < 0 = no predecessor (partial sums 0).  part: [n][256].
__global__ __launch_bounds__(256) void absdiff_pairs_kernel(const uint8_t* __restrict__ gray, const int* __restrict__ prev_index,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    const int f = blockIdx.y, pf = prev_index[f];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int d = pf < 0 ? 0 : abs((int)gray[(size_t)f * FPIX + i] - (int)gray[(size_t)pf * FPIX + i]);
    const double t = block_sum<256>((double)d, red);
    if (threadIdx.x == 0) part[(size_t)f * 256 + blockIdx.x] = t;
}

void launch_absdiff_pairs(const uint8_t* gray, const int* prev_index, double* part, int n, hipStream_t s) {
    hipLaunchKernelGGL(absdiff_pairs_kernel, dim3(256, n), dim3(256), 0, s, gray, prev_index, part);
}

void launch_absdiff(const uint8_t* gray, const uint8_t* prev, double* part, hipStream_t s) {
    hipLaunchKernelGGL(absdiff_kernel, dim3(256), dim3(256), 0, s, gray, prev, part);
}

}  // namespace dfd

namespace dfd {

namespace {
constexpr size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
struct Sizes {
    size_t rs = al(FPIX * 3), gray = al(FPIX), fft = al(FPIX * sizeof(float2)), fftp = al(256 * 7 * 8),
           grad = al(FPIX * sizeof(short2)), lapp = al(256 * 2 * 8), map = al(FPIX), ec = al(8), jy = al(FPIX),
           jc = al(FPIX / 4), hsvp = al(256 * 4 * 8), hue = al(6 * 4), st = al(FORENSIC_STATS * 8), blk = al(64 * 8);
    size_t total() const { return rs + gray + fft + fftp + grad + lapp + map + ec + jy + 2 * jc + hsvp + hue + st + 2 * blk; }
};
}  // namespace

size_t forensic_bytes_per_frame() { return Sizes().total(); }

// arrays are frame-major ([n][...]), so each kind gets one contiguous region of n * size bytes;
// the per-kind sizes above are multiples of the exact element counts the kernels index with
void forensic_carve(void* base, int n, ForensicBuffers* o) {
    char* p = static_cast<char*>(base);
    auto take = [&](size_t exact_per_frame) { char* r = p; p += al(exact_per_frame * n); return r; };
    o->rs = (uint8_t*)take(FPIX * 3);
    o->gray = (uint8_t*)take(FPIX);
    o->fft_tmp = (float2*)take(FPIX * sizeof(float2));
    o->fft_part = (double*)take(256 * 7 * 8);
    o->grad = (short2*)take(FPIX * sizeof(short2));
    o->lap_part = (double*)take(256 * 2 * 8);
    o->map = (uint8_t*)take(FPIX);
    o->edge_count = (double*)take(8);
    o->jy = (uint8_t*)take(FPIX);
    o->jcb = (uint8_t*)take(FPIX / 4);
    o->jcr = (uint8_t*)take(FPIX / 4);
    o->hsv_part = (double*)take(256 * 4 * 8);
    o->hue_bits = (unsigned*)take(6 * 4);
    o->stats = (double*)take(FORENSIC_STATS * 8);
    o->stats_noise = (double*)take(64 * 8);
    o->stats_ela = (double*)take(64 * 8);
}

}  // namespace dfd
