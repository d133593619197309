// SSD-style face detector: the kernels that are not plain convolutions (gfx950).
//
//   ssd_conv1_kernel   7x7 s2 conv straight from the resized u8 BGR image with the
//                      blobFromImage mean subtracted on the fly            face_detection.py:76-79
//   maxpool3s2_kernel  Caffe max pooling 3x3 s2, ceil mode
//   l2norm_kernel      SSD Normalize across channels (wave reduction)
//   ssd_decode_kernel  analytic PriorBox + CENTER_SIZE decode + 2-way softmax
//   ssd_nms_kernel     DetectionOutput: per-image bitonic sort in LDS (score desc, index asc =
//                      stable), top_k, greedy NMS, keep_top_k
// The 3x3 / 1x1 trunk and head convolutions run on pw_kernel<NT, true> (b0_kernels.hip).
#include "ssd_kernels.h"
#include "kernel_util.h"

namespace dfd {

typedef float v4f __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- conv1
// thread = one output pixel x 16 of the 32 output channels (blockIdx.y = channel half).  The 7x7x3 x 16 weights of the
// half are SGPR operands of v_pk_fma_f32 (scalar loads, one tap ahead, kernel_util.h): no LDS, no weight VGPRs; the
// products accumulate in (ky, kx, ci) order on top of the bias.  A tap outside the image contributes 0 (Caffe pads the
// mean-subtracted blob with zeros): the byte is read from a clamped address and the VALUE is masked, so every load of
// a kernel row is unconditional.  Replaces a thread-per-channel-quad kernel with LDS weights (365 us per 64 frames).
__global__ __launch_bounds__(256) void ssd_conv1_kernel(const uint8_t* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int n_img,
                                                        float sb, float sg, float sr, float hb, float hg, float hr, int relu) {
    const int half = blockIdx.y;
    const float* wh = w + half * 16;                         // weights [7][7][3][32]: a tap's 16 floats of this half
    const long long npix = (long long)n_img * 150 * 150;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = t < npix ? t : npix - 1;           // surplus threads redo the last pixel (uniform control flow)
    const int ox = (int)(pix % 150), oy = (int)((pix / 150) % 150), n = (int)(pix / 22500);
    const uint8_t* src = img + (size_t)n * 300 * 300 * 3;
    const float sc[3] = {sb, sg, sr}, sh[3] = {hb, hg, hr};
    float acc[16];
    {
        sf16 bv = sload16s(b, half * 64);
        swait1(bv);
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = bv[o];
    }
    int col[7];
    float mx[7];                                             // 1 inside the image, 0 outside (a multiplier, not a branch)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
        const int ix = 2 * ox - 3 + kx;
        mx[kx] = (unsigned)ix < 300u ? 1.f : 0.f;
        col[kx] = (ix < 0 ? 0 : (ix > 299 ? 299 : ix)) * 3;
    }
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = 2 * oy - 3 + ky;
        const float my = (unsigned)iy < 300u ? 1.f : 0.f;
        const uint8_t* row = src + (size_t)(iy < 0 ? 0 : (iy > 299 ? 299 : iy)) * 900;
        float v[21];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float f = (float)row[col[kx] + ci] * sc[ci] + sh[ci];
                v[kx * 3 + ci] = f * (my * mx[kx]);          // x 1 is exact; a select here makes hipcc branch around the load
            }
        sf16 wn = sload16s(wh, ky * 21 * 128);
#pragma unroll
        for (int tp = 0; tp < 21; ++tp) {
            swait1(wn);
            const sf16 wc = wn;
            if (tp + 1 < 21) wn = sload16s(wh, (ky * 21 + tp + 1) * 128);
#pragma unroll
            for (int o = 0; o < 16; o += 2) {
                const v2f x2 = {v[tp], v[tp]};
                v2f a = {acc[o], acc[o + 1]};
                a = __builtin_elementwise_fma(x2, (v2f){wc[o], wc[o + 1]}, a);
                acc[o] = a.x; acc[o + 1] = a.y;
            }
            __builtin_amdgcn_sched_barrier(0);               // the tap's FMAs stay before the next wait / load
        }
    }
    if (relu) {
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = fmaxf(acc[o], 0.f);
    }
    // the results are pinned before the guarded store (see mt_pnet_conv1_pool_kernel: otherwise the FMAs sink into the branch)
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]),
                 "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]));
    if (t < npix) {
        float* yp = y + (size_t)pix * 32 + half * 16;
#pragma unroll
        for (int o = 0; o < 16; o += 4) *reinterpret_cast<v4f*>(yp + o) = (v4f){acc[o], acc[o + 1], acc[o + 2], acc[o + 3]};
    }
}

void launch_ssd_conv1(const uint8_t* img, const float* w, const float* b, float* y, int n, const float in_scale[3],
                      const float in_shift[3], bool relu, hipStream_t s) {
    const long long threads = (long long)n * 150 * 150;
    hipLaunchKernelGGL(ssd_conv1_kernel, dim3((int)((threads + 255) / 256), 2), dim3(256), 0, s, img, w, b, y, n,
                       in_scale[0], in_scale[1], in_scale[2], in_shift[0], in_shift[1], in_shift[2], relu ? 1 : 0);
}

// ------------------------------------------------------------------- per-channel affine / add
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ add,
                                                             float* __restrict__ y, long long nvec, int c4, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int c = (int)(i % c4) * 4;
    v4f v = *reinterpret_cast<const v4f*>(x + i * 4);
    if (scale) v = v * *reinterpret_cast<const v4f*>(scale + c) + *reinterpret_cast<const v4f*>(shift + c);
    if (add) v += *reinterpret_cast<const v4f*>(add + i * 4);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<v4f*>(y + i * 4) = v;
}

void launch_channel_affine(const float* x, const float* scale, const float* shift, const float* add, float* y,
                           long long npix, int C, bool relu, hipStream_t s) {
    const long long nvec = npix * (C / 4);
    hipLaunchKernelGGL(channel_affine_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, x, scale, shift, add, y,
                       nvec, C / 4, relu ? 1 : 0);
}

// -------------------------------------------------------------------------------- maxpool
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int n_img,
                                                         int H, int Ho, int C) {
    const int c4 = C / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n_img * Ho * Ho * c4) return;
    const int cg = (int)(gid % c4);
    const long long pix = gid / c4;
    const int ox = (int)(pix % Ho), oy = (int)((pix / Ho) % Ho), n = (int)(pix / ((long long)Ho * Ho));
    v4f m = (v4f){-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    // ceil mode: the last window hangs over the edge; a clamped tap re-reads an element of the window (a max is
    // idempotent), so all nine loads are unconditional and in flight together
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky < H ? 2 * oy + ky : H - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox + kx < H ? 2 * ox + kx : H - 1;
            const v4f v = *reinterpret_cast<const v4f*>(x + (((size_t)n * H + iy) * H + ix) * C + 4 * cg);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<v4f*>(y + (size_t)pix * C + 4 * cg) = m;
}

void launch_maxpool3s2(const float* x, float* y, int n, int H, int Ho, int C, hipStream_t s) {
    const long long threads = (long long)n * Ho * Ho * (C / 4);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, x, y, n, H, Ho, C);
}

// --------------------------------------------------------------------------------- l2norm
// one 32-lane half-wave per pixel (C = 128 = 32 lanes x float4)
__global__ __launch_bounds__(256) void l2norm128_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        float* __restrict__ y, long long npix) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid >> 5;
    const int l = (int)(gid & 31);
    if (pix >= npix) return;                          // whole half-waves leave together (256 % 32 == 0)
    const v4f v = *reinterpret_cast<const v4f*>(x + pix * 128 + 4 * l);
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float inv = 1.0f / sqrtf(ss + 1e-10f);
    const v4f sc = *reinterpret_cast<const v4f*>(scale + 4 * l);
    *reinterpret_cast<v4f*>(y + pix * 128 + 4 * l) = v * inv * sc;
}

void launch_l2norm128(const float* x, const float* scale, float* y, long long npix, hipStream_t s) {
    const long long threads = npix * 32;
    hipLaunchKernelGGL(l2norm128_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, x, scale, y, npix);
}

// --------------------------------------------------------------------------------- decode
// thread per prior: PriorBox (offset .5, clip false) generated analytically, CENTER_SIZE decode
// with variances, softmax over (background, face).
__global__ __launch_bounds__(256) void ssd_decode_kernel(SsdHeads H, float* __restrict__ boxes, float* __restrict__ prob,
                                                         int n_priors, float image_size, float v0, float v1, float v2,
                                                         float v3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_priors) return;
    const int img = blockIdx.y;
    int s = 0;
    while (s < 5 && i >= H.first[s + 1]) ++s;
    const int local = i - H.first[s], p = H.priors[s], m = H.map[s];
    const int cell = local / p, k = local - cell * p;
    // the prior itself comes from the host-built table (PriorBox evaluated in double, rounded once)
    const float x1 = H.prior_tab[i * 4], y1 = H.prior_tab[i * 4 + 1], x2 = H.prior_tab[i * 4 + 2], y2 = H.prior_tab[i * 4 + 3];
    // head output [img][cell][p*6]: p*4 loc values then p*2 conf values
    const float* o = H.out[s] + ((size_t)img * m * m + cell) * (p * 6);
    const float l0 = o[k * 4], l1 = o[k * 4 + 1], l2 = o[k * 4 + 2], l3 = o[k * 4 + 3];
    const float c0 = o[p * 4 + k * 2], c1 = o[p * 4 + k * 2 + 1];
    const float pw = x2 - x1, ph = y2 - y1, pcx = (x1 + x2) * 0.5f, pcy = (y1 + y2) * 0.5f;
    const float dcx = v0 * l0 * pw + pcx, dcy = v1 * l1 * ph + pcy;
    const float dw = expf(v2 * l2) * pw, dh = expf(v3 * l3) * ph;
    float* b = boxes + ((size_t)img * n_priors + i) * 4;
    b[0] = dcx - dw * 0.5f; b[1] = dcy - dh * 0.5f; b[2] = dcx + dw * 0.5f; b[3] = dcy + dh * 0.5f;
    const float mxl = fmaxf(c0, c1);
    const float e0 = expf(c0 - mxl), e1 = expf(c1 - mxl);
    prob[(size_t)img * n_priors + i] = e1 / (e0 + e1);
}

void launch_ssd_decode(const SsdHeads& H, float* boxes, float* prob, int n, int n_priors, float image_size,
                       const float var[4], hipStream_t s) {
    hipLaunchKernelGGL(ssd_decode_kernel, dim3((n_priors + 255) / 256, n), dim3(256), 0, s, H, boxes, prob, n_priors,
                       image_size, var[0], var[1], var[2], var[3]);
}

// ------------------------------------------------------------------------------------ NMS
constexpr int NMS_SORT = 16384;      // next power of two >= 8732
constexpr int NMS_TOPK = 400;

__device__ __forceinline__ float box_area(const float* b) {
    return (b[2] < b[0] || b[3] < b[1]) ? 0.f : (b[2] - b[0]) * (b[3] - b[1]);
}
__device__ __forceinline__ float jaccard(const float* a, const float* b) {
    if (b[0] > a[2] || b[2] < a[0] || b[1] > a[3] || b[3] < a[1]) return 0.f;
    const float ix = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), iy = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
    const float inter = ix * iy;
    return inter / (box_area(a) + box_area(b) - inter);
}

// one 1024-thread block per image; key = score bits (positive floats order like uints) in the high
// word, ~index in the low word, sorted descending => score desc, index asc (stable order)
__global__ __launch_bounds__(1024) void ssd_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ prob,
                                                       int n_priors, float conf_thr, double nms_thr, int keep_top_k,
                                                       float* __restrict__ rows, int* __restrict__ count) {
    __shared__ unsigned long long key[NMS_SORT];
    __shared__ float cand[NMS_TOPK][4];
    __shared__ unsigned char dead[NMS_TOPK];
    __shared__ int kept[NMS_TOPK];
    __shared__ int n_kept;
    const int tid = threadIdx.x, img = blockIdx.x;
    const float* pr = prob + (size_t)img * n_priors;
    // priors above the confidence threshold are appended to the front of key[] (any order: the keys are unique and the
    // sort below orders them), then only the next power of two above their number is sorted - a few hundred for a
    // trained detector instead of all 16384 slots (105 barrier-separated bitonic stages down to ~40)
    __shared__ int n_valid;
    if (tid == 0) n_valid = 0;
    unsigned long long mine[NMS_SORT / 1024];
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r) {
        const int i = tid + r * 1024;
        unsigned long long k = 0ull;
        if (i < n_priors) {
            const float p = pr[i];
            if (p > conf_thr) k = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        }
        mine[r] = k;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r)
        if (mine[r] != 0ull) key[atomicAdd(&n_valid, 1)] = mine[r];
    __syncthreads();
    int sort_n = 2;
    while (sort_n < n_valid) sort_n <<= 1;
    if (sort_n < NMS_TOPK) sort_n = 512;                   // the candidate scan below reads the first NMS_TOPK slots
    for (int i = n_valid + tid; i < sort_n; i += 1024) key[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= sort_n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < sort_n / 2; t += 1024) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a < b) == desc) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    // candidates: first min(top_k, #valid) keys
    int ncand = 0;
    for (int i = tid; i < NMS_TOPK; i += 1024) {
        dead[i] = 0;
        if (key[i] != 0ull) {
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull);
            const float* b = boxes + ((size_t)img * n_priors + idx) * 4;
            cand[i][0] = b[0]; cand[i][1] = b[1]; cand[i][2] = b[2]; cand[i][3] = b[3];
        }
    }
    if (tid == 0) n_kept = 0;
    __syncthreads();
    ncand = n_valid < NMS_TOPK ? n_valid : NMS_TOPK;                // the valid keys sort to the front
    for (int i = 0; i < ncand; ++i) {
        if (!dead[i]) {                                             // uniform: all threads read the same flag
            if (tid == 0) kept[n_kept++] = i;
            for (int j = i + 1 + tid; j < ncand; j += 1024)
                if (!dead[j] && (double)jaccard(cand[i], cand[j]) > nms_thr) dead[j] = 1;
        }
        __syncthreads();
    }
    const int nk = n_kept < keep_top_k ? n_kept : keep_top_k;
    for (int r = tid; r < nk; r += 1024) {
        const int i = kept[r];
        float* o = rows + ((size_t)img * keep_top_k + r) * 5;
        o[0] = __uint_as_float((unsigned)(key[i] >> 32));
        o[1] = cand[i][0]; o[2] = cand[i][1]; o[3] = cand[i][2]; o[4] = cand[i][3];
    }
    if (tid == 0) count[img] = nk;
}

void launch_ssd_nms(const float* boxes, const float* prob, int n, int n_priors, float conf_thr, double nms_thr,
                    int keep_top_k, float* rows, int* count, hipStream_t s) {
    hipLaunchKernelGGL(ssd_nms_kernel, dim3(n), dim3(1024), 0, s, boxes, prob, n_priors, conf_thr, nms_thr, keep_top_k,
                       rows, count);
}

}  // namespace dfd
