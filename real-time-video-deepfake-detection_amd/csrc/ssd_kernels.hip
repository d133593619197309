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
#include <cstdlib>

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

// ---- conv1 on the matrix pipe (round 3) ------------------------------------------------------------------------
// The 7x7x3 -> 32 stride-2 convolution as a GEMM on v_mfma_f32_16x16x32_bf16, like the classifier's stem: D[co][pixel]
// = sum_k W[co][k] * X[k][pixel], k = (ky * 7 + kx) * 3 + ci = ky * 21 + (the 21 consecutive bytes of an image row),
// K = 147 in 5 steps of 32 (the weight planes are zero beyond 147).  Weights: the handle's three bf16 planes (exact
// split).  Input: u8 * scale + shift; with the reference's blobFromImage parameters (scale 1, integer means 104 / 177 /
// 123, face_detection.py:76-79) that is an integer of magnitude <= 255 - ONE exact bf16 term, three products per
// K-step (EXACT); any other input transform (a folded data-layer BatchNorm, caffe_io) takes the three-term split, six
// products.  Block = 8 output rows x 16 columns: the 21 x 37 x 3 input patch is transformed once into an fp32 LDS
// patch (0 outside the image: Caffe pads the mean-subtracted blob), each wave gathers the B operand of its two 16-pixel
// rows from it (lane = pixel, 8 consecutive k), and the A fragments of a K-step serve both rows.  The thread-per-pixel
// kernel above issued 147 x 16 FMAs per thread behind scalar weight loads: 195-240 us per 64 frames, VALU-bound.
template <bool EXACT>
__global__ __launch_bounds__(256, 3) void ssd_conv1_mfma_kernel(const uint8_t* __restrict__ img, const unsigned short* __restrict__ w3,
                                                             int plane, int Kp, const float* __restrict__ bias,
                                                             float* __restrict__ y, float sb, float sg, float sr, float hb,
                                                             float hg, float hr, int relu) {
    constexpr int TH = 32, TW = 16, RPW = TH / 4, PH = 2 * TH + 5, PW = (2 * TW + 5) * 3, PWP = PW + 1;      // 69 rows x 111 (+1) floats
    __shared__ float patch[PH * PWP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int n = blockIdx.z, ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const uint8_t* src = img + (size_t)n * 300 * 300 * 3;
    const int r0 = 2 * ty0 - 3, c0 = (2 * tx0 - 3) * 3;                  // patch origin in the image (row, byte column)
    const float sc[3] = {sb, sg, sr}, sh[3] = {hb, hg, hr};
    // patch: aligned dword loads (the row segment starts 3 bytes before the patch: c0 - 3 = 96 * blockIdx.x - 12 is a
    // multiple of 4), clamped and unconditional, all in flight before the first LDS store; 29 dwords cover the 111 bytes
    // of a row.  (One byte per load - 30 loads and 30 modulo-3 per thread - was a third of the block's time.)
    constexpr int DW = 29, NEL = PH * DW, NLD = (NEL + 255) / 256;
    unsigned pv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        const int r = i / DW, d = i - r * DW;
        const int iy = r0 + r, ib = c0 - 3 + 4 * d;                      // first byte column of the dword
        const bool inside = (unsigned)iy < 300u && ib >= 0 && ib < 900;
        pv[k] = *reinterpret_cast<const unsigned*>(src + (size_t)(inside ? iy : 0) * 900 + (inside ? ib : 0));
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        const int r = i / DW, d = i - r * DW;
        const int iy = r0 + r, ib0 = c0 - 3 + 4 * d;
        const bool row_in = (unsigned)iy < 300u;
        int ci = (ib0 + 900) % 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ib = ib0 + e, pc = 4 * d + e - 3;                  // patch column of this byte
            const bool inside = row_in && ib >= 0 && ib < 900;
            const float v = (float)((pv[k] >> (8 * e)) & 0xFFu) * sc[ci] + sh[ci];
            if (pc >= 0 && pc < PW) patch[r * PWP + pc] = inside ? v : 0.f;
            ci = ci == 2 ? 0 : ci + 1;
        }
    }
    // per-lane patch offsets of the 8 k values of each K-step (k >= 147: offset 0, zero weights)
    int koff[5][8];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = s5 * 32 + 8 * q + e, kc = k < 147 ? k : 0;
            const int ky = kc / 21;
            koff[s5][e] = ky * PWP + (kc - ky * 21);
        }
    const v4f b0 = *reinterpret_cast<const v4f*>(bias + 4 * q), b1 = *reinterpret_cast<const v4f*>(bias + 16 + 4 * q);
    __syncthreads();
    v4f acc[RPW][2];
#pragma unroll
    for (int t = 0; t < RPW; ++t) { acc[t][0] = b0; acc[t][1] = b1; }
    const unsigned short* wrow = w3 + (size_t)j * Kp + 8 * q;
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
        bf8 wf[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wf[nt][pl] = *reinterpret_cast<const bf8*>(wrow + (size_t)pl * plane + (size_t)nt * 16 * Kp + s5 * 32);
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
            const int py = wave * RPW + t;                               // tile row of this wave's t-th pixel row
            const float* pp = &patch[(2 * py) * PWP + (2 * j) * 3];
            v4f lo, hi;
            lo.x = pp[koff[s5][0]]; lo.y = pp[koff[s5][1]]; lo.z = pp[koff[s5][2]]; lo.w = pp[koff[s5][3]];
            hi.x = pp[koff[s5][4]]; hi.y = pp[koff[s5][5]]; hi.z = pp[koff[s5][6]]; hi.w = pp[koff[s5][7]];
            if constexpr (EXACT) {
                bf8 x0;
                x0[0] = (__bf16)lo.x; x0[1] = (__bf16)lo.y; x0[2] = (__bf16)lo.z; x0[3] = (__bf16)lo.w;
                x0[4] = (__bf16)hi.x; x0[5] = (__bf16)hi.y; x0[6] = (__bf16)hi.z; x0[7] = (__bf16)hi.w;
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)                           // smallest weight terms first
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][pl], x0, acc[t][nt], 0, 0, 0);
            } else {
                bf8 x0, x1, x2;
                split8(lo, hi, x0, x1, x2);
                const bf8* xs[3] = {&x0, &x1, &x2};
                const int wsel[6] = {2, 1, 0, 1, 0, 0}, xsel[6] = {0, 1, 2, 0, 1, 0};          // smallest terms first
#pragma unroll
                for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][wsel[p6]], *xs[xsel[p6]], acc[t][nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
        const int oy = ty0 + wave * RPW + t, ox = tx0 + j;
        if (oy < 150 && ox < 150) {
            float* yp = y + (((size_t)n * 150 + oy) * 150 + ox) * 32;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                v4f v = acc[t][nt];
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<v4f*>(yp + nt * 16 + 4 * q) = v;
            }
        }
    }
}

void launch_ssd_conv1_mfma(const uint8_t* img, const unsigned short* w3, int plane, int Kp, const float* b, float* y, int n,
                           const float in_scale[3], const float in_shift[3], bool relu, hipStream_t s) {
    bool exact = true;
    for (int c = 0; c < 3; ++c)
        exact = exact && in_scale[c] == 1.f && in_shift[c] == (float)(int)in_shift[c] && in_shift[c] >= -255.f && in_shift[c] <= 0.f;
    const dim3 grid((150 + 15) / 16, (150 + 31) / 32, n);
    if (exact)
        hipLaunchKernelGGL(ssd_conv1_mfma_kernel<true>, grid, dim3(256), 0, s, img, w3, plane, Kp, b, y, in_scale[0], in_scale[1],
                           in_scale[2], in_shift[0], in_shift[1], in_shift[2], relu ? 1 : 0);
    else
        hipLaunchKernelGGL(ssd_conv1_mfma_kernel<false>, grid, dim3(256), 0, s, img, w3, plane, Kp, b, y, in_scale[0], in_scale[1],
                           in_scale[2], in_shift[0], in_shift[1], in_shift[2], relu ? 1 : 0);
}

// ---- conv1 + ReLU + pool1 in one launch (round 3) ---------------------------------------------------------------
// The 150 x 150 x 32 conv1 map was written (184 MB per 64 frames) only for the 3x3 stride-2 max pool to read it back:
// 88 + 48 us.  Here a block owns an 8 x 8 tile of POOLED pixels = 17 x 17 conv pixels (13 % of them shared with the
// neighbours and recomputed): the conv runs as in ssd_conv1_mfma_kernel (19 sixteen-pixel MFMA column tiles over the
// block's fp32 input patch), its ReLU'd outputs go to an LDS tile that takes over the patch's storage, and a thread then
// folds the nine taps of a pooled pixel's channel quad.  Caffe's ceil-mode pooling: the last window hangs over the map's
// edge and a clamped tap re-reads an element of the window (maxpool3s2_kernel), so conv pixels outside the map are never
// read.
template <bool EXACT>
__global__ __launch_bounds__(256, 3) void ssd_conv1_pool_kernel(const uint8_t* __restrict__ img, const unsigned short* __restrict__ w3,
                                                             int plane, int Kp, const float* __restrict__ bias,
                                                             float* __restrict__ y, float sb, float sg, float sr, float hb,
                                                             float hg, float hr, int relu) {
    constexpr int PT = 8, CT = 2 * PT + 1, NPX = CT * CT, NMT = (NPX + 15) / 16, NMTW = (NMT + 3) / 4;   // 17 x 17 conv pixels, 19 tiles
    constexpr int PH = 2 * CT + 5, PW = (2 * CT + 5) * 3, PWP = PW + 1;                                    // 39 rows x 117 (+1) floats
    constexpr int CS = 36;                                                 // floats per pixel of the conv tile (32 + 4: bank spread)
    constexpr int LDSF = NPX * CS > PH * PWP ? NPX * CS : PH * PWP;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    float* patch = lds;
    float* ctile = lds;                                                    // after the last gather
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int n = blockIdx.z, py0 = blockIdx.y * PT, px0 = blockIdx.x * PT;       // pooled origin
    const int ty0 = 2 * py0, tx0 = 2 * px0;                                       // conv origin
    const uint8_t* src = img + (size_t)n * 300 * 300 * 3;
    const int r0 = 2 * ty0 - 3, c0 = (2 * tx0 - 3) * 3;
    const float sc[3] = {sb, sg, sr}, sh[3] = {hb, hg, hr};
    // patch: aligned dword loads (c0 - 3 = 96 * blockIdx.x - 12), 30 dwords cover the 117 bytes of a row
    constexpr int DW = 30, NEL = PH * DW, NLD = (NEL + 255) / 256;
    unsigned pv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        const int r = i / DW, d = i - r * DW;
        const int iy = r0 + r, ib = c0 - 3 + 4 * d;
        const bool inside = (unsigned)iy < 300u && ib >= 0 && ib < 900;
        pv[k] = *reinterpret_cast<const unsigned*>(src + (size_t)(inside ? iy : 0) * 900 + (inside ? ib : 0));
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        const int r = i / DW, d = i - r * DW;
        const int iy = r0 + r, ib0 = c0 - 3 + 4 * d;
        const bool row_in = (unsigned)iy < 300u;
        int ci = (ib0 + 900) % 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ib = ib0 + e, pc = 4 * d + e - 3;
            const bool inside = row_in && ib >= 0 && ib < 900;
            const float v = (float)((pv[k] >> (8 * e)) & 0xFFu) * sc[ci] + sh[ci];
            if (pc >= 0 && pc < PW) patch[r * PWP + pc] = inside ? v : 0.f;
            ci = ci == 2 ? 0 : ci + 1;
        }
    }
    int koff[5][8];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = s5 * 32 + 8 * q + e, kc = k < 147 ? k : 0;
            const int ky = kc / 21;
            koff[s5][e] = ky * PWP + (kc - ky * 21);
        }
    const v4f b0 = *reinterpret_cast<const v4f*>(bias + 4 * q), b1 = *reinterpret_cast<const v4f*>(bias + 16 + 4 * q);
    // this lane's pixel of each of the wave's tiles: patch offset of its top-left input sample
    int poff[NMTW];
#pragma unroll
    for (int t = 0; t < NMTW; ++t) {
        const int p = (wave + 4 * t) * 16 + j, pc = p < NPX ? p : NPX - 1;
        const int cy = pc / CT, cx = pc - cy * CT;
        poff[t] = (2 * cy) * PWP + (2 * cx) * 3;
    }
    __syncthreads();
    v4f acc[NMTW][2];
#pragma unroll
    for (int t = 0; t < NMTW; ++t) { acc[t][0] = b0; acc[t][1] = b1; }
    const unsigned short* wrow = w3 + (size_t)j * Kp + 8 * q;
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
        bf8 wf[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wf[nt][pl] = *reinterpret_cast<const bf8*>(wrow + (size_t)pl * plane + (size_t)nt * 16 * Kp + s5 * 32);
#pragma unroll
        for (int t = 0; t < NMTW; ++t) {
            if (wave + 4 * t >= NMT) continue;                             // wave-uniform (tile 19 does not exist)
            const float* pp = &patch[poff[t]];
            v4f lo, hi;
            lo.x = pp[koff[s5][0]]; lo.y = pp[koff[s5][1]]; lo.z = pp[koff[s5][2]]; lo.w = pp[koff[s5][3]];
            hi.x = pp[koff[s5][4]]; hi.y = pp[koff[s5][5]]; hi.z = pp[koff[s5][6]]; hi.w = pp[koff[s5][7]];
            if constexpr (EXACT) {
                bf8 x0;
                x0[0] = (__bf16)lo.x; x0[1] = (__bf16)lo.y; x0[2] = (__bf16)lo.z; x0[3] = (__bf16)lo.w;
                x0[4] = (__bf16)hi.x; x0[5] = (__bf16)hi.y; x0[6] = (__bf16)hi.z; x0[7] = (__bf16)hi.w;
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][pl], x0, acc[t][nt], 0, 0, 0);
            } else {
                bf8 x0, x1, x2;
                split8(lo, hi, x0, x1, x2);
                const bf8* xs[3] = {&x0, &x1, &x2};
                const int wsel[6] = {2, 1, 0, 1, 0, 0}, xsel[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][wsel[p6]], *xs[xsel[p6]], acc[t][nt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                       // every gather of the patch is done: the tile takes its place
#pragma unroll
    for (int t = 0; t < NMTW; ++t) {
        const int p = (wave + 4 * t) * 16 + j;
        if (p < NPX) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                v4f v = acc[t][nt];
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<v4f*>(&ctile[p * CS + nt * 16 + 4 * q]) = v;
            }
        }
    }
    __syncthreads();
    // pool: 64 pooled pixels x 8 channel quads = 512 items
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 256, cg = item & 7, pp = item >> 3, ppy = pp >> 3, ppx = pp & 7;
        const int oy = py0 + ppy, ox = px0 + ppx;
        v4f m = (v4f){-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int gy = 2 * oy + ky < 150 ? 2 * oy + ky : 149;          // conv row in the map, clamped (ceil mode)
            const int cy = gy - ty0 < CT ? gy - ty0 : CT - 1;              // pooled pixels past the map: any tile row (not stored)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gx = 2 * ox + kx < 150 ? 2 * ox + kx : 149;
                const int cx = gx - tx0 < CT ? gx - tx0 : CT - 1;
                const v4f v = *reinterpret_cast<const v4f*>(&ctile[((cy < 0 ? 0 : cy) * CT + (cx < 0 ? 0 : cx)) * CS + 4 * cg]);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        if (oy < 75 && ox < 75) *reinterpret_cast<v4f*>(y + (((size_t)n * 75 + oy) * 75 + ox) * 32 + 4 * cg) = m;
    }
}

bool launch_ssd_conv1_pool(const uint8_t* img, const unsigned short* w3, int plane, int Kp, const float* b, float* y_pool, int n,
                           const float in_scale[3], const float in_shift[3], bool relu, hipStream_t s) {
    bool exact = true;
    for (int c = 0; c < 3; ++c)
        exact = exact && in_scale[c] == 1.f && in_shift[c] == (float)(int)in_shift[c] && in_shift[c] >= -255.f && in_shift[c] <= 0.f;
    const dim3 grid((75 + 7) / 8, (75 + 7) / 8, n);
    if (exact)
        hipLaunchKernelGGL(ssd_conv1_pool_kernel<true>, grid, dim3(256), 0, s, img, w3, plane, Kp, b, y_pool, in_scale[0], in_scale[1],
                           in_scale[2], in_shift[0], in_shift[1], in_shift[2], relu ? 1 : 0);
    else
        hipLaunchKernelGGL(ssd_conv1_pool_kernel<false>, grid, dim3(256), 0, s, img, w3, plane, Kp, b, y_pool, in_scale[0], in_scale[1],
                           in_scale[2], in_shift[0], in_shift[1], in_shift[2], relu ? 1 : 0);
    return true;
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
// the same decision `(double)jaccard(a, b) > thr` with the areas given and the division avoided where the answer is
// clear: q = inter * rcp(union) is within a few ulp of the quotient, so only |q - thr| <= 1e-5 takes the division
__device__ __forceinline__ bool jaccard_above(const v4f a, float area_a, const v4f b, float area_b, float thr_f, double thr) {
    if (b.x > a.z || b.z < a.x || b.y > a.w || b.w < a.y) return 0.0 > thr;
    const float ix = fminf(a.z, b.z) - fmaxf(a.x, b.x), iy = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    const float inter = ix * iy, uni = area_a + area_b - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    if (fabsf(q - thr_f) > 1e-5f && uni > 0.f) return q > thr_f;
    return (double)(inter / uni) > thr;
}
__device__ __forceinline__ float jaccard(const float* a, const float* b) {
    if (b[0] > a[2] || b[2] < a[0] || b[1] > a[3] || b[3] < a[1]) return 0.f;
    const float ix = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), iy = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
    const float inter = ix * iy;
    return inter / (box_area(a) + box_area(b) - inter);
}

// One 1024-thread block per image; key = score bits (positive floats order like uints) in the high word, ~index in the
// low word: descending keys = score desc, index asc (Caffe's stable order).
//   1. priors above the confidence threshold -> keys; a 2048-bin histogram of the scores' leading bits finds the bin
//      that holds the top_k-th score, and only the keys at or above it are sorted (round 2 sorted every valid prior:
//      on a detector with many weak responses that was the full 16384-slot bitonic network, 105 barrier stages);
//   2. bitonic sort of those (a power of two >= 512), top_k = 400 candidates;
//   3. greedy NMS as a bitmask: all 400 x 400 overlap tests into 400 x 7 64-bit words by 15 waves, WHILE one wave walks
//      the candidates in order with the `removed` set in registers (lane w = word w; it waits only for the rows it reads next) - the 400 block-wide barriers of the
//      one-candidate-at-a-time loop were most of the kernel (192 us per 64 frames at VALU 0.01 busy).
// Same tests on the same values in the same order: rows identical to the round-2 kernel.
// score -> 2048 linear bins (monotone: p * 2048 is exact), and inside one bin 2048 sub-bins of width 2^-22
__device__ __forceinline__ int nms_bin(float p) { const int b = (int)(p * 2048.f); return b < 0 ? 0 : (b > 2047 ? 2047 : b); }
__device__ __forceinline__ int nms_sub(float p, int bin) {
    const int b = (int)((p * 2048.f - (float)bin) * 2048.f);         // exact for p in (0, 1]: 24-bit significand
    return b < 0 ? 0 : (b > 2047 ? 2047 : b);
}
// the lowest bin of hist[2048] whose suffix count (+ `above`) reaches `need`: wave 0, 32 bins per lane.  -> cut bin and the
// number of keys in the bins above it; (-1, total) when the whole histogram holds fewer than `need`
__device__ __forceinline__ void nms_cut(const int* hist, int lane, int above0, int need, int* cut, int* above_cut) {
    int mysum = 0;
    for (int b2 = 0; b2 < 32; ++b2) mysum += hist[lane * 32 + b2];
    int suffix = mysum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_down(suffix, off);
        if (lane + off < 64) suffix += v;
    }
    const int above = above0 + suffix - mysum;
    if (above < need && above + mysum >= need) {
        int acc = above, b2 = 31;
        for (; b2 >= 0; --b2) {
            if (acc + hist[lane * 32 + b2] >= need) break;
            acc += hist[lane * 32 + b2];
        }
        *cut = lane * 32 + (b2 < 0 ? 0 : b2);
        *above_cut = acc;
    }
}

__global__ __launch_bounds__(1024) void ssd_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ prob,
                                                       int n_priors, float conf_thr, double nms_thr, int keep_top_k,
                                                       float* __restrict__ rows, int* __restrict__ count, int spin_bound) {
    constexpr int WORDS = (NMS_TOPK + 63) / 64, SEL_CAP = 4096, MASK_AT = 512;
    __shared__ unsigned long long key[NMS_SORT];
    __shared__ unsigned long long ckey[NMS_TOPK];
    __shared__ __attribute__((aligned(16))) float cand[NMS_TOPK][4];
    __shared__ int kept[NMS_TOPK];
    __shared__ int hist[2048];
    __shared__ int n_kept, n_valid, n_sel, cut_bin, cut_above, cut_sub;
    __shared__ int rows_done[16];                                   // overlap rows finished by each producing wave
    __shared__ int spin_expired;                                    // the walker gave up waiting for a row: count[img] = -1
    static_assert(MASK_AT + NMS_TOPK * WORDS <= NMS_SORT, "the overlap words live in the sort buffer");
    const int tid = threadIdx.x, img = blockIdx.x, lane = tid & 63;
    const float* pr = prob + (size_t)img * n_priors;
    if (tid == 0) { n_valid = 0; n_sel = 0; n_kept = 0; spin_expired = 0; }
    if (tid < 16) rows_done[tid] = 0;
    for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
    unsigned long long mine[NMS_SORT / 1024];
    __syncthreads();
    int local = 0;
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r) {
        const int i = tid + r * 1024;
        unsigned long long k = 0ull;
        if (i < n_priors) {
            const float p = pr[i];
            if (p > conf_thr) {
                k = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
                atomicAdd(&hist[nms_bin(p)], 1);
                ++local;
            }
        }
        mine[r] = k;
    }
    if (local) atomicAdd(&n_valid, local);
    __syncthreads();
    // two-level select: the bin that holds the top_k-th score, then (a second histogram over that bin's keys) the sub-bin
    if (tid == 0) { cut_bin = -1; cut_above = 0; cut_sub = -1; }
    __syncthreads();
    if (tid < 64 && n_valid > NMS_TOPK) nms_cut(hist, lane, 0, NMS_TOPK, &cut_bin, &cut_above);
    __syncthreads();
    const int cb = cut_bin, ca = cut_above;
    for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
    __syncthreads();
    if (cb >= 0) {
#pragma unroll
        for (int r = 0; r < NMS_SORT / 1024; ++r)
            if (mine[r] != 0ull) {
                const float p = __uint_as_float((unsigned)(mine[r] >> 32));
                if (nms_bin(p) == cb) atomicAdd(&hist[nms_sub(p, cb)], 1);
            }
    }
    __syncthreads();
    if (tid < 64 && cb >= 0) {
        int dummy = 0;
        nms_cut(hist, lane, ca, NMS_TOPK, &cut_sub, &dummy);
    }
    __syncthreads();
    const int cs = cut_sub;
    // selected keys to the front of key[] (any order: unique keys, the sort orders them)
    int sel_local = 0;
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r)
        if (mine[r] != 0ull && cb >= 0) {
            const float p = __uint_as_float((unsigned)(mine[r] >> 32));
            const int bin = nms_bin(p);
            const bool take = bin > cb || (bin == cb && nms_sub(p, cb) >= cs);
            if (!take) mine[r] = 0ull;
        }
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r) sel_local += mine[r] != 0ull ? 1 : 0;
    int base = 0;
    if (sel_local) base = atomicAdd(&n_sel, sel_local);
#pragma unroll
    for (int r = 0; r < NMS_SORT / 1024; ++r)
        if (mine[r] != 0ull) key[base++] = mine[r];
    __syncthreads();
    const int nsel = n_sel;                                         // >= min(n_valid, top_k); a crowded cut bin can make it large
    (void)SEL_CAP;
    int sort_n = 512;
    while (sort_n < nsel) sort_n <<= 1;
    for (int i = nsel + tid; i < sort_n; i += 1024) key[i] = 0ull;
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
    // candidates: the first min(top_k, #valid) keys, with their boxes
    const int ncand = nsel < NMS_TOPK ? nsel : NMS_TOPK;
    for (int i = tid; i < NMS_TOPK; i += 1024) {
        const unsigned long long k = i < ncand ? key[i] : 0ull;
        ckey[i] = k;
        if (k != 0ull) {
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const float* b = boxes + ((size_t)img * n_priors + idx) * 4;
            cand[i][0] = b[0]; cand[i][1] = b[1]; cand[i][2] = b[2]; cand[i][3] = b[3];
        }
    }
    __syncthreads();                                                // key[] beyond the candidates is free from here on
    // overlap words: bit (j & 63) of ovl[i][j >> 6] = candidate j > i overlaps candidate i above the threshold.  A wave
    // takes a row i, its lanes 64 consecutive j: one box per lane (conflict-free 16-byte LDS reads), one ballot per word.
    // (One thread per (row, word) with a 64-step loop read boxes 1 KB apart from every lane - the same LDS bank - and
    // took 153k of the kernel's 270k cycles.)
    // The mask and the walk run TOGETHER: waves 1-15 produce the rows in increasing order (wave v: rows v - 1, v + 14, ...)
    // and publish how many each has finished; wave 0 walks the candidates and waits - inside the workgroup, on LDS words -
    // only for the eight rows it is about to read.  As two phases behind a barrier they were ~60k + ~65k cycles of a block
    // that owns one CU; the producers never wait for anything, so the walker's wait always ends.
    unsigned long long* ovl = key + MASK_AT;
    {
        const int wave = tid >> 6;
        const float thr_f = (float)nms_thr;
        if (wave > 0) {
            int done = 0;
            for (int i = wave - 1; i < ncand; i += 15) {
                const v4f bi = *reinterpret_cast<const v4f*>(cand[i]);
                const float ai = box_area(cand[i]);
                for (int w = i >> 6; w < WORDS; ++w) {
                    const int j = w * 64 + lane;
                    const int jc = j < ncand ? j : i;
                    const v4f bj = *reinterpret_cast<const v4f*>(cand[jc]);
                    const float b4[4] = {bj.x, bj.y, bj.z, bj.w};
                    const bool hit = j > i && j < ncand && jaccard_above(bi, ai, bj, box_area(b4), thr_f, nms_thr);
                    const unsigned long long bits = __ballot(hit);
                    if (lane == 0) ovl[i * WORDS + w] = bits;
                }
                if (lane < (i >> 6)) ovl[i * WORDS + lane] = 0ull;        // words entirely below the diagonal
                ++done;
                // the row's words are written (LDS operations of a wave complete in order); publish the count
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&rows_done[wave], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {                                                    // one wave: lane w holds word w of the removed set
            unsigned long long removed = 0ull;
            int nk = 0;
            for (int w = 0; w * 64 < ncand; ++w) {
                // the word that covers candidates 64 w .. 64 w + 63, wave-uniform (two readlanes: no LDS round trip)
                unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(removed >> 32), w) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane((int)(removed & 0xFFFFFFFFull), w);
                for (int b0 = 0; b0 < 64 && w * 64 + b0 < ncand; b0 += 8) {
                    // rows i0 .. i0 + 7 must have been published by their producers (row i: wave i % 15 + 1, its
                    // (i / 15 + 1)-th row); lane e < 8 checks row i0 + e.  The spin is bounded: a guard against a hang,
                    // not a code path - producers cannot stall.  If the bound ever expires the rows below were NOT
                    // published: the walk goes on (every wave must reach the barrier) but the image's count becomes -1
                    // and the host turns that into DFD_ERR_HIP - wrong boxes never leave with rc 0
                    {
                        const int i = w * 64 + b0 + (lane & 7);
                        const bool mine = lane < 8 && i < ncand;
                        bool ready = false;
                        for (int spin = 0; spin < spin_bound; ++spin) {
                            const int have = mine ? __hip_atomic_load(&rows_done[i % 15 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
                            if (!__any(mine && have <= i / 15)) { ready = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!ready && lane == 0) spin_expired = 1;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    unsigned long long rows8[8];                        // eight rows requested together: one LDS latency per eight steps
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int i = w * 64 + b0 + e;
                        rows8[e] = (lane < WORDS && i < ncand) ? ovl[i * WORDS + lane] : 0ull;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int b2 = b0 + e, i = w * 64 + b2;
                        if (i < ncand && !((cur >> b2) & 1ull)) {       // wave-uniform
                            if (lane == 0) kept[nk] = i;
                            ++nk;
                            removed |= rows8[e];
                            cur |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(rows8[e] >> 32), w) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(rows8[e] & 0xFFFFFFFFull), w);
                        }
                    }
                }
            }
            if (lane == 0) n_kept = nk;
        }
    }
    __syncthreads();
    const int nk = n_kept < keep_top_k ? n_kept : keep_top_k;
    for (int r = tid; r < nk; r += 1024) {
        const int i = kept[r];
        float* o = rows + ((size_t)img * keep_top_k + r) * 5;
        o[0] = __uint_as_float((unsigned)(ckey[i] >> 32));
        o[1] = cand[i][0]; o[2] = cand[i][1]; o[3] = cand[i][2]; o[4] = cand[i][3];
    }
    if (tid == 0) count[img] = spin_expired ? -1 : nk;
}

void launch_ssd_nms(const float* boxes, const float* prob, int n, int n_priors, float conf_thr, double nms_thr,
                    int keep_top_k, float* rows, int* count, hipStream_t s) {
    // DFD_NMS_SPIN_BOUND: the walker's wait bound in polls (default 2^22 ~ seconds); 0 = "expired at once", the
    // switch tests/test_ssd_gpu.py uses to prove that an expired wait is reported and not walked over
    static const int spin_bound = getenv("DFD_NMS_SPIN_BOUND") ? atoi(getenv("DFD_NMS_SPIN_BOUND")) : (1 << 22);
    hipLaunchKernelGGL(ssd_nms_kernel, dim3(n), dim3(1024), 0, s, boxes, prob, n_priors, conf_thr, nms_thr, keep_top_k,
                       rows, count, spin_bound);
}

}  // namespace dfd
