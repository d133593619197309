// Kernels of the MTCNN align/crop stage (reference deepfake_detection.py:376-380 calls
// facenet-pytorch's MTCNN.forward on the cropped face).  NHWC fp32, weights laid out [ci][ky][kx][co].  The layers with
// 3-16 input channels do not fill an MFMA K and run on the VALU, register-blocked; every later convolution and the wide
// FC layers run on the split-precision MFMA GEMM (gemm_split.hip) from mtcnn_api.hip.
//   mt_area_resize_ragged / _multi   interpolate(mode="area") (= adaptive average pool) of u8 BGR windows -> RGB fp32,
//                                    (x - 127.5) * 0.0078125; exact: window sums are integers < 2^24
//   mt_pnet_conv1_pool               P-Net conv1 + PReLU + MaxPool(2, 2, ceil) over all pyramid levels of all crops
//   mt_convpx (ragged)               P-Net conv2, conv3 (+ both heads, softmax, candidate compaction)
//   mt_conv1_pool                    R-/O-Net conv1 + PReLU + MaxPool(3, 2, ceil)
//   mt_maxpool                       MaxPool2d(k, s, ceil_mode=True) behind the GEMM convolutions
//   mt_heads                         the 2- / 4-wide heads of R-/O-Net + softmax in one launch (sixteen lanes per window)
//   mt_extract_h / _v                extract_face of all crops: Pillow's 8-bit fixed-point Image.resize passes
// Every convolution accumulates its products as fmaf in (ci, ky, kx) order, then + bias, then PReLU - the order a
// direct convolution over the weight tensor [co][ci][ky][kx] walks.
#include "mtcnn_kernels.h"
#include "kernel_util.h"

#include <cstdlib>

namespace dfd {

// thread = output pixel x 4 channels (c % 4 == 0: every pooled map of R-/O-Net has 32 or 64 channels; other widths take
// the scalar path).  Ceil mode: a tap outside the map re-reads the window's first element (a max is idempotent), so
// all k x k vector loads are unconditional and in flight together.
template <int V>
__global__ __launch_bounds__(256) void mt_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int ih,
                                                         int iw, int c, int k, int st, int oh, int ow) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cv = c / V;
    const long long total = (long long)n * oh * ow * cv;
    if (t >= total) return;
    const int ch = (int)(t % cv) * V;
    const int ox = (int)((t / cv) % ow), oy = (int)((t / cv / ow) % oh), i = (int)(t / cv / ow / oh);
    const float* xi = x + (size_t)i * ih * iw * c + ch;
    float m[V];
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = -INFINITY;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            int yy = oy * st + ky, xx = ox * st + kx;
            if (yy >= ih) yy = oy * st;                                  // clipped window: repeat an element of it
            if (xx >= iw) xx = ox * st;
            const float* p = xi + ((size_t)yy * iw + xx) * c;
            if (V == 4) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                m[0] = fmaxf(m[0], v.x); m[1 % V] = fmaxf(m[1 % V], v.y); m[2 % V] = fmaxf(m[2 % V], v.z); m[3 % V] = fmaxf(m[3 % V], v.w);
            } else {
                m[0] = fmaxf(m[0], p[0]);
            }
        }
    float* yp = y + (((size_t)i * oh + oy) * ow + ox) * c + ch;
    if (V == 4) *reinterpret_cast<float4*>(yp) = make_float4(m[0], m[1 % V], m[2 % V], m[3 % V]);
    else yp[0] = m[0];
}

int mt_pool_out(int in, int k, int st) {
    int o = (in - k + st - 1) / st + 1;              // ceil((in - k) / st) + 1
    if ((o - 1) * st >= in) --o;                     // PyTorch: the last window must start inside the input
    return o;
}

void launch_mt_maxpool(const float* x, float* y, int n, int ih, int iw, int c, int k, int st, hipStream_t s) {
    const int oh = mt_pool_out(ih, k, st), ow = mt_pool_out(iw, k, st);
    if (c % 4 == 0) {
        const long long total = (long long)n * oh * ow * (c / 4);
        if (total <= 0) return;
        hipLaunchKernelGGL(mt_maxpool_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, n, ih, iw, c, k, st, oh, ow);
    } else {
        const long long total = (long long)n * oh * ow * c;
        if (total <= 0) return;
        hipLaunchKernelGGL(mt_maxpool_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, n, ih, iw, c, k, st, oh, ow);
    }
}

// The two output heads of R-Net / O-Net in ONE launch: face probability = softmax(f . W1 + b1)[1], regression = f . W2 + b2
// for `n` feature rows f[in] (in = 128 / 256).  Sixteen lanes share a row: lane l sums the terms c = l, l + 16, ... of all six
// dot products, four xor-shuffles fold them, lane 0 of the group finishes.  As three launches (a thread per output
// element walking `in` dependent FMAs, softmax, again) the heads took 36 + 64 us per step for a few hundred windows.
__global__ __launch_bounds__(256) void mt_heads_kernel(const float* __restrict__ f, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ prob,
                                                       float* __restrict__ reg, int n, int in) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    constexpr int LANES = 16;
    const int l = (int)(t & (LANES - 1));
    const long long row = t / LANES;
    const long long rc = row < n ? row : (long long)n - 1;     // surplus groups redo the last row (uniform shuffles)
    const float* fp = f + (size_t)rc * in;
    float z0 = 0.f, z1 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    for (int c = l; c < in; c += LANES) {
        const float v = fp[c];
        const float2 a = *reinterpret_cast<const float2*>(w1 + (size_t)c * 2);
        const float4 q = *reinterpret_cast<const float4*>(w2 + (size_t)c * 4);
        z0 = fmaf(v, a.x, z0); z1 = fmaf(v, a.y, z1);
        r0 = fmaf(v, q.x, r0); r1 = fmaf(v, q.y, r1); r2 = fmaf(v, q.z, r2); r3 = fmaf(v, q.w, r3);
    }
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) {
        z0 += __shfl_xor(z0, off); z1 += __shfl_xor(z1, off);
        r0 += __shfl_xor(r0, off); r1 += __shfl_xor(r1, off); r2 += __shfl_xor(r2, off); r3 += __shfl_xor(r3, off);
    }
    if (l == 0 && row < n) {
        z0 += b1[0]; z1 += b1[1];
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        prob[row] = e1 / (e0 + e1);
        *reinterpret_cast<float4*>(reg + (size_t)row * 4) = make_float4(r0 + b2[0], r1 + b2[1], r2 + b2[2], r3 + b2[3]);
    }
}

void launch_mt_heads(const float* f, const float* w1, const float* b1, const float* w2, const float* b2, float* prob, float* reg,
                     int n, int in, hipStream_t s) {
    if (n <= 0) return;
    const long long threads = (long long)n * 16;
    hipLaunchKernelGGL(mt_heads_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, f, w1, b1, w2, b2, prob, reg, n, in);
}

// extract_face for all crops of a step in two launches (blockIdx.y = crop): the horizontal pass of every crop whose
// window is not 160 wide, then the vertical pass / plain copy / zero fill that completes each 160x160x3 face slot.
// Per element the arithmetic is mt_pil_pass_kernel's.
__device__ __forceinline__ uint8_t pil_tap_sum(const uint8_t* __restrict__ p, size_t step, const int* __restrict__ kk, int cnt) {
    int ss = 1 << 21;
    for (int k = 0; k < cnt; ++k) ss += kk[k] * (int)p[(size_t)k * step];
    int v = ss >> 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void mt_extract_h_kernel(const MtFaceJob* __restrict__ jobs, const int* __restrict__ tables,
                                                           uint8_t* __restrict__ faces, uint8_t* __restrict__ tmp) {
    const MtFaceJob j = jobs[blockIdx.y];
    if (!j.found || j.cw == 160) return;
    uint8_t* dst = j.ch == 160 ? faces + (size_t)blockIdx.y * 160 * 160 * 3 : tmp + j.tmp_off;
    const int total = j.ch * 160 * 3;
    const int* coeff = tables + j.cx;
    const int* bounds = tables + j.bx;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int c = t % 3, col = (t / 3) % 160, row = t / 3 / 160;
        const int xmin = bounds[2 * col], cnt = bounds[2 * col + 1];
        dst[t] = pil_tap_sum(j.src + (size_t)(j.y1 + row) * j.stride + (size_t)(j.x1 + xmin) * 3 + c, 3, coeff + (size_t)col * j.kx, cnt);
    }
}

__global__ __launch_bounds__(256) void mt_extract_v_kernel(const MtFaceJob* __restrict__ jobs, const int* __restrict__ tables,
                                                           uint8_t* __restrict__ faces, const uint8_t* __restrict__ tmp) {
    const MtFaceJob j = jobs[blockIdx.y];
    uint8_t* dst = faces + (size_t)blockIdx.y * 160 * 160 * 3;
    const int total = 160 * 160 * 3;
    if (j.found && j.ch == 160 && j.cw != 160) return;                      // completed by the horizontal pass
    const bool resized = j.cw != 160;                                       // the vertical pass reads tmp [ch][160][3]
    const uint8_t* src = resized ? tmp + j.tmp_off : j.src + (size_t)j.y1 * j.stride + (size_t)j.x1 * 3;
    const size_t sstride = resized ? 160 * 3 : (size_t)j.stride;
    const int* coeff = tables + j.cy;
    const int* bounds = tables + j.by;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int c = t % 3, col = (t / 3) % 160, row = t / 3 / 160;
        uint8_t v = 0;
        if (j.found) {
            if (j.ch == 160) v = src[(size_t)row * sstride + (size_t)col * 3 + c];                 // 160 x 160 window: copy
            else {
                const int ymin = bounds[2 * row], cnt = bounds[2 * row + 1];
                v = pil_tap_sum(src + (size_t)ymin * sstride + (size_t)col * 3 + c, sstride, coeff + (size_t)row * j.ky, cnt);
            }
        }
        dst[t] = v;
    }
}

void launch_mt_extract_faces(const MtFaceJob* jobs_dev, int n, const int* tables_dev, uint8_t* faces, uint8_t* tmp, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mt_extract_h_kernel, dim3(64, n), dim3(256), 0, s, jobs_dev, tables_dev, faces, tmp);
    hipLaunchKernelGGL(mt_extract_v_kernel, dim3(64, n), dim3(256), 0, s, jobs_dev, tables_dev, faces, tmp);
}

// u8 BGR [n][hw][3] -> float RGB CHW (0..255), the tensor MTCNN.forward returns with post_process=False
__global__ __launch_bounds__(256) void mt_face_chw_kernel(const uint8_t* __restrict__ bgr, float* __restrict__ out, int hw) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * hw) return;
    const int c = t / hw, p = t % hw;
    out[t] = (float)bgr[(size_t)p * 3 + (2 - c)];
}

void launch_mt_face_chw(const uint8_t* bgr, float* out, int hw, hipStream_t s) {
    hipLaunchKernelGGL(mt_face_chw_kernel, dim3((3 * hw + 255) / 256), dim3(256), 0, s, bgr, out, hw);
}

// ------------------------------------------------------------------------------------------------ ragged
// Stage 1 runs P-Net over every pyramid level of every crop of a step: a few hundred small images of different
// sizes.  One launch per layer covers them all: `items` describes each image (offsets into the layer's input and
// output arenas, input size), `pre` is the running total of output elements; a thread finds its image by binary
// search.  Arithmetic per element is exactly that of the single-image kernels above.
__device__ __forceinline__ int mt_find(const long long* __restrict__ pre, int n, long long t) {
    int lo = 0, hi = n;                     // pre[lo] <= t < pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The same search by one WAVE for two elements at once (lanes 0-31: t0, lanes 32-63: t1): every round the 32 lanes of a
// half probe 32 evenly spaced entries of the current range together and a ballot picks the sub-range - two or three
// rounds of loads for n <= 32768 instead of log2(n) dependent ones.  The binary search by threads 0 and 1 in front of
// the block's first barrier was 11 dependent loads for ~2000 pyramid levels: ~5 us during which none of the block's
// other waves could start (s_memtime; a third of the pyramid-resize launch).  Call with all 64 lanes of a wave;
// returns the item of t0 in every lane's .x, of t1 in .y.
__device__ __forceinline__ int2 mt_find2_wave(const long long* __restrict__ pre, int n, long long t0, long long t1) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const long long t = half ? t1 : t0;
    int lo = 0, hi = n;                                            // pre[lo] <= t < pre[hi], per half
    while (__any(hi - lo > 1)) {
        const int span = hi - lo, step = (span + 31) >> 5;         // >= 1
        const int idx = lo + l32 * step;
        const bool le = idx < hi && pre[idx] <= t;                 // monotone in l32; lane 0 of a half is always true
        const unsigned long long m = __ballot(le);
        const unsigned mh = (unsigned)(half ? (m >> 32) : (m & 0xFFFFFFFFull));
        const int k = __popc(mh) - 1;                              // last probe at or below t
        if (hi - lo > 1) {
            const int nlo = lo + k * step, nhi = nlo + step < hi ? nlo + step : hi;
            lo = nlo;
            hi = nhi;
        }
    }
    const int r0 = __shfl(lo, 0), r1 = __shfl(lo, 32);
    return make_int2(r0, r1);
}

// Exact integer sums of the three channels over the window rows [y0, y1) x pixels [x0, x1) of a packed BGR image.
// A row's 3 * (x1 - x0) bytes are read as unaligned dwords, 12 bytes (four pixels) per step, instead of one byte per
// load: the ragged pyramid resize issued 740 M byte loads per 256 crops (every level reads the whole crop) and was bound
// by load instructions.  `safe_rows`: rows below this index may read up to 3 bytes past their last pixel (they lie inside
// the image buffer); the rows from it on fall back to byte loads.
__device__ __forceinline__ void mt_window_sums(const uint8_t* __restrict__ img, size_t stride, int y0, int y1, int x0, int x1,
                                               int safe_rows, int& sb, int& sg, int& sr) {
    const int nbytes = 3 * (x1 - x0);
    for (int y = y0; y < y1; ++y) {
        const uint8_t* p = img + (size_t)y * stride + (size_t)x0 * 3;
        if (y < safe_rows) {
            int k = 0;
            for (; k + 12 <= nbytes; k += 12) {                     // b g r b | g r b g | r b g r
                unsigned d0, d1, d2;
                memcpy(&d0, p + k, 4); memcpy(&d1, p + k + 4, 4); memcpy(&d2, p + k + 8, 4);
                sb += (d0 & 0xFF) + (d0 >> 24) + ((d1 >> 16) & 0xFF) + ((d2 >> 8) & 0xFF);
                sg += ((d0 >> 8) & 0xFF) + (d1 & 0xFF) + (d1 >> 24) + ((d2 >> 16) & 0xFF);
                sr += ((d0 >> 16) & 0xFF) + ((d1 >> 8) & 0xFF) + (d2 & 0xFF) + (d2 >> 24);
            }
            // tail: 3, 6 or 9 bytes (one to three pixels) from up to three dwords, bytes past the window masked
            if (k < nbytes) {
                const int rem = nbytes - k;                         // 3, 6, 9
                unsigned d0, d1 = 0, d2 = 0;
                memcpy(&d0, p + k, 4);
                if (rem > 4) memcpy(&d1, p + k + 4, 4);
                if (rem > 8) memcpy(&d2, p + k + 8, 4);
                sb += (d0 & 0xFF);
                sg += ((d0 >> 8) & 0xFF);
                sr += ((d0 >> 16) & 0xFF);
                if (rem > 3) { sb += (d0 >> 24); sg += (d1 & 0xFF); sr += ((d1 >> 8) & 0xFF); }
                if (rem > 6) { sb += ((d1 >> 16) & 0xFF); sg += (d1 >> 24); sr += (d2 & 0xFF); }
            }
        } else {
            for (int x = 0; x < nbytes; x += 3) { sb += p[x]; sg += p[x + 1]; sr += p[x + 2]; }
        }
    }
}

// The item that holds element t, searched only between the items of the block's first and last element (lohi[0..1],
// found by two threads with the full search and shared through LDS): a block's 256 consecutive elements span one or two
// pyramid levels, so the per-thread search is 0-2 steps instead of log2(n) = 11 dependent loads of `pre`.
__device__ __forceinline__ int mt_find_in(const long long* __restrict__ pre, const int* lohi, long long t) {
    int lo = lohi[0], hi = lohi[1] + 1;                        // pre[lo] <= t < pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one thread per output pixel (three channel sums share the window walk and the index arithmetic; the sums are exact
// integers below 2^24, so their order is free); pre = running totals of output ELEMENTS (pixels x 3)
__global__ __launch_bounds__(256) void mt_area_resize_ragged_kernel(const MtLevel* __restrict__ lv,
                                                                    const long long* __restrict__ pre, int n,
                                                                    float* __restrict__ dst) {
    __shared__ int lohi[2];
    // XCD-aware block order (guide T1, bijective form): the eight pyramid levels of a crop each read the WHOLE crop, and
    // outputs are ordered crop by crop - with the dispatcher's round-robin every XCD read every crop (8 x 8 fetches of
    // it); ids that share an XCD now walk a contiguous range of the output space, i.e. a few crops, whose bytes stay in
    // that XCD's L2 across their levels
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, qq = nb >> 3, rr = nb & 7;
    const unsigned bidx = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    const long long t = (long long)bidx * 256 + threadIdx.x;
    if (threadIdx.x < 64) {
        const long long tot = pre[n], e0 = (long long)bidx * 256 * 3, e1 = ((long long)bidx * 256 + 255) * 3;
        const int2 f = mt_find2_wave(pre, n, e0 < tot ? e0 : tot - 1, e1 < tot ? e1 : tot - 1);
        if (threadIdx.x == 0) { lohi[0] = f.x; lohi[1] = f.y; }
    }
    __syncthreads();
    if (t * 3 >= pre[n]) return;
    const int i = mt_find_in(pre, lohi, t * 3);
    const MtLevel L = lv[i];
    // 32-bit index arithmetic: a level has < 2^31 pixels and (pixel index) x (image edge) < 2^31 for any image this path
    // accepts (edges < 2^15) - the 64-bit divisions of the first version were several hundred instructions per thread,
    // more than the window loads
    const long long r = t - pre[i] / 3;
    const unsigned ru = (unsigned)r, uow = (unsigned)L.ow, uoh = (unsigned)L.oh;
    const unsigned oyu = ru / uow, oxu = ru - oyu * uow;
    const int y0 = (int)(oyu * (unsigned)L.h / uoh), y1 = (int)(((oyu + 1u) * (unsigned)L.h + uoh - 1u) / uoh);
    const int x0 = (int)(oxu * (unsigned)L.w / uow), x1 = (int)(((oxu + 1u) * (unsigned)L.w + uow - 1u) / uow);
    int sb = 0, sg = 0, sr = 0;
    mt_window_sums(L.src, (size_t)L.stride, y0, y1, x0, x1, L.h - 1, sb, sg, sr);      // the last row: byte loads (nothing behind it)
    const float area = (float)((y1 - y0) * (x1 - x0));
    float* d = dst + L.out_off + r * 3;                      // RGB order
    d[0] = ((float)sr / area - 127.5f) * 0.0078125f;
    d[1] = ((float)sg / area - 127.5f) * 0.0078125f;
    d[2] = ((float)sb / area - 127.5f) * 0.0078125f;
}

void launch_mt_area_resize_ragged(const MtLevel* lv_dev, const long long* pre_dev, int n, long long total, float* dst,
                                  hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_area_resize_ragged_kernel, dim3((unsigned)((total / 3 + 255) / 256)), dim3(256), 0, s, lv_dev, pre_dev, n, dst);
}

// Register-blocked valid convolution, one thread per output pixel (P pixels per thread, all CO channels of each in
// registers), weights [ci][ky][kx][co] staged once per block in LDS and read as wave-uniform broadcasts.  Replaces the
// thread-per-output-element kernels above for the layers that carry the cascade's arithmetic (P-Net on every pyramid
// level of every crop; the 3 -> 32 first convolution of R-/O-Net on every candidate window): those issue two loads
// per FMA, this one CI*K*K activation loads per CI*K*K*CO FMAs.  Every output still accumulates its products as
// fmaf in (ci, ky, kx) order, then + bias, then PReLU (the order of the one-thread-per-output kernel it replaced: same bits).
//   items != null: ragged launch (pre = running totals of output ELEMENTS, as for the kernels above);
//   items == null: `n` maps of ih x iw.
//   HEADS (P-Net conv3, CO = 32): the two 1x1 heads and the softmax are evaluated from the registers - the launch
//   writes prob [cell] and reg [cell][4] and never stores the 32-channel map.
template <int CI, int CO, int K, int P, bool HEADS>
__global__ __launch_bounds__(256) void mt_convpx_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ slope,
                                                        float* __restrict__ y, const MtItem* __restrict__ items,
                                                        const long long* __restrict__ pre, int n, int ih_u, int iw_u,
                                                        const float* __restrict__ w41, const float* __restrict__ b41,
                                                        const float* __restrict__ w42, const float* __restrict__ b42,
                                                        float* __restrict__ prob, float* __restrict__ reg,
                                                        MtCand* __restrict__ cand, unsigned* __restrict__ cand_count,
                                                        unsigned cand_cap, float thr) {
    constexpr int CC = CI % 2 == 0 ? 2 : 1;                  // channels per activation load
    constexpr int NW = CI * K * K * CO;
    __shared__ __attribute__((aligned(16))) float sw[NW + (HEADS ? CO * 6 + 8 : 0)];
    __shared__ int lohi[P][2];
    if (items && threadIdx.x < 2 * P) {                        // the items of each pixel slot's first / last element of this block
        const int p = threadIdx.x >> 1, e = threadIdx.x & 1;
        const long long pix = (long long)blockIdx.x * 256 + (e ? 255 : 0) + (long long)p * gridDim.x * 256;
        const long long last = pre[n] / CO - 1;
        lohi[p][e] = mt_find(pre, n, (pix < last ? pix : last) * CO);
    }
    for (int i = threadIdx.x; i < NW; i += 256) sw[i] = w[i];
    if (HEADS) {
        for (int i = threadIdx.x; i < CO * 2; i += 256) sw[NW + i] = w41[i];
        for (int i = threadIdx.x; i < CO * 4; i += 256) sw[NW + CO * 2 + i] = w42[i];
        if (threadIdx.x < 2) sw[NW + CO * 6 + threadIdx.x] = b41[threadIdx.x];
        if (threadIdx.x < 4) sw[NW + CO * 6 + 2 + threadIdx.x] = b42[threadIdx.x];
    }
    __syncthreads();
    const int oh_u = ih_u - K + 1, ow_u = iw_u - K + 1;
    const long long npix = items ? pre[n] / CO : (long long)n * oh_u * ow_u;
    const long long G = (long long)gridDim.x * 256, g = (long long)blockIdx.x * 256 + threadIdx.x;
    const float* xp[P];
    long long opix[P];                                       // output pixel index in the arena (element offset / CO)
    int iw[P];
    bool ok[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const long long pix = g + p * G;
        ok[p] = pix < npix;
        const long long q = ok[p] ? pix : 0;
        if (items) {
            const int i = ok[p] ? mt_find_in(pre, lohi[p], q * CO) : 0;
            const MtItem it = items[i];
            const long long r = q - pre[i] / CO;
            const int ow = it.iw - K + 1;
            const unsigned oyu = (unsigned)r / (unsigned)ow;          // a level has < 2^31 pixels: 32-bit division
            const int oy = (int)oyu, ox = (int)((unsigned)r - oyu * (unsigned)ow);
            iw[p] = it.iw;
            xp[p] = x + it.in_off + ((size_t)oy * it.iw + ox) * CI;
            opix[p] = it.out_off / CO + r;
        } else {
            const long long per = (long long)oh_u * ow_u;
            const long long i = q / per;
            const int r = (int)(q % per), oy = r / ow_u, ox = r % ow_u;
            iw[p] = iw_u;
            xp[p] = x + ((i * ih_u + oy) * iw_u + ox) * CI;
            opix[p] = q;
        }
    }
    float acc[P][CO];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[p][o] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < CI; c0 += CC) {
        float xv[K][K][P][CC];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float* q = xp[p] + ((size_t)ky * iw[p] + kx) * CI + c0;
                    if (CC == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(q);
                        xv[ky][kx][p][0] = v.x;
                        xv[ky][kx][p][CC - 1] = v.y;
                    } else {
                        xv[ky][kx][p][0] = *q;
                    }
                }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* wr = sw + (((c0 + cc) * K + ky) * K + kx) * CO;
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        const float wv = wr[o];
#pragma unroll
                        for (int p = 0; p < P; ++p) acc[p][o] = fmaf(xv[ky][kx][p][cc], wv, acc[p][o]);
                    }
                }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            float v = acc[p][o] + b[o];
            if (slope) v = v >= 0.f ? v : v * slope[o];
            acc[p][o] = v;
        }
        if (!HEADS && !ok[p]) continue;
        if (HEADS) {                                        // (every lane runs this block: the append below uses a ballot)
            const float* h41 = sw + NW;
            const float* h42 = sw + NW + CO * 2;
            float z[2] = {0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CO; ++c) {
#pragma unroll
                for (int j = 0; j < 2; ++j) z[j] = fmaf(acc[p][c], h41[c * 2 + j], z[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = fmaf(acc[p][c], h42[c * 4 + j], r4[j]);
            }
            const float z0 = z[0] + sw[NW + CO * 6], z1 = z[1] + sw[NW + CO * 6 + 1];
            const float m = fmaxf(z0, z1);
            const float e0 = expf(z0 - m), e1 = expf(z1 - m);
            const float pf = e1 / (e0 + e1);
            float4 rv;
            rv.x = r4[0] + sw[NW + CO * 6 + 2]; rv.y = r4[1] + sw[NW + CO * 6 + 3];
            rv.z = r4[2] + sw[NW + CO * 6 + 4]; rv.w = r4[3] + sw[NW + CO * 6 + 5];
            if (ok[p]) {
                prob[opix[p]] = pf;
                *reinterpret_cast<float4*>(reg + opix[p] * 4) = rv;
            }
            // one atomic per wave: the lanes that pass take consecutive slots behind the wave's base (with a dense
            // funnel tens of thousands of per-lane atomics on one counter held this kernel at 0.7 of its cycles waiting)
            const bool pass = cand && ok[p] && pf >= thr;
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(cand_count, (unsigned)__popcll(mask));
                base = __shfl(base, leader);
                const unsigned slot = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                if (pass && slot < cand_cap) cand[slot] = MtCand{(unsigned)opix[p], pf, {rv.x, rv.y, rv.z, rv.w}};
            }
        } else {
            float* yp = y + opix[p] * CO;
            if (CO % 4 == 0) {
#pragma unroll
                for (int o = 0; o < CO; o += 4)
                    *reinterpret_cast<float4*>(yp + o) = make_float4(acc[p][o], acc[p][o + 1], acc[p][o + 2], acc[p][o + 3]);
            } else {
#pragma unroll
                for (int o = 0; o < CO; o += 2) *reinterpret_cast<float2*>(yp + o) = make_float2(acc[p][o], acc[p][o + 1]);
            }
        }
    }
}

// P-Net conv1 (3 -> 10, 3x3) + PReLU + MaxPool2d(2, 2, ceil_mode) over every pyramid level of every crop in one ragged
// launch: a thread owns one pooled pixel = the 2x2 convolution pixels under it (no overlap at stride 2: nothing is
// recomputed), 4 x 10 accumulators, the 4x4x3 input patch in registers, the 270 weights as SGPR operands (sload16s, one
// group of 16 ahead; `w` must be readable up to 272 floats).  Products accumulate as fmaf in (ci, ky, kx) order, then
// + bias, PReLU, max over the pixels that exist (ceil mode clips the last row / column): the bits of an unfused convolution + the
// separate pool.  items[i] = {pyramid offset, pooled-map offset (elements), level height, width}; pre = running totals
// of pooled ELEMENTS (pixels x 10).  The 10-channel conv map (1.4 GB per 256 crops, written and read back) is gone.
__global__ __launch_bounds__(256) void mt_pnet_conv1_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, const float* __restrict__ slope,
                                                                 float* __restrict__ y, const MtItem* __restrict__ items,
                                                                 const long long* __restrict__ pre, int n) {
    constexpr int CO = 10;
    __shared__ int lohi[2];
    const long long npix = pre[n] / CO;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long q = t < npix ? t : npix - 1;             // surplus threads redo the last pixel (uniform control flow)
    if (threadIdx.x < 64) {
        const long long e0 = (long long)blockIdx.x * 256, e1 = e0 + 255;
        const int2 f = mt_find2_wave(pre, n, (e0 < npix ? e0 : npix - 1) * CO, (e1 < npix ? e1 : npix - 1) * CO);
        if (threadIdx.x == 0) { lohi[0] = f.x; lohi[1] = f.y; }
    }
    __syncthreads();
    const int i = mt_find_in(pre, lohi, q * CO);
    const MtItem it = items[i];
    const long long r = q - pre[i] / CO;
    const int c1h = it.ih - 2, c1w = it.iw - 2;
    int pw = (c1w - 2 + 1) / 2 + 1;                          // MaxPool2d(2, 2, ceil_mode): as mt_pool_out
    if ((pw - 1) * 2 >= c1w) --pw;
    const unsigned pyu = (unsigned)r / (unsigned)pw;                 // a level has < 2^31 pixels: 32-bit division
    const int py = (int)pyu, px = (int)((unsigned)r - pyu * (unsigned)pw);
    const bool vy = 2 * py + 1 < c1h, vx = 2 * px + 1 < c1w;
    const float* xi = x + it.in_off;
    float patch[4][4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int row = min(2 * py + a, it.ih - 1);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int col = min(2 * px + c4, it.iw - 1);
            const float* p = xi + ((size_t)row * it.iw + col) * 3;
            patch[a][c4][0] = p[0]; patch[a][c4][1] = p[1]; patch[a][c4][2] = p[2];
        }
    }
    float acc[4][CO];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[p][o] = 0.f;
    // 17 groups of 16 weights (flat index = tap * 10 + channel), two register groups used in turn: while one feeds the
    // FMAs the other is in flight.  Two output channels per instruction: v_pk_fma_f32 with the SGPR weight pair as one
    // operand (each half an IEEE fma: the bits of fmaf); 10 and 16 are even, so (idx, idx + 1) is a channel pair of one
    // tap inside one group.
    auto use = [&](const sf16& wc, int g) {
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const int idx = g * 16 + j;
            if (idx < 270) {
                const int tap = idx / CO, o = idx % CO, c = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xv = patch[(p >> 1) + ky][(p & 1) + kx][c];
                    const v2f x2 = {xv, xv};
                    v2f a = {acc[p][o], acc[p][o + 1]};
                    a = __builtin_elementwise_fma(x2, (v2f){wc[j], wc[j + 1]}, a);
                    acc[p][o] = a.x; acc[p][o + 1] = a.y;
                }
            }
        }
    };
    sf16 wa = sload16<0>(w), wb;
#pragma unroll
    for (int g = 0; g < 17; g += 2) {
        swait1(wa);
        if (g + 1 < 17) wb = sload16s(w, (g + 1) * 64);
        use(wa, g);
        __builtin_amdgcn_sched_barrier(0);                   // the group's FMAs stay before the next wait / load
        if (g + 1 < 17) {
            swait1(wb);
            if (g + 2 < 17) wa = sload16s(w, (g + 2) * 64);
            use(wb, g + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the accumulators are pinned here: pixels 1-3 are used under `vx` / `vy` below, and with that as their only use the
    // optimiser sinks their FMAs into those branches - behind every scalar load, with all 17 weight groups spilled lane by
    // lane and restored there (208 v_writelane + 208 v_readlane in the packed-FMA build)
#pragma unroll
    for (int p = 0; p < 4; ++p)
        asm volatile("" : "+v"(acc[p][0]), "+v"(acc[p][1]), "+v"(acc[p][2]), "+v"(acc[p][3]), "+v"(acc[p][4]), "+v"(acc[p][5]),
                          "+v"(acc[p][6]), "+v"(acc[p][7]), "+v"(acc[p][8]), "+v"(acc[p][9]));
    float out[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float bo = b[o], so = slope[o];
        float v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p] = acc[p][o] + bo;
            v[p] = v[p] >= 0.f ? v[p] : v[p] * so;
        }
        float m = v[0];
        if (vx) m = fmaxf(m, v[1]);
        if (vy) m = fmaxf(m, v[2]);
        if (vx && vy) m = fmaxf(m, v[3]);
        out[o] = m;
    }
    // the results are pinned here: with the guarded store as their only use the optimiser sinks the whole FMA sequence into
    // the branch, below every scalar load, and then spills all 17 weight groups lane by lane
    asm volatile("" ::"v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7]),
                 "v"(out[8]), "v"(out[9]));
    if (t < npix) {
        float* yp = y + it.out_off + r * CO;
#pragma unroll
        for (int o = 0; o < CO; o += 2) *reinterpret_cast<float2*>(yp + o) = make_float2(out[o], out[o + 1]);
    }
}

void launch_mt_pnet_conv1_pool(const float* x, const float* w_padded, const float* b, const float* slope, float* y,
                               const MtItem* items_dev, const long long* pre_dev, int n, long long total, hipStream_t s) {
    const long long npix = total / 10;
    if (npix <= 0) return;
    hipLaunchKernelGGL(mt_pnet_conv1_pool_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, w_padded, b, slope, y,
                       items_dev, pre_dev, n);
}

// First layer of R-Net / O-Net with its pooling: conv 3x3 (3 -> 32) + bias + PReLU + MaxPool2d(3, 2, ceil_mode) in one
// launch.  A thread owns one pooled column of one window and walks down the convolution rows: per row it evaluates the
// three convolution pixels of its pooling window (all 32 channels in registers, products accumulated as fmaf in
// (ci, ky, kx) order: the bits of an unfused convolution), keeps their maximum, and folds rows 2p, 2p+1, 2p+2 into pooled
// row p.  The 46x46x32 (22x22x32) map is never stored: 8.5 GB of writes and as many reads per 25k windows gone, for
// 1.5x the convolution arithmetic (columns 2p+2 / 2p are evaluated by both neighbours).  Weights, bias and slopes are
// SGPR operands (sload16, one tap ahead): no LDS or VGPR traffic for them.
// blockIdx.y = channel half: 16 of the 32 output channels per thread (3 waves per SIMD without spills)
// blockIdx.z = row segment: pooled rows [seg * pseg, (seg + 1) * pseg) - a few hundred windows of a selective cascade are
// 158 blocks of threads that each walk 46 rows; split in segments the same work is four times as many, shorter walks
// (the convolution row two segments share is evaluated by both)
__global__ __launch_bounds__(256, 3) void mt_conv1_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ slope,
                                                               float* __restrict__ y, int n, int ih, int iw, int ph, int pw,
                                                               int pseg) {
    constexpr int CO = 32, CH = 16;
    const int half = blockIdx.y;
    const int p0 = blockIdx.z * pseg, p1 = min(p0 + pseg, ph);          // this block's pooled rows
    if (p0 >= ph) return;
    const int y_begin = 2 * p0;                                          // first convolution row of the segment (even)
    w += half * CH;
    b += half * CH;
    slope += half * CH;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tc = min(t, (long long)n * pw - 1);     // surplus threads of the last block redo its last column
    const int px = (int)(tc % pw);
    const long long i = tc / pw;
    const int oh = ih - 2, ow = iw - 2;
    const float* xi = x + (size_t)i * ih * iw * 3;
    float* yo = y + ((size_t)i * ph * pw + px) * CO + half * CH;
    int col[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) col[j] = min(2 * px + j, iw - 1) * 3;
    const bool v1 = 2 * px + 1 < ow, v2 = 2 * px + 2 < ow;
    float rows[3][5][3], nxt[5][3];                        // input rows y, y+1, y+2 (x 5 columns x RGB) and y+3 in flight
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rows[1][j][c] = xi[(size_t)min(y_begin, ih - 1) * iw * 3 + col[j] + c];
            rows[2][j][c] = xi[(size_t)min(y_begin + 1, ih - 1) * iw * 3 + col[j] + c];
            nxt[j][c] = xi[(size_t)min(y_begin + 2, ih - 1) * iw * 3 + col[j] + c];
        }
    float m[CH];
#pragma unroll
    for (int o = 0; o < CH; ++o) m[o] = 0.f;
    // rows 2 p0 .. 2 p1 close the pooled rows p0 .. p1 - 1 (row 2 p1, if it exists, only completes row p1 - 1); the last
    // segment runs to the map's end and also writes the clipped last window
    const int y_end = p1 == ph ? oh : min(2 * p1 + 1, oh);
#pragma unroll 1
    for (int yy = y_begin; yy < y_end; ++yy) {
        sf16 wn = sload16<0>(w);
        // the row needed by the NEXT iteration is requested now: a whole row of arithmetic hides its latency (the SQ
        // counters of the version that loaded row y+2 at the top of iteration y showed 69 % of the wave cycles waiting)
        const size_t rn = (size_t)min(yy + 3, ih - 1) * iw * 3;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rows[0][j][c] = rows[1][j][c];
                rows[1][j][c] = rows[2][j][c];
                rows[2][j][c] = nxt[j][c];
                nxt[j][c] = xi[rn + col[j] + c];
            }
        float acc[3][CH];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int o = 0; o < CH; ++o) acc[p][o] = 0.f;
        // one tap per step: the 16 weights of this thread's channel half as SGPR operands, the next tap's in flight
        // meanwhile.  The compiler must never spill an in-flight group - a spill right after the asm would copy registers
        // the load has not written yet (tests/test_build_isa.py checks the row loop's ISA for v_writelane).
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int c = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
            swait1(wn);
            const sf16 wc = wn;
            if (tap + 1 < 27) wn = sload16s(w, (tap + 1) * 128);
            // two output channels per instruction: v_pk_fma_f32 with the SGPR weight pair as one operand (each half an IEEE fma)
#pragma unroll
            for (int o = 0; o < CH; o += 2)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const float xv = rows[ky][p + kx][c];
                    const v2f x2 = {xv, xv};
                    v2f a = {acc[p][o], acc[p][o + 1]};
                    a = __builtin_elementwise_fma(x2, (v2f){wc[o], wc[o + 1]}, a);
                    acc[p][o] = a.x; acc[p][o + 1] = a.y;
                }
            __builtin_amdgcn_sched_barrier(0);               // the tap's FMAs stay before the next wait / load: two groups live
        }
        const bool even = (yy & 1) == 0;
        sf16 bv = sload16<0>(b), sv = sload16<0>(slope);
        swait2(bv, sv);
#pragma unroll
        for (int o = 0; o < CH; ++o) {
            float v0 = acc[0][o] + bv[o], u1 = acc[1][o] + bv[o], u2 = acc[2][o] + bv[o];
            v0 = v0 >= 0.f ? v0 : v0 * sv[o];
            u1 = u1 >= 0.f ? u1 : u1 * sv[o];
            u2 = u2 >= 0.f ? u2 : u2 * sv[o];
            float h = v0;
            if (v1) h = fmaxf(h, u1);
            if (v2) h = fmaxf(h, u2);
            const float full = fmaxf(m[o], h);              // odd row: second row of the window; even row: its third
            acc[0][o] = full;
            m[o] = even ? h : full;                         // an even row also opens pooled row yy / 2
        }
        if (even && yy > y_begin && t == tc) {                   // (row y_begin opens the segment's first window: nothing to store)
            float* yp = yo + (size_t)(yy / 2 - 1) * pw * CO;
#pragma unroll
            for (int o = 0; o < CH; o += 4)
                *reinterpret_cast<float4*>(yp + o) = make_float4(acc[0][o], acc[0][o + 1], acc[0][o + 2], acc[0][o + 3]);
        }
    }
    // the clipped last window (ceil mode): rows 2(ph-1) .. oh-1 are in m
    if (t == tc && p1 == ph) {
        float* yp = yo + (size_t)(ph - 1) * pw * CO;
#pragma unroll
        for (int o = 0; o < CH; o += 4) *reinterpret_cast<float4*>(yp + o) = make_float4(m[o], m[o + 1], m[o + 2], m[o + 3]);
    }
}

bool launch_mt_conv1_pool(const float* x, const float* w, const float* b, const float* slope, float* y, int n, int ih, int iw,
                          int co, hipStream_t s) {
    const int oh = ih - 2, ow = iw - 2, ph = mt_pool_out(oh, 3, 2), pw = mt_pool_out(ow, 3, 2);
    // the row walk above assumes the last pooled row / column is a clipped window that starts inside the map
    if (co != 32 || !slope || oh < 3 || ow < 3 || 2 * (ph - 1) > oh - 1 || 2 * (ph - 1) + 2 < oh - 1 || 2 * (pw - 1) > ow - 1 ||
        2 * (pw - 1) + 2 < ow - 1)
        return false;
    const long long threads = (long long)n * pw;
    if (threads <= 0) return true;
    // row segments while the launch would not fill the chip otherwise (256 CUs x 3 blocks): up to 4, at least 3 pooled rows each
    const long long blocks = (threads + 255) / 256 * 2;
    int segs = (int)std::min<long long>(4, std::max<long long>(1, 768 / std::max<long long>(blocks, 1)));
    segs = std::max(1, std::min(segs, ph / 3));
    const int pseg = (ph + segs - 1) / segs;
    hipLaunchKernelGGL(mt_conv1_pool_kernel, dim3((unsigned)((threads + 255) / 256), 2, (unsigned)((ph + pseg - 1) / pseg)), dim3(256), 0, s,
                       x, w, b, slope, y, n, ih, iw, ph, pw, pseg);
    return true;
}

template <int CI, int CO, int K, int P, bool HEADS>
static void convpx_launch(const float* x, const float* w, const float* b, const float* slope, float* y, const MtItem* items,
                          const long long* pre, int n, long long npix, int ih, int iw, const MtPnetHeads* hd, hipStream_t s) {
    if (npix <= 0) return;
    const long long threads = (npix + P - 1) / P;
    hipLaunchKernelGGL((mt_convpx_kernel<CI, CO, K, P, HEADS>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, w, b,
                       slope, y, items, pre, n, ih, iw, hd ? hd->w41 : nullptr, hd ? hd->b41 : nullptr, hd ? hd->w42 : nullptr,
                       hd ? hd->b42 : nullptr, hd ? hd->prob : nullptr, hd ? hd->reg : nullptr, hd ? hd->cand : nullptr,
                       hd ? hd->cand_count : nullptr, hd ? hd->cand_cap : 0u, hd ? hd->thr : 0.f);
}

// ---- P-Net conv2 / conv3 on the matrix pipe (round 3) ---------------------------------------------------------------
// The two convolutions that carry P-Net's arithmetic (10 -> 16 and 16 -> 32 channels, 3x3 valid, over every pyramid
// level of every crop: 17 of the cascade's ~20 GFLOP per 256 crops) as ragged implicit GEMMs on
// v_mfma_f32_16x16x32_bf16 with exact three-term operands, like the classifier's GEMMs.  D[co][pixel] = sum_k W[co][k] *
// X[k][pixel]; a kernel ROW of a pixel's 3 x 3 x CI window is 3 * CI contiguous floats of the NHWC map, so with k = ky *
// KR + (kx * CI + ci) (KR = 32 for CI = 10 - two zero-weight slots per row - and 48 for CI = 16) a lane's 8 consecutive
// k are 8 consecutive floats in memory: the B operand is two 16-byte loads per K-step, no gather.  Block = 4 waves x 4
// tiles of 16 consecutive output pixels of the ragged index space (a tile may straddle rows and levels: every lane finds
// its own pixel); the weight planes [3][CO][K] sit in LDS (rows padded by 8 elements: conflict-free 16-byte reads) and
// a K-step's A fragments serve the wave's four tiles; the next K-step's activations are in flight during the MFMAs.
// HEADS (conv3): both 1x1 heads, the softmax and the candidate append from the accumulators (the 32-channel map is
// never stored): a pixel's 32 channels sit on the 4 lanes j, j + 16, j + 32, j + 48 - partial dot products, two
// shuffles.  (The register-blocked VALU kernels above stay as DFD_MT_PNET_MFMA=0: conv3 792 us, conv2 365 us per 256 crops.)
template <int CI, int CO, bool HEADS>
__global__ __launch_bounds__(256) void mt_pnet_mfma_kernel(const float* __restrict__ x, const unsigned short* __restrict__ w3,
                                                           int plane, int Kp, const float* __restrict__ b,
                                                           const float* __restrict__ slope, float* __restrict__ y,
                                                           const MtItem* __restrict__ items, const long long* __restrict__ pre, int n,
                                                           const float* __restrict__ w41, const float* __restrict__ b41,
                                                           const float* __restrict__ w42, const float* __restrict__ b42,
                                                           float* __restrict__ prob, float* __restrict__ reg,
                                                           MtCand* __restrict__ cand, unsigned* __restrict__ cand_count,
                                                           unsigned cand_cap, float thr, int tiles_per_wave) {
    constexpr int KR = CI == 10 ? 32 : 48, KTOT = (3 * KR + 31) / 32 * 32, KS = KTOT / 32, NT = CO / 16;
    constexpr int KROW = KTOT + 8;                                   // LDS row stride (elements): 16-byte reads of 16 rows hit 16 bank groups
    __shared__ __attribute__((aligned(16))) unsigned short wl[3 * CO * KROW];
    __shared__ float hw[HEADS ? CO * 6 + 8 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const long long npix = pre[n] / CO;
    // weight planes -> LDS once per block (16-byte chunks; rows of the global planes are Kp long, zero beyond 3 * KR);
    // the block then walks 256-pixel chunks of the ragged index space (staged per 256 pixels, the 30 KB of conv3 planes
    // were 270 MB of L2 -> LDS traffic per launch and a barrier-bound 2-3 us in front of 1.6 us of MFMA work)
    for (int c = tid; c < 3 * CO * (KTOT / 8); c += 256) {
        const int row = c / (KTOT / 8), oc = c - row * (KTOT / 8), pl = row / CO, r = row - pl * CO;
        *reinterpret_cast<uint4*>(&wl[row * KROW + oc * 8]) = *reinterpret_cast<const uint4*>(w3 + (size_t)pl * plane + (size_t)r * Kp + oc * 8);
    }
    if (HEADS) {
        for (int i = tid; i < CO * 2; i += 256) hw[i] = w41[i];
        for (int i = tid; i < CO * 4; i += 256) hw[CO * 2 + i] = w42[i];
        if (tid < 2) hw[CO * 6 + tid] = b41[tid];
        if (tid < 4) hw[CO * 6 + 2 + tid] = b42[tid];
    }
    __syncthreads();
    v4f bv[NT], sv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bv[nt] = *reinterpret_cast<const v4f*>(b + nt * 16 + 4 * q);
        sv[nt] = *reinterpret_cast<const v4f*>(slope + nt * 16 + 4 * q);
    }
    // A wave owns a CONTIGUOUS run of 16-pixel tiles of the ragged index space (tiles_per_wave each, the launcher's
    // split): the item (pyramid level) of a pixel only moves forward along the run, so after ONE binary search at the
    // start the lookup is "advance while the next level begins at or before this pixel" - the per-chunk searches of the
    // grid-stride version were 8k of its 22k cycles per chunk (s_memtime).
    const long long ntiles = (npix + 15) / 16;
    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long t_begin = wave_id * tiles_per_wave, t_end = t_begin + tiles_per_wave < ntiles ? t_begin + tiles_per_wave : ntiles;
    if (t_begin >= ntiles) return;                                   // (after the only barrier)
    // wave-uniform cursor: the item of the current tile's first pixel with its bounds and descriptor IN REGISTERS - the
    // common lane (same item as the cursor) does no dependent load before its window loads; only a lane past the item's
    // end walks on through `pre` (levels are thousands of pixels long: rare)
    int item = mt_find(pre, n, (t_begin * 16 < npix ? t_begin * 16 : npix - 1) * CO);
    long long cur_start = pre[item], next_start = pre[item + 1];
    MtItem cur_it = items[item];
    struct Tile { v4f lo[KS], hi[KS]; long long opix; bool ok; };
    auto prep = [&](long long tile, Tile& T) {
        const long long pix = tile * 16 + j;
        T.ok = pix < npix;
        const long long pq = T.ok ? pix : npix - 1;
        MtItem it = cur_it;
        long long st = cur_start;
        if (pq * CO >= next_start) {                                 // this lane's pixel lies in a later item
            int i = item + 1;
            while (i + 1 < n && pre[i + 1] <= pq * CO) ++i;
            it = items[i];
            st = pre[i];
        }
        const long long r = pq - st / CO;
        const int ow = it.iw - 2;
        const unsigned oyu = (unsigned)r / (unsigned)ow;             // a level has < 2^31 pixels: 32-bit division
        const int oy = (int)oyu, ox = (int)((unsigned)r - oyu * (unsigned)ow);
        const float* xb = x + it.in_off + ((size_t)oy * it.iw + ox) * CI;
        T.opix = it.out_off / CO + r;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = s * 4 + q, ky = o / (KR / 8), r0 = (o - ky * (KR / 8)) * 8;
            const float* p = xb + (ky < 3 ? (size_t)ky * it.iw * CI + r0 : 0);    // octets past the third row: zero weights
            T.lo[s] = ldg4u(p);
            T.hi[s] = ldg4u(p + 4);
        }
    };
    auto advance_item = [&](long long tile) {                       // move the cursor to the item of the tile's first pixel
        const long long e = (tile * 16 < npix ? tile * 16 : npix - 1) * CO;
        while (item + 1 < n && next_start <= e) {
            ++item;
            cur_start = next_start;
            next_start = pre[item + 1];
            cur_it = items[item];
        }
    };
    Tile ring[2];
    prep(t_begin, ring[0]);
    for (long long tile = t_begin; tile < t_end; tile += 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const long long tcur = tile + d;
        if (tcur >= t_end) break;                                    // wave-uniform
        Tile& T = ring[d];
        if (tcur + 1 < t_end) { advance_item(tcur + 1); prep(tcur + 1, ring[d ^ 1]); }
        v4f acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf8 wf[NT][3];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wf[nt][pl] = *reinterpret_cast<const bf8*>(&wl[(pl * CO + nt * 16 + j) * KROW + s * 32 + 8 * q]);
            bf8 x0, x1, x2;
            split8(T.lo[s], T.hi[s], x0, x1, x2);
            const bf8* xs[3] = {&x0, &x1, &x2};
            const int wsel[6] = {2, 1, 0, 1, 0, 0}, xsel[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first
#pragma unroll
            for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][wsel[p6]], *xs[xsel[p6]], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                       // a K-step's fragments and split terms die here (256 -> VGPRs otherwise)
        }
        // epilogue: bias, PReLU; the lane holds channels nt * 16 + 4 q .. + 3 of its pixel
        float v[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const v4f a = acc[nt] + bv[nt];
            const float av[4] = {a.x, a.y, a.z, a.w}, sl[4] = {sv[nt].x, sv[nt].y, sv[nt].z, sv[nt].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[nt][e] = av[e] >= 0.f ? av[e] : av[e] * sl[e];
        }
        if constexpr (!HEADS) {
            if (T.ok)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    *reinterpret_cast<v4f*>(y + T.opix * CO + nt * 16 + 4 * q) = (v4f){v[nt][0], v[nt][1], v[nt][2], v[nt][3]};
        } else {
            float z[2] = {0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = nt * 16 + 4 * q + e;
#pragma unroll
                    for (int k = 0; k < 2; ++k) z[k] = fmaf(v[nt][e], hw[c * 2 + k], z[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) r4[k] = fmaf(v[nt][e], hw[CO * 2 + c * 4 + k], r4[k]);
                }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {                // fold the four channel groups of the pixel (a fixed order)
#pragma unroll
                for (int k = 0; k < 2; ++k) z[k] += __shfl_xor(z[k], off);
#pragma unroll
                for (int k = 0; k < 4; ++k) r4[k] += __shfl_xor(r4[k], off);
            }
            const float z0 = z[0] + hw[CO * 6], z1 = z[1] + hw[CO * 6 + 1];
            const float m = fmaxf(z0, z1);
            const float e0 = expf(z0 - m), e1 = expf(z1 - m);
            const float pf = e1 / (e0 + e1);
            float4 rv;
            rv.x = r4[0] + hw[CO * 6 + 2]; rv.y = r4[1] + hw[CO * 6 + 3];
            rv.z = r4[2] + hw[CO * 6 + 4]; rv.w = r4[3] + hw[CO * 6 + 5];
            const bool mine = T.ok && q == 0;                        // one lane per pixel writes
            if (mine) {
                prob[T.opix] = pf;
                *reinterpret_cast<float4*>(reg + T.opix * 4) = rv;
            }
            const bool pass = cand && mine && pf >= thr;
            const unsigned long long mask = __ballot(pass);
            if (mask) {                                              // one atomic per wave (see mt_convpx_kernel)
                const int leader = __ffsll((long long)mask) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(cand_count, (unsigned)__popcll(mask));
                base = __shfl(base, leader);
                const unsigned slot = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                if (pass && slot < cand_cap) cand[slot] = MtCand{(unsigned)T.opix, pf, {rv.x, rv.y, rv.z, rv.w}};
            }
        }
      }
    }
}

bool launch_mt_pnet_mfma(const float* x, const unsigned short* w3, int plane, int Kp, const float* b, const float* slope, float* y,
                         const MtItem* items_dev, const long long* pre_dev, int n, long long total, int ci, int co,
                         const MtPnetHeads* hd, hipStream_t s) {
    const long long npix = total / co;
    if (npix <= 0) return true;
    // 1024 blocks x 4 waves when there is enough work (each wave a contiguous run of tiles), at least 4 tiles per wave
    const long long ntiles = (npix + 15) / 16;
    long long tpw = (ntiles + 4095) / 4096;
    if (tpw < 4) tpw = 4;
    const unsigned grid = (unsigned)((ntiles + 4 * tpw - 1) / (4 * tpw));
    if (ci == 10 && co == 16 && !hd)
        hipLaunchKernelGGL((mt_pnet_mfma_kernel<10, 16, false>), dim3(grid), dim3(256), 0, s, x, w3, plane, Kp, b, slope, y, items_dev, pre_dev,
                           n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0.f, (int)tpw);
    else if (ci == 16 && co == 32 && hd)
        hipLaunchKernelGGL((mt_pnet_mfma_kernel<16, 32, true>), dim3(grid), dim3(256), 0, s, x, w3, plane, Kp, b, slope, y, items_dev, pre_dev,
                           n, hd->w41, hd->b41, hd->w42, hd->b42, hd->prob, hd->reg, hd->cand, hd->cand_count, hd->cand_cap, hd->thr, (int)tpw);
    else return false;
    return true;
}

bool launch_mt_convpx_ragged(const float* x, const float* w, const float* b, const float* slope, float* y,
                             const MtItem* items_dev, const long long* pre_dev, int n, long long total, int ci, int co, int k,
                             const MtPnetHeads* heads, hipStream_t s) {
    const long long npix = total / co;
    if (ci == 10 && co == 16 && k == 3 && !heads) convpx_launch<10, 16, 3, 4, false>(x, w, b, slope, y, items_dev, pre_dev, n, npix, 0, 0, nullptr, s);
    else if (ci == 16 && co == 32 && k == 3 && heads) convpx_launch<16, 32, 3, 4, true>(x, w, b, slope, y, items_dev, pre_dev, n, npix, 0, 0, heads, s);
    else return false;
    return true;
}


// windows that live in different images (the crops of a step): per-window source pointer and stride
__global__ __launch_bounds__(256) void mt_area_resize_multi_kernel(const MtSrcWindow* __restrict__ win, int n, int oh, int ow,
                                                                   float* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;          // one output pixel (three exact integer sums)
    if (t >= (long long)n * oh * ow) return;
    const unsigned tu = (unsigned)t, uow = (unsigned)ow, uoh = (unsigned)oh;   // n * oh * ow < 2^31 (kChunk windows of <= 48 x 48)
    const unsigned rowi = tu / uow, oxu = tu - rowi * uow, iu = rowi / uoh, oyu = rowi - iu * uoh;
    const int i = (int)iu;
    const MtSrcWindow w = win[i];
    const int y0 = (int)(oyu * (unsigned)w.h / uoh), y1 = (int)(((oyu + 1u) * (unsigned)w.h + uoh - 1u) / uoh);
    const int x0 = (int)(oxu * (unsigned)w.w / uow), x1 = (int)(((oxu + 1u) * (unsigned)w.w + uow - 1u) / uow);
    int sb = 0, sg = 0, sr = 0;
    // (the window lies inside its source image: only the image's LAST row has nothing behind it; the source height is
    // not known here, so the window's own last row takes the byte loads)
    mt_window_sums(w.src + (size_t)w.y * w.stride + (size_t)w.x * 3, (size_t)w.stride, y0, y1, x0, x1, w.h - 1, sb, sg, sr);
    const float area = (float)((y1 - y0) * (x1 - x0));
    float* d = dst + t * 3;
    d[0] = ((float)sr / area - 127.5f) * 0.0078125f;
    d[1] = ((float)sg / area - 127.5f) * 0.0078125f;
    d[2] = ((float)sb / area - 127.5f) * 0.0078125f;
}

void launch_mt_area_resize_multi(const MtSrcWindow* win_dev, int n, int oh, int ow, float* dst, hipStream_t s) {
    const long long total = (long long)n * oh * ow;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_area_resize_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, win_dev, n, oh, ow, dst);
}


}  // namespace dfd
