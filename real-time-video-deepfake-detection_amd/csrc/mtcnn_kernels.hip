// Kernels of the MTCNN align/crop stage (reference deepfake_detection.py:376-380 calls
// facenet-pytorch's MTCNN.forward on the cropped face).  The cascade's tensors are tiny (12x12 receptive
// fields, <= 128 channels, a few hundred candidate windows), so these are plain NHWC fp32 kernels - one thread
// per output element, weights laid out [ci][ky][kx][co] so that consecutive threads read consecutive
// addresses - not MFMA tiles: the stage is launch/latency bound, not arithmetic bound.
//   mt_area_resize    interpolate(mode="area") (= adaptive average pool) of a u8 BGR window -> RGB fp32 NHWC,
//                     (x - 127.5) * 0.0078125; exact: window sums are integers < 2^24
//   mt_conv           valid k x k convolution + bias (+ PReLU)
//   mt_maxpool        MaxPool2d(k, s, ceil_mode=True)
//   mt_dense          fully connected + bias (+ PReLU)
//   mt_softmax_face   softmax over the two classes, face probability only
//   mt_pil_pass       one pass of Pillow's 8-bit fixed-point Image.resize (coefficients from the host)
#include "mtcnn_kernels.h"

namespace dfd {

__global__ __launch_bounds__(256) void mt_area_resize_kernel(const uint8_t* __restrict__ src, size_t stride,
                                                             const MtWindow* __restrict__ win, int n, int oh, int ow,
                                                             float* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n * oh * ow * 3;
    if (t >= total) return;
    const int c = (int)(t % 3);                        // RGB channel of the output
    const int ox = (int)((t / 3) % ow), oy = (int)((t / 3 / ow) % oh), i = (int)(t / 3 / ow / oh);
    const MtWindow w = win[i];
    // adaptive pooling window: [floor(o * in / out), ceil((o + 1) * in / out))
    const int y0 = (int)(((long long)oy * w.h) / oh), y1 = (int)((((long long)oy + 1) * w.h + oh - 1) / oh);
    const int x0 = (int)(((long long)ox * w.w) / ow), x1 = (int)((((long long)ox + 1) * w.w + ow - 1) / ow);
    const uint8_t* p = src + (size_t)w.y * stride + (size_t)w.x * 3 + (2 - c);      // BGR bytes
    float sum = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sum += (float)p[(size_t)y * stride + (size_t)x * 3];
    const float mean = sum / (float)((y1 - y0) * (x1 - x0));
    dst[t] = (mean - 127.5f) * 0.0078125f;
}

void launch_mt_area_resize(const uint8_t* src, size_t stride, const MtWindow* win_dev, int n, int oh, int ow, float* dst,
                           hipStream_t s) {
    const long long total = (long long)n * oh * ow * 3;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_area_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, stride, win_dev,
                       n, oh, ow, dst);
}

__global__ __launch_bounds__(256) void mt_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, const float* __restrict__ slope,
                                                      float* __restrict__ y, int n, int ih, int iw, int ci, int co, int k) {
    const int oh = ih - k + 1, ow = iw - k + 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n * oh * ow * co;
    if (t >= total) return;
    const int o = (int)(t % co);
    const int ox = (int)((t / co) % ow), oy = (int)((t / co / ow) % oh), i = (int)(t / co / ow / oh);
    const float* xp = x + (((size_t)i * ih + oy) * iw + ox) * ci;
    float acc = 0.f;
    // accumulation order (ci, ky, kx): the order of the weight tensor [co][ci][ky][kx] a direct convolution walks
    for (int c = 0; c < ci; ++c)
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                acc = fmaf(xp[((size_t)ky * iw + kx) * ci + c], w[(((size_t)c * k + ky) * k + kx) * co + o], acc);
    acc += b[o];
    if (slope) acc = acc >= 0.f ? acc : acc * slope[o];
    y[t] = acc;
}

void launch_mt_conv(const float* x, const float* w, const float* b, const float* slope, float* y, int n, int ih, int iw,
                    int ci, int co, int k, hipStream_t s) {
    const long long total = (long long)n * (ih - k + 1) * (iw - k + 1) * co;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, b, slope, y, n, ih, iw,
                       ci, co, k);
}

__global__ __launch_bounds__(256) void mt_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int ih,
                                                         int iw, int c, int k, int st, int oh, int ow) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n * oh * ow * c;
    if (t >= total) return;
    const int ch = (int)(t % c);
    const int ox = (int)((t / c) % ow), oy = (int)((t / c / ow) % oh), i = (int)(t / c / ow / oh);
    float m = -INFINITY;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int yy = oy * st + ky, xx = ox * st + kx;
            if (yy < ih && xx < iw) m = fmaxf(m, x[(((size_t)i * ih + yy) * iw + xx) * c + ch]);      // ceil mode: clipped window
        }
    y[t] = m;
}

int mt_pool_out(int in, int k, int st) {
    int o = (in - k + st - 1) / st + 1;              // ceil((in - k) / st) + 1
    if ((o - 1) * st >= in) --o;                     // PyTorch: the last window must start inside the input
    return o;
}

void launch_mt_maxpool(const float* x, float* y, int n, int ih, int iw, int c, int k, int st, hipStream_t s) {
    const int oh = mt_pool_out(ih, k, st), ow = mt_pool_out(iw, k, st);
    const long long total = (long long)n * oh * ow * c;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, n, ih, iw, c, k, st,
                       oh, ow);
}

__global__ __launch_bounds__(256) void mt_dense_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ slope,
                                                       float* __restrict__ y, int n, int in, int out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * out) return;
    const int o = (int)(t % out), i = (int)(t / out);
    const float* xp = x + (size_t)i * in;
    float acc = 0.f;
    for (int c = 0; c < in; ++c) acc = fmaf(xp[c], w[(size_t)c * out + o], acc);
    acc += b[o];
    if (slope) acc = acc >= 0.f ? acc : acc * slope[o];
    y[t] = acc;
}

void launch_mt_dense(const float* x, const float* w, const float* b, const float* slope, float* y, int n, int in, int out,
                     hipStream_t s) {
    const long long total = (long long)n * out;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, b, slope, y, n, in, out);
}

__global__ __launch_bounds__(256) void mt_softmax_face_kernel(const float* __restrict__ z, float* __restrict__ p, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float z0 = z[2 * t], z1 = z[2 * t + 1];
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    p[t] = e1 / (e0 + e1);
}

void launch_mt_softmax_face(const float* z, float* p, long long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mt_softmax_face_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, p, n);
}

// One pass of ImagingResample (8 bits per channel): out = clip8((2^21 + sum_k coeff[k] * in[xmin + k]) >> 22).
// horizontal: src window (x0, y0, sw x sh) of a BGR image -> dst [sh][out][3]; vertical: src [sh][sw][3] -> dst [out][sw][3].
__global__ __launch_bounds__(256) void mt_pil_pass_kernel(const uint8_t* __restrict__ src, size_t stride, int x0, int y0,
                                                          int sw, int sh, const int* __restrict__ coeff,
                                                          const int* __restrict__ bounds, int ksize, int out, int vertical,
                                                          uint8_t* __restrict__ dst) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int rows = vertical ? out : sh, cols = vertical ? sw : out;
    if (t >= rows * cols * 3) return;
    const int c = t % 3, col = (t / 3) % cols, row = t / 3 / cols;
    const int o = vertical ? row : col;
    const int xmin = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* kk = coeff + (size_t)o * ksize;
    int ss = 1 << 21;
    for (int k = 0; k < cnt; ++k) {
        const int yy = vertical ? xmin + k : row, xx = vertical ? col : xmin + k;
        ss += kk[k] * (int)src[(size_t)(y0 + yy) * stride + (size_t)(x0 + xx) * 3 + c];
    }
    int v = ss >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[t] = (uint8_t)v;
}

void launch_mt_pil_pass(const uint8_t* src, size_t stride, int x0, int y0, int sw, int sh, const int* coeff_dev,
                        const int* bounds_dev, int ksize, int out, int vertical, uint8_t* dst, hipStream_t s) {
    const int total = (vertical ? out * sw : sh * out) * 3;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_pil_pass_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, stride, x0, y0, sw, sh, coeff_dev,
                       bounds_dev, ksize, out, vertical, dst);
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

__global__ __launch_bounds__(256) void mt_area_resize_ragged_kernel(const MtLevel* __restrict__ lv,
                                                                    const long long* __restrict__ pre, int n,
                                                                    float* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= pre[n]) return;
    const int i = mt_find(pre, n, t);
    const MtLevel L = lv[i];
    const long long r = t - pre[i];
    const int c = (int)(r % 3), ox = (int)((r / 3) % L.ow), oy = (int)(r / 3 / L.ow);
    const int y0 = (int)(((long long)oy * L.h) / L.oh), y1 = (int)((((long long)oy + 1) * L.h + L.oh - 1) / L.oh);
    const int x0 = (int)(((long long)ox * L.w) / L.ow), x1 = (int)((((long long)ox + 1) * L.w + L.ow - 1) / L.ow);
    const uint8_t* p = L.src + (2 - c);
    float sum = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sum += (float)p[(size_t)y * L.stride + (size_t)x * 3];
    const float mean = sum / (float)((y1 - y0) * (x1 - x0));
    dst[L.out_off + r] = (mean - 127.5f) * 0.0078125f;
}

void launch_mt_area_resize_ragged(const MtLevel* lv_dev, const long long* pre_dev, int n, long long total, float* dst,
                                  hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_area_resize_ragged_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, lv_dev, pre_dev, n, dst);
}

__global__ __launch_bounds__(256) void mt_conv_ragged_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ slope,
                                                             float* __restrict__ y, const MtItem* __restrict__ items,
                                                             const long long* __restrict__ pre, int n, int ci, int co, int k) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= pre[n]) return;
    const int i = mt_find(pre, n, t);
    const MtItem it = items[i];
    const long long r = t - pre[i];
    const int ow = it.iw - k + 1;
    const int o = (int)(r % co), ox = (int)((r / co) % ow), oy = (int)(r / co / ow);
    const float* xp = x + it.in_off + ((size_t)oy * it.iw + ox) * ci;
    float acc = 0.f;
    for (int c = 0; c < ci; ++c)
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                acc = fmaf(xp[((size_t)ky * it.iw + kx) * ci + c], w[(((size_t)c * k + ky) * k + kx) * co + o], acc);
    acc += b[o];
    if (slope) acc = acc >= 0.f ? acc : acc * slope[o];
    y[it.out_off + r] = acc;
}

void launch_mt_conv_ragged(const float* x, const float* w, const float* b, const float* slope, float* y,
                           const MtItem* items_dev, const long long* pre_dev, int n, long long total, int ci, int co, int k,
                           hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_conv_ragged_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, b, slope, y, items_dev,
                       pre_dev, n, ci, co, k);
}

__global__ __launch_bounds__(256) void mt_maxpool_ragged_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                const MtItem* __restrict__ items, const long long* __restrict__ pre,
                                                                int n, int c, int k, int st) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= pre[n]) return;
    const int i = mt_find(pre, n, t);
    const MtItem it = items[i];
    const long long r = t - pre[i];
    int ow = (it.iw - k + st - 1) / st + 1;
    if ((ow - 1) * st >= it.iw) --ow;
    const int ch = (int)(r % c), ox = (int)((r / c) % ow), oy = (int)(r / c / ow);
    float m = -INFINITY;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int yy = oy * st + ky, xx = ox * st + kx;
            if (yy < it.ih && xx < it.iw) m = fmaxf(m, x[it.in_off + ((size_t)yy * it.iw + xx) * c + ch]);
        }
    y[it.out_off + r] = m;
}

void launch_mt_maxpool_ragged(const float* x, float* y, const MtItem* items_dev, const long long* pre_dev, int n, long long total,
                              int c, int k, int st, hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_maxpool_ragged_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, items_dev, pre_dev, n, c,
                       k, st);
}

// windows that live in different images (the crops of a step): per-window source pointer and stride
__global__ __launch_bounds__(256) void mt_area_resize_multi_kernel(const MtSrcWindow* __restrict__ win, int n, int oh, int ow,
                                                                   float* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n * oh * ow * 3;
    if (t >= total) return;
    const int c = (int)(t % 3);
    const int ox = (int)((t / 3) % ow), oy = (int)((t / 3 / ow) % oh), i = (int)(t / 3 / ow / oh);
    const MtSrcWindow w = win[i];
    const int y0 = (int)(((long long)oy * w.h) / oh), y1 = (int)((((long long)oy + 1) * w.h + oh - 1) / oh);
    const int x0 = (int)(((long long)ox * w.w) / ow), x1 = (int)((((long long)ox + 1) * w.w + ow - 1) / ow);
    const uint8_t* p = w.src + (size_t)w.y * w.stride + (size_t)w.x * 3 + (2 - c);
    float sum = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sum += (float)p[(size_t)y * w.stride + (size_t)x * 3];
    const float mean = sum / (float)((y1 - y0) * (x1 - x0));
    dst[t] = (mean - 127.5f) * 0.0078125f;
}

void launch_mt_area_resize_multi(const MtSrcWindow* win_dev, int n, int oh, int ow, float* dst, hipStream_t s) {
    const long long total = (long long)n * oh * ow * 3;
    if (total <= 0) return;
    hipLaunchKernelGGL(mt_area_resize_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, win_dev, n, oh, ow, dst);
}

__global__ __launch_bounds__(256) void mt_prelu_kernel(float* __restrict__ x, const float* __restrict__ slope, long long n, int c) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float v = x[t];
    x[t] = v >= 0.f ? v : v * slope[t % c];
}

void launch_mt_prelu(float* x, const float* slope, long long n, int c, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mt_prelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, slope, n, c);
}

}  // namespace dfd
