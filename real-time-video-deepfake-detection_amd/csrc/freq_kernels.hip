// compute_frequency_features (reference model.py:105-149): gray -> resize 224x224 -> channel 0 =
// min-max-normalised log1p|fftshift(fft2)|, channel 1 = min-max-normalised log1p|dct2(gray/255)|.
// The model ignores this input; it is kept for API parity.  224 = 2^5 * 7 is not a
// power of two, and the image is tiny, so both transforms are evaluated as two passes of a direct
// 224-point transform from a twiddle table (2 x 224^3 complex MACs: ~45 MFLOP, latency-bound);
// pass 1 writes transposed so pass 2 reads rows.
#include <cmath>

#include "dfd_common.h"

namespace dfd {

constexpr int FN = 224;

__global__ __launch_bounds__(256) void gray_resize_kernel(const uint8_t* __restrict__ src, int sh, int sw, size_t sstride,
                                                          int channels, uint8_t* __restrict__ gray_full) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sh * sw) return;
    const int y = i / sw, x = i % sw;
    const uint8_t* p = src + (size_t)y * sstride + (size_t)x * channels;
    gray_full[i] = channels == 3 ? (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14) : p[0];
}

// one block per row: out[k][row] = sum_n in[row][n] * tw[(k*n) mod 224]   (complex; re-only input when im == null)
__global__ __launch_bounds__(256) void dft_pass_kernel(const float* __restrict__ in_re, const float* __restrict__ in_im,
                                                       float* __restrict__ out_re, float* __restrict__ out_im,
                                                       const float2* __restrict__ tw) {
    __shared__ float re[FN], im[FN];
    __shared__ float2 t[FN];
    const int row = blockIdx.x, k = threadIdx.x;
    if (k < FN) {
        re[k] = in_re[row * FN + k];
        im[k] = in_im ? in_im[row * FN + k] : 0.f;
        t[k] = tw[k];
    }
    __syncthreads();
    if (k >= FN) return;
    float ar = 0.f, ai = 0.f;
    int idx = 0;                                   // (k*n) mod 224, updated incrementally
    for (int n = 0; n < FN; ++n) {
        const float2 w = t[idx];
        ar += re[n] * w.x - im[n] * w.y;
        ai += re[n] * w.y + im[n] * w.x;
        idx += k;
        if (idx >= FN) idx -= FN;
    }
    out_re[k * FN + row] = ar;
    out_im[k * FN + row] = ai;
}

// DCT-II, orthonormal (cv2.dct): out[k][row] = a_k * sum_n in[row][n] * cos(pi*(2n+1)*k / 448)
__global__ __launch_bounds__(256) void dct_pass_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       const float* __restrict__ ctab /*cos(pi*j/448), j<896*/) {
    __shared__ float v[FN];
    const int row = blockIdx.x, k = threadIdx.x;
    if (k < FN) v[k] = in[row * FN + k];
    __syncthreads();
    if (k >= FN) return;
    float a = 0.f;
    int idx = k;                                   // (2n+1)*k mod 896
    for (int n = 0; n < FN; ++n) {
        a += v[n] * ctab[idx];
        idx += 2 * k;
        if (idx >= 4 * FN) idx -= 4 * FN;
    }
    out[k * FN + row] = a * (k == 0 ? sqrtf(1.0f / FN) : sqrtf(2.0f / FN));
}

// value = log1p(|.|) with fftshift for the FFT channel; per-block min/max partials
__global__ __launch_bounds__(256) void logmag_kernel(const float* __restrict__ re, const float* __restrict__ im, int shift,
                                                     float* __restrict__ out, float* __restrict__ part) {
    __shared__ float smin[256], smax[256];
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    float v = 0.f, lo = 3.4e38f, hi = -3.4e38f;
    if (i < FN * FN) {
        const int y = i / FN, x = i % FN;
        const int sy = shift ? (y + FN - FN / 2) % FN : y, sx = shift ? (x + FN - FN / 2) % FN : x;   // fftshift
        const float a = re[sy * FN + sx], b = im ? im[sy * FN + sx] : 0.f;
        v = log1pf(im ? hypotf(a, b) : fabsf(a));
        out[i] = v;
        lo = hi = v;
    }
    smin[tid] = lo; smax[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { smin[tid] = fminf(smin[tid], smin[tid + s]); smax[tid] = fmaxf(smax[tid], smax[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

__global__ __launch_bounds__(256) void minmax_norm_kernel(float* __restrict__ v, const float* __restrict__ part, int nparts) {
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int p = 0; p < nparts; ++p) { lo = fminf(lo, part[2 * p]); hi = fmaxf(hi, part[2 * p + 1]); }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= FN * FN) return;
    v[i] = (hi - lo > 1e-6f) ? (v[i] - lo) / (hi - lo) : 0.f;
}

struct FreqState {
    float2* tw = nullptr;       // exp(-2 pi i k / 224)
    float* ctab = nullptr;      // cos(pi j / 448)
    DevBuf work, gray_full, gray;
};

void freq_destroy(dfd_handle* h) {
    delete h->freq;
    h->freq = nullptr;
}

static int freq_init(dfd_handle* h) {
    if (h->freq) return DFD_OK;
    h->freq = new FreqState();
    std::vector<float2> tw(FN);
    for (int k = 0; k < FN; ++k) {
        const double a = -2.0 * M_PI * k / FN;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    std::vector<float> ct(4 * FN);
    for (int j = 0; j < 4 * FN; ++j) ct[j] = (float)std::cos(M_PI * j / (2.0 * FN));
    void* d = nullptr;
    DFD_HIP_TRY(h, hipMalloc(&d, tw.size() * sizeof(float2)));
    h->owned.push_back(d);
    DFD_HIP_TRY(h, hipMemcpy(d, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    h->freq->tw = static_cast<float2*>(d);
    DFD_HIP_TRY(h, hipMalloc(&d, ct.size() * 4));
    h->owned.push_back(d);
    DFD_HIP_TRY(h, hipMemcpy(d, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
    h->freq->ctab = static_cast<float*>(d);
    return DFD_OK;
}

}  // namespace dfd

using namespace dfd;

extern "C" int dfd_frequency_features(dfd_handle* h, const uint8_t* img, int hh, int ww, int stride, int channels,
                                      float* out /* [2][224][224] */) {
    if (!h) return DFD_ERR_ARG;
    if (!img || !out || hh <= 0 || ww <= 0 || (channels != 1 && channels != 3) || stride < ww * channels)
        return fail(h, DFD_ERR_ARG, "frequency_features: bad pointer or geometry");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = freq_init(h);
    if (rc) return rc;
    FreqState& F = *h->freq;
    constexpr size_t PL = (size_t)FN * FN;
    if ((rc = ensure(h, &h->frame_buf, (size_t)hh * stride))) return rc;
    if ((rc = ensure(h, &F.gray_full, (size_t)hh * ww))) return rc;
    if ((rc = ensure(h, &F.gray, PL))) return rc;
    if ((rc = ensure(h, &F.work, PL * 4 * 8 + 4096))) return rc;
    float* w = static_cast<float*>(F.work.p);
    float *g = w, *a_re = w + PL, *a_im = w + 2 * PL, *b_re = w + 3 * PL, *b_im = w + 4 * PL, *o0 = w + 5 * PL, *o1 = w + 6 * PL,
          *part = w + 7 * PL;
    hipStream_t s = h->stream;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, img, (size_t)hh * stride, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gray_resize_kernel, dim3((hh * ww + 255) / 256), dim3(256), 0, s, (const uint8_t*)h->frame_buf.p, hh, ww,
                       (size_t)stride, channels, (uint8_t*)F.gray_full.p);
    launch_resize_gray((const uint8_t*)F.gray_full.p, hh, ww, (uint8_t*)F.gray.p, FN, FN, s);      // cv2.resize(gray,(224,224))
    u8_to_float((const uint8_t*)F.gray.p, g, (int)PL, 1.0f, s);
    // channel 0: fft2 -> fftshift -> log1p|.| -> min-max
    hipLaunchKernelGGL(dft_pass_kernel, dim3(FN), dim3(256), 0, s, g, nullptr, a_re, a_im, F.tw);
    hipLaunchKernelGGL(dft_pass_kernel, dim3(FN), dim3(256), 0, s, a_re, a_im, b_re, b_im, F.tw);
    const int nb = (int)((PL + 255) / 256);
    hipLaunchKernelGGL(logmag_kernel, dim3(nb), dim3(256), 0, s, b_re, b_im, 1, o0, part);
    hipLaunchKernelGGL(minmax_norm_kernel, dim3(nb), dim3(256), 0, s, o0, part, nb);
    // channel 1: dct2(gray / 255) -> log1p|.| -> min-max
    u8_to_float((const uint8_t*)F.gray.p, g, (int)PL, 1.0f / 255.0f, s);
    hipLaunchKernelGGL(dct_pass_kernel, dim3(FN), dim3(256), 0, s, g, a_re, F.ctab);
    hipLaunchKernelGGL(dct_pass_kernel, dim3(FN), dim3(256), 0, s, a_re, b_re, F.ctab);
    hipLaunchKernelGGL(logmag_kernel, dim3(nb), dim3(256), 0, s, b_re, nullptr, 0, o1, part);
    hipLaunchKernelGGL(minmax_norm_kernel, dim3(nb), dim3(256), 0, s, o1, part, nb);
    DFD_HIP_TRY(h, hipMemcpyAsync(out, o0, PL * 4, hipMemcpyDeviceToHost, s));
    DFD_HIP_TRY(h, hipMemcpyAsync(out + PL, o1, PL * 4, hipMemcpyDeviceToHost, s));
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}
