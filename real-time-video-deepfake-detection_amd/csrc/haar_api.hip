// Haar-cascade face detection: the reference's fallback when the SSD files are missing (SURVEY section 8(f) N4;
// reference face_detection.py:108-123: face_cascade.detectMultiScale(gray, scaleFactor=1.1, minNeighbors=5,
// minSize=(30, 30))).  OpenCV's CascadeClassifier for a new-format stump cascade with upright HAAR features, restated:
//
//   for factor = 1, 1.1, 1.21, ...                      window = round(win * factor), image = round(size / factor)
//       scaled = resize(gray, image, INTER_LINEAR); integral images of scaled and scaled^2
//       for every y, x on a grid of step (factor > 2 ? 1 : 2):      (a window rejected by stage 0 also skips its
//           nf = area * sqsum - sum^2 over the window shrunk by 1 px;    right-hand neighbour)
//           norm = nf > 0 ? 1 / sqrt(nf) : 1
//           per stage: sum of (feature * norm < threshold ? left : right) over its stumps; < stage threshold -> reject
//           all stages passed -> candidate (round(x * factor), round(y * factor), window, window)
//   groupRectangles(candidates, minNeighbors, 0.2)
//
// Device: gray, the per-scale resize, both integral images (row scan, column scan), one thread per window for the
// cascade, one thread per row for the skip rule; host: the scale loop and groupRectangles (a union-find over a few
// hundred candidates).  The cascade comes from the weights blob ("haar.win", "haar.stages", "haar.stumps",
// "haar.rects": haar.py reads OpenCV's XML).
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>

#include "dfd_common.h"
#include "host_boxes.h"

using namespace dfd;

namespace dfd {

struct HaarState {
    int win_w = 24, win_h = 24, n_stages = 0, n_stumps = 0, n_feats = 0;
    const float *stages = nullptr, *stumps = nullptr, *rects = nullptr;       // device (blob tensors)
    DevBuf gray, scaled, sum, sqsum, result, cand, count;
};

void haar_destroy(dfd_handle* h) {
    delete h->haar;
    h->haar = nullptr;
}

int haar_init(dfd_handle* h) {
    auto w = h->tensors.find("haar.win");
    if (w == h->tensors.end()) return DFD_OK;
    auto st = h->tensors.find("haar.stages"), sp = h->tensors.find("haar.stumps"), rc = h->tensors.find("haar.rects");
    if (st == h->tensors.end() || sp == h->tensors.end() || rc == h->tensors.end() || w->second.count != 2 ||
        st->second.count % 3 || sp->second.count % 4 || rc->second.count % 15)
        return fail(h, DFD_ERR_BLOB, "weights blob: malformed haar.* tensors");
    HaarState* S = new HaarState();
    h->haar = S;
    float wh[2];
    DFD_HIP_TRY(h, hipMemcpy(wh, w->second.dev, 8, hipMemcpyDeviceToHost));
    S->win_w = (int)wh[0];
    S->win_h = (int)wh[1];
    S->n_stages = (int)(st->second.count / 3);
    S->n_stumps = (int)(sp->second.count / 4);
    S->n_feats = (int)(rc->second.count / 15);
    S->stages = st->second.dev;
    S->stumps = sp->second.dev;
    S->rects = rc->second.dev;
    if (S->win_w < 4 || S->win_h < 4 || S->n_stages < 1) return fail(h, DFD_ERR_BLOB, "weights blob: empty Haar cascade");
    return DFD_OK;
}

}  // namespace dfd

namespace {

__global__ __launch_bounds__(256) void haar_gray_kernel(const uint8_t* __restrict__ bgr, int W, int stride, uint8_t* __restrict__ g,
                                                        long long npix) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const uint8_t* p = bgr + (size_t)y * stride + 3 * x;
    g[i] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);       // cv2 COLOR_BGR2GRAY
}

// integral images with a zero first row and column: sum [h+1][w+1] u32, sqsum [h+1][w+1] u64
__global__ __launch_bounds__(64) void haar_rowscan_kernel(const uint8_t* __restrict__ img, int w, int h, unsigned* __restrict__ sum,
                                                          unsigned long long* __restrict__ sq) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y > h) return;
    unsigned* so = sum + (size_t)y * (w + 1);
    unsigned long long* qo = sq + (size_t)y * (w + 1);
    so[0] = 0;
    qo[0] = 0;
    if (y == 0) {
        for (int x = 1; x <= w; ++x) { so[x] = 0; qo[x] = 0; }
        return;
    }
    const uint8_t* r = img + (size_t)(y - 1) * w;
    unsigned a = 0;
    unsigned long long b = 0;
    for (int x = 0; x < w; ++x) {
        const unsigned v = r[x];
        a += v;
        b += v * v;
        so[x + 1] = a;
        qo[x + 1] = b;
    }
}

__global__ __launch_bounds__(256) void haar_colscan_kernel(int w, int h, unsigned* __restrict__ sum, unsigned long long* __restrict__ sq) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x > w) return;
    unsigned a = 0;
    unsigned long long b = 0;
    for (int y = 1; y <= h; ++y) {
        const size_t i = (size_t)y * (w + 1) + x;
        a += sum[i];
        b += sq[i];
        sum[i] = a;
        sq[i] = b;
    }
}

// result per grid position: 1 = passed every stage, 0 = rejected by stage 0, -s = rejected by stage s
__global__ __launch_bounds__(256) void haar_eval_kernel(const unsigned* __restrict__ sum, const unsigned long long* __restrict__ sq, int w,
                                                        int nx, int ny, int step, int win_w, int win_h,
                                                        const float* __restrict__ stages, int n_stages,
                                                        const float* __restrict__ stumps, const float* __restrict__ rects,
                                                        int* __restrict__ result) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int gy = i / nx, gx = i - gy * nx;
    const int x = gx * step, y = gy * step, W1 = w + 1;
    auto rs = [&](int rx, int ry, int rw, int rh) -> int {
        const size_t a = (size_t)(y + ry) * W1 + x + rx;
        return (int)(sum[a] - sum[a + rw] - sum[a + (size_t)rh * W1] + sum[a + (size_t)rh * W1 + rw]);
    };
    // normalisation over the window shrunk by one pixel on every side (HaarEvaluator::setWindow)
    const int nw = win_w - 2, nh = win_h - 2;
    const size_t a = (size_t)(y + 1) * W1 + x + 1;
    const double vs = (double)rs(1, 1, nw, nh);
    const double vq = (double)(sq[a] - sq[a + nw] - sq[a + (size_t)nh * W1] + sq[a + (size_t)nh * W1 + nw]);
    double nf = (double)(nw * nh) * vq - vs * vs;
    const float norm = nf > 0.0 ? (float)(1.0 / sqrt(nf)) : 1.0f;
    int out = 1;
    for (int s = 0; s < n_stages; ++s) {
        const int first = (int)stages[3 * s], cnt = (int)stages[3 * s + 1];
        const float thr = stages[3 * s + 2];
        double acc = 0.0;
        for (int k = first; k < first + cnt; ++k) {
            const float* sp = stumps + 4 * k;
            const float* r = rects + 15 * (int)sp[0];
            float v = r[4] * (float)rs((int)r[0], (int)r[1], (int)r[2], (int)r[3]) +
                      r[9] * (float)rs((int)r[5], (int)r[6], (int)r[7], (int)r[8]);
            if (r[14] != 0.f) v += r[14] * (float)rs((int)r[10], (int)r[11], (int)r[12], (int)r[13]);
            acc += (double)((v * norm) < sp[1] ? sp[2] : sp[3]);
        }
        if (acc < (double)thr) { out = -s; break; }
    }
    result[i] = out;
}

// one thread per grid row: a window rejected by stage 0 makes OpenCV skip the next grid position too
__global__ __launch_bounds__(64) void haar_collect_kernel(const int* __restrict__ result, int nx, int ny, int step, int* __restrict__ cand,
                                                          int* __restrict__ count, int cap, int scale_idx) {
    const int gy = blockIdx.x * 64 + threadIdx.x;
    if (gy >= ny) return;
    for (int gx = 0; gx < nx; ++gx) {
        const int r = result[gy * nx + gx];
        if (r > 0) {
            const int k = atomicAdd(count, 1);
            if (k < cap) { cand[3 * k] = scale_idx; cand[3 * k + 1] = gy * step; cand[3 * k + 2] = gx * step; }
        }
        if (r == 0) ++gx;
    }
}

}  // namespace

extern "C" {

int dfd_has_haar(const dfd_handle* h) { return h && h->haar ? 1 : 0; }

int dfd_detect_faces_haar(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, float scale_factor, int min_neighbors,
                          int min_size, int32_t* xywh_out, int max_out, int* n_out, int* n_candidates) {
    if (!h) return DFD_ERR_ARG;
    if (!bgr || !xywh_out || !n_out || max_out <= 0 || hh <= 0 || ww <= 0 || stride < ww * 3 || !(scale_factor > 1.0f))
        return fail(h, DFD_ERR_ARG, "detect_faces_haar: bad pointer, geometry or scale factor");
    *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (!h->haar) return fail(h, DFD_ERR_STATE, "no Haar cascade in the weights blob (weights.pack_all(..., haar=...))");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, &h->frame_buf, (size_t)hh * stride))) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return haar_run(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, scale_factor, min_neighbors, min_size, xywh_out, max_out,
                    n_out, n_candidates);
}

}  // extern "C"

namespace dfd {

// detectMultiScale + groupRectangles on a frame already resident in HBM (the fallback inside dfd_analyze_frame /
// dfd_analyze_jpeg when the handle has no SSD or the SSD pass failed: reference face_detection.py:58-66).
// *n_total (optional) = number of grouped rectangles before max_out cut the list.
int haar_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, float scale_factor, int min_neighbors,
             int min_size, int32_t* xywh_out, int max_out, int* n_out, int* n_candidates, int* n_total) {
    *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (n_total) *n_total = 0;
    HaarState* S = h->haar;
    if (!S) return fail(h, DFD_ERR_STATE, "no Haar cascade in the weights blob (weights.pack_all(..., haar=...))");
    int rc;
    const size_t npix = (size_t)hh * ww;
    constexpr int CAP = 1 << 16;
    if ((rc = ensure(h, &S->gray, npix))) return rc;
    if ((rc = ensure(h, &S->scaled, npix))) return rc;
    if ((rc = ensure(h, &S->sum, (size_t)(hh + 1) * (ww + 1) * 4))) return rc;
    if ((rc = ensure(h, &S->sqsum, (size_t)(hh + 1) * (ww + 1) * 8))) return rc;
    if ((rc = ensure(h, &S->result, npix * 4))) return rc;
    if ((rc = ensure(h, &S->cand, (size_t)CAP * 12))) return rc;
    if ((rc = ensure(h, &S->count, 4))) return rc;
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(haar_gray_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, frame_dev, ww,
                       stride, (uint8_t*)S->gray.p, (long long)npix);
    DFD_HIP_TRY(h, hipMemsetAsync(S->count.p, 0, 4, s));
    std::vector<double> factors;
    for (double factor = 1.0;; factor *= (double)scale_factor) {
        const int win_w = cv_round(S->win_w * factor), win_h = cv_round(S->win_h * factor);
        const int sw = cv_round(ww / factor), sh = cv_round(hh / factor);
        if (sw - S->win_w <= 0 || sh - S->win_h <= 0) break;
        factors.push_back(factor);
        if (win_w < min_size || win_h < min_size) continue;
        const int step = factor > 2.0 ? 1 : 2;
        const int nx = (sw - S->win_w + step - 1) / step, ny = (sh - S->win_h + step - 1) / step;
        const uint8_t* img = (const uint8_t*)S->gray.p;
        if (sw != ww || sh != hh) {
            launch_resize_gray((const uint8_t*)S->gray.p, hh, ww, (uint8_t*)S->scaled.p, sh, sw, s);
            img = (const uint8_t*)S->scaled.p;
        }
        hipLaunchKernelGGL(haar_rowscan_kernel, dim3((sh + 1 + 63) / 64), dim3(64), 0, s, img, sw, sh, (unsigned*)S->sum.p,
                           (unsigned long long*)S->sqsum.p);
        hipLaunchKernelGGL(haar_colscan_kernel, dim3((sw + 1 + 255) / 256), dim3(256), 0, s, sw, sh, (unsigned*)S->sum.p,
                           (unsigned long long*)S->sqsum.p);
        hipLaunchKernelGGL(haar_eval_kernel, dim3((nx * ny + 255) / 256), dim3(256), 0, s, (const unsigned*)S->sum.p,
                           (const unsigned long long*)S->sqsum.p, sw, nx, ny, step, S->win_w, S->win_h, S->stages, S->n_stages,
                           S->stumps, S->rects, (int*)S->result.p);
        hipLaunchKernelGGL(haar_collect_kernel, dim3((ny + 63) / 64), dim3(64), 0, s, (const int*)S->result.p, nx, ny, step,
                           (int*)S->cand.p, (int*)S->count.p, CAP, (int)factors.size() - 1);
    }
    int count = 0;
    DFD_HIP_TRY(h, hipMemcpyAsync(&count, S->count.p, 4, hipMemcpyDeviceToHost, s));
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());
    if (count > CAP) return fail(h, DFD_ERR_CAPACITY, "detect_faces_haar: %d candidate windows exceed the buffer of %d", count, CAP);
    std::vector<int> cand((size_t)count * 3);
    if (count) DFD_HIP_TRY(h, hipMemcpy(cand.data(), S->cand.p, cand.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::array<int, 3>> c3(count);
    for (int i = 0; i < count; ++i) c3[i] = {cand[3 * i], cand[3 * i + 1], cand[3 * i + 2]};
    std::sort(c3.begin(), c3.end());                              // (scale, y, x): the sequential scan order
    std::vector<Rect> rects;
    for (const auto& c : c3) {
        const double f = factors[c[0]];
        rects.push_back(Rect{cv_round(c[2] * f), cv_round(c[1] * f), cv_round(S->win_w * f), cv_round(S->win_h * f)});
    }
    if (n_candidates) *n_candidates = count;
    const std::vector<Rect> out = group_rectangles(rects, min_neighbors, 0.2);
    int k = 0;
    for (const Rect& r : out) {
        if (k >= max_out) break;
        xywh_out[4 * k] = r.x; xywh_out[4 * k + 1] = r.y; xywh_out[4 * k + 2] = r.w; xywh_out[4 * k + 3] = r.h;
        ++k;
    }
    *n_out = k;
    if (n_total) *n_total = (int)out.size();
    return DFD_OK;
}

}  // namespace dfd
