// Host-only box logic of the two detectors (plain C++, no HIP; also built into the sanitizer harness):
// the integer post-processing of the SSD rows (reference face_detection.py:84-105) and cv::groupRectangles of the
// Haar fallback (reference face_detection.py:108-123 -> detectMultiScale).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace dfd {

// reference face_detection.py:84-105 on one image's DetectionOutput rows
inline int ssd_postprocess(const float* rows, int nrows, int hh, int ww, float conf_thr, int32_t* xywh, float* conf,
                    int max_out, int* total = nullptr) {
    int k = 0, all = 0;
    for (int i = 0; i < nrows; ++i) {
        const float* r = rows + (size_t)i * 5;
        if (!(r[0] > conf_thr)) continue;                                   // strict '>'
        // a non-finite coordinate (broken weights) has no integer value: numpy's astype(int) of it is unspecified and the
        // cast below would be undefined behaviour - such a row is no detection
        if (!std::isfinite(r[1]) || !std::isfinite(r[2]) || !std::isfinite(r[3]) || !std::isfinite(r[4])) continue;
        // float32 box * int64 [w,h,w,h] is a float64 product in numpy; astype(int) truncates toward zero
        long long x1 = (long long)((double)r[1] * ww), y1 = (long long)((double)r[2] * hh);
        long long x2 = (long long)((double)r[3] * ww), y2 = (long long)((double)r[4] * hh);
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > ww) x2 = ww;
        if (y2 > hh) y2 = hh;
        const long long bw = x2 - x1, bh = y2 - y1;
        if (bw > 20 && bh > 20) {
            ++all;                                                   // every detection counts (len(faces)) ...
            if (k < max_out) {                                       // ... the first max_out are returned
                xywh[4 * k] = (int32_t)x1; xywh[4 * k + 1] = (int32_t)y1; xywh[4 * k + 2] = (int32_t)bw; xywh[4 * k + 3] = (int32_t)bh;
                if (conf) conf[k] = r[0];
                ++k;
            }
        }
    }
    if (total) *total = all;
    return k;
}

struct Rect { int x, y, w, h; };

inline int cv_round(double v) { return (int)std::nearbyint(v); }            // round half to even, as cvRound

// cv::groupRectangles(rects, group_threshold, eps)
inline std::vector<Rect> group_rectangles(const std::vector<Rect>& in, int group_threshold, double eps) {
    const int n = (int)in.size();
    if (group_threshold <= 0 || n == 0) return in;
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
    auto similar = [&](const Rect& a, const Rect& b) {
        const double delta = eps * (std::min(a.w, b.w) + std::min(a.h, b.h)) * 0.5;
        return std::abs(a.x - b.x) <= delta && std::abs(a.y - b.y) <= delta && std::abs(a.x + a.w - b.x - b.w) <= delta &&
               std::abs(a.y + a.h - b.y - b.h) <= delta;
    };
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (similar(in[i], in[j])) {
                const int a = find(i), b = find(j);
                if (a != b) parent[b] = a;
            }
    std::vector<int> label(n, -1);
    int nclasses = 0;
    std::vector<int> cls(n);
    for (int i = 0; i < n; ++i) {                                // classes numbered by first appearance
        const int r = find(i);
        if (label[r] < 0) label[r] = nclasses++;
        cls[i] = label[r];
    }
    std::vector<long long> sx(nclasses, 0), sy(nclasses, 0), sw(nclasses, 0), sh(nclasses, 0);
    std::vector<int> cnt(nclasses, 0);
    for (int i = 0; i < n; ++i) { sx[cls[i]] += in[i].x; sy[cls[i]] += in[i].y; sw[cls[i]] += in[i].w; sh[cls[i]] += in[i].h; ++cnt[cls[i]]; }
    std::vector<Rect> mean(nclasses);
    for (int c = 0; c < nclasses; ++c) {
        const float s = 1.f / (float)cnt[c];
        mean[c] = Rect{cv_round((float)sx[c] * s), cv_round((float)sy[c] * s), cv_round((float)sw[c] * s), cv_round((float)sh[c] * s)};
    }
    std::vector<Rect> out;
    for (int i = 0; i < nclasses; ++i) {
        const Rect& r1 = mean[i];
        const int n1 = cnt[i];
        if (n1 <= group_threshold) continue;
        int j = 0;
        for (; j < nclasses; ++j) {                              // a small rectangle inside a stronger large one goes
            const int n2 = cnt[j];
            if (j == i || n2 <= group_threshold) continue;
            const Rect& r2 = mean[j];
            const int dx = cv_round(r2.w * eps), dy = cv_round(r2.h * eps);
            if (r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.w <= r2.x + r2.w + dx && r1.y + r1.h <= r2.y + r2.h + dy &&
                (n2 > std::max(3, n1) || n1 < 3))
                break;
        }
        if (j == nclasses) out.push_back(r1);
    }
    return out;
}

}  // namespace dfd
