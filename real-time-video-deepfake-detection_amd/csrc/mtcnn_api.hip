// MTCNN align/crop stage: the cascade the reference runs on every cropped face before the
// classifier (reference deepfake_detection.py:24-28, 376-380 -> facenet-pytorch MTCNN.forward: detect_face,
// select by probability, extract_face to 160x160).  All crops of a call go through the three stages together: the
// networks, the resamplers and the P-Net candidate compaction run as HIP kernels (mtcnn_kernels.hip, gemm_split.hip).
// The box bookkeeping between the stages (NMS, regression, squaring, clipping, selection, extract_face geometry - float32
// arithmetic in the package's operation order, restated line by line from oracle/mtcnn_ref.py) runs on the device too
// (mtcnn_boxes.hip, Cascade::run_device: one block per crop and stage; the host reads back two window counts and one
// result row per crop - three stream waits per step).  The same logic on the host side of the library (Cascade::run:
// grid-local per-level NMS, SoA sweeps, host threads when the funnel is dense) is the reference the device path is
// tested against bit for bit (DFD_MT_DEVICE_BOXES=0) and takes over when a crop exceeds the device blocks' capacity.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <atomic>
#include <numeric>
#include <thread>

#include "b0_kernels.h"
#include "dfd_common.h"
#include "mtcnn_kernels.h"

namespace dfd {

struct MtConv { const float *w, *b, *a; int co, ci, k; };
struct MtDense { const float *w, *b, *a; int out, in; const float* ba = nullptr; };      // ba: [out bias][out slope] (GEMM epilogue)
struct MtGemmConv { const float *w, *b, *a; int co, ci, k; const float* ba = nullptr; };   // [co][ky][kx][ci], channels padded to 32 / 64

struct MtcnnState {
    bool ready = false;
    MtConv p1, p2, p3, p41, p42;
    MtConv r1, r2, r3;
    MtDense r4, r51, r52;
    MtConv o1, o2, o3, o4;
    MtDense o5, o61, o62, o63;
    MtConv r1p;                           // R-Net conv1 with 32 output channels (4 zero filters)
    const float* p1w_pad = nullptr;       // P-Net conv1 weights in a 288-float buffer (scalar loads read 16 at a time)
    const float *p2m = nullptr, *p3m = nullptr;   // P-Net conv2 / conv3 in the MFMA kernel's K layout ([16][96], [32][160])
    MtGemmConv r2g, r3g, o2g, o3g, o4g;
    DevBuf in, a0, a1, prob, reg, win, coef, bnd, tmp, face, d_lv, bs, cand;       // d_lv: every descriptor table of a step
    // box bookkeeping on the device (mtcnn_boxes.hip): crop / level tables, counts + prefix arrays + meta words, segmented and
    // compact row / window arenas, per-crop result rows, rows of crop 0 for the parity taps
    DevBuf cnt, rows_a, wins_a, rows_b, wins_b, res, taprows;
};

void mtcnn_destroy(dfd_handle* h) {
    delete h->mtcnn;
    h->mtcnn = nullptr;
}

namespace {

const float* mt_tensor(dfd_handle* h, const std::string& name, size_t count, bool* ok) {
    auto it = h->tensors.find(name);
    if (it == h->tensors.end() || it->second.count != count) {
        if (*ok) fail(h, DFD_ERR_BLOB, "weights blob: MTCNN tensor '%s' missing or wrong size", name.c_str());
        *ok = false;
        return nullptr;
    }
    return it->second.dev;
}

MtConv mt_conv(dfd_handle* h, const std::string& q, int co, int ci, int k, const char* prelu, bool* ok) {
    MtConv c{};
    c.co = co; c.ci = ci; c.k = k;
    c.w = mt_tensor(h, "mtcnn." + q + ".w", (size_t)co * ci * k * k, ok);
    c.b = mt_tensor(h, "mtcnn." + q + ".b", co, ok);
    c.a = prelu ? mt_tensor(h, std::string("mtcnn.") + prelu + ".a", co, ok) : nullptr;
    return c;
}

MtDense mt_dense(dfd_handle* h, const std::string& q, int out, int in, const char* prelu, bool* ok) {
    MtDense d{};
    d.out = out; d.in = in;
    d.w = mt_tensor(h, "mtcnn." + q + ".w", (size_t)out * in, ok);
    d.b = mt_tensor(h, "mtcnn." + q + ".b", out, ok);
    d.a = prelu ? mt_tensor(h, std::string("mtcnn.") + prelu + ".a", out, ok) : nullptr;
    return d;
}

MtGemmConv mt_gemm_conv(dfd_handle* h, const std::string& q, int co, int ci, int k, bool* ok) {
    MtGemmConv c{};
    c.co = co; c.ci = ci; c.k = k;
    c.w = mt_tensor(h, "mtcnn." + q + ".wg", (size_t)co * ci * k * k, ok);
    c.b = mt_tensor(h, "mtcnn." + q + ".bg", co, ok);
    c.a = mt_tensor(h, "mtcnn." + q + ".ag", co, ok);
    return c;
}

// valid k x k conv of `n` maps [ih][iw][ci] on the split-precision MFMA GEMM (fp32-exact products), PReLU in its epilogue
int gemm_conv_prelu(dfd_handle* h, const MtGemmConv& c, const float* x, float* y, int n, int ih, int iw) {
    ConvGeom g;
    g.H = ih; g.W = iw; g.Ho = ih - c.k + 1; g.Wo = iw - c.k + 1; g.Cin = c.ci; g.ksize = c.k; g.stride = 1; g.pad = 0; g.dil = 1;
    const int K = c.k * c.k * c.ci;
    const unsigned short* w3 = split_weights(h, c.w, c.co, K);
    if (!w3) return DFD_ERR_HIP;
    if (!launch_conv_gemm_split<float>(h->gemm, x, w3, c.ba, nullptr, y, n, g, c.co, ACT_PRELU, false, 3, h->stream))
        return fail(h, DFD_ERR_STATE, "mtcnn: conv shape not supported by the GEMM kernel");
    return DFD_OK;
}

// y[n][out] = prelu(x[n][in] . w[in][out] + b) on the same GEMM (the two wide layers: R-Net dense4, O-Net dense5)
int gemm_dense_prelu(dfd_handle* h, const MtDense& d, const float* x, float* y, int n) {
    const unsigned short* w3 = split_weights(h, d.w, d.out, d.in, true);        // stored [in][out]
    if (!w3) return DFD_ERR_HIP;
    if (!launch_pointwise_split<float>(h->gemm, x, w3, d.ba, nullptr, nullptr, y, n, d.in, d.out, 1, ACT_PRELU, 3, h->stream))
        return fail(h, DFD_ERR_STATE, "mtcnn: dense shape not supported by the GEMM kernel");
    return DFD_OK;
}

// The box bookkeeping of independent crops / pyramid levels on a few host threads when there is enough of it (a dense
// cascade hands thousands of candidates per crop to the O(kept x n) NMS loops); `fn(i)` must only write slot i.
template <typename F>
void parallel_for(int count, size_t work, F fn) {
    const int hw = (int)std::thread::hardware_concurrency();
    const char* cap = getenv("DFD_HOST_THREADS");            // 1 = everything on the calling thread
    const int nt = std::min(std::min(count, cap ? std::max(atoi(cap), 1) : 8), std::max(hw / 2, 1));
    if (nt <= 1 || work < 200000) {
        for (int i = 0; i < count; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    auto run = [&] {
        for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1)) fn(i);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(run);
    run();
    for (auto& th : pool) th.join();
}

struct Box { float x1, y1, x2, y2, score, r[4]; };

// Greedy NMS over boxes already in processing order (structure of arrays: the inner loop is branch-free and
// vectorises).  PLUS1 = false: torchvision.ops.nms - suppress IoU > thr, areas without the +1;
// PLUS1 = true: the package's nms_numpy(method="Min") - suppress inter / min(area) > thr, +1 on every extent.
struct SortedBoxes {
    std::vector<float> x1, y1, x2, y2, area;
    std::vector<unsigned char> dead;
    void resize(size_t n) { x1.resize(n); y1.resize(n); x2.resize(n); y2.resize(n); area.resize(n); dead.assign(n, 0); }
};

template <bool PLUS1>
std::vector<int> greedy_nms(const std::vector<Box>& b, const std::vector<int>& order, float thr) {
    const size_t n = order.size();
    SortedBoxes S;
    S.resize(n);
    const float one = PLUS1 ? 1.f : 0.f;
    for (size_t k = 0; k < n; ++k) {
        const Box& q = b[order[k]];
        S.x1[k] = q.x1; S.y1[k] = q.y1; S.x2[k] = q.x2; S.y2[k] = q.y2;
        S.area[k] = (q.x2 - q.x1 + one) * (q.y2 - q.y1 + one);
    }
    std::vector<int> keep;
    const float *X1 = S.x1.data(), *Y1 = S.y1.data(), *X2 = S.x2.data(), *Y2 = S.y2.data(), *A = S.area.data();
    unsigned char* dead = S.dead.data();
    for (size_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep.push_back(order[i]);
        const float ix1 = X1[i], iy1 = Y1[i], ix2 = X2[i], iy2 = Y2[i], ai = A[i];
        for (size_t j = i + 1; j < n; ++j) {
            const float xx1 = std::max(ix1, X1[j]), yy1 = std::max(iy1, Y1[j]);
            const float xx2 = std::min(ix2, X2[j]), yy2 = std::min(iy2, Y2[j]);
            const float w = std::max(0.f, xx2 - xx1 + one), hgt = std::max(0.f, yy2 - yy1 + one);
            const float inter = w * hgt;
            const float o = PLUS1 ? inter / std::min(ai, A[j]) : inter / (ai + A[j] - inter);
            dead[j] |= (unsigned char)(PLUS1 ? !(o <= thr) : (o > thr));      // NaN: dropped by 'Min', kept by torchvision
        }
    }
    return keep;
}

// torchvision.ops.nms: descending score (stable)
std::vector<int> nms_iou(const std::vector<Box>& b, float thr) {
    std::vector<int> order(b.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int i, int j) { return b[i].score > b[j].score; });
    return greedy_nms<false>(b, order, thr);
}

// nms_numpy(method='Min'): ascending stable argsort, taken from the end (among equal scores the later box first)
std::vector<int> nms_min(const std::vector<Box>& b, float thr) {
    std::vector<int> order(b.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int i, int j) { return b[i].score < b[j].score; });
    std::reverse(order.begin(), order.end());
    return greedy_nms<true>(b, order, thr);
}

// nms_iou for the candidates of ONE P-Net level, which sit on the level's cell grid: equal-sized boxes (12 / scale,
// +-1 from the floors) spaced 2 / scale apart can only reach IoU > 0.5 within two cells of each other (scale <= 0.6:
// overlap width < (11 - 2 dx) / scale + 2 must exceed 2/3 (11 / scale - 1)^2 / (11 / scale + 1)), so every candidate is
// tested against the kept boxes of its 9 x 9 cell neighbourhood only - same tests, same order, same result as the
// all-pairs loop, in O(n) instead of O(kept x n).  cell[k] = y * ow + x of candidate k.
std::vector<int> nms_iou_level(const std::vector<Box>& b, const std::vector<int>& cell, int oh, int ow, float thr,
                               std::vector<int>* grid) {
    constexpr int R = 4;
    std::vector<int> order(b.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int i, int j) { return b[i].score > b[j].score; });
    grid->assign((size_t)oh * ow, -1);                       // cell -> index of the KEPT candidate there
    std::vector<int> keep;
    for (int i : order) {
        const int cy = cell[i] / ow, cx = cell[i] % ow;
        const Box& q = b[i];
        const float aq = (q.x2 - q.x1) * (q.y2 - q.y1);
        bool dead = false;
        for (int y = std::max(cy - R, 0); y <= std::min(cy + R, oh - 1) && !dead; ++y)
            for (int x = std::max(cx - R, 0); x <= std::min(cx + R, ow - 1); ++x) {
                const int k = (*grid)[(size_t)y * ow + x];
                if (k < 0) continue;
                const Box& p = b[k];
                const float xx1 = std::max(p.x1, q.x1), yy1 = std::max(p.y1, q.y1);
                const float xx2 = std::min(p.x2, q.x2), yy2 = std::min(p.y2, q.y2);
                const float inter = std::max(0.f, xx2 - xx1) * std::max(0.f, yy2 - yy1);
                const float ap = (p.x2 - p.x1) * (p.y2 - p.y1);
                if (inter / (ap + aq - inter) > thr) { dead = true; break; }
            }
        if (dead) continue;
        (*grid)[cell[i]] = i;
        keep.push_back(i);
    }
    return keep;
}

void bbreg(Box& b) {
    const float w = b.x2 - b.x1 + 1.f, hgt = b.y2 - b.y1 + 1.f;
    const float x1 = b.x1 + b.r[0] * w, y1 = b.y1 + b.r[1] * hgt, x2 = b.x2 + b.r[2] * w, y2 = b.y2 + b.r[3] * hgt;
    b.x1 = x1; b.y1 = y1; b.x2 = x2; b.y2 = y2;
}

void rerec(Box& b) {
    const float hgt = b.y2 - b.y1, w = b.x2 - b.x1;
    const float l = std::max(w, hgt);
    b.x1 = b.x1 + w * 0.5f - l * 0.5f;
    b.y1 = b.y1 + hgt * 0.5f - l * 0.5f;
    b.x2 = b.x1 + l;
    b.y2 = b.y1 + l;
}

// pad(): truncate, clip to the image; the 1-based (y, ey, x, ex) become the 0-based window [y-1, ey) x [x-1, ex)
bool window_of(const Box& b, int w, int hgt, MtWindow* out) {
    int x = (int)std::trunc(b.x1), y = (int)std::trunc(b.y1), ex = (int)std::trunc(b.x2), ey = (int)std::trunc(b.y2);
    if (x < 1) x = 1;
    if (y < 1) y = 1;
    if (ex > w) ex = w;
    if (ey > hgt) ey = hgt;
    if (!(ey > y - 1 && ex > x - 1)) return false;
    *out = MtWindow{x - 1, y - 1, ex - (x - 1), ey - (y - 1)};
    return true;
}

// precompute_coeffs + normalize_coeffs_8bpc of Pillow's bilinear filter (src/libImaging/Resample.c)
void pil_coeffs(int in_size, int out_size, std::vector<int>* coeff, std::vector<int>* bounds, int* ksize_out) {
    const double scale = (double)in_size / out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fs;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    coeff->assign((size_t)out_size * ksize, 0);
    bounds->assign((size_t)out_size * 2, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale, ss = 1.0 / fs;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? k[x] / ww : k[x];
            (*coeff)[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
        }
        (*bounds)[2 * xx] = xmin;
        (*bounds)[2 * xx + 1] = xmax;
    }
    *ksize_out = ksize;
}

// The cascade for all crops of a step at once.  Stage 1 runs P-Net over every pyramid level of every crop in one
// ragged launch per layer and downloads all maps together; stages 2 and 3 batch the candidate windows of all
// crops; the host does the per-crop box logic in between.  Four stream synchronisations per step instead of
// ~30 per crop.
struct Cascade {
    dfd_handle* h;
    MtcnnState* S;
    const MtImage* imgs;
    int n;
    const char* tap_name;          // parity taps (crop 0 only)
    std::vector<float>* tap;
    int* tap_dims;

    bool want(const std::string& nm) const { return tap_name && nm == tap_name; }

    std::chrono::steady_clock::time_point t_mark = std::chrono::steady_clock::now();
    void mark(const char* what) {                          // DFD_MT_VERBOSE=2: wall time of each phase
        static const bool on = getenv("DFD_MT_VERBOSE") && atoi(getenv("DFD_MT_VERBOSE")) >= 2;
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[dfd]   %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
    }

    // descriptor tables and result rows are small: through the handle's mailbox (dfd_common.h), not the DMA queues
    template <typename T>
    int upload(DevBuf* buf, const std::vector<T>& v) {
        int rc = ensure(h, buf, std::max<size_t>(v.size() * sizeof(T), 16));
        if (rc) return rc;
        return v.empty() ? DFD_OK : mailbox_h2d(h, buf->p, v.data(), v.size() * sizeof(T));
    }
    int download(const void* dev, size_t floats, std::vector<float>* out) {
        out->resize(floats);
        if (!floats) return DFD_OK;
        const float* p = (const float*)mailbox_d2h(h, dev, floats * 4);
        if (!p) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
        DFD_HIP_TRY(h, hipGetLastError());
        DFD_HIP_TRY(h, stream_sync(h));
        std::copy(p, p + floats, out->begin());
        return DFD_OK;
    }
    void tap_boxes(const std::string& name, const std::vector<Box>& b) {
        if (!want(name)) return;
        tap->clear();
        for (const Box& x : b) { tap->push_back(x.x1); tap->push_back(x.y1); tap->push_back(x.x2); tap->push_back(x.y2); tap->push_back(x.score); }
        tap_dims[0] = (int)b.size(); tap_dims[1] = 5; tap_dims[2] = 1;
    }

    struct Level { int crop; double scale; int sh, sw, oh, ow; long long cell_off; };

    bool stage1_done = false;              // the P-Net launches of this step are already queued (device path fell back)
    std::vector<Level> levels;             // pyramid levels of all crops (stage1_gpu)
    long long cells = 0;                   // P-Net output cells of all levels
    std::vector<float> tap_prob, tap_reg;  // whole P-Net maps (parity taps only)
    std::vector<MtCropGeo> cg;             // device box path: per-crop / per-level tables, output arena sizes
    std::vector<MtLevelGeo> lg;
    bool geo_ok = false;
    long long seg = 0, tmp_bytes = 0, tab = 0;
    const MtLevel* d_levels = nullptr;     // device copies of the step's tables (one upload)
    const MtItem* d_items = nullptr;
    const long long* d_pre = nullptr;
    const MtCropGeo* dcg = nullptr;
    const MtLevelGeo* dlg = nullptr;

    // Pillow's kernel size for a 160-wide axis read from in_size pixels (pil_coeffs)
    static int pil_ksize(int in_size) {
        const double scale = (double)in_size / 160;
        return (int)std::ceil(scale < 1.0 ? 1.0 : scale) * 2 + 1;
    }

    // ---- stage 1, device half: P-Net over all pyramid levels of all crops -> prob / reg maps + candidate list in HBM
    int stage1_gpu() {
        hipStream_t s = h->stream;
        int rc;
        std::vector<MtLevel> lv;
        // per launch: items + running totals of output elements (arena offsets are the same running totals):
        //   pyramid (resize) -> pooled conv1 map (conv1 + PReLU + pool) -> conv2 map -> conv3 cells (heads only)
        std::vector<MtItem> it_f, it_c2, it_c3;
        std::vector<long long> pre_in{0}, pre_p{0}, pre_c2{0}, pre_c3{0};
        levels.clear();
        cells = 0;
        for (int c = 0; c < n; ++c) {
            const int hh = imgs[c].h, ww = imgs[c].w;
            const double m = 12.0 / 20.0;                         // scale pyramid in double, as the package's Python floats
            double minl = std::min(hh, ww) * m, scale_i = m;
            while (minl >= 12) {
                const int sh = (int)(hh * scale_i + 1), sw = (int)(ww * scale_i + 1);
                const int c1h = sh - 2, c1w = sw - 2, ph = mt_pool_out(c1h, 2, 2), pw = mt_pool_out(c1w, 2, 2);
                const int c2h = ph - 2, c2w = pw - 2, c3h = c2h - 2, c3w = c2w - 2;
                levels.push_back(Level{c, scale_i, sh, sw, c3h, c3w, cells});
                lv.push_back(MtLevel{imgs[c].src, (long long)imgs[c].stride, hh, ww, sh, sw, pre_in.back()});
                it_f.push_back(MtItem{pre_in.back(), pre_p.back(), sh, sw});            // conv1 + pool: pyramid -> pooled map
                it_c2.push_back(MtItem{pre_p.back(), pre_c2.back(), ph, pw});
                it_c3.push_back(MtItem{pre_c2.back(), pre_c3.back(), c2h, c2w});
                pre_in.push_back(pre_in.back() + (long long)sh * sw * 3);
                pre_p.push_back(pre_p.back() + (long long)ph * pw * 10);
                pre_c2.push_back(pre_c2.back() + (long long)c2h * c2w * 16);
                pre_c3.push_back(pre_c3.back() + (long long)c3h * c3w * 32);
                cells += (long long)c3h * c3w;
                scale_i *= 0.709;
                minl *= 0.709;
            }
        }
        // crop / level tables of the device box path (mtcnn_boxes.hip); geo_ok = false: a field would overflow - host path
        cg.assign(n, MtCropGeo{});
        lg.assign(levels.size(), MtLevelGeo{});
        geo_ok = true;
        seg = tmp_bytes = tab = 0;
        {
            size_t li = 0;
            for (int c = 0; c < n; ++c) {
                MtCropGeo& g = cg[c];
                g = MtCropGeo{imgs[c].src, (long long)imgs[c].stride, imgs[c].h, imgs[c].w, (int)li, 0, seg, tmp_bytes, (int)tab, 0};
                long long crop_cells = 0;
                while (li < levels.size() && levels[li].crop == c) {
                    const Level& L = levels[li];
                    const long long lc = (long long)std::max(L.oh, 0) * std::max(L.ow, 0);
                    if (lc >= (1ll << 27)) geo_ok = false;           // cell index field of the sort key
                    lg[li] = MtLevelGeo{L.cell_off, L.oh, L.ow, (float)L.scale, 0};
                    crop_cells += lc;
                    ++li;
                    ++g.nlevels;
                }
                if (g.nlevels > 31) geo_ok = false;
                seg += std::min<long long>(crop_cells, kMtCap2);
                tmp_bytes += ((long long)imgs[c].h * 160 * 3 + 255) & ~255ll;
                tab += 160ll * pil_ksize(imgs[c].w) + 320 + 160ll * pil_ksize(imgs[c].h) + 320;
                if (tab > (1ll << 30) || seg > (1ll << 30)) geo_ok = false;
            }
        }
        mark("s1 host: pyramid tables");
        const int nl = (int)levels.size();
        // counters of the step in ONE zeroed block: [candidate count (4 words)][counts n][first_a n + 1][first_b n + 1][meta 4]
        if ((rc = ensure(h, &S->cnt, ((size_t)n * 3 + 2 + 8) * 4))) return rc;
        DFD_HIP_TRY(h, hipMemsetAsync(S->cnt.p, 0, ((size_t)n * 3 + 2 + 8) * 4, s));
        {
            // every descriptor table of the step in ONE upload through the mailbox (copied before the call returns):
            // levels | items of the three ragged launches | their running totals | crop geometry | level geometry
            std::vector<MtItem> items;
            std::vector<long long> pres;
            const std::vector<MtItem>* its[3] = {&it_f, &it_c2, &it_c3};
            const std::vector<long long>* prs[4] = {&pre_in, &pre_p, &pre_c2, &pre_c3};
            for (auto* v : its) items.insert(items.end(), v->begin(), v->end());
            for (auto* v : prs) pres.insert(pres.end(), v->begin(), v->end());
            auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
            const size_t o_lv = 0, o_items = al(o_lv + lv.size() * sizeof(MtLevel)), o_pre = al(o_items + items.size() * sizeof(MtItem)),
                         o_cg = al(o_pre + pres.size() * 8), o_lg = al(o_cg + cg.size() * sizeof(MtCropGeo)),
                         total = al(o_lg + lg.size() * sizeof(MtLevelGeo)) + 16;
            std::vector<char> blob(total, 0);
            if (!lv.empty()) memcpy(blob.data() + o_lv, lv.data(), lv.size() * sizeof(MtLevel));
            if (!items.empty()) memcpy(blob.data() + o_items, items.data(), items.size() * sizeof(MtItem));
            memcpy(blob.data() + o_pre, pres.data(), pres.size() * 8);
            memcpy(blob.data() + o_cg, cg.data(), cg.size() * sizeof(MtCropGeo));
            if (!lg.empty()) memcpy(blob.data() + o_lg, lg.data(), lg.size() * sizeof(MtLevelGeo));
            if ((rc = upload(&S->d_lv, blob))) return rc;
            const char* base = (const char*)S->d_lv.p;
            d_levels = (const MtLevel*)(base + o_lv);
            d_items = (const MtItem*)(base + o_items);
            d_pre = (const long long*)(base + o_pre);
            dcg = (const MtCropGeo*)(base + o_cg);
            dlg = (const MtLevelGeo*)(base + o_lg);
        }
        if (nl) {
            const MtItem* di = d_items;
            const long long* dp = d_pre;
            auto item_at = [&](int k) { return di + (size_t)k * nl; };
            auto pre_at = [&](int k) { return dp + (size_t)k * (nl + 1); };
            if ((rc = ensure(h, &S->in, pre_in.back() * 4))) return rc;
            if ((rc = ensure(h, &S->a0, std::max(pre_c2.back(), (long long)4) * 4))) return rc;
            if ((rc = ensure(h, &S->a1, std::max(pre_p.back(), (long long)4) * 4 + 64))) return rc;   // + the MFMA kernel's 2-float row overrun
            if ((rc = ensure(h, &S->prob, cells * 4))) return rc;
            if ((rc = ensure(h, &S->reg, cells * 16))) return rc;
            float *in = (float*)S->in.p, *a0 = (float*)S->a0.p, *a1 = (float*)S->a1.p;
            // conv2's MFMA rows (KR = 32 for 3 x 10 floats) read two floats past a window row; zero weight planes cancel
            // them - unless they are NaN / Inf, which the last pixel of the last level would pick up from this pad
            // (fresh from hipMalloc, or R-/O-Net leftovers): 0 x NaN = NaN, and `p >= thr` would drop the cell silently
            DFD_HIP_TRY(h, hipMemsetAsync((char*)a1 + (size_t)std::max(pre_p.back(), (long long)4) * 4, 0, 64, s));
            launch_mt_area_resize_ragged(d_levels, pre_at(0), nl, pre_in.back(), in, s);
            // candidate list: 16 bytes of counter, then the records
            if ((rc = ensure(h, &S->cand, 16 + (size_t)cells * sizeof(MtCand)))) return rc;
            unsigned* d_count = (unsigned*)S->cnt.p;                 // zeroed above
            MtCand* d_cand = (MtCand*)((char*)S->cand.p + 16);
            const MtPnetHeads heads{S->p41.w, S->p41.b, S->p42.w, S->p42.b, (float*)S->prob.p, (float*)S->reg.p,
                                    d_cand, d_count, (unsigned)cells, 0.6f};                  // thresholds[0], >=, float32
            // conv1 + PReLU + pool in one launch (the 10-channel conv map is never stored), conv2, then conv3 with both
            // 1x1 heads and the softmax evaluated from its registers (prob [cell], reg [cell][4], candidates; the
            // 32-channel map is never stored either)
            launch_mt_pnet_conv1_pool(in, S->p1w_pad, S->p1.b, S->p1.a, a1, item_at(0), pre_at(1), nl, pre_p.back(), s);
            bool ok = true;
            // conv2 / conv3 on the bf16 MFMA with exact three-term operands (DFD_MT_PNET_MFMA=0: the register-blocked VALU kernels)
            static const bool pnet_mfma = !(getenv("DFD_MT_PNET_MFMA") && atoi(getenv("DFD_MT_PNET_MFMA")) == 0);
            if (pnet_mfma) {
                const unsigned short* w2 = split_weights(h, S->p2m, 16, 96);
                const unsigned short* w3 = split_weights(h, S->p3m, 32, 160);
                if (!w2 || !w3) return DFD_ERR_HIP;
                ok = ok && launch_mt_pnet_mfma(a1, w2, (int)split_weights_count(16, 96), 128, S->p2.b, S->p2.a, a0, item_at(1), pre_at(2), nl,
                                               pre_c2.back(), 10, 16, nullptr, s);
                ok = ok && launch_mt_pnet_mfma(a0, w3, (int)split_weights_count(32, 160), 192, S->p3.b, S->p3.a, nullptr, item_at(2), pre_at(3), nl,
                                               pre_c3.back(), 16, 32, &heads, s);
            } else {
                ok = ok && launch_mt_convpx_ragged(a1, S->p2.w, S->p2.b, S->p2.a, a0, item_at(1), pre_at(2), nl, pre_c2.back(), 10, 16, 3, nullptr, s);
                ok = ok && launch_mt_convpx_ragged(a0, S->p3.w, S->p3.b, S->p3.a, nullptr, item_at(2), pre_at(3), nl, pre_c3.back(), 16, 32, 3, &heads, s);
            }
            if (!ok) return fail(h, DFD_ERR_STATE, "mtcnn: no P-Net kernel instance for this layer shape");
            DFD_HIP_TRY(h, hipGetLastError());
            if (tap_name) {                                   // parity taps read whole maps
                tap_prob.resize(cells);
                DFD_HIP_TRY(h, hipMemcpyAsync(tap_prob.data(), S->prob.p, cells * 4, hipMemcpyDeviceToHost, s));
                if ((rc = download(S->reg.p, (size_t)cells * 4, &tap_reg))) return rc;
            }
        }
        int level_in_crop = 0, prev_crop = -1;
        for (const Level& L : levels) {                      // parity taps (crop 0 only)
            level_in_crop = L.crop == prev_crop ? level_in_crop + 1 : 0;
            prev_crop = L.crop;
            if (L.crop != 0 || !tap_name) continue;
            const float* P = tap_prob.data() + L.cell_off;
            const float* R = tap_reg.data() + L.cell_off * 4;
            if (want("pnet.prob." + std::to_string(level_in_crop))) {
                tap->assign(P, P + (size_t)std::max(L.oh, 0) * std::max(L.ow, 0));
                tap_dims[0] = L.oh; tap_dims[1] = L.ow; tap_dims[2] = 1;
            }
            if (want("pnet.reg." + std::to_string(level_in_crop))) {
                tap->assign(R, R + (size_t)std::max(L.oh, 0) * std::max(L.ow, 0) * 4);
                tap_dims[0] = L.oh; tap_dims[1] = L.ow; tap_dims[2] = 4;
            }
        }
        mark("s1 gpu: P-Net launches");
        return DFD_OK;
    }

    // ---- stage 1, host half (host path): candidates -> per-crop boxes after NMS, regression and rerec
    int stage1_host(std::vector<std::vector<Box>>* out) {
        out->assign(n, {});
        std::vector<MtCand> cands;
        if (!levels.empty()) {
            const unsigned* d_count = (const unsigned*)S->cnt.p;
            const MtCand* d_cand = (const MtCand*)((const char*)S->cand.p + 16);
            // the candidates (cells at or above the threshold), not the maps: count first, then that many records, both
            // through the mailbox; the atomic append order is restored to (level, y, x) by sorting on the cell
            const unsigned* pc = (const unsigned*)mailbox_d2h(h, d_count, 4);
            if (!pc) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
            DFD_HIP_TRY(h, stream_sync(h));
            const size_t nc = std::min<size_t>(*pc, (size_t)cells);
            if (nc) {
                const MtCand* pr = (const MtCand*)mailbox_d2h(h, d_cand, nc * sizeof(MtCand));
                if (!pr) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
                DFD_HIP_TRY(h, stream_sync(h));
                cands.assign(pr, pr + nc);
            }
            std::sort(cands.begin(), cands.end(), [](const MtCand& a, const MtCand& b) { return a.cell < b.cell; });
        }
        mark("s1 host: candidate download");
        // host: generateBoundingBox per level, per-scale NMS, cross-scale NMS, regression, rerec - per crop
        std::vector<std::vector<Box>> all(n);
        std::vector<std::vector<Box>> kept(levels.size());
        // first candidate of each level in the cell-sorted list (levels own consecutive cell ranges)
        std::vector<size_t> lfirst(levels.size() + 1, cands.size());
        {
            size_t k = 0;
            for (size_t li = 0; li < levels.size(); ++li) {
                while (k < cands.size() && (long long)cands[k].cell < levels[li].cell_off) ++k;
                lfirst[li] = k;
            }
        }
        parallel_for((int)levels.size(), cands.size() * 100, [&](int li) {
            const Level& L = levels[li];
            std::vector<Box> bs;
            std::vector<int> cell, grid;
            const float fs = (float)L.scale;
            for (size_t k = lfirst[li]; k < lfirst[li + 1]; ++k) {          // (y, x) order within the level
                const MtCand& q = cands[k];
                const int rel = (int)((long long)q.cell - L.cell_off), y = rel / L.ow, x = rel % L.ow;
                Box b{};
                b.x1 = std::floor((2.f * (float)x + 1.f) / fs);
                b.y1 = std::floor((2.f * (float)y + 1.f) / fs);
                b.x2 = std::floor((2.f * (float)x + 12.f) / fs);
                b.y2 = std::floor((2.f * (float)y + 12.f) / fs);
                b.score = q.p;
                for (int r = 0; r < 4; ++r) b.r[r] = q.r[r];
                bs.push_back(b);
                cell.push_back(rel);
            }
            for (int i : nms_iou_level(bs, cell, L.oh, L.ow, 0.5f, &grid)) kept[li].push_back(bs[i]);
        });
        size_t pairs = 0;
        for (size_t li = 0; li < levels.size(); ++li) all[levels[li].crop].insert(all[levels[li].crop].end(), kept[li].begin(), kept[li].end());
        for (int c = 0; c < n; ++c) pairs += all[c].size() * all[c].size();
        mark("s1 host: scan + level NMS");
        parallel_for(n, pairs, [&](int c) {
            for (int i : nms_iou(all[c], 0.7f)) {
                Box b = all[c][i];
                const float regw = b.x2 - b.x1, regh = b.y2 - b.y1;
                const float x1 = b.x1 + b.r[0] * regw, y1 = b.y1 + b.r[1] * regh, x2 = b.x2 + b.r[2] * regw, y2 = b.y2 + b.r[3] * regh;
                b.x1 = x1; b.y1 = y1; b.x2 = x2; b.y2 = y2;
                rerec(b);
                (*out)[c].push_back(b);
            }
        });
        mark("s1 host: cross-level NMS");
        tap_boxes("stage1", (*out)[0]);
        return DFD_OK;
    }

    // R-Net (24) / O-Net (48) over `total` windows on the device (at most kChunk per launch set): face probability
    // S->prob [total], regression S->reg [total][4]; nothing is read back
    int refine_gpu(bool onet, const MtSrcWindow* wd_all, int total) {
        hipStream_t s = h->stream;
        const int sz = onet ? 48 : 24;
        constexpr int kChunk = 4096;
        int rc;
        const int mmax = std::min(kChunk, total);
        const size_t big = (size_t)mmax * (sz - 2) * (sz - 2) * 32 * 4;
        if ((rc = ensure(h, &S->in, (size_t)mmax * sz * sz * 3 * 4))) return rc;
        if ((rc = ensure(h, &S->a0, big))) return rc;
        if ((rc = ensure(h, &S->a1, big))) return rc;
        if ((rc = ensure(h, &S->prob, (size_t)total * 4))) return rc;
        if ((rc = ensure(h, &S->reg, (size_t)total * 4 * 4))) return rc;
        for (int start = 0; start < total; start += kChunk) {
            const int m = std::min(kChunk, total - start);
            const MtSrcWindow* wd = wd_all + start;                  // (the ensure calls above never move the window list)
            float *in = (float*)S->in.p, *a0 = (float*)S->a0.p, *a1 = (float*)S->a1.p;
            float *pr = (float*)S->prob.p + start, *rg = (float*)S->reg.p + (size_t)start * 4;
            launch_mt_area_resize_multi(wd, m, sz, sz, in, s);
            if (!onet) {
                // conv1 (22, 32 ch = 28 + 4 zero) + PReLU + pool (11) in one launch
                if (!launch_mt_conv1_pool(in, S->r1p.w, S->r1p.b, S->r1p.a, a1, m, 24, 24, 32, s)) return fail(h, DFD_ERR_STATE, "mtcnn: conv1+pool shape");
                if ((rc = gemm_conv_prelu(h, S->r2g, a1, a0, m, 11, 11))) return rc;                // 9, 64 ch (48 + 16 zero)
                launch_mt_maxpool(a0, a1, m, 9, 9, 64, 3, 2, s);                                    // 4
                if ((rc = gemm_conv_prelu(h, S->r3g, a1, a0, m, 4, 4))) return rc;                  // 3 -> [m][3][3][64]
                if ((rc = gemm_dense_prelu(h, S->r4, a0, a1, m))) return rc;                        // 576 -> 128
                launch_mt_heads(a1, S->r51.w, S->r51.b, S->r52.w, S->r52.b, pr, rg, m, 128, s);
            } else {
                // conv1 (46) + PReLU + pool (23) in one launch
                if (!launch_mt_conv1_pool(in, S->o1.w, S->o1.b, S->o1.a, a1, m, 48, 48, 32, s)) return fail(h, DFD_ERR_STATE, "mtcnn: conv1+pool shape");
                if ((rc = gemm_conv_prelu(h, S->o2g, a1, a0, m, 23, 23))) return rc;                // 21
                launch_mt_maxpool(a0, a1, m, 21, 21, 64, 3, 2, s);                                  // 10
                if ((rc = gemm_conv_prelu(h, S->o3g, a1, a0, m, 10, 10))) return rc;                // 8
                launch_mt_maxpool(a0, a1, m, 8, 8, 64, 2, 2, s);                                    // 4
                if ((rc = gemm_conv_prelu(h, S->o4g, a1, a0, m, 4, 4))) return rc;                  // 3 -> [m][3][3][128]
                if ((rc = gemm_dense_prelu(h, S->o5, a0, a1, m))) return rc;                        // 1152 -> 256
                launch_mt_heads(a1, S->o61.w, S->o61.b, S->o62.w, S->o62.b, pr, rg, m, 256, s);
                // dense6_3 (landmarks) does not influence the selected crop: not evaluated
            }
            DFD_HIP_TRY(h, hipGetLastError());
        }
        return DFD_OK;
    }

    // host path: upload the windows, run the network, read probability [n] and regression [n][4] back
    int refine(bool onet, const std::vector<MtSrcWindow>& wins, std::vector<float>* prob, std::vector<float>* reg) {
        const int total = (int)wins.size();
        prob->clear();
        reg->clear();
        int rc;
        if ((rc = upload(&S->win, wins))) return rc;
        if ((rc = refine_gpu(onet, (const MtSrcWindow*)S->win.p, total))) return rc;
        const float* pp = (const float*)mailbox_d2h(h, S->prob.p, (size_t)total * 4);
        const float* rr = (const float*)mailbox_d2h(h, S->reg.p, (size_t)total * 16);
        if (!pp || !rr) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
        DFD_HIP_TRY(h, hipGetLastError());
        DFD_HIP_TRY(h, stream_sync(h));
        prob->assign(pp, pp + total);
        reg->assign(rr, rr + (size_t)total * 4);
        return DFD_OK;
    }

    // detect_face for all crops -> per-crop boxes after the three stages
    int run(std::vector<std::vector<Box>>* out) {
        int rc;
        static const bool verbose = getenv("DFD_MT_VERBOSE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::vector<std::vector<Box>> boxes;
        if (!stage1_done && (rc = stage1_gpu())) return rc;
        if ((rc = stage1_host(&boxes))) return rc;
        if (verbose) {
            size_t tot = 0;
            for (auto& b : boxes) tot += b.size();
            fprintf(stderr, "[dfd] mtcnn %d crops: stage 1 -> %zu boxes, %.2f ms\n", n, tot, since());
        }
        for (int stage = 2; stage <= 3; ++stage) {
            std::vector<MtSrcWindow> wins;
            std::vector<std::vector<Box>> live(n);
            for (int c = 0; c < n; ++c)
                for (const Box& b : boxes[c]) {
                    MtWindow w;
                    if (!window_of(b, imgs[c].w, imgs[c].h, &w)) continue;
                    wins.push_back(MtSrcWindow{imgs[c].src, (long long)imgs[c].stride, w.x, w.y, w.w, w.h});
                    live[c].push_back(b);
                }
            std::vector<float> prob, reg;
            mark("host: windows");
            if (!wins.empty() && (rc = refine(stage == 3, wins, &prob, &reg))) return rc;
            mark("gpu: refine + download");
            std::vector<size_t> first(n + 1, 0);
            size_t pairs = 0;
            for (int c = 0; c < n; ++c) {
                first[c + 1] = first[c] + live[c].size();
                pairs += live[c].size() * live[c].size();
            }
            if (!live[0].empty()) {
                if (want(stage == 2 ? "rnet.prob" : "onet.prob")) {
                    tap->assign(prob.begin(), prob.begin() + first[1]);
                    tap_dims[0] = (int)first[1]; tap_dims[1] = 1; tap_dims[2] = 1;
                }
                if (want(stage == 2 ? "rnet.reg" : "onet.reg")) {
                    tap->assign(reg.begin(), reg.begin() + first[1] * 4);
                    tap_dims[0] = (int)first[1]; tap_dims[1] = 4; tap_dims[2] = 1;
                }
            }
            parallel_for(n, pairs, [&](int c) {
                size_t k = first[c];
                std::vector<Box> pass;
                for (const Box& lb : live[c]) {
                    if (prob[k] > 0.7f) {                        // thresholds[1] = thresholds[2] = 0.7, strict
                        Box b = lb;
                        b.score = prob[k];
                        for (int r = 0; r < 4; ++r) b.r[r] = reg[k * 4 + r];
                        pass.push_back(b);
                    }
                    ++k;
                }
                boxes[c].clear();
                if (stage == 2) {
                    for (int i : nms_iou(pass, 0.7f)) { Box b = pass[i]; bbreg(b); rerec(b); boxes[c].push_back(b); }
                } else {
                    for (Box& b : pass) bbreg(b);
                    for (int i : nms_min(pass, 0.7f)) boxes[c].push_back(pass[i]);
                }
            });
            mark("host: threshold + NMS");
            tap_boxes(stage == 2 ? "stage2" : "stage3", boxes[0]);
            if (verbose) {
                size_t tot = 0;
                for (auto& b : boxes) tot += b.size();
                fprintf(stderr, "[dfd] mtcnn stage %d: %zu windows -> %zu boxes, %.2f ms so far\n", stage, wins.size(), tot, since());
            }
        }
        *out = boxes;
        return DFD_OK;
    }

    // The whole step with the box bookkeeping on the device (mtcnn_boxes.hip): per stage one block per crop does what the
    // host path does between the networks; the host reads back the window count of all crops after stages 1 and 2
    // (launch sizes of the next network) and one result row per crop at the end - three stream waits, no box on the
    // host.  *done = false: a crop exceeded the blocks' capacity (overflow flag) - nothing was written to the outputs
    // and the caller runs the host path on the P-Net results that are already in HBM.
    int run_device(uint8_t* faces_out, float* boxes_out, char* found, bool* done) {
        hipStream_t s = h->stream;
        int rc;
        *done = false;
        static const bool verbose = getenv("DFD_MT_VERBOSE") != nullptr;
        if ((rc = stage1_gpu())) return rc;
        stage1_done = true;
        if (!geo_ok) return DFD_OK;
        const size_t segn = (size_t)std::max<long long>(seg, 1);
        if ((rc = ensure(h, &S->rows_a, segn * sizeof(MtRow)))) return rc;
        if ((rc = ensure(h, &S->wins_a, segn * sizeof(MtSrcWindow)))) return rc;
        if ((rc = ensure(h, &S->rows_b, segn * sizeof(MtRow)))) return rc;
        if ((rc = ensure(h, &S->wins_b, segn * sizeof(MtSrcWindow)))) return rc;
        if ((rc = ensure(h, &S->res, (size_t)n * 8 * 4))) return rc;
        if ((rc = ensure(h, &S->coef, (size_t)std::max<long long>(tab, 4) * 4))) return rc;
        if ((rc = ensure(h, &S->bnd, (size_t)n * sizeof(MtFaceJob)))) return rc;
        if ((rc = ensure(h, &S->tmp, (size_t)std::max<long long>(tmp_bytes, 16)))) return rc;
        if (tap_name && (rc = ensure(h, &S->taprows, (size_t)kMtCap1 * sizeof(MtRow)))) return rc;
        int* counts = (int*)S->cnt.p + 4;                        // (behind the candidate counter; zeroed by stage1_gpu)
        int *first_a = counts + n, *first_b = first_a + n + 1, *meta = first_b + n + 1;
        MtRow *rows_a = (MtRow*)S->rows_a.p, *rows_b = (MtRow*)S->rows_b.p, *taprows = tap_name ? (MtRow*)S->taprows.p : nullptr;
        MtSrcWindow *wins_a = (MtSrcWindow*)S->wins_a.p, *wins_b = (MtSrcWindow*)S->wins_b.p;
        // rows of crop 0 after a stage (parity taps): meta[2] of them at taprows
        auto tap_stage = [&](const char* name, const int* meta_host) -> int {
            if (!want(name)) return DFD_OK;
            const int k = std::min(meta_host[2], kMtCap1);
            std::vector<float> rows;
            int rc2 = download(S->taprows.p, (size_t)k * 5, &rows);
            if (rc2) return rc2;
            *tap = rows;
            tap_dims[0] = k; tap_dims[1] = 5; tap_dims[2] = 1;
            return DFD_OK;
        };
        // first_x[0 .. cnt) of crop 0: the network's outputs of its windows (parity taps)
        auto tap_net = [&](const char* pname, const char* rname, int cnt0) -> int {
            if (cnt0 <= 0) return DFD_OK;
            if (want(pname)) {
                int rc2 = download(S->prob.p, (size_t)cnt0, tap);
                if (rc2) return rc2;
                tap_dims[0] = cnt0; tap_dims[1] = 1; tap_dims[2] = 1;
            }
            if (want(rname)) {
                int rc2 = download(S->reg.p, (size_t)cnt0 * 4, tap);
                if (rc2) return rc2;
                tap_dims[0] = cnt0; tap_dims[1] = 4; tap_dims[2] = 1;
            }
            return DFD_OK;
        };
        // the count words after a stage: {windows of all crops, overflow, rows of crop 0, windows of crop 0}
        auto read_meta = [&](int out[4]) -> int {
            const int* pm = (const int*)mailbox_d2h(h, meta, 16);
            if (!pm) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
            DFD_HIP_TRY(h, hipGetLastError());
            DFD_HIP_TRY(h, stream_sync(h));
            out[0] = pm[0]; out[1] = pm[1]; out[2] = pm[2]; out[3] = pm[3];
            return DFD_OK;
        };
        int mh[4] = {0, 0, 0, 0};
        int m2 = 0, m3 = 0;
        if (!levels.empty()) {
            launch_mt_stage1_boxes(dcg, dlg, n, (const float*)S->prob.p, (const float*)S->reg.p, 0.6f,
                                   rows_a, wins_a, counts, meta, taprows, s);
            launch_mt_compact(counts, n, dcg, nullptr, rows_a, wins_a, rows_b, wins_b, first_a, meta, s);
            if ((rc = read_meta(mh))) return rc;
            mark("s1 gpu: boxes + count");
            if (mh[1]) {
                if (verbose) fprintf(stderr, "[dfd] mtcnn: a crop exceeds the device box capacity - host path\n");
                return DFD_OK;
            }
            if ((rc = tap_stage("stage1", mh))) return rc;
            m2 = mh[0];
        }
        else if (want("stage1")) {
            tap->clear();
            tap_dims[0] = 0; tap_dims[1] = 5; tap_dims[2] = 1;
        }
        if (verbose) fprintf(stderr, "[dfd] mtcnn %d crops (device boxes): stage 1 -> %d windows\n", n, m2);
        if (m2 > 0) {
            if ((rc = refine_gpu(false, wins_b, m2))) return rc;
            if ((rc = tap_net("rnet.prob", "rnet.reg", mh[3]))) return rc;
            launch_mt_refine_boxes(2, dcg, first_a, n, rows_b, (const float*)S->prob.p, (const float*)S->reg.p, 0.7f, 0.7f, rows_a, wins_a,
                                   counts, nullptr, nullptr, nullptr, taprows, meta, s);
            launch_mt_compact(counts, n, dcg, first_a, rows_a, wins_a, rows_b, wins_b, first_b, meta, s);
            if ((rc = read_meta(mh))) return rc;
            mark("s2 gpu: R-Net + boxes + count");
            if ((rc = tap_stage("stage2", mh))) return rc;
            m3 = mh[0];
            if (verbose) fprintf(stderr, "[dfd] mtcnn stage 2 (device boxes): %d windows -> %d\n", m2, m3);
        } else if (want("stage2")) {
            tap->clear();
            tap_dims[0] = 0; tap_dims[1] = 5; tap_dims[2] = 1;
        }
        if (m3 > 0) {
            if ((rc = refine_gpu(true, wins_b, m3))) return rc;
            if ((rc = tap_net("onet.prob", "onet.reg", mh[3]))) return rc;
        }
        // (with no window left first_b is all zero: every crop gets a zero-filled face and found = 0)
        launch_mt_refine_boxes(3, dcg, first_b, n, rows_b, (const float*)S->prob.p, (const float*)S->reg.p, 0.7f, 0.7f, nullptr, nullptr,
                               nullptr, (MtFaceJob*)S->bnd.p, (float*)S->res.p, (int*)S->coef.p, taprows, meta, s);
        launch_mt_extract_faces((const MtFaceJob*)S->bnd.p, n, (const int*)S->coef.p, faces_out, (uint8_t*)S->tmp.p, s);
        DFD_HIP_TRY(h, hipGetLastError());
        const float* pr = (const float*)mailbox_d2h(h, S->res.p, (size_t)n * 8 * 4);
        const int* pm = tap_name ? (const int*)mailbox_d2h(h, meta, 12) : nullptr;
        if (!pr || (tap_name && !pm)) return fail(h, DFD_ERR_HIP, "mtcnn: mailbox allocation failed");
        DFD_HIP_TRY(h, stream_sync(h));
        mark("s3 gpu: O-Net + boxes + extract");
        int m3h[3] = {0, 0, pm ? pm[2] : 0};
        for (int i = 0; i < n; ++i) {
            const float* r = pr + (size_t)i * 8;
            found[i] = r[0] != 0.f;
            if (boxes_out && r[1] != 0.f) memcpy(boxes_out + 5 * i, r + 2, 5 * sizeof(float));
        }
        if (want("stage3")) {
            if (m3 > 0) {
                if ((rc = tap_stage("stage3", m3h))) return rc;
            } else {
                tap->clear();
                tap_dims[0] = 0; tap_dims[1] = 5; tap_dims[2] = 1;
            }
        }
        *done = true;
        return DFD_OK;
    }
};

}  // namespace

int mtcnn_init(dfd_handle* h) {
    if (h->tensors.find("mtcnn.pnet.conv1.w") == h->tensors.end()) return DFD_OK;      // blob without the cascade
    MtcnnState* S = new MtcnnState();
    h->mtcnn = S;
    bool ok = true;
    S->p1 = mt_conv(h, "pnet.conv1", 10, 3, 3, "pnet.prelu1", &ok);
    S->p2 = mt_conv(h, "pnet.conv2", 16, 10, 3, "pnet.prelu2", &ok);
    S->p3 = mt_conv(h, "pnet.conv3", 32, 16, 3, "pnet.prelu3", &ok);
    S->p41 = mt_conv(h, "pnet.conv4_1", 2, 32, 1, nullptr, &ok);
    S->p42 = mt_conv(h, "pnet.conv4_2", 4, 32, 1, nullptr, &ok);
    S->r1 = mt_conv(h, "rnet.conv1", 28, 3, 3, "rnet.prelu1", &ok);
    S->r2 = mt_conv(h, "rnet.conv2", 48, 28, 3, "rnet.prelu2", &ok);
    S->r3 = mt_conv(h, "rnet.conv3", 64, 48, 2, "rnet.prelu3", &ok);
    S->r4 = mt_dense(h, "rnet.dense4", 128, 576, "rnet.prelu4", &ok);
    S->r51 = mt_dense(h, "rnet.dense5_1", 2, 128, nullptr, &ok);
    S->r52 = mt_dense(h, "rnet.dense5_2", 4, 128, nullptr, &ok);
    S->o1 = mt_conv(h, "onet.conv1", 32, 3, 3, "onet.prelu1", &ok);
    S->o2 = mt_conv(h, "onet.conv2", 64, 32, 3, "onet.prelu2", &ok);
    S->o3 = mt_conv(h, "onet.conv3", 64, 64, 3, "onet.prelu3", &ok);
    S->o4 = mt_conv(h, "onet.conv4", 128, 64, 2, "onet.prelu4", &ok);
    S->o5 = mt_dense(h, "onet.dense5", 256, 1152, "onet.prelu5", &ok);
    S->o61 = mt_dense(h, "onet.dense6_1", 2, 256, nullptr, &ok);
    S->o62 = mt_dense(h, "onet.dense6_2", 4, 256, nullptr, &ok);
    S->o63 = mt_dense(h, "onet.dense6_3", 10, 256, nullptr, &ok);
    S->p2m = mt_tensor(h, "mtcnn.pnet.conv2.wm", 16 * 96, &ok);
    S->p3m = mt_tensor(h, "mtcnn.pnet.conv3.wm", 32 * 160, &ok);
    S->r1p.co = 32; S->r1p.ci = 3; S->r1p.k = 3;
    S->r1p.w = mt_tensor(h, "mtcnn.rnet.conv1.wp", 3 * 3 * 3 * 32, &ok);
    S->r1p.b = mt_tensor(h, "mtcnn.rnet.conv1.bp", 32, &ok);
    S->r1p.a = mt_tensor(h, "mtcnn.rnet.conv1.ap", 32, &ok);
    S->r2g = mt_gemm_conv(h, "rnet.conv2", 64, 32, 3, &ok);
    S->r3g = mt_gemm_conv(h, "rnet.conv3", 64, 64, 2, &ok);
    S->o2g = mt_gemm_conv(h, "onet.conv2", 64, 32, 3, &ok);
    S->o3g = mt_gemm_conv(h, "onet.conv3", 64, 64, 3, &ok);
    S->o4g = mt_gemm_conv(h, "onet.conv4", 128, 64, 2, &ok);
    if (!ok) return DFD_ERR_BLOB;
    // [bias | slope] pairs for the GEMM epilogue's PReLU
    struct Pair { const float *b, *a; int n; const float** out; };
    const Pair pairs[] = {{S->r2g.b, S->r2g.a, S->r2g.co, &S->r2g.ba}, {S->r3g.b, S->r3g.a, S->r3g.co, &S->r3g.ba},
                          {S->o2g.b, S->o2g.a, S->o2g.co, &S->o2g.ba}, {S->o3g.b, S->o3g.a, S->o3g.co, &S->o3g.ba},
                          {S->o4g.b, S->o4g.a, S->o4g.co, &S->o4g.ba}, {S->r4.b, S->r4.a, S->r4.out, &S->r4.ba},
                          {S->o5.b, S->o5.a, S->o5.out, &S->o5.ba}};
    size_t floats = 288;
    for (const Pair& p : pairs) floats += 2 * (size_t)p.n;
    int rc = ensure(h, &S->bs, floats * 4);
    if (rc) return rc;
    float* dst = (float*)S->bs.p;
    DFD_HIP_TRY(h, hipMemsetAsync(dst, 0, 288 * 4, h->stream));
    DFD_HIP_TRY(h, hipMemcpyAsync(dst, S->p1.w, 270 * 4, hipMemcpyDeviceToDevice, h->stream));
    S->p1w_pad = dst;
    dst += 288;
    for (const Pair& p : pairs) {
        DFD_HIP_TRY(h, hipMemcpyAsync(dst, p.b, (size_t)p.n * 4, hipMemcpyDeviceToDevice, h->stream));
        DFD_HIP_TRY(h, hipMemcpyAsync(dst + p.n, p.a, (size_t)p.n * 4, hipMemcpyDeviceToDevice, h->stream));
        *p.out = dst;
        dst += 2 * p.n;
    }
    S->ready = true;
    return DFD_OK;
}

// MTCNN.forward on `n` BGR images already in HBM: per image the selected box, found flag, and the 160x160 BGR u8
// crop written to faces_out + i * 160*160*3 (zero-filled when not found).  found = 0: no face passed the cascade,
// or the selected box is degenerate (the package raises there and the reference call site returns None).
int mtcnn_align_batch_device(dfd_handle* h, const MtImage* imgs, int n, uint8_t* faces_out, float* boxes_out, char* found,
                             const char* tap_name, std::vector<float>* tap, int* tap_dims) {
    MtcnnState* S = h->mtcnn;
    if (!S || !S->ready) return fail(h, DFD_ERR_STATE, "the weights blob holds no MTCNN cascade");
    for (int i = 0; i < n; ++i)
        if (imgs[i].h <= 0 || imgs[i].w <= 0) return fail(h, DFD_ERR_ARG, "mtcnn: empty image");
    Cascade c{h, S, imgs, n, tap_name, tap, tap_dims};
    std::vector<std::vector<Box>> boxes;
    int rc;
    // box bookkeeping on the device unless DFD_MT_DEVICE_BOXES=0 (read per call: the tests switch it); a step that does
    // not fit the device blocks falls through to the host path with its P-Net results already in HBM
    const char* dv = getenv("DFD_MT_DEVICE_BOXES");
    if (!(dv && atoi(dv) == 0)) {
        bool done = false;
        if ((rc = c.run_device(faces_out, boxes_out, found, &done))) return rc;
        if (done) return DFD_OK;
    }
    if ((rc = c.run(&boxes))) return rc;
    hipStream_t s = h->stream;
    // selection + extract_face geometry on the host; all resize coefficient tables and the job list in one upload each,
    // all crops resampled by two launches
    std::vector<MtFaceJob> jobs(n);
    std::vector<int> tables;
    size_t tmp_bytes = 0;
    for (int i = 0; i < n; ++i) {
        found[i] = 0;
        MtFaceJob& j = jobs[i];
        j = MtFaceJob{imgs[i].src, (long long)imgs[i].stride, 0, 0, 160, 160, 0, 0, 0, 0, 0, 0, 0, 0};
        if (boxes[i].empty()) continue;
        // select_boxes(method="probability"): np.argsort(probs)[::-1][0] = the LAST of the ascending stable order
        int best = 0;
        for (int k = 1; k < (int)boxes[i].size(); ++k)
            if (boxes[i][k].score >= boxes[i][best].score) best = k;
        const Box& b = boxes[i][best];
        if (boxes_out) { float* o = boxes_out + 5 * i; o[0] = b.x1; o[1] = b.y1; o[2] = b.x2; o[3] = b.y2; o[4] = b.score; }
        // extract_face(margin 0): int() of the clipped float corners
        const int x1 = (int)std::max(b.x1, 0.f), y1 = (int)std::max(b.y1, 0.f);
        const int x2 = (int)std::min(b.x2, (float)imgs[i].w), y2 = (int)std::min(b.y2, (float)imgs[i].h);
        if (x2 <= x1 || y2 <= y1) continue;
        j.x1 = x1; j.y1 = y1; j.cw = x2 - x1; j.ch = y2 - y1;
        j.tmp_off = (long long)tmp_bytes;
        std::vector<int> coeff, bounds;
        if (j.cw != 160) {                                   // crop.resize((160, 160), BILINEAR): horizontal pass into tmp [ch][160][3]
            pil_coeffs(j.cw, 160, &coeff, &bounds, &j.kx);
            j.cx = (int)tables.size(); tables.insert(tables.end(), coeff.begin(), coeff.end());
            j.bx = (int)tables.size(); tables.insert(tables.end(), bounds.begin(), bounds.end());
            if (j.ch != 160) tmp_bytes += ((size_t)j.ch * 160 * 3 + 255) & ~(size_t)255;
        }
        if (j.ch != 160) {                                   // ... vertical pass into the face slot
            pil_coeffs(j.ch, 160, &coeff, &bounds, &j.ky);
            j.cy = (int)tables.size(); tables.insert(tables.end(), coeff.begin(), coeff.end());
            j.by = (int)tables.size(); tables.insert(tables.end(), bounds.begin(), bounds.end());
        }
        j.found = 1;
        found[i] = 1;
    }
    if ((rc = c.upload(&S->coef, tables))) return rc;
    if ((rc = c.upload(&S->bnd, jobs))) return rc;
    if ((rc = ensure(h, &S->tmp, std::max<size_t>(tmp_bytes, 16)))) return rc;
    launch_mt_extract_faces((const MtFaceJob*)S->bnd.p, n, (const int*)S->coef.p, faces_out, (uint8_t*)S->tmp.p, s);
    DFD_HIP_TRY(h, hipGetLastError());
    DFD_HIP_TRY(h, stream_sync(h));          // `tables` / `jobs` (host) feed the copies above
    return DFD_OK;
}

int mtcnn_align_device(dfd_handle* h, const uint8_t* img_dev, int hh, int ww, size_t stride, float* box_out, int* found,
                       const char* tap_name, std::vector<float>* tap, int* tap_dims) {
    MtcnnState* S = h->mtcnn;
    if (!S || !S->ready) return fail(h, DFD_ERR_STATE, "the weights blob holds no MTCNN cascade");
    int rc = ensure(h, &S->face, 160 * 160 * 3);
    if (rc) return rc;
    const MtImage img{img_dev, hh, ww, stride};
    char f = 0;
    float box[5] = {0, 0, 0, 0, 0};
    if ((rc = mtcnn_align_batch_device(h, &img, 1, (uint8_t*)S->face.p, box, &f, tap_name, tap, tap_dims))) return rc;
    if (box_out) memcpy(box_out, box, sizeof box);
    *found = f;
    return DFD_OK;
}

const uint8_t* mtcnn_face_dev(dfd_handle* h) { return h->mtcnn ? (const uint8_t*)h->mtcnn->face.p : nullptr; }

}  // namespace dfd

using namespace dfd;

extern "C" {

int dfd_has_mtcnn(const dfd_handle* h) { return h && h->mtcnn && h->mtcnn->ready ? 1 : 0; }

static int mt_upload(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride) {
    if (!bgr || hh <= 0 || ww <= 0 || stride < ww * 3) return fail(h, DFD_ERR_ARG, "mtcnn: bad pointer or geometry");
    int rc = ensure(h, &h->frame_buf, (size_t)hh * stride);
    if (rc) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return DFD_OK;
}

int dfd_mtcnn_align(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, float* face_chw_out, float* box_out,
                    int* found) {
    if (!h) return DFD_ERR_ARG;
    if (!found) return fail(h, DFD_ERR_ARG, "mtcnn_align: null found");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = mt_upload(h, bgr, hh, ww, stride);
    if (rc) return rc;
    if ((rc = mtcnn_align_device(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, box_out, found, nullptr, nullptr, nullptr)))
        return rc;
    if (*found && face_chw_out) {
        if ((rc = ensure(h, &h->mtcnn->in, 3 * 160 * 160 * 4))) return rc;
        launch_mt_face_chw(mtcnn_face_dev(h), (float*)h->mtcnn->in.p, 160 * 160, h->stream);
        DFD_HIP_TRY(h, hipMemcpyAsync(face_chw_out, h->mtcnn->in.p, 3 * 160 * 160 * 4, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, stream_sync(h));
    }
    return DFD_OK;
}

int dfd_mtcnn_tap(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, const char* name, float* out,
                  size_t capacity, size_t* count, int* dims) {
    if (!h) return DFD_ERR_ARG;
    if (!name || !out || !count || !dims) return fail(h, DFD_ERR_ARG, "mtcnn_tap: null argument");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = mt_upload(h, bgr, hh, ww, stride);
    if (rc) return rc;
    std::vector<float> tap;
    int found = 0;
    dims[0] = dims[1] = dims[2] = -1;
    float box[5];
    if ((rc = mtcnn_align_device(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, box, &found, name, &tap, dims))) return rc;
    if (dims[0] < 0) return fail(h, DFD_ERR_ARG, "mtcnn_tap: no stage named '%s' for this image", name);
    if (tap.size() > capacity) return fail(h, DFD_ERR_ARG, "mtcnn_tap '%s' needs %zu floats, capacity %zu", name, tap.size(), capacity);
    memcpy(out, tap.data(), tap.size() * 4);
    *count = tap.size();
    return DFD_OK;
}

}  // extern "C"
