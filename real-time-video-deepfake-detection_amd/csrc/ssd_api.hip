// SSD-style face detector: layer plan, workspace and the dfd_detect_faces entry point.
// The layer table restates ssd_arch.py (kept in step by tests/test_ssd_gpu.py, which checks
// every intermediate tensor against the oracle run on the Python table).
#include <cmath>

#include "b0_kernels.h"
#include "dfd_common.h"
#include "ssd_kernels.h"

using namespace dfd;

namespace dfd {

enum SsdKind { SK_CONV1, SK_POOL, SK_CONV, SK_NORM };

struct SsdLayer {
    const char* name;
    SsdKind kind;
    const char* src;
    int cin, cout, k, stride, pad, dil;
    bool relu;
    const char* res;
};

static const SsdLayer kSsdLayers[] = {
    {"conv1", SK_CONV1, "data", 3, 32, 7, 2, 3, 1, true, nullptr},
    {"pool1", SK_POOL, "conv1", 32, 32, 3, 2, 0, 1, false, nullptr},
    {"res2a", SK_CONV, "pool1", 32, 32, 3, 1, 1, 1, true, nullptr},
    {"res2b", SK_CONV, "res2a", 32, 32, 3, 1, 1, 1, true, "pool1"},
    {"res3p", SK_CONV, "res2b", 32, 128, 1, 2, 0, 1, false, nullptr},
    {"res3a", SK_CONV, "res2b", 32, 128, 3, 2, 1, 1, true, nullptr},
    {"res3b", SK_CONV, "res3a", 128, 128, 3, 1, 1, 1, true, "res3p"},
    {"res4p", SK_CONV, "res3b", 128, 256, 1, 2, 0, 1, false, nullptr},
    {"res4a", SK_CONV, "res3b", 128, 256, 3, 2, 1, 1, true, nullptr},
    {"res4b", SK_CONV, "res4a", 256, 256, 3, 1, 1, 1, true, "res4p"},
    {"res5a", SK_CONV, "res4b", 256, 256, 3, 1, 2, 2, true, nullptr},
    {"res5b", SK_CONV, "res5a", 256, 256, 3, 1, 2, 2, true, "res4b"},
    {"conv6_1", SK_CONV, "res5b", 256, 128, 1, 1, 0, 1, true, nullptr},
    {"conv6_2", SK_CONV, "conv6_1", 128, 256, 3, 2, 1, 1, true, nullptr},
    {"conv7_1", SK_CONV, "conv6_2", 256, 64, 1, 1, 0, 1, true, nullptr},
    {"conv7_2", SK_CONV, "conv7_1", 64, 128, 3, 2, 1, 1, true, nullptr},
    {"conv8_1", SK_CONV, "conv7_2", 128, 64, 1, 1, 0, 1, true, nullptr},
    {"conv8_2", SK_CONV, "conv8_1", 64, 128, 3, 1, 0, 1, true, nullptr},
    {"conv9_1", SK_CONV, "conv8_2", 128, 64, 1, 1, 0, 1, true, nullptr},
    {"conv9_2", SK_CONV, "conv9_1", 64, 128, 3, 1, 0, 1, true, nullptr},
    {"norm3", SK_NORM, "res3b", 128, 128, 1, 1, 0, 1, false, nullptr},
};

struct SsdSource { const char* tensor; int c, map; double mn, mx; int nar; double ar[2]; double step; };
static const SsdSource kSsdSources[6] = {
    {"norm3", 128, 38, 30, 60, 1, {2, 0}, 8},     {"res5b", 256, 19, 60, 111, 2, {2, 3}, 16},
    {"conv6_2", 256, 10, 111, 162, 2, {2, 3}, 32}, {"conv7_2", 128, 5, 162, 213, 2, {2, 3}, 64},
    {"conv8_2", 128, 3, 213, 264, 1, {2, 0}, 100}, {"conv9_2", 128, 1, 264, 315, 1, {2, 0}, 300},
};
constexpr int SSD_IN = 300, SSD_KEEP = 200;
constexpr float SSD_CONF = 0.01f;
constexpr double SSD_NMS = 0.45;

struct SsdTensor { int c = 0, size = 0; size_t off = 0; };    // NHWC [n][size][size][c], offset in floats per image

struct SsdState {
    bool ready = false;
    std::map<std::string, SsdTensor> t;
    size_t floats_per_image = 0;
    int n_priors = 0;
    float* prior_tab = nullptr;
    DevBuf work, in_u8, boxes, prob, rows, count;
    int cap = 0;
};

void ssd_destroy(dfd_handle* h) {
    delete h->ssd;
    h->ssd = nullptr;
}

static const float* wt(dfd_handle* h, const std::string& name, size_t count, bool* ok) {
    auto it = h->tensors.find(name);
    if (it == h->tensors.end() || it->second.count != count) {
        if (*ok) fail(h, DFD_ERR_BLOB, "weights blob: detector tensor '%s' missing or wrong size", name.c_str());
        *ok = false;
        return nullptr;
    }
    return it->second.dev;
}

// k x k conv on the split-precision MFMA path when enabled, else on the fp32 MFMA kernel
static bool conv_gemm(dfd_handle* h, const float* X, const float* W, const float* bias, const float* R, float* Y,
                      int n, const ConvGeom& g, int Cout, int act, bool res_first) {
    if (!W) return true;                                     // missing tensor: reported by wt()
    const int K = g.ksize * g.ksize * g.Cin;
    if (h->split_gemm && g.Cin % 32 == 0 && split_gemm_supports(K, Cout)) {
        const unsigned short* w3 = split_weights(h, W, Cout, K);
        if (w3 && launch_conv_gemm_split<float>(h->gemm, X, w3, bias, R, Y, n, g, Cout, act, res_first, 3, h->stream)) return true;
    }
    return launch_conv_gemm(X, W, bias, R, Y, n, g, Cout, act, res_first, h->stream);
}

// shapes, per-image workspace layout and the prior table
int ssd_init(dfd_handle* h) {
    if (h->tensors.find("ssd.conv1.w") == h->tensors.end()) return DFD_OK;      // blob without a detector
    SsdState* S = new SsdState();
    h->ssd = S;
    SsdTensor data;
    data.c = 3; data.size = SSD_IN;
    S->t["data"] = data;
    size_t off = 0;
    for (const SsdLayer& L : kSsdLayers) {
        const SsdTensor& src = S->t[L.src];
        SsdTensor o;
        o.c = L.cout;
        if (L.kind == SK_POOL) o.size = (src.size - L.k + L.stride - 1) / L.stride + 1;            // ceil mode
        else if (L.kind == SK_NORM) o.size = src.size;
        else o.size = (src.size + 2 * L.pad - L.dil * (L.k - 1) - 1) / L.stride + 1;
        o.off = off;
        off += ((size_t)o.size * o.size * o.c + 63) & ~(size_t)63;
        S->t[L.name] = o;
    }
    int first = 0;
    for (int s = 0; s < 6; ++s) {
        const SsdSource& src = kSsdSources[s];
        const int p = 2 + 2 * src.nar;
        SsdTensor o;
        o.c = p * 6; o.size = src.map; o.off = off;
        off += ((size_t)o.size * o.size * o.c + 63) & ~(size_t)63;
        S->t[std::string(src.tensor) + ".head"] = o;
        if (S->t[src.tensor].size != src.map || S->t[src.tensor].c != src.c)
            return fail(h, DFD_ERR_STATE, "detector plan: source %s has the wrong shape", src.tensor);
        first += src.map * src.map * p;
    }
    S->floats_per_image = off;
    S->n_priors = first;
    // Caffe PriorBox (offset 0.5, clip false), evaluated in double and rounded once
    std::vector<float> tab((size_t)first * 4);
    size_t k = 0;
    for (int s = 0; s < 6; ++s) {
        const SsdSource& src = kSsdSources[s];
        std::vector<std::pair<double, double>> sizes = {{src.mn, src.mn}, {std::sqrt(src.mn * src.mx), std::sqrt(src.mn * src.mx)}};
        for (int a = 0; a < src.nar; ++a) {
            const double r = std::sqrt(src.ar[a]);
            sizes.push_back({src.mn * r, src.mn / r});
            sizes.push_back({src.mn / r, src.mn * r});
        }
        for (int y = 0; y < src.map; ++y)
            for (int x = 0; x < src.map; ++x) {
                const double cx = (x + 0.5) * src.step, cy = (y + 0.5) * src.step;
                for (auto& wh : sizes) {
                    tab[k++] = (float)((cx - wh.first / 2.0) / SSD_IN);
                    tab[k++] = (float)((cy - wh.second / 2.0) / SSD_IN);
                    tab[k++] = (float)((cx + wh.first / 2.0) / SSD_IN);
                    tab[k++] = (float)((cy + wh.second / 2.0) / SSD_IN);
                }
            }
    }
    void* d = nullptr;
    DFD_HIP_TRY(h, hipMalloc(&d, tab.size() * 4));
    h->owned.push_back(d);
    DFD_HIP_TRY(h, hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    S->prior_tab = static_cast<float*>(d);
    // every weight tensor must be present with the planned size
    bool ok = true;
    for (const SsdLayer& L : kSsdLayers) {
        const std::string q = std::string("ssd.") + L.name;
        if (L.kind == SK_CONV1 || L.kind == SK_CONV) {
            wt(h, q + ".w", (size_t)L.cout * L.k * L.k * L.cin, &ok);
            wt(h, q + ".b", L.cout, &ok);
        } else if (L.kind == SK_NORM) wt(h, q + ".scale", L.cout, &ok);
    }
    for (const SsdSource& src : kSsdSources) {
        const int p = 2 + 2 * src.nar;
        wt(h, std::string("ssd.") + src.tensor + ".head.w", (size_t)p * 6 * 9 * src.c, &ok);
        wt(h, std::string("ssd.") + src.tensor + ".head.b", p * 6, &ok);
    }
    if (!ok) return DFD_ERR_BLOB;
    S->ready = true;
    return DFD_OK;
}

// frames_u8: [n][300][300][3] resized BGR on the device.  Leaves DetectionOutput rows in S->rows/count.
int ssd_forward(dfd_handle* h, const uint8_t* in300, int n, const char* tap_name, float* tap_out, size_t tap_cap,
                size_t* tap_count) {
    SsdState* S = h->ssd;
    if (!S || !S->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob (weights.pack_all)");
    int rc;
    if (n > S->cap) {
        if ((rc = ensure(h, &S->work, S->floats_per_image * 4 * n))) return rc;
        if ((rc = ensure(h, &S->boxes, (size_t)n * S->n_priors * 16))) return rc;
        if ((rc = ensure(h, &S->prob, (size_t)n * S->n_priors * 4))) return rc;
        if ((rc = ensure(h, &S->rows, (size_t)n * SSD_KEEP * 5 * 4))) return rc;
        if ((rc = ensure(h, &S->count, (size_t)n * 4))) return rc;
        S->cap = n;
    }
    hipStream_t s = h->stream;
    float* base = static_cast<float*>(S->work.p);
    // tensor t of image i lives at base + off(t)*n + i*size(t)   (frame-major per tensor)
    auto ptr = [&](const std::string& name) { return base + S->t[name].off * n; };
    bool ok = true;
    auto W_ = [&](const std::string& nm, size_t c) { return wt(h, nm, c, &ok); };
    const float mean[3] = {104.0f, 177.0f, 123.0f};
    bool tapped = false;
    auto tap = [&](const std::string& name) -> int {
        if (!tap_name || tapped || name != tap_name) return DFD_OK;
        const SsdTensor& t = S->t[name];
        const size_t cnt = (size_t)n * t.size * t.size * t.c;
        if (cnt > tap_cap) return fail(h, DFD_ERR_ARG, "detector tap '%s' needs %zu floats", tap_name, cnt);
        DFD_HIP_TRY(h, hipMemcpyAsync(tap_out, ptr(name), cnt * 4, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, hipStreamSynchronize(s));
        *tap_count = cnt;
        tapped = true;
        return DFD_OK;
    };
    for (const SsdLayer& L : kSsdLayers) {
        const SsdTensor& src = S->t[L.src];
        const SsdTensor& dst = S->t[L.name];
        const std::string q = std::string("ssd.") + L.name;
        switch (L.kind) {
            case SK_CONV1:
                launch_ssd_conv1(in300, W_(q + ".w", 147 * 32), W_(q + ".b", 32), ptr(L.name), n, mean, s);
                break;
            case SK_POOL:
                launch_maxpool3s2(ptr(L.src), ptr(L.name), n, src.size, dst.size, src.c, s);
                break;
            case SK_NORM:
                launch_l2norm128(ptr(L.src), W_(q + ".scale", 128), ptr(L.name), (long long)n * src.size * src.size, s);
                break;
            case SK_CONV: {
                ConvGeom g;
                g.H = g.W = src.size; g.Ho = g.Wo = dst.size; g.Cin = L.cin; g.ksize = L.k; g.stride = L.stride;
                g.pad = L.pad; g.dil = L.dil;
                if (!conv_gemm(h, ptr(L.src), W_(q + ".w", (size_t)L.cout * L.k * L.k * L.cin), W_(q + ".b", L.cout),
                               L.res ? ptr(L.res) : nullptr, ptr(L.name), n, g, L.cout,
                               L.relu ? ACT_RELU : ACT_NONE, true))
                    return fail(h, DFD_ERR_STATE, "detector layer %s: C_in not a multiple of 32", L.name);
                break;
            }
        }
        if ((rc = tap(L.name))) return rc;
    }
    SsdHeads H{};
    H.prior_tab = S->prior_tab;
    int first = 0;
    for (int i = 0; i < 6; ++i) {
        const SsdSource& src = kSsdSources[i];
        const int p = 2 + 2 * src.nar;
        const std::string hn = std::string(src.tensor) + ".head", q = std::string("ssd.") + src.tensor + ".head";
        ConvGeom g;
        g.H = g.W = g.Ho = g.Wo = src.map; g.Cin = src.c; g.ksize = 3; g.stride = 1; g.pad = 1; g.dil = 1;
        conv_gemm(h, ptr(src.tensor), W_(q + ".w", (size_t)p * 6 * 9 * src.c), W_(q + ".b", p * 6), nullptr, ptr(hn), n,
                  g, p * 6, ACT_NONE, false);
        if ((rc = tap(hn))) return rc;
        H.out[i] = ptr(hn);
        H.first[i] = first;
        H.priors[i] = p;
        H.map[i] = src.map;
        first += src.map * src.map * p;
    }
    H.first[6] = first;
    if (!ok) return DFD_ERR_BLOB;
    const float var[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    launch_ssd_decode(H, (float*)S->boxes.p, (float*)S->prob.p, n, S->n_priors, (float)SSD_IN, var, s);
    launch_ssd_nms((const float*)S->boxes.p, (const float*)S->prob.p, n, S->n_priors, SSD_CONF, SSD_NMS, SSD_KEEP,
                   (float*)S->rows.p, (int*)S->count.p, s);
    if (tap_name && !tapped) {
        const std::string tn = tap_name;
        const float* srcp = tn == "prob" ? (const float*)S->prob.p : tn == "boxes" ? (const float*)S->boxes.p : nullptr;
        if (!srcp) return fail(h, DFD_ERR_ARG, "detector tap: unknown tensor '%s'", tap_name);
        const size_t cnt = (size_t)n * S->n_priors * (tn == "boxes" ? 4 : 1);
        if (cnt > tap_cap) return fail(h, DFD_ERR_ARG, "detector tap '%s' needs %zu floats", tap_name, cnt);
        DFD_HIP_TRY(h, hipMemcpyAsync(tap_out, srcp, cnt * 4, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, hipStreamSynchronize(s));
        *tap_count = cnt;
    }
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}

// dfd_warmup: the detector's layer shapes at batch n, on a byte pattern (sizes the workspace, splits the weights and
// - with the tile table in tuning mode - measures the GEMM tiles)
int ssd_warmup(dfd_handle* h, int n) {
    if (!h->ssd || !h->ssd->ready || n <= 0) return DFD_OK;
    int rc;
    const size_t bytes = (size_t)n * SSD_IN * SSD_IN * 3;
    if ((rc = ensure(h, &h->ssd->in_u8, bytes))) return rc;
    std::vector<uint8_t> pat(bytes);
    uint32_t st = 12345u;
    for (size_t i = 0; i < bytes; ++i) { st = st * 1664525u + 1013904223u; pat[i] = (uint8_t)(50 + ((st >> 24) * 150 >> 8)); }
    DFD_HIP_TRY(h, hipMemcpyAsync(h->ssd->in_u8.p, pat.data(), bytes, hipMemcpyHostToDevice, h->stream));
    DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, n, nullptr, nullptr, 0, nullptr);
}

// reference face_detection.py:84-105 on one image's DetectionOutput rows
int ssd_postprocess(const float* rows, int nrows, int hh, int ww, float conf_thr, int32_t* xywh, float* conf,
                    int max_out, int* total = nullptr) {
    int k = 0, all = 0;
    for (int i = 0; i < nrows; ++i) {
        const float* r = rows + (size_t)i * 5;
        if (!(r[0] > conf_thr)) continue;                                   // strict '>'
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

// detector on a frame already in HBM (shared by dfd_detect_faces and dfd_analyze_frame)
int detect_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, float conf_thr, int32_t* xywh_out,
               float* conf_out, int max_out, int* n_out) {
    *n_out = 0;
    h->last_detections = 0;
    if (hh < 30 || ww < 30) return DFD_OK;                                    // face_detection.py:55-56
    if (!h->ssd || !h->ssd->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob (weights.pack_all)");
    int rc;
    if ((rc = ensure(h, &h->ssd->in_u8, (size_t)SSD_IN * SSD_IN * 3))) return rc;
    launch_resize_bgr(frame_dev, 1, hh, ww, stride, 0, (uint8_t*)h->ssd->in_u8.p, SSD_IN, SSD_IN, h->stream);
    if ((rc = ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, 1, nullptr, nullptr, 0, nullptr))) return rc;
    float rows[SSD_KEEP * 5];
    int cnt = 0;
    DFD_HIP_TRY(h, hipMemcpyAsync(&cnt, h->ssd->count.p, 4, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, hipMemcpyAsync(rows, h->ssd->rows.p, sizeof rows, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
    *n_out = ssd_postprocess(rows, cnt, hh, ww, conf_thr, xywh_out, conf_out, max_out, &h->last_detections);
    return DFD_OK;
}

// the same for n frames resident in HBM (frame f at frames_dev + f * frame_bytes): one launch set
int detect_batch_run(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes,
                     float conf_thr, int max_faces, int32_t* xywh_out, int* n_out) {
    for (int f = 0; f < n; ++f) n_out[f] = 0;
    if (hh < 30 || ww < 30) return DFD_OK;
    if (!h->ssd || !h->ssd->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob (weights.pack_all)");
    int rc;
    if ((rc = ensure(h, &h->ssd->in_u8, (size_t)n * SSD_IN * SSD_IN * 3))) return rc;
    launch_resize_bgr(frames_dev, n, hh, ww, stride, frame_bytes, (uint8_t*)h->ssd->in_u8.p, SSD_IN, SSD_IN, h->stream);
    if ((rc = ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, n, nullptr, nullptr, 0, nullptr))) return rc;
    std::vector<float> rows((size_t)n * SSD_KEEP * 5);
    std::vector<int> cnt(n);
    DFD_HIP_TRY(h, hipMemcpyAsync(cnt.data(), h->ssd->count.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, hipMemcpyAsync(rows.data(), h->ssd->rows.p, rows.size() * 4, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int f = 0; f < n; ++f)
        n_out[f] = ssd_postprocess(rows.data() + (size_t)f * SSD_KEEP * 5, cnt[f], hh, ww, conf_thr,
                                   xywh_out + (size_t)f * max_faces * 4, nullptr, max_faces);
    return DFD_OK;
}

}  // namespace dfd

extern "C" {

int dfd_has_detector(const dfd_handle* h) { return h && h->ssd && h->ssd->ready ? 1 : 0; }

int dfd_last_detection_count(const dfd_handle* h) { return h ? h->last_detections : DFD_ERR_ARG; }

int dfd_detect_faces(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, float conf_thr, int32_t* xywh_out,
                     float* conf_out, int max_out, int* n_out) {
    if (!h) return DFD_ERR_ARG;
    if (!bgr || !xywh_out || !n_out || max_out <= 0 || hh <= 0 || ww <= 0 || stride < ww * 3)
        return fail(h, DFD_ERR_ARG, "detect_faces: bad pointer or geometry");
    *n_out = 0;
    if (hh < 30 || ww < 30) return DFD_OK;                                    // face_detection.py:55-56
    if (!h->ssd || !h->ssd->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob (weights.pack_all)");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    const int rc = ensure(h, &h->frame_buf, (size_t)hh * stride);
    if (rc) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return detect_run(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, conf_thr, xywh_out, conf_out, max_out, n_out);
}

int dfd_ssd_tap(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, const char* name, float* out,
                size_t capacity, size_t* count) {
    if (!h) return DFD_ERR_ARG;
    if (!bgr || !name || !out || !count) return fail(h, DFD_ERR_ARG, "ssd_tap: null pointer");
    if (!h->ssd || !h->ssd->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, &h->frame_buf, (size_t)hh * stride))) return rc;
    if ((rc = ensure(h, &h->ssd->in_u8, (size_t)SSD_IN * SSD_IN * 3))) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    launch_resize_bgr((const uint8_t*)h->frame_buf.p, 1, hh, ww, stride, 0, (uint8_t*)h->ssd->in_u8.p, SSD_IN, SSD_IN, h->stream);
    *count = 0;
    if (std::string(name) == "rows") {
        if ((rc = ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, 1, nullptr, nullptr, 0, nullptr))) return rc;
        int cnt = 0;
        DFD_HIP_TRY(h, hipMemcpyAsync(&cnt, h->ssd->count.p, 4, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
        if ((size_t)cnt * 5 > capacity) return fail(h, DFD_ERR_ARG, "ssd_tap: capacity");
        DFD_HIP_TRY(h, hipMemcpyAsync(out, h->ssd->rows.p, (size_t)cnt * 20, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
        *count = (size_t)cnt * 5;
        return DFD_OK;
    }
    return ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, 1, name, out, capacity, count);
}

}  // extern "C"
