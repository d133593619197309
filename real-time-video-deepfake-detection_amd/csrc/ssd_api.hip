// SSD-style face detector: layer plan, workspace and the dfd_detect_faces entry point.
// The layer table restates ssd_arch.py (kept in step by tests/test_ssd_gpu.py, which checks
// every intermediate tensor against the oracle run on the Python table).
#include <cmath>

#include "b0_kernels.h"
#include "dfd_common.h"
#include "host_boxes.h"
#include "ssd_kernels.h"

using namespace dfd;

namespace dfd {

enum SsdKind { SK_CONV1 = 0, SK_POOL = 1, SK_CONV = 2, SK_NORM = 3, SK_AFFINE = 4, SK_ADD = 5 };

// One layer of the detector's trunk.  The table is either the built-in one below (this build's statement of the
// res10 family: BatchNorm/Scale folded, post-activation blocks) or comes from the weights blob ("ssd.plan" +
// "ssd.names", written by weights.pack_ssd_tensors for an architecture that caffe_io built from a deploy.prototxt:
// SURVEY section 8(f) N3).  src / res name the producing layers ("data" = the 300x300 input).
struct SsdLayer {
    std::string name;
    SsdKind kind;
    std::string src;
    int cin, cout, k, stride, pad, dil;
    bool relu;
    std::string res;           // conv: tensor added before the activation; add: second operand; "" = none
};

static const SsdLayer kSsdLayers[] = {
    {"conv1", SK_CONV1, "data", 3, 32, 7, 2, 3, 1, true, ""},
    {"pool1", SK_POOL, "conv1", 32, 32, 3, 2, 0, 1, false, ""},
    {"res2a", SK_CONV, "pool1", 32, 32, 3, 1, 1, 1, true, ""},
    {"res2b", SK_CONV, "res2a", 32, 32, 3, 1, 1, 1, true, "pool1"},
    {"res3p", SK_CONV, "res2b", 32, 128, 1, 2, 0, 1, false, ""},
    {"res3a", SK_CONV, "res2b", 32, 128, 3, 2, 1, 1, true, ""},
    {"res3b", SK_CONV, "res3a", 128, 128, 3, 1, 1, 1, true, "res3p"},
    {"res4p", SK_CONV, "res3b", 128, 256, 1, 2, 0, 1, false, ""},
    {"res4a", SK_CONV, "res3b", 128, 256, 3, 2, 1, 1, true, ""},
    {"res4b", SK_CONV, "res4a", 256, 256, 3, 1, 1, 1, true, "res4p"},
    {"res5a", SK_CONV, "res4b", 256, 256, 3, 1, 2, 2, true, ""},
    {"res5b", SK_CONV, "res5a", 256, 256, 3, 1, 2, 2, true, "res4b"},
    {"conv6_1", SK_CONV, "res5b", 256, 128, 1, 1, 0, 1, true, ""},
    {"conv6_2", SK_CONV, "conv6_1", 128, 256, 3, 2, 1, 1, true, ""},
    {"conv7_1", SK_CONV, "conv6_2", 256, 64, 1, 1, 0, 1, true, ""},
    {"conv7_2", SK_CONV, "conv7_1", 64, 128, 3, 2, 1, 1, true, ""},
    {"conv8_1", SK_CONV, "conv7_2", 128, 64, 1, 1, 0, 1, true, ""},
    {"conv8_2", SK_CONV, "conv8_1", 64, 128, 3, 1, 0, 1, true, ""},
    {"conv9_1", SK_CONV, "conv8_2", 128, 64, 1, 1, 0, 1, true, ""},
    {"conv9_2", SK_CONV, "conv9_1", 64, 128, 3, 1, 0, 1, true, ""},
    {"norm3", SK_NORM, "res3b", 128, 128, 1, 1, 0, 1, false, ""},
};

struct SsdSource { std::string tensor; int c, map; double mn, mx; int nar; double ar[2]; double step; };
static const SsdSource kSsdSources[6] = {
    {"norm3", 128, 38, 30, 60, 1, {2, 0}, 8},     {"res5b", 256, 19, 60, 111, 2, {2, 3}, 16},
    {"conv6_2", 256, 10, 111, 162, 2, {2, 3}, 32}, {"conv7_2", 128, 5, 162, 213, 2, {2, 3}, 64},
    {"conv8_2", 128, 3, 213, 264, 1, {2, 0}, 100}, {"conv9_2", 128, 1, 264, 315, 1, {2, 0}, 300},
};
constexpr int SSD_IN = 300, SSD_KEEP = 200, SSD_TOPK = 400;

struct SsdTensor { int c = 0, size = 0; size_t off = 0; };    // NHWC [n][size][size][c], offset in floats per image

struct SsdState {
    bool ready = false;
    std::vector<SsdLayer> layers;
    std::vector<SsdSource> sources;          // exactly six
    float in_scale[3] = {1.f, 1.f, 1.f}, in_shift[3] = {-104.f, -177.f, -123.f};      // conv1 input: x * scale + shift
    float var[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    float conf_thr = 0.01f;
    double nms_thr = 0.45;
    int keep_top_k = SSD_KEEP;
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

// The layer table: "ssd.plan" [L][10] (kind, src index, res index, cin, cout, k, stride, pad, dil, relu; index -1 =
// "data" / none), "ssd.names" [L][32] (character codes), "ssd.srcs" [6][9] (layer index, channels, map, min, max,
// number of aspect ratios, ar0, ar1, step), "ssd.det" [15] (input size, conv1 input scale b,g,r, shift b,g,r,
// variances x4, nms threshold, top_k, keep_top_k, confidence threshold) - all float32, written by
// weights.pack_ssd_tensors.  A blob without them runs the built-in table.
static int load_plan(dfd_handle* h, SsdState* S) {
    auto plan = h->tensors.find("ssd.plan");
    if (plan == h->tensors.end()) {
        S->layers.assign(std::begin(kSsdLayers), std::end(kSsdLayers));
        S->sources.assign(std::begin(kSsdSources), std::end(kSsdSources));
        return DFD_OK;
    }
    auto names = h->tensors.find("ssd.names"), srcs = h->tensors.find("ssd.srcs"), det = h->tensors.find("ssd.det");
    if (names == h->tensors.end() || srcs == h->tensors.end() || det == h->tensors.end())
        return fail(h, DFD_ERR_BLOB, "detector plan: ssd.names / ssd.srcs / ssd.det missing");
    const size_t L = plan->second.count / 10;
    if (plan->second.count != L * 10 || names->second.count != L * 32 || srcs->second.count != 6 * 9 || det->second.count != 15 || L == 0 || L > 256)
        return fail(h, DFD_ERR_BLOB, "detector plan: malformed tables");
    std::vector<float> pv(L * 10), nv(L * 32), sv(54), dv(15);
    DFD_HIP_TRY(h, hipMemcpy(pv.data(), plan->second.dev, pv.size() * 4, hipMemcpyDeviceToHost));
    DFD_HIP_TRY(h, hipMemcpy(nv.data(), names->second.dev, nv.size() * 4, hipMemcpyDeviceToHost));
    DFD_HIP_TRY(h, hipMemcpy(sv.data(), srcs->second.dev, sv.size() * 4, hipMemcpyDeviceToHost));
    DFD_HIP_TRY(h, hipMemcpy(dv.data(), det->second.dev, dv.size() * 4, hipMemcpyDeviceToHost));
    auto name_of = [&](int i) -> std::string {
        if (i < 0) return "data";
        std::string s;
        for (int k = 0; k < 32 && nv[(size_t)i * 32 + k] != 0.f; ++k) s.push_back((char)(int)nv[(size_t)i * 32 + k]);
        return s;
    };
    for (size_t i = 0; i < L; ++i) {
        const float* r = &pv[i * 10];
        const int kind = (int)r[0], src = (int)r[1], res = (int)r[2];
        if (kind < 0 || kind > 5 || src >= (int)i || res >= (int)i) return fail(h, DFD_ERR_BLOB, "detector plan: bad row %zu", i);
        SsdLayer Ly{name_of((int)i), (SsdKind)kind, name_of(src), (int)r[3], (int)r[4], (int)r[5], (int)r[6], (int)r[7], (int)r[8],
                    r[9] != 0.f, res >= 0 ? name_of(res) : std::string()};
        if (Ly.name.empty() || Ly.name == "data") return fail(h, DFD_ERR_BLOB, "detector plan: bad layer name in row %zu", i);
        S->layers.push_back(Ly);
    }
    for (int s = 0; s < 6; ++s) {
        const float* r = &sv[(size_t)s * 9];
        const int li = (int)r[0];
        if (li < 0 || li >= (int)L || (int)r[5] < 1 || (int)r[5] > 2) return fail(h, DFD_ERR_BLOB, "detector plan: bad source %d", s);
        S->sources.push_back(SsdSource{name_of(li), (int)r[1], (int)r[2], r[3], r[4], (int)r[5], {r[6], r[7]}, r[8]});
    }
    if ((int)dv[0] != SSD_IN) return fail(h, DFD_ERR_BLOB, "detector plan: input size %d (this build resizes to %d)", (int)dv[0], SSD_IN);
    for (int c = 0; c < 3; ++c) { S->in_scale[c] = dv[1 + c]; S->in_shift[c] = dv[4 + c]; }
    for (int c = 0; c < 4; ++c) S->var[c] = dv[7 + c];
    S->nms_thr = (double)dv[11];
    S->keep_top_k = (int)dv[13];
    S->conf_thr = dv[14];
    if ((int)dv[12] != SSD_TOPK || S->keep_top_k < 1 || S->keep_top_k > SSD_KEEP)
        return fail(h, DFD_ERR_BLOB, "detector plan: DetectionOutput top_k must be %d and keep_top_k <= %d", SSD_TOPK, SSD_KEEP);
    return DFD_OK;
}

// shapes, per-image workspace layout and the prior table
int ssd_init(dfd_handle* h) {
    if (h->tensors.find("ssd.conv1.w") == h->tensors.end() && h->tensors.find("ssd.plan") == h->tensors.end())
        return DFD_OK;                                                          // blob without a detector
    SsdState* S = new SsdState();
    h->ssd = S;
    int prc = load_plan(h, S);
    if (prc) return prc;
    SsdTensor data;
    data.c = 3; data.size = SSD_IN;
    S->t["data"] = data;
    size_t off = 0;
    for (const SsdLayer& L : S->layers) {
        if (S->t.find(L.src) == S->t.end() || (!L.res.empty() && S->t.find(L.res) == S->t.end()))
            return fail(h, DFD_ERR_BLOB, "detector plan: layer %s reads a tensor that no earlier layer produces", L.name.c_str());
        const SsdTensor& src = S->t[L.src];
        if (src.c != L.cin) return fail(h, DFD_ERR_BLOB, "detector plan: layer %s expects %d input channels, %s has %d", L.name.c_str(), L.cin, L.src.c_str(), src.c);
        SsdTensor o;
        o.c = L.cout;
        if (L.kind == SK_POOL) o.size = (src.size - L.k + L.stride - 1) / L.stride + 1;            // ceil mode
        else if (L.kind == SK_NORM || L.kind == SK_AFFINE || L.kind == SK_ADD) o.size = src.size;
        else o.size = (src.size + 2 * L.pad - L.dil * (L.k - 1) - 1) / L.stride + 1;
        // what the kernels are built for
        if (L.kind == SK_CONV1 && !(L.k == 7 && L.stride == 2 && L.pad == 3 && L.cin == 3 && L.cout == 32 && L.src == "data"))
            return fail(h, DFD_ERR_BLOB, "detector plan: the first convolution must be 7x7 stride 2 pad 3, 3 -> 32 channels");
        if (L.kind == SK_POOL && !(L.k == 3 && L.stride == 2 && L.pad == 0 && L.cin % 4 == 0))
            return fail(h, DFD_ERR_BLOB, "detector plan: pooling %s must be 3x3 stride 2 (ceil mode)", L.name.c_str());
        if (L.kind == SK_NORM && L.cin != 128) return fail(h, DFD_ERR_BLOB, "detector plan: Normalize is built for 128 channels");
        if (L.kind == SK_CONV && L.cin % 32 != 0) return fail(h, DFD_ERR_BLOB, "detector plan: convolution %s: C_in %d is not a multiple of 32", L.name.c_str(), L.cin);
        if ((L.kind == SK_AFFINE || L.kind == SK_ADD) && L.cin % 4 != 0) return fail(h, DFD_ERR_BLOB, "detector plan: %s: channels must be a multiple of 4", L.name.c_str());
        o.off = off;
        off += ((size_t)o.size * o.size * o.c + 63) & ~(size_t)63;
        S->t[L.name] = o;
    }
    int first = 0;
    for (int s = 0; s < 6; ++s) {
        const SsdSource& src = S->sources[s];
        if (S->t.find(src.tensor) == S->t.end()) return fail(h, DFD_ERR_BLOB, "detector plan: source %s is not a layer", src.tensor.c_str());
        const int p = 2 + 2 * src.nar;
        SsdTensor o;
        o.c = p * 6; o.size = src.map; o.off = off;
        off += ((size_t)o.size * o.size * o.c + 63) & ~(size_t)63;
        S->t[std::string(src.tensor) + ".head"] = o;
        if (S->t[src.tensor].size != src.map || S->t[src.tensor].c != src.c)
            return fail(h, DFD_ERR_STATE, "detector plan: source %s has the wrong shape", src.tensor.c_str());
        first += src.map * src.map * p;
    }
    S->floats_per_image = off;
    S->n_priors = first;
    // Caffe PriorBox (offset 0.5, clip false), evaluated in double and rounded once
    std::vector<float> tab((size_t)first * 4);
    size_t k = 0;
    for (int s = 0; s < 6; ++s) {
        const SsdSource& src = S->sources[s];
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
    for (const SsdLayer& L : S->layers) {
        const std::string q = std::string("ssd.") + L.name;
        if (L.kind == SK_CONV1 || L.kind == SK_CONV) {
            wt(h, q + ".w", (size_t)L.cout * L.k * L.k * L.cin, &ok);
            wt(h, q + ".b", L.cout, &ok);
        } else if (L.kind == SK_NORM) wt(h, q + ".scale", L.cout, &ok);
        else if (L.kind == SK_AFFINE) { wt(h, q + ".scale", L.cout, &ok); wt(h, q + ".shift", L.cout, &ok); }
    }
    for (const SsdSource& src : S->sources) {
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
    bool tapped = false;
    auto tap = [&](const std::string& name) -> int {
        if (!tap_name || tapped || name != tap_name) return DFD_OK;
        const SsdTensor& t = S->t[name];
        const size_t cnt = (size_t)n * t.size * t.size * t.c;
        if (cnt > tap_cap) return fail(h, DFD_ERR_ARG, "detector tap '%s' needs %zu floats", tap_name, cnt);
        DFD_HIP_TRY(h, hipMemcpyAsync(tap_out, ptr(name), cnt * 4, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, stream_sync(h));
        *tap_count = cnt;
        tapped = true;
        return DFD_OK;
    };
    bool skip_pool = false;                  // pool1 was computed inside conv1's launch
    for (size_t li = 0; li < S->layers.size(); ++li) {
        const SsdLayer& L = S->layers[li];
        const SsdTensor& src = S->t[L.src];
        const SsdTensor& dst = S->t[L.name];
        const std::string q = std::string("ssd.") + L.name;
        if (skip_pool && L.kind == SK_POOL) {
            skip_pool = false;
            if ((rc = tap(L.name))) return rc;
            continue;
        }
        switch (L.kind) {
            case SK_CONV1: {
                // the matrix-pipe kernel (DFD_SSD_CONV1_MFMA=0: the thread-per-pixel kernel, for A/B runs)
                static const bool mfma = !(getenv("DFD_SSD_CONV1_MFMA") && atoi(getenv("DFD_SSD_CONV1_MFMA")) == 0);
                static const bool pool_fused = !(getenv("DFD_SSD_CONV1_POOL") && atoi(getenv("DFD_SSD_CONV1_POOL")) == 0);
                const float* wc = W_(q + ".w", 147 * 32);
                const unsigned short* w3 = mfma ? split_weights(h, wc, 32, 147, true) : nullptr;
                if (mfma && !w3) return DFD_ERR_HIP;
                // conv1 -> pool1 with nothing else reading conv1 (and no tap on it): one launch, the 150 x 150 map stays on chip
                bool fuse = mfma && pool_fused && li + 1 < S->layers.size() && S->layers[li + 1].kind == SK_POOL &&
                            S->layers[li + 1].src == L.name && dst.size == 150 && dst.c == 32 &&
                            S->t[S->layers[li + 1].name].size == 75 && !(tap_name && L.name == tap_name);
                for (size_t o = li + 2; fuse && o < S->layers.size(); ++o)
                    if (S->layers[o].src == L.name || S->layers[o].res == L.name) fuse = false;
                for (int i6 = 0; fuse && i6 < 6; ++i6)
                    if (L.name == S->sources[i6].tensor) fuse = false;
                if (fuse) {
                    launch_ssd_conv1_pool(in300, w3, (int)split_weights_count(32, 147), 192, W_(q + ".b", 32), ptr(S->layers[li + 1].name), n,
                                          S->in_scale, S->in_shift, L.relu, s);
                    skip_pool = true;
                    break;
                }
                if (mfma)
                    launch_ssd_conv1_mfma(in300, w3, (int)split_weights_count(32, 147), 192, W_(q + ".b", 32), ptr(L.name), n,
                                          S->in_scale, S->in_shift, L.relu, s);
                else
                    launch_ssd_conv1(in300, wc, W_(q + ".b", 32), ptr(L.name), n, S->in_scale, S->in_shift, L.relu, s);
                break;
            }
            case SK_AFFINE:
                launch_channel_affine(ptr(L.src), W_(q + ".scale", L.cout), W_(q + ".shift", L.cout), nullptr, ptr(L.name),
                                      (long long)n * src.size * src.size, L.cout, L.relu, s);
                break;
            case SK_ADD:
                launch_channel_affine(ptr(L.src), nullptr, nullptr, ptr(L.res), ptr(L.name),
                                      (long long)n * src.size * src.size, L.cout, L.relu, s);
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
                               !L.res.empty() ? ptr(L.res) : nullptr, ptr(L.name), n, g, L.cout,
                               L.relu ? ACT_RELU : ACT_NONE, true))
                    return fail(h, DFD_ERR_STATE, "detector layer %s: C_in not a multiple of 32", L.name.c_str());
                break;
            }
        }
        if ((rc = tap(L.name))) return rc;
    }
    SsdHeads H{};
    H.prior_tab = S->prior_tab;
    int first = 0;
    for (int i = 0; i < 6; ++i) {
        const SsdSource& src = S->sources[i];
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
    launch_ssd_decode(H, (float*)S->boxes.p, (float*)S->prob.p, n, S->n_priors, (float)SSD_IN, S->var, s);
    launch_ssd_nms((const float*)S->boxes.p, (const float*)S->prob.p, n, S->n_priors, S->conf_thr, S->nms_thr, S->keep_top_k,
                   (float*)S->rows.p, (int*)S->count.p, s);
    if (tap_name && !tapped) {
        const std::string tn = tap_name;
        const float* srcp = tn == "prob" ? (const float*)S->prob.p : tn == "boxes" ? (const float*)S->boxes.p : nullptr;
        if (!srcp) return fail(h, DFD_ERR_ARG, "detector tap: unknown tensor '%s'", tap_name);
        const size_t cnt = (size_t)n * S->n_priors * (tn == "boxes" ? 4 : 1);
        if (cnt > tap_cap) return fail(h, DFD_ERR_ARG, "detector tap '%s' needs %zu floats", tap_name, cnt);
        DFD_HIP_TRY(h, hipMemcpyAsync(tap_out, srcp, cnt * 4, hipMemcpyDeviceToHost, s));
        DFD_HIP_TRY(h, stream_sync(h));
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
    DFD_HIP_TRY(h, stream_sync(h));
    return ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, n, nullptr, nullptr, 0, nullptr);
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
    DFD_HIP_TRY(h, stream_sync(h));
    if (cnt < 0) return fail(h, DFD_ERR_HIP, "detector: DetectionOutput gave up waiting for its overlap rows (ssd_nms_kernel); no boxes returned");
    *n_out = ssd_postprocess(rows, cnt, hh, ww, conf_thr, xywh_out, conf_out, max_out, &h->last_detections);
    return DFD_OK;
}

// the same for n frames resident in HBM (frame f at frames_dev + f * frame_bytes): one launch set
int detect_batch_run(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes,
                     float conf_thr, int max_faces, int32_t* xywh_out, int* n_out, int* n_total_out) {
    for (int f = 0; f < n; ++f) n_out[f] = 0;
    if (n_total_out) for (int f = 0; f < n; ++f) n_total_out[f] = 0;
    if (hh < 30 || ww < 30) return DFD_OK;
    if (!h->ssd || !h->ssd->ready) return fail(h, DFD_ERR_STATE, "detector weights were not packed into the blob (weights.pack_all)");
    int rc;
    if ((rc = ensure(h, &h->ssd->in_u8, (size_t)n * SSD_IN * SSD_IN * 3))) return rc;
    launch_resize_bgr(frames_dev, n, hh, ww, stride, frame_bytes, (uint8_t*)h->ssd->in_u8.p, SSD_IN, SSD_IN, h->stream);
    if ((rc = ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, n, nullptr, nullptr, 0, nullptr))) return rc;
    // DetectionOutput rows and counts through the mailbox (dfd_common.h): not behind the next batch's frame upload
    const int* cnt = (const int*)mailbox_d2h(h, h->ssd->count.p, (size_t)n * 4);
    const float* rows = (const float*)mailbox_d2h(h, h->ssd->rows.p, (size_t)n * SSD_KEEP * 5 * 4);
    if (!cnt || !rows) return fail(h, DFD_ERR_HIP, "detect_batch: mailbox allocation failed");
    DFD_HIP_TRY(h, hipGetLastError());
    DFD_HIP_TRY(h, stream_sync(h));
    for (int f = 0; f < n; ++f)
        if (cnt[f] < 0)
            return fail(h, DFD_ERR_HIP, "detector: DetectionOutput of frame %d gave up waiting for its overlap rows (ssd_nms_kernel); no boxes returned", f);
    for (int f = 0; f < n; ++f)
        n_out[f] = ssd_postprocess(rows + (size_t)f * SSD_KEEP * 5, cnt[f], hh, ww, conf_thr,
                                   xywh_out + (size_t)f * max_faces * 4, nullptr, max_faces, n_total_out ? n_total_out + f : nullptr);
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
        DFD_HIP_TRY(h, stream_sync(h));
        if (cnt < 0) return fail(h, DFD_ERR_HIP, "detector: DetectionOutput gave up waiting for its overlap rows (ssd_nms_kernel)");
        if ((size_t)cnt * 5 > capacity) return fail(h, DFD_ERR_ARG, "ssd_tap: capacity");
        DFD_HIP_TRY(h, hipMemcpyAsync(out, h->ssd->rows.p, (size_t)cnt * 20, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, stream_sync(h));
        *count = (size_t)cnt * 5;
        return DFD_OK;
    }
    return ssd_forward(h, (const uint8_t*)h->ssd->in_u8.p, 1, name, out, capacity, count);
}

}  // extern "C"
