// MTCNN align/crop stage: the cascade the reference runs on every cropped face before the
// classifier (reference deepfake_detection.py:24-28, 376-380 -> facenet-pytorch MTCNN.forward: detect_face,
// select by probability, extract_face to 160x160).  Networks and the two resamplers run as HIP kernels
// (mtcnn_kernels.hip); the box bookkeeping between the stages (threshold scan, NMS, regression, squaring,
// clipping: a few hundred rows of float32 arithmetic in the package's operation order) runs here on the host
// of the library, as the package does it in numpy/torch glue.  Operation order and float32/double choices
// follow the restatement in oracle/mtcnn_ref.py line by line.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "dfd_common.h"
#include "mtcnn_kernels.h"

namespace dfd {

struct MtConv { const float *w, *b, *a; int co, ci, k; };
struct MtDense { const float *w, *b, *a; int out, in; };

struct MtcnnState {
    bool ready = false;
    MtConv p1, p2, p3, p41, p42;
    MtConv r1, r2, r3;
    MtDense r4, r51, r52;
    MtConv o1, o2, o3, o4;
    MtDense o5, o61, o62, o63;
    DevBuf in, a0, a1, prob, reg, win, coef, bnd, tmp, face;
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

struct Box { float x1, y1, x2, y2, score, r[4]; };

// torchvision.ops.nms: descending score (stable), suppress IoU > thr, areas without the +1
std::vector<int> nms_iou(const std::vector<Box>& b, float thr) {
    std::vector<int> order(b.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int i, int j) { return b[i].score > b[j].score; });
    std::vector<char> dead(b.size(), 0);
    std::vector<int> keep;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int i = order[oi];
        if (dead[i]) continue;
        keep.push_back(i);
        const float ai = (b[i].x2 - b[i].x1) * (b[i].y2 - b[i].y1);
        for (size_t oj = oi + 1; oj < order.size(); ++oj) {
            const int j = order[oj];
            if (dead[j]) continue;
            const float xx1 = std::max(b[i].x1, b[j].x1), yy1 = std::max(b[i].y1, b[j].y1);
            const float xx2 = std::min(b[i].x2, b[j].x2), yy2 = std::min(b[i].y2, b[j].y2);
            const float inter = std::max(0.f, xx2 - xx1) * std::max(0.f, yy2 - yy1);
            const float aj = (b[j].x2 - b[j].x1) * (b[j].y2 - b[j].y1);
            const float iou = inter / (ai + aj - inter);
            if (iou > thr) dead[j] = 1;
        }
    }
    return keep;
}

// nms_numpy(method='Min'): ascending stable argsort, take from the end; +1 areas; keep o <= thr
std::vector<int> nms_min(const std::vector<Box>& b, float thr) {
    std::vector<int> idx(b.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int i, int j) { return b[i].score < b[j].score; });
    std::vector<int> pick;
    while (!idx.empty()) {
        const int i = idx.back();
        pick.push_back(i);
        idx.pop_back();
        const float ai = (b[i].x2 - b[i].x1 + 1.f) * (b[i].y2 - b[i].y1 + 1.f);
        std::vector<int> rest;
        for (int j : idx) {
            const float xx1 = std::max(b[i].x1, b[j].x1), yy1 = std::max(b[i].y1, b[j].y1);
            const float xx2 = std::min(b[i].x2, b[j].x2), yy2 = std::min(b[i].y2, b[j].y2);
            const float w = std::max(0.f, xx2 - xx1 + 1.f), hgt = std::max(0.f, yy2 - yy1 + 1.f);
            const float aj = (b[j].x2 - b[j].x1 + 1.f) * (b[j].y2 - b[j].y1 + 1.f);
            const float o = (w * hgt) / std::min(ai, aj);
            if (o <= thr) rest.push_back(j);
        }
        idx.swap(rest);
    }
    return pick;
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

struct Cascade {
    dfd_handle* h;
    MtcnnState* S;
    const uint8_t* img;      // device, BGR
    int hh, ww;
    size_t stride;
    const char* tap_name;
    std::vector<float>* tap;
    int* tap_dims;

    bool want(const std::string& n) const { return tap_name && n == tap_name; }

    int upload_windows(const std::vector<MtWindow>& w) {
        int rc = ensure(h, &S->win, w.size() * sizeof(MtWindow));
        if (rc) return rc;
        DFD_HIP_TRY(h, hipMemcpyAsync(S->win.p, w.data(), w.size() * sizeof(MtWindow), hipMemcpyHostToDevice, h->stream));
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
        return DFD_OK;
    }
    int download(const void* dev, size_t floats, std::vector<float>* out) {
        out->resize(floats);
        if (!floats) return DFD_OK;
        DFD_HIP_TRY(h, hipMemcpyAsync(out->data(), dev, floats * 4, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
        return DFD_OK;
    }

    // P-Net over one pyramid level: face-probability map [oh][ow] and regression map [oh][ow][4]
    int pnet(int sh, int sw, std::vector<float>* prob, std::vector<float>* reg, int* oh, int* ow) {
        hipStream_t s = h->stream;
        int rc;
        const std::vector<MtWindow> whole{MtWindow{0, 0, ww, hh}};
        if ((rc = upload_windows(whole))) return rc;
        const int c1h = sh - 2, c1w = sw - 2, ph = mt_pool_out(c1h, 2, 2), pw = mt_pool_out(c1w, 2, 2);
        const int c2h = ph - 2, c2w = pw - 2, c3h = c2h - 2, c3w = c2w - 2;
        *oh = c3h; *ow = c3w;
        if (c3h <= 0 || c3w <= 0) { prob->clear(); reg->clear(); return DFD_OK; }
        if ((rc = ensure(h, &S->in, (size_t)sh * sw * 3 * 4))) return rc;
        if ((rc = ensure(h, &S->a0, (size_t)c1h * c1w * 32 * 4))) return rc;
        if ((rc = ensure(h, &S->a1, (size_t)c1h * c1w * 32 * 4))) return rc;
        if ((rc = ensure(h, &S->prob, (size_t)c3h * c3w * 4))) return rc;
        if ((rc = ensure(h, &S->reg, (size_t)c3h * c3w * 4 * 4))) return rc;
        float *in = (float*)S->in.p, *a0 = (float*)S->a0.p, *a1 = (float*)S->a1.p;
        launch_mt_area_resize(img, stride, (const MtWindow*)S->win.p, 1, sh, sw, in, s);
        launch_mt_conv(in, S->p1.w, S->p1.b, S->p1.a, a0, 1, sh, sw, 3, 10, 3, s);
        launch_mt_maxpool(a0, a1, 1, c1h, c1w, 10, 2, 2, s);
        launch_mt_conv(a1, S->p2.w, S->p2.b, S->p2.a, a0, 1, ph, pw, 10, 16, 3, s);
        launch_mt_conv(a0, S->p3.w, S->p3.b, S->p3.a, a1, 1, c2h, c2w, 16, 32, 3, s);
        launch_mt_conv(a1, S->p41.w, S->p41.b, nullptr, a0, 1, c3h, c3w, 32, 2, 1, s);
        launch_mt_softmax_face(a0, (float*)S->prob.p, (long long)c3h * c3w, s);
        launch_mt_conv(a1, S->p42.w, S->p42.b, nullptr, (float*)S->reg.p, 1, c3h, c3w, 32, 4, 1, s);
        DFD_HIP_TRY(h, hipGetLastError());
        if ((rc = download(S->prob.p, (size_t)c3h * c3w, prob))) return rc;
        return download(S->reg.p, (size_t)c3h * c3w * 4, reg);
    }

    // R-Net (size 24) / O-Net (size 48) over `wins`: face probability [n], regression [n][4]
    int refine(bool onet, const std::vector<MtWindow>& wins, std::vector<float>* prob, std::vector<float>* reg) {
        hipStream_t s = h->stream;
        const int n = (int)wins.size(), sz = onet ? 48 : 24;
        int rc;
        if ((rc = upload_windows(wins))) return rc;
        const size_t big = (size_t)n * (sz - 2) * (sz - 2) * (onet ? 32 : 28) * 4;
        if ((rc = ensure(h, &S->in, (size_t)n * sz * sz * 3 * 4))) return rc;
        if ((rc = ensure(h, &S->a0, big))) return rc;
        if ((rc = ensure(h, &S->a1, big))) return rc;
        if ((rc = ensure(h, &S->prob, (size_t)n * 4))) return rc;
        if ((rc = ensure(h, &S->reg, (size_t)n * 4 * 4))) return rc;
        float *in = (float*)S->in.p, *a0 = (float*)S->a0.p, *a1 = (float*)S->a1.p;
        launch_mt_area_resize(img, stride, (const MtWindow*)S->win.p, n, sz, sz, in, s);
        if (!onet) {
            launch_mt_conv(in, S->r1.w, S->r1.b, S->r1.a, a0, n, 24, 24, 3, 28, 3, s);       // 22
            launch_mt_maxpool(a0, a1, n, 22, 22, 28, 3, 2, s);                                // 11
            launch_mt_conv(a1, S->r2.w, S->r2.b, S->r2.a, a0, n, 11, 11, 28, 48, 3, s);     // 9
            launch_mt_maxpool(a0, a1, n, 9, 9, 48, 3, 2, s);                                  // 4
            launch_mt_conv(a1, S->r3.w, S->r3.b, S->r3.a, a0, n, 4, 4, 48, 64, 2, s);       // 3 -> [n][3][3][64]
            launch_mt_dense(a0, S->r4.w, S->r4.b, S->r4.a, a1, n, 576, 128, s);
            launch_mt_dense(a1, S->r51.w, S->r51.b, nullptr, a0, n, 128, 2, s);
            launch_mt_softmax_face(a0, (float*)S->prob.p, n, s);
            launch_mt_dense(a1, S->r52.w, S->r52.b, nullptr, (float*)S->reg.p, n, 128, 4, s);
        } else {
            launch_mt_conv(in, S->o1.w, S->o1.b, S->o1.a, a0, n, 48, 48, 3, 32, 3, s);       // 46
            launch_mt_maxpool(a0, a1, n, 46, 46, 32, 3, 2, s);                                // 23
            launch_mt_conv(a1, S->o2.w, S->o2.b, S->o2.a, a0, n, 23, 23, 32, 64, 3, s);     // 21
            launch_mt_maxpool(a0, a1, n, 21, 21, 64, 3, 2, s);                                // 10
            launch_mt_conv(a1, S->o3.w, S->o3.b, S->o3.a, a0, n, 10, 10, 64, 64, 3, s);     // 8
            launch_mt_maxpool(a0, a1, n, 8, 8, 64, 2, 2, s);                                  // 4
            launch_mt_conv(a1, S->o4.w, S->o4.b, S->o4.a, a0, n, 4, 4, 64, 128, 2, s);      // 3 -> [n][3][3][128]
            launch_mt_dense(a0, S->o5.w, S->o5.b, S->o5.a, a1, n, 1152, 256, s);
            launch_mt_dense(a1, S->o61.w, S->o61.b, nullptr, a0, n, 256, 2, s);
            launch_mt_softmax_face(a0, (float*)S->prob.p, n, s);
            launch_mt_dense(a1, S->o62.w, S->o62.b, nullptr, (float*)S->reg.p, n, 256, 4, s);
            // dense6_3 (landmarks) does not influence the selected crop: not evaluated
        }
        DFD_HIP_TRY(h, hipGetLastError());
        if ((rc = download(S->prob.p, n, prob))) return rc;
        return download(S->reg.p, (size_t)n * 4, reg);
    }

    void tap_boxes(const std::string& name, const std::vector<Box>& b) {
        if (!want(name)) return;
        tap->clear();
        for (const Box& x : b) { tap->push_back(x.x1); tap->push_back(x.y1); tap->push_back(x.x2); tap->push_back(x.y2); tap->push_back(x.score); }
        tap_dims[0] = (int)b.size(); tap_dims[1] = 5; tap_dims[2] = 1;
    }

    // detect_face for one image -> boxes after the three stages
    int run(std::vector<Box>* out) {
        int rc;
        // scale pyramid (double arithmetic, as the package's Python floats)
        std::vector<double> scales;
        {
            const double m = 12.0 / 20.0;
            double minl = std::min(hh, ww) * m, scale_i = m;
            while (minl >= 12) { scales.push_back(scale_i); scale_i *= 0.709; minl *= 0.709; }
        }
        std::vector<Box> all;
        for (size_t si = 0; si < scales.size(); ++si) {
            const double scale = scales[si];
            const int sh = (int)(hh * scale + 1), sw = (int)(ww * scale + 1);
            std::vector<float> prob, reg;
            int oh = 0, ow = 0;
            if ((rc = pnet(sh, sw, &prob, &reg, &oh, &ow))) return rc;
            if (want("pnet.prob." + std::to_string(si))) { *tap = prob; tap_dims[0] = oh; tap_dims[1] = ow; tap_dims[2] = 1; }
            if (want("pnet.reg." + std::to_string(si))) { *tap = reg; tap_dims[0] = oh; tap_dims[1] = ow; tap_dims[2] = 4; }
            // generateBoundingBox: cells with prob >= 0.6 in (y, x) order; float32 arithmetic
            std::vector<Box> bs;
            const float fs = (float)scale;
            for (int y = 0; y < oh; ++y)
                for (int x = 0; x < ow; ++x) {
                    const float p = prob[(size_t)y * ow + x];
                    if (!(p >= 0.6f)) continue;
                    Box b{};
                    b.x1 = std::floor((2.f * (float)x + 1.f) / fs);
                    b.y1 = std::floor((2.f * (float)y + 1.f) / fs);
                    b.x2 = std::floor((2.f * (float)x + 12.f) / fs);
                    b.y2 = std::floor((2.f * (float)y + 12.f) / fs);
                    b.score = p;
                    for (int r = 0; r < 4; ++r) b.r[r] = reg[((size_t)y * ow + x) * 4 + r];
                    bs.push_back(b);
                }
            for (int i : nms_iou(bs, 0.5f)) all.push_back(bs[i]);
        }
        std::vector<Box> boxes;
        for (int i : nms_iou(all, 0.7f)) {
            Box b = all[i];
            const float regw = b.x2 - b.x1, regh = b.y2 - b.y1;
            const float x1 = b.x1 + b.r[0] * regw, y1 = b.y1 + b.r[1] * regh, x2 = b.x2 + b.r[2] * regw, y2 = b.y2 + b.r[3] * regh;
            b.x1 = x1; b.y1 = y1; b.x2 = x2; b.y2 = y2;
            rerec(b);
            boxes.push_back(b);
        }
        tap_boxes("stage1", boxes);
        // second and third stage
        for (int stage = 2; stage <= 3 && !boxes.empty(); ++stage) {
            std::vector<MtWindow> wins;
            std::vector<Box> live;
            for (const Box& b : boxes) {
                MtWindow w;
                if (window_of(b, ww, hh, &w)) { wins.push_back(w); live.push_back(b); }
            }
            boxes.clear();
            if (!wins.empty()) {
                std::vector<float> prob, reg;
                if ((rc = refine(stage == 3, wins, &prob, &reg))) return rc;
                if (want(stage == 2 ? "rnet.prob" : "onet.prob")) { *tap = prob; tap_dims[0] = (int)prob.size(); tap_dims[1] = 1; tap_dims[2] = 1; }
                if (want(stage == 2 ? "rnet.reg" : "onet.reg")) { *tap = reg; tap_dims[0] = (int)prob.size(); tap_dims[1] = 4; tap_dims[2] = 1; }
                const float thr = 0.7f;
                std::vector<Box> pass;
                for (size_t i = 0; i < live.size(); ++i) {
                    if (!(prob[i] > thr)) continue;
                    Box b = live[i];
                    b.score = prob[i];
                    for (int r = 0; r < 4; ++r) b.r[r] = reg[i * 4 + r];
                    pass.push_back(b);
                }
                if (stage == 2) {
                    for (int i : nms_iou(pass, 0.7f)) { Box b = pass[i]; bbreg(b); rerec(b); boxes.push_back(b); }
                } else {
                    for (Box& b : pass) bbreg(b);
                    for (int i : nms_min(pass, 0.7f)) boxes.push_back(pass[i]);
                }
            }
            tap_boxes(stage == 2 ? "stage2" : "stage3", boxes);
        }
        if (tap_name && tap_dims[0] < 0) {                     // a stage that was never reached is an empty list
            const std::vector<Box> none;
            tap_boxes("stage2", none);
            tap_boxes("stage3", none);
        }
        *out = boxes;
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
    if (!ok) return DFD_ERR_BLOB;
    S->ready = true;
    return DFD_OK;
}

// MTCNN.forward on a BGR image already in HBM: selected box + the 160x160 BGR u8 crop in S->face.
// *found = 0: no face passed the cascade, or the selected box is degenerate (the package raises there and the
// reference call site returns None).
int mtcnn_align_device(dfd_handle* h, const uint8_t* img_dev, int hh, int ww, size_t stride, float* box_out, int* found,
                       const char* tap_name, std::vector<float>* tap, int* tap_dims) {
    MtcnnState* S = h->mtcnn;
    if (!S || !S->ready) return fail(h, DFD_ERR_STATE, "the weights blob holds no MTCNN cascade");
    if (hh <= 0 || ww <= 0) return fail(h, DFD_ERR_ARG, "mtcnn: empty image");
    *found = 0;
    Cascade c{h, S, img_dev, hh, ww, stride, tap_name, tap, tap_dims};
    std::vector<Box> boxes;
    int rc = c.run(&boxes);
    if (rc) return rc;
    if (boxes.empty()) return DFD_OK;
    // select_boxes(method="probability"): np.argsort(probs)[::-1][0] = the LAST of the ascending stable order
    int best = 0;
    for (int i = 1; i < (int)boxes.size(); ++i)
        if (boxes[i].score >= boxes[best].score) best = i;
    const Box& b = boxes[best];
    if (box_out) { box_out[0] = b.x1; box_out[1] = b.y1; box_out[2] = b.x2; box_out[3] = b.y2; box_out[4] = b.score; }
    // extract_face(margin 0): int() of the clipped float corners
    const int x1 = (int)std::max(b.x1, 0.f), y1 = (int)std::max(b.y1, 0.f);
    const int x2 = (int)std::min(b.x2, (float)ww), y2 = (int)std::min(b.y2, (float)hh);
    if (x2 <= x1 || y2 <= y1) return DFD_OK;
    const int cw = x2 - x1, ch = y2 - y1;
    if ((rc = ensure(h, &S->face, 160 * 160 * 3))) return rc;
    hipStream_t s = h->stream;
    // crop.resize((160, 160), BILINEAR): horizontal pass into tmp [ch][160][3], vertical pass into face
    const uint8_t* src = img_dev;
    size_t sstride = stride;
    int sx = x1, sy = y1;
    std::vector<int> coeff, bounds;
    int ksize = 0;
    if (cw != 160) {
        pil_coeffs(cw, 160, &coeff, &bounds, &ksize);
        if ((rc = ensure(h, &S->coef, coeff.size() * 4))) return rc;
        if ((rc = ensure(h, &S->bnd, bounds.size() * 4))) return rc;
        if ((rc = ensure(h, &S->tmp, (size_t)ch * 160 * 3))) return rc;
        DFD_HIP_TRY(h, hipMemcpyAsync(S->coef.p, coeff.data(), coeff.size() * 4, hipMemcpyHostToDevice, s));
        DFD_HIP_TRY(h, hipMemcpyAsync(S->bnd.p, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice, s));
        const bool last = ch == 160;
        launch_mt_pil_pass(src, sstride, sx, sy, cw, ch, (const int*)S->coef.p, (const int*)S->bnd.p, ksize, 160, 0,
                           (uint8_t*)(last ? S->face.p : S->tmp.p), s);
        DFD_HIP_TRY(h, hipStreamSynchronize(s));        // coeff/bounds are stack temporaries
        src = (const uint8_t*)S->tmp.p; sstride = 160 * 3; sx = 0; sy = 0;
    }
    if (ch != 160) {
        pil_coeffs(ch, 160, &coeff, &bounds, &ksize);
        if ((rc = ensure(h, &S->coef, coeff.size() * 4))) return rc;
        if ((rc = ensure(h, &S->bnd, bounds.size() * 4))) return rc;
        DFD_HIP_TRY(h, hipMemcpyAsync(S->coef.p, coeff.data(), coeff.size() * 4, hipMemcpyHostToDevice, s));
        DFD_HIP_TRY(h, hipMemcpyAsync(S->bnd.p, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice, s));
        launch_mt_pil_pass(src, sstride, sx, sy, cw == 160 ? cw : 160, ch, (const int*)S->coef.p, (const int*)S->bnd.p, ksize,
                           160, 1, (uint8_t*)S->face.p, s);
        DFD_HIP_TRY(h, hipStreamSynchronize(s));
    } else if (cw == 160) {                              // already 160 x 160: plain copy of the window
        DFD_HIP_TRY(h, hipMemcpy2DAsync(S->face.p, 160 * 3, img_dev + (size_t)y1 * stride + (size_t)x1 * 3, stride, 160 * 3, 160,
                                        hipMemcpyDeviceToDevice, s));
    }
    DFD_HIP_TRY(h, hipGetLastError());
    *found = 1;
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
        DFD_HIP_TRY(h, hipStreamSynchronize(h->stream));
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
