// Host side of the classifier: weights blob -> device tensors -> per-layer launch plan.
// The block table restates efficientnet_pytorch's B0 arguments (the dependency the
// reference builds its backbone from, reference model.py:39-43) and must agree with
// b0_arch.py; dfd_create checks every tensor's shape against it.
#include <algorithm>

#include "b0_kernels.h"
#include "dfd_common.h"

namespace dfd {

thread_local std::string g_create_error;

int fail(dfd_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

namespace {

struct Stage { int rep, k, s, e, ci, co; };
const Stage kStages[] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40},
                         {3, 3, 2, 6, 40, 80}, {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192},
                         {1, 3, 1, 6, 192, 320}};

const float* need(dfd_handle* h, const std::string& name, std::initializer_list<uint32_t> dims, bool* ok) {
    auto it = h->tensors.find(name);
    if (it == h->tensors.end()) {
        if (*ok) fail(h, DFD_ERR_BLOB, "weights blob: tensor '%s' missing", name.c_str());
        *ok = false;
        return nullptr;
    }
    const Tensor& t = it->second;
    size_t want = 1;
    for (uint32_t d : dims) want *= d;
    if (t.count != want) {
        if (*ok) fail(h, DFD_ERR_BLOB, "weights blob: tensor '%s' has %zu elements, expected %zu", name.c_str(), t.count, want);
        *ok = false;
        return nullptr;
    }
    return t.dev;
}

}  // namespace

int b0_build_plan(dfd_handle* h) {
    B0Plan& P = h->b0;
    bool ok = true;
    P.stem_w = need(h, "stem.w", {3, 3, 3, 32}, &ok);
    P.stem_b = need(h, "stem.b", {32}, &ok);
    int hcur = 112, idx = 0;
    P.io_floats = (size_t)112 * 112 * 32;
    for (const Stage& st : kStages) {
        for (int r = 0; r < st.rep; ++r, ++idx) {
            B0Block b{};
            b.kernel = st.k;
            b.stride = r == 0 ? st.s : 1;
            b.expand = st.e;
            b.c_in = r == 0 ? st.ci : st.co;
            b.c_out = st.co;
            b.c_exp = b.c_in * st.e;
            b.c_se = b.c_in / 4 > 0 ? b.c_in / 4 : 1;
            b.h_in = hcur;
            b.h_out = (hcur + b.stride - 1) / b.stride;
            const int total = (b.h_out - 1) * b.stride + b.kernel - b.h_in;
            b.pad_lo = (total > 0 ? total : 0) / 2;
            b.skip = b.stride == 1 && b.c_in == b.c_out;
            const std::string q = "b" + std::to_string(idx);
            const uint32_t ce = b.c_exp, ci = b.c_in, co = b.c_out, cs = b.c_se, k = b.kernel;
            if (b.expand != 1) {
                b.exp_w = need(h, q + ".exp.w", {ce, ci}, &ok);
                b.exp_b = need(h, q + ".exp.b", {ce}, &ok);
            }
            b.dw_w = need(h, q + ".dw.w", {k, k, ce}, &ok);
            b.dw_b = need(h, q + ".dw.b", {ce}, &ok);
            b.se_w1 = need(h, q + ".se.w1", {cs, ce}, &ok);
            b.se_b1 = need(h, q + ".se.b1", {cs}, &ok);
            b.se_w2 = need(h, q + ".se.w2", {cs, ce}, &ok);
            b.se_b2 = need(h, q + ".se.b2", {ce}, &ok);
            b.proj_w = need(h, q + ".proj.w", {co, ce}, &ok);
            b.proj_b = need(h, q + ".proj.b", {co}, &ok);
            b.dw_tiles = depthwise_tiles(b.h_in, b.c_exp, b.kernel, b.stride);
            if (b.dw_tiles < 0) return fail(h, DFD_ERR_STATE, "no depthwise kernel for block %d", idx);
            b.dw_tiles = std::max(b.dw_tiles, mbconv_tiles(b.h_in, b.c_exp, b.kernel, b.stride, b.c_in, true));
            auto mx = [](size_t& a, size_t v) { if (v > a) a = v; };
            mx(P.io_floats, (size_t)b.h_out * b.h_out * b.c_out);
            if (b.expand != 1) mx(P.exp_floats, (size_t)b.h_in * b.h_in * b.c_exp);
            mx(P.dw_floats, (size_t)b.h_out * b.h_out * b.c_exp);
            mx(P.pool_floats, (size_t)b.dw_tiles * b.c_exp);
            mx(P.gate_floats, (size_t)b.c_exp);
            P.blocks.push_back(b);
            hcur = b.h_out;
        }
    }
    P.head_w = need(h, "head.w", {1280, 320}, &ok);
    P.head_b = need(h, "head.b", {1280}, &ok);
    P.fc1_w = need(h, "fc1.w", {512, 1280}, &ok);
    P.fc1_b = need(h, "fc1.b", {512}, &ok);
    P.fc2_w = need(h, "fc2.w", {256, 512}, &ok);
    P.fc2_b = need(h, "fc2.b", {256}, &ok);
    P.fc3_w = need(h, "fc3.w", {1, 256}, &ok);
    P.fc3_b = need(h, "fc3.b", {1}, &ok);
    return ok ? DFD_OK : DFD_ERR_BLOB;
}

const unsigned short* split_weights(dfd_handle* h, const float* W, int N, int K, bool transposed) {
    const size_t count = split_weights_count(N, K);
    auto it = h->wsplit.find(W);
    if (it != h->wsplit.end()) return it->second;
    void* p = nullptr;
    if (hipMalloc(&p, count * 3 * sizeof(unsigned short)) != hipSuccess) {
        fail(h, DFD_ERR_HIP, "hipMalloc of %zu bytes for split weights failed", count * 6);
        return nullptr;
    }
    h->owned.push_back(p);
    launch_split_weights(W, static_cast<unsigned short*>(p), N, K, h->stream, transposed);
    h->wsplit[W] = static_cast<unsigned short*>(p);
    return static_cast<unsigned short*>(p);
}

// 1x1 conv.  XT = float: split path when enabled and the shape allows, else the fp32 MFMA kernel.
// XT = bf16_t ("bf16_activations"): always the split GEMM's bf16-activation instances.
// does this 1x1 conv run on the split GEMM (the only kernels that can evaluate the squeeze-excite gate themselves)?
template <typename XT>
static bool pointwise_on_split(const dfd_handle* h, int K, int N) {
    return (sizeof(XT) == 2 || h->split_gemm) && N >= 16 && split_gemm_supports(K, N);
}

template <typename XT>
int pointwise_t(dfd_handle* h, const XT* X, const float* W, const float* bias, const float* gate, const XT* R,
                XT* Y, int M, int K, int N, int HW, int act, const SeFuse* se = nullptr) {
    constexpr bool BF = sizeof(XT) == 2;
    if ((BF || h->split_gemm) && N >= 16 && split_gemm_supports(K, N)) {       // never on M: batch-invariant results
        const unsigned short* w3 = split_weights(h, W, N, K);
        if (!w3) return DFD_ERR_HIP;
        if (!launch_pointwise_split<XT>(h->gemm, X, w3, bias, gate, R, Y, M, K, N, HW, act, BF ? h->bf16_planes : 3, h->stream, se))
            return fail(h, DFD_ERR_CAPACITY, "1x1 conv M=%d K=%d: one image exceeds the 2^31-byte addressing of the split GEMM", M, K);
    } else {
        if constexpr (BF) return fail(h, DFD_ERR_STATE, "1x1 conv K=%d N=%d has no bf16-activation kernel", K, N);
        else launch_pointwise(X, W, bias, gate, R, Y, M, K, N, HW, act, h->stream);
    }
    return DFD_OK;
}

int pointwise(dfd_handle* h, const float* X, const float* W, const float* bias, const float* gate, const float* R,
              float* Y, int M, int K, int N, int HW, int act) {
    return pointwise_t<float>(h, X, W, bias, gate, R, Y, M, K, N, HW, act);
}

namespace {

struct Marks {
    B0Prof* prof;
    hipStream_t s;
    void mark(const char* name) {
        if (!prof || !prof->enabled) return;
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, s);
        prof->events.push_back(e);
        prof->names.push_back(name);
    }
};

// copy a device buffer out if it is the requested tap (bf16 activations are widened to fp32 on the device first)
template <typename XT>
int tap_out(dfd_handle* h, B0Tap* tap, const std::string& name, const XT* dev, size_t count) {
    if (!tap || !tap->name || tap->found || name != tap->name) return DFD_OK;
    tap->found = true;
    if (count > tap->capacity) return fail(h, DFD_ERR_ARG, "tap '%s' needs %zu floats, capacity %zu", tap->name, count, tap->capacity);
    const float* src = nullptr;
    if constexpr (sizeof(XT) == 2) {
        const int rc = ensure(h, &h->tap_buf, count * 4);
        if (rc) return rc;
        launch_bf16_to_f32(dev, static_cast<float*>(h->tap_buf.p), count, h->stream);
        src = static_cast<const float*>(h->tap_buf.p);
    } else {
        src = dev;
    }
    DFD_HIP_TRY(h, hipMemcpyAsync(tap->out, src, count * 4, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    tap->count = count;
    return DFD_OK;
}

// static names for the profiler (one per launch, in launch order)
const char* layer_name(int blk, const char* what) {
    static std::map<std::string, std::string> pool;
    const std::string key = "b" + std::to_string(blk) + "." + what;
    return pool.emplace(key, key).first->second.c_str();
}

}  // namespace

template <typename XT>
static int b0_forward_t(dfd_handle* h, const float* x, int n, float* logits_dev, B0Tap* tap, B0Prof* prof) {
    if (n <= 0) return fail(h, DFD_ERR_ARG, "batch must be positive");
    if (n > h->max_batch) return fail(h, DFD_ERR_CAPACITY, "batch %d exceeds handle capacity %d", n, h->max_batch);
    const B0Plan& P = h->b0;
    hipStream_t s = h->stream;
    Marks mk{prof, s};
    int rc;
    // the activation workspace is sized for fp32; bf16 storage uses the front half of each buffer
    XT* const io0 = reinterpret_cast<XT*>(h->io0);
    XT* const io1 = reinterpret_cast<XT*>(h->io1);
    XT* const expbuf = reinterpret_cast<XT*>(h->expbuf);
    XT* const dwbuf = reinterpret_cast<XT*>(h->dwbuf);
    XT* const headbuf = reinterpret_cast<XT*>(h->headbuf);
    // squeeze-excite inside the depthwise-family launch (option "fuse_se"): the last block of an image writes the gate
    auto se_of = [&](const B0Block& b) {
        SeTail t;
        if (h->fuse_se && h->se_counter) {
            t.w1 = b.se_w1; t.b1 = b.se_b1; t.w2t = b.se_w2; t.b2 = b.se_b2;
            t.gate = h->gate; t.counter = h->se_counter;
            t.inv_hw = 1.0f / (float)(b.h_out * b.h_out);
            t.c_se = b.c_se;
        }
        return t;
    };
    const bool se_fused = h->fuse_se && h->se_counter;
    mk.mark("start");
    // block 0 has no expand conv: its depthwise input IS the stem output, so the two fuse (option "fuse_stem")
    const bool stem_fused = h->fuse_stem && P.blocks[0].expand == 1;
    int stem_tiles = 0;
    if (stem_fused) {
        const bool want_stem = tap && tap->name && std::string(tap->name) == "stem";
        const unsigned short* ws3 = split_weights(h, P.stem_w, 32, 27, true);          // [ky][kx][ci][co] = [27][32], transposed
        if (!ws3) return DFD_ERR_HIP;
        launch_stem_dw<XT>(x, ws3, (int)split_weights_count(32, 27), 64, P.stem_b, P.blocks[0].dw_w, P.blocks[0].dw_b, dwbuf,
                           h->pool, want_stem ? io0 : (XT*)nullptr, n, &stem_tiles, s, se_of(P.blocks[0]));
        mk.mark("b0.dw");                               // stem + depthwise of block 0 in one launch
    } else {
        launch_stem<XT>(x, P.stem_w, P.stem_b, io0, n, s);
        mk.mark("stem");
    }
    if ((rc = tap_out(h, tap, "stem", io0, (size_t)n * 112 * 112 * 32))) return rc;
    XT* cur = io0;
    XT* nxt = io1;
    int bi = 0;
    for (const B0Block& b : P.blocks) {
        const int m_in = n * b.h_in * b.h_in, m_out = n * b.h_out * b.h_out;
        const std::string q = "b" + std::to_string(bi);
        const XT* dw_in = cur;
        int tiles = 0;
        bool fused = false;
        const unsigned short* we3 = nullptr;
        // whole-image launches (blocks 6-15) where they measure faster: per block, batch 256 (section 5, round 4) - blocks 8
        // and 9 (k5, 480 / 672 channels at 14 x 14) are 2-4 us quicker as expand GEMM + depthwise kernel ("fuse_late_skip")
        const bool late_here = h->fuse_late && !((h->fuse_late_skip >> bi) & 1u);
        if (b.expand != 1 && h->fuse_expand && mbconv_tiles(b.h_in, b.c_exp, b.kernel, b.stride, b.c_in, late_here) > 0 &&
            !(we3 = split_weights(h, b.exp_w, b.c_exp, b.c_in))) return DFD_ERR_HIP;
        if (we3 && launch_mbconv_front<XT>(cur, b.c_in, we3, (int)split_weights_count(b.c_exp, b.c_in), (b.c_in + 63) / 64 * 64,
                                           b.exp_w, b.exp_b, b.dw_w, b.dw_b, dwbuf, h->pool, n, b.h_in, b.c_exp,
                                           b.kernel, b.stride, b.pad_lo, &tiles, s, se_of(b), late_here)) {
            fused = true;
            mk.mark(layer_name(bi, "dw"));            // expand + depthwise in one launch
            if (tap && tap->name && q + ".exp" == tap->name)
                return fail(h, DFD_ERR_STATE, "tap '%s': the expanded tensor is not materialised when expand is fused "
                                              "(dfd_set_option(h, \"fuse_expand\", 0))", tap->name);
        }
        if (!fused && b.expand != 1) {
            if ((rc = pointwise_t<XT>(h, cur, b.exp_w, b.exp_b, nullptr, nullptr, expbuf, m_in, b.c_in, b.c_exp,
                                      b.h_in * b.h_in, ACT_SWISH))) return rc;
            mk.mark(layer_name(bi, "exp"));
            if ((rc = tap_out(h, tap, q + ".exp", expbuf, (size_t)m_in * b.c_exp))) return rc;
            dw_in = expbuf;
        }
        if (bi == 0 && stem_fused) {
            tiles = stem_tiles;
        } else if (!fused) {
            if (!launch_depthwise<XT>(dw_in, b.dw_w, b.dw_b, dwbuf, h->pool, n, b.h_in, b.c_exp, b.kernel,
                                      b.stride, b.pad_lo, &tiles, s, se_of(b)))
                return fail(h, DFD_ERR_STATE, "no depthwise kernel for block %d", bi);
            mk.mark(layer_name(bi, "dw"));
        }
        if ((rc = tap_out(h, tap, q + ".dw", dwbuf, (size_t)m_out * b.c_exp))) return rc;
        // option "se_in_proj": where the depthwise launch left FINAL per-image pool sums (tiles == 1: the whole-image
        // launches of blocks 6-10 / 12-15) the projection GEMM evaluates the gate itself - no se_kernel launch
        SeFuse sef;
        // option "se_thin" (round 4; measured slower, off by default): the narrow projections of blocks 0-4 run on pw8_kernel,
        // whose blocks evaluate the gate of the images they meet themselves - no se_kernel launch in front of them
        const bool se_thin = !se_fused && h->se_thin && pointwise_on_split<XT>(h, b.c_exp, b.c_out) && !h->se_in_proj &&
                             split_gemm_thin_supports(b.c_exp, b.c_out, b.h_out * b.h_out) && se_thin_supported(b.c_exp, b.c_se);
        const bool se_in_proj = se_thin || (!se_fused && h->se_in_proj && tiles == 1 && se_fuse_supported(b.h_out * b.h_out, b.c_se) &&
                                            pointwise_on_split<XT>(h, b.c_exp, b.c_out));
        if (se_thin) {
            sef.P = h->pool; sef.w1 = b.se_w1; sef.b1 = b.se_b1; sef.w2t = b.se_w2; sef.b2 = b.se_b2;
            sef.inv_hw = 1.0f / (float)(b.h_out * b.h_out); sef.c_se = b.c_se; sef.tiles = tiles; sef.thin = true;
        } else if (se_in_proj) {
            sef.P = h->pool; sef.w1 = b.se_w1; sef.b1 = b.se_b1; sef.w2t = b.se_w2; sef.b2 = b.se_b2;
            sef.inv_hw = 1.0f / (float)(b.h_out * b.h_out); sef.c_se = b.c_se;
        } else if (!se_fused) {
            launch_se(h->pool, tiles, 1.0f / (float)(b.h_out * b.h_out), b.se_w1, b.se_b1, b.se_w2, b.se_b2,
                      h->gate, n, b.c_exp, b.c_se, s);
            mk.mark(layer_name(bi, "se"));
        }
        if (!se_in_proj && (rc = tap_out(h, tap, q + ".gate", h->gate, (size_t)n * b.c_exp))) return rc;
        if ((rc = pointwise_t<XT>(h, dwbuf, b.proj_w, b.proj_b, h->gate, b.skip ? cur : (const XT*)nullptr, nxt, m_out,
                                  b.c_exp, b.c_out, b.h_out * b.h_out, ACT_NONE, se_in_proj ? &sef : nullptr))) return rc;
        mk.mark(layer_name(bi, "proj"));
        if (se_in_proj && (rc = tap_out(h, tap, q + ".gate", h->gate, (size_t)n * b.c_exp))) return rc;   // written by the GEMM's blocks
        if ((rc = tap_out(h, tap, q + ".out", nxt, (size_t)m_out * b.c_out))) return rc;
        XT* t = cur; cur = nxt; nxt = t;
        ++bi;
    }
    const B0Block& last = P.blocks.back();
    const int hw = last.h_out * last.h_out;
    if ((rc = pointwise_t<XT>(h, cur, P.head_w, P.head_b, nullptr, nullptr, headbuf, n * hw, last.c_out, 1280, hw,
                              ACT_SWISH))) return rc;
    mk.mark("head");
    if ((rc = tap_out(h, tap, "head", headbuf, (size_t)n * hw * 1280))) return rc;
    launch_avgpool<XT>(headbuf, h->feat, n, hw, 1280, s);
    mk.mark("avgpool");
    if ((rc = tap_out(h, tap, "feat", h->feat, (size_t)n * 1280))) return rc;
    if (!logits_dev) return DFD_OK;   // extract_features stops here
    if ((rc = pointwise(h, h->feat, P.fc1_w, P.fc1_b, nullptr, nullptr, h->fc1, n, 1280, 512, 1, ACT_RELU))) return rc;
    if ((rc = pointwise(h, h->fc1, P.fc2_w, P.fc2_b, nullptr, nullptr, h->fc2, n, 512, 256, 1, ACT_RELU))) return rc;
    launch_pointwise(h->fc2, P.fc3_w, P.fc3_b, nullptr, nullptr, logits_dev, n, 256, 1, 1, ACT_NONE, s);
    mk.mark("mlp");
    if ((rc = tap_out(h, tap, "logit", logits_dev, (size_t)n))) return rc;
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}

// fp32 activation storage, or - dfd_set_option(h, "bf16_activations", 1) - bf16 storage of every activation tensor
// that reaches HBM (depthwise / block / expand / head outputs) with fp32 arithmetic and accumulation throughout:
// BASELINE.json configs[3].  Squeeze-excite pools, gates, the pooled feature vector and the MLP head stay fp32.
int b0_forward(dfd_handle* h, const float* x, int n, float* logits_dev, B0Tap* tap, B0Prof* prof) {
    if (n > 0) h->classifier_crops += (unsigned long long)n;
    return h->act_bf16 ? b0_forward_t<bf16_t>(h, x, n, logits_dev, tap, prof)
                       : b0_forward_t<float>(h, x, n, logits_dev, tap, prof);
}

}  // namespace dfd
