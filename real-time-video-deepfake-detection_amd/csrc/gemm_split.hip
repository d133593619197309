// Host side of the split-precision GEMM (kernels: gemm_split_impl.h): weight splitting, the candidate list, the
// per-handle tile table with its measurement pass (dfd_warmup only), the 2^31-byte chunking and the launchers.
#include "gemm_split_impl.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

namespace dfd {

DFD_S6_INSTANTIATE(float, 3)

// W [N][K] fp32 -> three planes [Np][Kp] bf16, zero outside N x K (Kp = K rounded up to 64, Np = s6_np(N)):
// the GEMM's weight loads need neither clamps nor zero-fill selects.
// (transposed: W is stored [K][N] - the stem convolution's [ky][kx][ci][co] tensor read as a [27][32] matrix)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, __bf16* __restrict__ out,
                                                            int N, int K, int Np, int Kp, int transposed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, plane = (size_t)Np * Kp;
    if (i >= plane) return;
    const int n = (int)(i / Kp), k = (int)(i - (size_t)n * Kp);
    const float a = n < N && k < K ? (transposed ? W[(size_t)k * N + n] : W[(size_t)n * K + k]) : 0.f;
    const __bf16 h0 = (__bf16)a;
    const float r1 = a - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    out[i] = h0;
    out[plane + i] = h1;
    out[2 * plane + i] = (__bf16)r2;
}

size_t split_weights_count(int N, int K) {
    return (size_t)s6_np(N) * ((K + S6_KPAD - 1) / S6_KPAD * S6_KPAD);
}

void launch_split_weights(const float* W, unsigned short* out, int N, int K, hipStream_t s, bool transposed) {
    const int Np = s6_np(N), Kp = (K + S6_KPAD - 1) / S6_KPAD * S6_KPAD;
    const size_t plane = (size_t)Np * Kp;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, W,
                       reinterpret_cast<__bf16*>(out), N, K, Np, Kp, transposed ? 1 : 0);
}

// Heuristic tile (shapes nobody warmed up - e.g. the cascade's R-/O-Net layers, whose row count is the number of
// windows of the step): the biggest per-wave pw6 tile that still fills the chip; when even that tile leaves the chip
// under-filled (fewer blocks than CUs) the launch lasts as long as ONE block, so the rule turns around: rounds of
// resident blocks x (tile area + a fixed per-block prologue / epilogue share) / useful share of the padded width -
// R-Net's dense4 at 1,193 windows ran as 10 blocks of 128 x 128 on 256 CUs (36 us), now as 152 blocks of 64 x 16 (10 us).
static S6Tile pick_tile6(int M, int N) {
    const int tiles = (N + 15) / 16;
    S6Tile best = make_tile6(M, N, 1, 1);
    double best_score = -1.0;
    for (int mt = 1; mt <= 2; ++mt)
        for (int nt = 1; nt <= 8; ++nt) {
            const S6Tile t = make_tile6(M, N, mt, nt);
            const double blocks = (double)t.mblocks * t.nblocks;
            const double useful = (double)tiles / ((double)t.nblocks * nt);
            const double fill = blocks >= 512.0 ? 1.0 : blocks / 512.0;
            const double score = mt * nt * useful * fill * (t.nblocks == 1 ? 1.15 : 1.0);
            if (score > best_score) { best_score = score; best = t; }
        }
    if ((long long)best.mblocks * best.nblocks >= 256) return best;
    double best_cost = 1e300;
    for (int mt = 1; mt <= 2; ++mt)
        for (int nt = 1; nt <= 8; ++nt) {
            const S6Tile t = make_tile6(M, N, mt, nt);
            const double blocks = (double)t.mblocks * t.nblocks;
            const double useful = (double)tiles / ((double)t.nblocks * nt);
            const double rounds = std::max(1.0, std::ceil(blocks / 512.0));
            const double cost = rounds * (mt * nt + 1.5) / useful;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = t; }
        }
    return best;
}

// Every instance the library can launch for a shape, in a fixed order (what the tuner measures and what
// dfd_set_option(h, "gemm_tile", i) indexes): pw6 with 4 waves (KS x MT x NT), pw6 with 8 waves (KS x NT), pw7.
// thin: the shape and call may also run pw8 (all of N per wave, weights resident in LDS; s6_thin below) - appended last
// wide: the call may run pw9 (activations resident in registers, all column blocks per block; s6_wide in s6_run_one)
static std::vector<S6Tile> s6_candidates(int M, int K, int N, bool thin = false, bool wide = false) {
    const int tiles = (N + 15) / 16;
    std::vector<S6Tile> raw, out;
    for (int ks = 1; ks <= (K > 32 ? 2 : 1); ++ks)
        for (int mt = 1; mt <= 2; ++mt)
            for (int nt = 1; nt <= 8 && nt <= tiles; ++nt) raw.push_back(make_tile6(M, N, mt, nt, ks));
    for (int ks = 1; ks <= (K > 32 ? 2 : 1); ++ks)              // eight waves per block (MT = 1)
        for (int nt = 2; nt <= 8 && nt <= tiles; ++nt) raw.push_back(make_tile(M, N, 0, 8, 1, 1, nt, ks));
#define DFD_S7_CAND(WMV, WNV, MTV, NTV) raw.push_back(make_tile(M, N, 1, WMV, WNV, MTV, NTV));
    DFD_S7_CONFIGS(DFD_S7_CAND)
#undef DFD_S7_CAND
    for (const S6Tile& t : raw) {
        const int bn_tiles = t.kind == 0 ? t.nt : t.wn * t.nt;
        if ((double)tiles / ((double)t.nblocks * bn_tiles) < 0.7) continue;      // mostly padding
        if ((long long)t.mblocks * t.nblocks > (1 << 20)) continue;
        out.push_back(t);
    }
    if (out.empty()) out.push_back(make_tile6(M, N, 1, 1));
    if (wide)
    {
        for (int nt : {6, 4}) out.push_back(make_tile(M, N, 3, 4, 1, 2, nt, 1));
        out.push_back(make_tile(M, N, 3, 4, 1, 2, 4, 2));              // ks = 2: the pipelined epilogue (two accumulator sets)
    }
    if (thin)
        for (int bpc : {2, 3, 4, 6, 8, 12, 16, 24}) out.push_back(make_tile(M, N, 2, s8_waves(K), 1, 1, tiles, bpc));      // ks = blocks per CU (grid: dispatcher)
    return out;
}

// The best tile depends on how the block count quantises into rounds of resident blocks (8 XCDs x 32 CUs x
// 1-6 blocks, by VGPRs and LDS of the instance), on K (prologue/epilogue share) and on the L2 re-reads of X:
// measured rather than modelled - but only inside dfd_warmup (table->tuning), which is allowed to synchronise.
// Everywhere else a shape that was never warmed up runs the heuristic tile and nothing blocks.  Every tile
// computes each output with the same MFMA sequence, so the choice never changes a result bit
// (tests/test_gemm_tiles_gpu.py walks every candidate through dfd_set_option(h, "gemm_tile", i)).
struct S6Key {
    int M, K, N, mode;
    bool operator<(const S6Key& o) const {
        return std::tie(M, K, N, mode) < std::tie(o.M, o.K, o.N, o.mode);
    }
};
struct S6Table {
    std::map<S6Key, S6Tile> tiles;
    int force = -1;          // >= 0: candidate index (mod the shape's candidate count) for every call
    bool tuning = false;     // measure unseen shapes (synchronises the stream): dfd_warmup only
};

S6Table* s6_table_create() { return new (std::nothrow) S6Table(); }
void s6_table_destroy(S6Table* t) { delete t; }
void s6_table_set_force(S6Table* t, int idx) { if (t) t->force = idx; }
void s6_table_set_tuning(S6Table* t, bool on) { if (t) t->tuning = on; }
int s6_table_measured(const S6Table* t) {
    int n = 0;
    if (t) for (const auto& kv : t->tiles) n += kv.second.measured ? 1 : 0;
    return n;
}
// the measured entries as text, one "M K N mode kind wm wn mt nt ks" line each (dfd_tiles_export / _import: lets a
// profiled run reuse the tiles of an earlier run instead of launching ~100 candidates per shape under the profiler)
std::string s6_table_export(const S6Table* t) {
    std::string out;
    if (!t) return out;
    char line[128];
    for (const auto& kv : t->tiles) {
        if (!kv.second.measured) continue;
        snprintf(line, sizeof line, "%d %d %d %d %d %d %d %d %d %d\n", kv.first.M, kv.first.K, kv.first.N, kv.first.mode,
                 kv.second.kind, kv.second.wm, kv.second.wn, kv.second.mt, kv.second.nt, kv.second.ks);
        out += line;
    }
    return out;
}
int s6_table_import(S6Table* t, const char* text, size_t len) {
    if (!t || !text) return -1;
    int n = 0;
    const std::string all(text, len);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t e = all.find('\n', pos);
        if (e == std::string::npos) e = all.size();
        int v[10];
        if (sscanf(all.substr(pos, e - pos).c_str(), "%d %d %d %d %d %d %d %d %d %d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5],
                   &v[6], &v[7], &v[8], &v[9]) == 10) {
            // accept only tiles the dispatcher can launch for this shape
            bool known = false;
            for (const S6Tile& c : s6_candidates(v[0], v[1], v[2], (v[3] & 128) != 0, (v[3] & 64) != 0))
                known |= c.kind == v[4] && c.wm == v[5] && c.wn == v[6] && c.mt == v[7] && c.nt == v[8] && c.ks == v[9];
            if (known) {
                S6Tile tile = make_tile(v[0], v[2], v[4], v[5], v[6], v[7], v[8], v[9]);
                tile.measured = true;
                t->tiles[S6Key{v[0], v[1], v[2], v[3]}] = tile;
                ++n;
            }
        }
        pos = e + 1;
    }
    return n;
}

int s6_max_candidates() { return (int)s6_candidates(1 << 20, 1152, 1280).size(); }

// Rows of one kernel call: activations (and gates) are addressed with 32-bit byte offsets and sized with a
// 32-bit num_records, so a call never spans 2^31 bytes of X.  Chunks are whole images (HW rows) so that the
// gate index m / HW and the convolution's image index stay relative to the chunk base.
long long s6_chunk_rows(long long M, long long row_bytes, long long HW) {
    const long long lim = (1ll << 31) - 1;
    if (HW <= 0) HW = 1;
    if (M * row_bytes <= lim) return M;
    long long imgs = lim / (row_bytes * HW);
    if (imgs < 1) return -1;                                   // one image alone is too large (never for B0 / SSD)
    return imgs * HW;
}


template <typename XT, int NP>
static void s6_measure(bool conv, bool gated, const std::vector<S6Tile>& cands, S6Tile* tile, const S6Key& key, const XT* X,
                       const unsigned short* W3, const float* bias, const float* gate, const XT* R, XT* Y, int M,
                       int K, int N, int HW, int act, const ConvGeom& g, int res_first, hipStream_t s, const SeFuse& se) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return; }
    float best_ms = 1e30f;
    for (const S6Tile& t : cands) {
        s6_dispatch_any<XT, NP>(conv, gated, t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
        float ms = 1e30f;
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {          // best of two groups of three: robust to a stray hiccup
            hipEventRecord(e0, s);
            for (int r = 0; r < 3; ++r)
                s6_dispatch_any<XT, NP>(conv, gated, t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
            hipEventRecord(e1, s);
            float m1 = 0.f;
            ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&m1, e0, e1) == hipSuccess;
            if (ok && m1 < ms) ms = m1;
        }
        if (!ok) continue;
        if (getenv("DFD_S6_VERBOSE") && atoi(getenv("DFD_S6_VERBOSE")) > 1)
            fprintf(stderr, "[dfd]   kind %d %dx%d mt=%d nt=%d ks=%d: %.1f us\n", t.kind, t.wm, t.wn, t.mt, t.nt, t.ks, ms * 1000.f / 3.f);
        if (ms < best_ms) { best_ms = ms; *tile = t; tile->measured = true; }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (getenv("DFD_S6_VERBOSE"))
        fprintf(stderr, "[dfd] split gemm M=%d K=%d N=%d mode=%d -> kind %d %dx%d mt=%d nt=%d ks=%d (%.1f us)\n", M, K, N,
                key.mode, tile->kind, tile->wm, tile->wn, tile->mt, tile->nt, tile->ks, best_ms * 1000.f / 3.f);
}

template <typename XT, int NP>
static void s6_run_one(S6Table* tab, bool conv, bool gated, const XT* X, const unsigned short* W3, const float* bias,
                       const float* gate, const XT* R, XT* Y, int M, int K, int N, int HW, int act, const ConvGeom& g,
                       int res_first, hipStream_t s, const SeFuse& se) {
    static const bool tune_env = !(getenv("DFD_S6_TUNE") && atoi(getenv("DFD_S6_TUNE")) == 0);
    static const bool thin_env = !(getenv("DFD_S6_THIN") && atoi(getenv("DFD_S6_THIN")) == 0);
    // pw8 takes the call when the whole weight matrix fits its LDS image, a 16-row tile never straddles two images (the
    // gate row is block-uniform) and nothing but bias / activation / residual happens in the epilogue
    const bool thin = thin_env && !conv && s8_supports(K, N) && (!gated || (HW % 16 == 0 && M % HW == 0)) && act != ACT_PRELU &&
                      !(R && res_first) && (!se.P || (se.thin && se_thin_supported(K, se.c_se)));
    // se.thin: nobody launched se_kernel - only pw8 can take the call (the plan asked split_gemm_thin_supports first)
    static const bool wide_env = !(getenv("DFD_S6_WIDE") && atoi(getenv("DFD_S6_WIDE")) == 0);
    const bool wide = wide_env && !conv && !gated && sizeof(XT) == 4 && NP == 3 && s9_supports(K, N) && act != ACT_PRELU && !se.P && !R;
    auto candidates = [&]() {
        std::vector<S6Tile> c = s6_candidates(M, K, N, thin, wide);
        if (se.P && se.thin) c.erase(std::remove_if(c.begin(), c.end(), [](const S6Tile& t) { return t.kind != 2; }), c.end());
        return c;
    };
    S6Tile tile;
    if (tab && tab->force >= 0) {
        const std::vector<S6Tile> cands = candidates();
        tile = cands[(size_t)tab->force % cands.size()];
    } else {
        // M in 8 buckets per octave: data-dependent row counts (the MTCNN candidate windows) share an entry
        int mkey = M;
        if (M > 64) {
            int sh = 0;
            while ((M >> sh) > 15) ++sh;
            mkey = ((M + (1 << sh) - 1) >> sh) << sh;
        }
        const S6Key key{mkey, K, N, (conv ? 1 : 0) | (gated ? 2 : 0) | (sizeof(XT) == 2 ? 4 : 0) | (NP == 1 ? 8 : 0) |
                                        (se.P ? 16 : 0) | (thin ? 128 : 0) | (wide ? 64 : 0) | (conv ? (g.ksize << 8) | (g.stride << 4) : 0)};
        const bool tuning = tab && tab->tuning && tune_env;
        auto it = tab ? tab->tiles.find(key) : std::map<S6Key, S6Tile>::iterator();
        if (tab && it != tab->tiles.end() && (it->second.measured || !tuning)) {
            tile = it->second;
        } else {
            // A shape nobody warmed up (the classifier at the batch of the faces a step keeps, a request's few crops) keeps the
            // pw6 heuristic: pw8 / pw9 are picked by MEASUREMENT only - un-warmed at small batches they lose (batch 71 / 16 / 4 / 1:
            // 1.34 / 0.73 / 0.59 / 0.56 ms per forward against 1.31 / 0.66 / 0.50 / 0.47 with pw6; profiles/unwarmed_batch_probe.py):
            // their blocks amortise a prologue over many tiles, and a short matrix has few.
            tile = se.P && se.thin ? candidates()[2] : pick_tile6(M, N);      // (thin-only list: 2, 3, 4 ... blocks per CU)
            if (tuning) s6_measure<XT, NP>(conv, gated, candidates(), &tile, key, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
            if (tab) tab->tiles[key] = tile;
        }
        tile = make_tile(M, N, tile.kind, tile.wm, tile.wn, tile.mt, tile.nt, tile.ks);      // block counts for this call's M
    }
    s6_dispatch_any<XT, NP>(conv, gated, tile, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
}

template <typename XT>
static void s6_run_np(int planes, S6Table* tab, bool conv, bool gated, const XT* X, const unsigned short* W3, const float* bias,
                      const float* gate, const XT* R, XT* Y, int M, int K, int N, int HW, int act, const ConvGeom& g,
                      int res_first, hipStream_t s, const SeFuse& se = SeFuse()) {
    if constexpr (sizeof(XT) == 2) {
        if (planes == 1) {
            s6_run_one<XT, 1>(tab, conv, gated, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
            return;
        }
    }
    s6_run_one<XT, 3>(tab, conv, gated, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);
}

bool split_gemm_supports(int K, int N) { return K % 8 == 0 && K >= 16 && split_weights_count(N, K) * 6 < (1ull << 31); }
bool split_gemm_thin_supports(int K, int N, int HW) {
    static const bool thin_env = !(getenv("DFD_S6_THIN") && atoi(getenv("DFD_S6_THIN")) == 0);
    return thin_env && split_gemm_supports(K, N) && s8_supports(K, N) && HW > 0 && HW % 16 == 0;
}

template <typename XT>
bool launch_pointwise_split(S6Table* tab, const XT* X, const unsigned short* W3, const float* bias, const float* gate,
                            const XT* R, XT* Y, int M, int K, int N, int HW, int act, int planes, hipStream_t s,
                            const SeFuse* se) {
    const ConvGeom none{};
    if (HW <= 0) HW = 1;
    if (se && se->P && (!gate || !se_fuse_supported(HW, se->c_se))) return false;
    if (se && se->P && se->thin && !(split_gemm_thin_supports(K, N, HW) && se_thin_supported(K, se->c_se) && M % HW == 0)) return false;
    const long long chunk = s6_chunk_rows(M, (long long)K * (long long)sizeof(XT), gate ? HW : 1);
    if (chunk <= 0) return false;
    for (long long m0 = 0; m0 < M; m0 += chunk) {
        const int mc = (int)std::min<long long>(chunk, M - m0);
        SeFuse sec = se ? *se : SeFuse();
        if (sec.P) sec.P += (size_t)(m0 / HW) * K * (size_t)sec.tiles;      // chunks are whole images
        s6_run_np<XT>(planes, tab, false, gate != nullptr, X + (size_t)m0 * K, W3, bias,
                      gate ? gate + (size_t)(m0 / HW) * K : nullptr, R ? R + (size_t)m0 * N : nullptr, Y + (size_t)m0 * N,
                      mc, K, N, HW, act, none, 0, s, sec);
    }
    return true;
}
template bool launch_pointwise_split<float>(S6Table*, const float*, const unsigned short*, const float*, const float*,
                                            const float*, float*, int, int, int, int, int, int, hipStream_t, const SeFuse*);
template bool launch_pointwise_split<bf16_t>(S6Table*, const bf16_t*, const unsigned short*, const float*, const float*,
                                             const bf16_t*, bf16_t*, int, int, int, int, int, int, hipStream_t, const SeFuse*);

template <typename XT>
bool launch_conv_gemm_split(S6Table* tab, const XT* X, const unsigned short* W3, const float* bias, const XT* R,
                            XT* Y, int n_img, const ConvGeom& g, int Cout, int act, bool res_first, int planes, hipStream_t s) {
    if (g.Cin % S6_BK != 0) return false;                  // a K stage must not straddle two taps
    const int K = g.ksize * g.ksize * g.Cin;
    if (!split_gemm_supports(K, Cout)) return false;
    const long long in_img = (long long)g.H * g.W * g.Cin * (long long)sizeof(XT), out_rows = (long long)g.Ho * g.Wo;
    const long long imgs = s6_chunk_rows(n_img, in_img, 1);        // "rows" = images of in_img bytes
    if (imgs <= 0) return false;
    for (long long i0 = 0; i0 < n_img; i0 += imgs) {
        const int ni = (int)std::min<long long>(imgs, n_img - i0);
        s6_run_np<XT>(planes, tab, true, false, X + (size_t)i0 * g.H * g.W * g.Cin, W3, bias, nullptr,
                      R ? R + (size_t)i0 * out_rows * Cout : nullptr, Y + (size_t)i0 * out_rows * Cout,
                      (int)(ni * out_rows), K, Cout, 1, act, g, res_first ? 1 : 0, s);
    }
    return true;
}
template bool launch_conv_gemm_split<float>(S6Table*, const float*, const unsigned short*, const float*, const float*, float*,
                                            int, const ConvGeom&, int, int, bool, int, hipStream_t);
template bool launch_conv_gemm_split<bf16_t>(S6Table*, const bf16_t*, const unsigned short*, const float*, const bf16_t*,
                                             bf16_t*, int, const ConvGeom&, int, int, bool, int, hipStream_t);

#ifdef S6_TRACE
extern "C" int dfd_debug_s6_trace(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s6_trace), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

}  // namespace dfd
