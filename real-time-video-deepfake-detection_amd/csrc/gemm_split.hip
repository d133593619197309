// Split-precision fp32 GEMM for gfx950: the 1x1 convolutions at 14x14 / 7x7 and the detector's k x k
// convolutions are bound by the fp32 MFMA rate (v_mfma_f32_16x16x4_f32: 256 cycles per 16x16x32 block of
// products).  An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits):
//     a = a0 + a1 + a2,  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (all subtractions exact)
// so a*b = sum_{i,j} ai*bj, every ai*bj is exact in fp32, and the three terms with i + j >= 3 are below
// 2^-24 |a*b|.  The kernel forms the other six on v_mfma_f32_16x16x32_bf16 (16 cycles each, fp32 accumulate):
// 96 cycles per block instead of 256, with the rounding error of an fp32 dot product (measured against
// float64: 2e-7 of max|y| at K = 1152, the plain fp32 MFMA chain gives 6e-7).  This is not a reduced-precision
// mode: no operand bit is dropped.
//
// Weights are split once per handle (split_weights_kernel -> three bf16 planes); activations are split in
// registers right after the load (and after the squeeze-excite gate multiply).  Tile structure, XCD-aware
// block order, epilogue and the implicit-GEMM convolution mode are those of pw_kernel (b0_kernels.hip).
//
// Operand layout of v_mfma_f32_16x16x32_bf16: lane l holds 8 consecutive k (k = 8*(l>>4) .. +7) of row
// (A) / column (B) l & 15; D as for every 16x16 MFMA: column l & 15, rows 4*(l>>4) + r.  A = weights
// (row = output channel), B = activations (column = pixel): a lane ends with 4 consecutive channels of one
// pixel = one 16-byte NHWC store.
#include "b0_kernels.h"
#include "kernel_util.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

namespace dfd {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 struct does not always leave the stack)

constexpr int S6_BK = 32;                 // K per LDS stage = K of one MFMA
// rows of a zero-padded weight plane: the last n-block of any tile (NT <= 8) stays inside it
__host__ __device__ constexpr int s6_np(int N) { return ((N + 15) / 16 + 7) * 16; }
constexpr int S6_ROWB = 3 * 64;           // bytes per weight row per stage: 3 planes x 32 bf16 = twelve 16-byte chunks
// LDS image of a row: chunk c (= plane * 4 + k-octet) sits at chunk position (c + 6 * ((row >> 2) & 1)) % 12.
// ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) with bank = dword % 64
// (MI355X_MICROARCH.md, LDS table); with 192-byte rows this rotation gives every lane of a group its own four
// banks for all three plane reads (checked exhaustively; the unrotated image is 2-way conflicted: 42 % extra LDS
// cycles measured).  No padding, so two buffers of the widest block are 48 KB: three blocks per CU.
__host__ __device__ constexpr int s6_chunk_pos(int row, int c) { return (c + 6 * ((row >> 2) & 1)) % 12; }

// W [N][K] fp32 -> three planes [Np][Kp] bf16, zero outside N x K (Kp = K rounded up to a stage, Np = s6_np(N)):
// the GEMM's weight loads need neither clamps nor zero-fill selects.
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, __bf16* __restrict__ out,
                                                            int N, int K, int Np, int Kp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, plane = (size_t)Np * Kp;
    if (i >= plane) return;
    const int n = (int)(i / Kp), k = (int)(i - (size_t)n * Kp);
    const float a = n < N && k < K ? W[(size_t)n * K + k] : 0.f;
    const __bf16 h0 = (__bf16)a;
    const float r1 = a - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    out[i] = h0;
    out[plane + i] = h1;
    out[2 * plane + i] = (__bf16)r2;
}

size_t split_weights_count(int N, int K) {
    return (size_t)s6_np(N) * ((K + S6_BK - 1) / S6_BK * S6_BK);
}

void launch_split_weights(const float* W, unsigned short* out, int N, int K, hipStream_t s) {
    const int Np = s6_np(N), Kp = (K + S6_BK - 1) / S6_BK * S6_BK;
    const size_t plane = (size_t)Np * Kp;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, W,
                       reinterpret_cast<__bf16*>(out), N, K, Np, Kp);
}

// 8 fp32 values -> three bf16x8 terms (exact: see the header)
__device__ __forceinline__ void split8(const v4f lo, const v4f hi, bf8& s0, bf8& s1, bf8& s2) {
    const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h0 = (__bf16)f[i];
        const float r1 = f[i] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        s0[i] = h0;
        s1[i] = h1;
        s2[i] = (__bf16)r2;
    }
}

template <int NT, bool CONV, int MT, bool GATE>
__global__ __launch_bounds__(256, 2) void pw6_kernel(const float* __restrict__ X,
                                                     const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gate,
                                                     const float* __restrict__ R,
                                                     float* __restrict__ Y, int M, int K, int N,
                                                     int HW, int act, int mblocks, int nblocks,
                                                     ConvGeom cg, int res_first) {
    constexpr int BK = S6_BK;
    constexpr int BN = NT * 16, BM = 4 * MT * 16;
    constexpr int CHUNKS = BN * 12;                       // 16-byte chunks per stage: row x plane x k-octet
    constexpr int WLOADS = (CHUNKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][BN * S6_ROWB];

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int n0 = nblk * BN;

    int m[MT];
    size_t gbase[MT];
    int iy0[MT], ix0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        m[mt] = mblk * BM + wave * (MT * 16) + mt * 16 + j;
        if constexpr (CONV) {
            const int mm = m[mt] < M ? m[mt] : 0;
            const int img = mm / (cg.Ho * cg.Wo), r = mm - img * (cg.Ho * cg.Wo);
            const int oy = r / cg.Wo, ox = r - oy * cg.Wo;
            gbase[mt] = (size_t)img * cg.H * cg.W * cg.Cin;
            iy0[mt] = oy * cg.stride - cg.pad;
            ix0[mt] = ox * cg.stride - cg.pad;
        } else {
            gbase[mt] = GATE ? (size_t)(m[mt] < M ? m[mt] / HW : 0) * K : 0;
            iy0[mt] = ix0[mt] = 0;
        }
    }

    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    // weight chunk t of this thread: (row, plane, k-octet) -> fixed global / LDS offsets.  The planes are
    // zero-padded to [Np][Kp], so loads and LDS stores are unconditional and select-free (k >= K meets zero
    // weights, whatever the clamped X load returned); threads past the last chunk repeat the last chunk (same
    // value to the same address).  A store under a branch makes hipcc sink the global load into that branch
    // with a vmcnt(0) behind it: one exposed memory latency per K-step.
    int woff[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * 256 < CHUNKS ? tid + t * 256 : CHUNKS - 1;
        const int row = e / 12, rem = e - row * 12, pl = rem >> 2, c = rem & 3;
        woff[t] = pl * plane + (n0 + row) * Kp + 8 * c;
        wlds[t] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + c) * 16;
    }
    u4 wreg[WLOADS];
    auto load_w = [&](int kc) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) wreg[t] = *reinterpret_cast<const u4*>(W3 + woff[t] + kc);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) *reinterpret_cast<u4*>(&ws[buf][wlds[t]]) = wreg[t];
    };

    int mclamp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mclamp[mt] = m[mt] < M ? m[mt] : M - 1;

    v4f xcur[MT][2], xnext[MT][2];
    v4f gcur[MT][2], gnext[MT][2];      // GATE: raw squeeze-excite gate fragments, multiplied in at use
    bool okcur[MT], oknext[MT];         // CONV: tap inside the image (zero padding applied at use)
    auto load_x = [&](int kc, v4f (&xf)[MT][2], v4f (&gf)[MT][2], bool (&okf)[MT]) {
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // wave-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int iy = iy0[mt] + ky * cg.dil, ix = ix0[mt] + kx * cg.dil;
                const bool ok = m[mt] < M && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const float* p = X + gbase[mt] + ((size_t)(ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin + ci0 + 8 * q;
                xf[mt][0] = ldg4(p);
                xf[mt][1] = ldg4(p + 4);
                okf[mt] = ok;
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int k = kc + 8 * q;
                k = k < K ? k : K - 8;
                const float* p = X + (size_t)mclamp[mt] * K + k;
                xf[mt][0] = ldg4(p);
                xf[mt][1] = ldg4(p + 4);
                if constexpr (GATE) {
                    gf[mt][0] = ldg4(gate + gbase[mt] + k);
                    gf[mt][1] = ldg4(gate + gbase[mt] + k + 4);
                }
                okf[mt] = true;
            }
        }
    };

    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    const int nk = (K + BK - 1) / BK;
    load_w(0);
    load_x(0, xcur, gcur, okcur);
    store_w(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        // branch-free body; the last iteration re-loads its own step (see pw_kernel)
        const int kn = (kt + 1 < nk ? kt + 1 : kt) * BK;
        load_w(kn);
        load_x(kn, xnext, gnext, oknext);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch at the top of the step: a whole step to land

        bf8 xs[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v4f lo = xcur[mt][0], hi = xcur[mt][1];
            if constexpr (GATE) { lo *= gcur[mt][0]; hi *= gcur[mt][1]; }
            if constexpr (CONV) {
                if (!okcur[mt]) { lo = (v4f){0.f, 0.f, 0.f, 0.f}; hi = lo; }
            }
            split8(lo, hi, xs[mt][0], xs[mt][1], xs[mt][2]);
        }
        const unsigned char* wb = ws[kt & 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned char* wp = wb + (nt * 16 + j) * S6_ROWB;       // (nt * 16 + j) >> 2 has the parity of j >> 2
            const bf8 w0 = *reinterpret_cast<const bf8*>(wp + rd0);
            const bf8 w1 = *reinterpret_cast<const bf8*>(wp + rd1);
            const bf8 w2 = *reinterpret_cast<const bf8*>(wp + rd2);
            // smallest terms first; the MT accumulators alternate
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xs[mt][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xs[mt][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xs[mt][2], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xs[mt][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xs[mt][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xs[mt][0], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // ... and its consumers at the bottom
        store_w((kt + 1) & 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xcur[mt][0] = xnext[mt][0];
            xcur[mt][1] = xnext[mt][1];
            if constexpr (GATE) {
                gcur[mt][0] = gnext[mt][0];
                gcur[mt][1] = gnext[mt][1];
            }
            okcur[mt] = oknext[mt];
        }
        __syncthreads();
    }

    // epilogue: lane holds Y[m][n .. n+3]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 16 + 4 * q;
        if (n >= N) continue;
        const bool vec = n + 3 < N;
        v4f bv = (v4f){0.f, 0.f, 0.f, 0.f};
        if (vec) bv = ldg4(bias + n);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < N) bv[r] = bias[n + r];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (m[mt] >= M) continue;
            v4f v = acc[mt][nt] + bv;
            v4f rv = (v4f){0.f, 0.f, 0.f, 0.f};
            if (R) {
                if (vec) rv = ldg4(R + (size_t)m[mt] * N + n);
                else
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) rv[r] = R[(size_t)m[mt] * N + n + r];
            }
            if (res_first) v += rv;
            if (act == ACT_SWISH) v = swish4(v);
            else if (act == ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (!res_first) v += rv;
            float* yp = Y + (size_t)m[mt] * N + n;
            if (vec) stg4(yp, v);
            else
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) yp[r] = v[r];
        }
    }
}

struct S6Tile { int mt, nt, mblocks, nblocks; };
static S6Tile make_tile(int M, int N, int mt, int nt) {
    return S6Tile{mt, nt, (M + 64 * mt - 1) / (64 * mt), ((N + 15) / 16 + nt - 1) / nt};
}

// Heuristic tile (used when measuring is switched off): the biggest per-wave tile that still fills the chip.
static S6Tile pick_tile6(int M, int N) {
    const int tiles = (N + 15) / 16;
    S6Tile best = make_tile(M, N, 1, 1);
    double best_score = -1.0;
    for (int mt = 1; mt <= 2; ++mt)
        for (int nt = 1; nt <= 8; ++nt) {
            const S6Tile t = make_tile(M, N, mt, nt);
            const double blocks = (double)t.mblocks * t.nblocks;
            const double useful = (double)tiles / ((double)t.nblocks * nt);
            const double fill = blocks >= 512.0 ? 1.0 : blocks / 512.0;
            const double score = mt * nt * useful * fill * (t.nblocks == 1 ? 1.15 : 1.0);
            if (score > best_score) { best_score = score; best = t; }
        }
    return best;
}

#define DFD_S6_NT_CASES(OP) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8)

template <bool CONV, bool GATE>
static void s6_dispatch(const S6Tile& t, const float* X, const unsigned short* W3, const float* bias,
                        const float* gate, const float* R, float* Y, int M, int K, int N, int HW, int act,
                        const ConvGeom& g, int res_first, hipStream_t s) {
    const int grid = ((t.mblocks + 7) / 8) * 8 * t.nblocks;
    const int Kp = (K + S6_BK - 1) / S6_BK * S6_BK, plane = s6_np(N) * Kp;
#define DFD_S6_CASE(NTV)                                                                                          \
    case NTV:                                                                                                     \
        if (t.mt == 2)                                                                                            \
            hipLaunchKernelGGL((pw6_kernel<NTV, CONV, 2, GATE>), dim3(grid), dim3(256), 0, s, X, W3, plane, Kp, bias, \
                               gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first);                 \
        else                                                                                                      \
            hipLaunchKernelGGL((pw6_kernel<NTV, CONV, 1, GATE>), dim3(grid), dim3(256), 0, s, X, W3, plane, Kp, bias, \
                               gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first);                 \
        break;
    switch (t.nt) { DFD_S6_NT_CASES(DFD_S6_CASE) }
#undef DFD_S6_CASE
}

// The best tile depends on how the block count quantises into rounds of resident blocks (8 XCDs x 32 CUs x
// 2-6 blocks, by VGPRs and LDS of the instance), on K (prologue/epilogue share) and on the L2 re-reads of X:
// measured rather than modelled.  The first call with a new (M, K, N, mode) times every tile on the caller's
// own operands (the kernel is idempotent: Y never aliases X or R) and keeps the fastest; every tile computes
// each output with the same MFMA sequence, so the choice never changes a result bit.  DFD_S6_TUNE=0: heuristic.
struct S6Key {
    int M, K, N, mode;
    bool operator<(const S6Key& o) const {
        return std::tie(M, K, N, mode) < std::tie(o.M, o.K, o.N, o.mode);
    }
};
static std::map<S6Key, S6Tile> g_tiles;
static std::mutex g_tiles_mu;

template <bool CONV, bool GATE>
static void s6_run(const float* X, const unsigned short* W3, const float* bias, const float* gate, const float* R,
                   float* Y, int M, int K, int N, int HW, int act, const ConvGeom& g, int res_first, hipStream_t s) {
    static const bool tune = !(getenv("DFD_S6_TUNE") && atoi(getenv("DFD_S6_TUNE")) == 0);
    const S6Key key{M, K, N, (CONV ? 1 : 0) | (GATE ? 2 : 0) | (CONV ? (g.ksize << 8) | (g.stride << 4) : 0)};
    S6Tile tile;
    bool have = false;
    {
        std::lock_guard<std::mutex> lk(g_tiles_mu);
        auto it = g_tiles.find(key);
        if (it != g_tiles.end()) { tile = it->second; have = true; }
    }
    if (!have) {
        tile = pick_tile6(M, N);
        hipEvent_t e0, e1;
        if (tune && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            const int tiles = (N + 15) / 16;
            float best_ms = 1e30f;
            for (int mt = 1; mt <= 2; ++mt)
                for (int nt = 1; nt <= 8 && nt <= tiles; ++nt) {
                    const S6Tile t = make_tile(M, N, mt, nt);
                    if ((double)tiles / ((double)t.nblocks * nt) < 0.7) continue;      // mostly padding
                    if ((long long)t.mblocks * t.nblocks > (1 << 20)) continue;
                    s6_dispatch<CONV, GATE>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
                    hipEventRecord(e0, s);
                    for (int r = 0; r < 3; ++r)
                        s6_dispatch<CONV, GATE>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
                    hipEventRecord(e1, s);
                    float ms = 0.f;
                    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
                    if (ms < best_ms) { best_ms = ms; tile = t; }
                }
            hipEventDestroy(e0);
            hipEventDestroy(e1);
            if (getenv("DFD_S6_VERBOSE"))
                fprintf(stderr, "[dfd] split gemm M=%d K=%d N=%d mode=%d -> mt=%d nt=%d (%.1f us)\n", M, K, N, key.mode,
                        tile.mt, tile.nt, best_ms * 1000.f / 3.f);
        }
        std::lock_guard<std::mutex> lk(g_tiles_mu);
        g_tiles[key] = tile;
    }
    s6_dispatch<CONV, GATE>(tile, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
}

bool split_gemm_supports(int K, int N) { return K % 8 == 0 && K >= 16 && split_weights_count(N, K) * 3 < (1ull << 31); }

void launch_pointwise_split(const float* X, const unsigned short* W3, const float* bias, const float* gate,
                            const float* R, float* Y, int M, int K, int N, int HW, int act, hipStream_t s) {
    const ConvGeom none{};
    if (gate) s6_run<false, true>(X, W3, bias, gate, R, Y, M, K, N, HW, act, none, 0, s);
    else s6_run<false, false>(X, W3, bias, nullptr, R, Y, M, K, N, HW, act, none, 0, s);
}

bool launch_conv_gemm_split(const float* X, const unsigned short* W3, const float* bias, const float* R, float* Y,
                            int n_img, const ConvGeom& g, int Cout, int act, bool res_first, hipStream_t s) {
    if (g.Cin % S6_BK != 0) return false;                  // a K stage must not straddle two taps
    const int M = n_img * g.Ho * g.Wo, K = g.ksize * g.ksize * g.Cin;
    if (!split_gemm_supports(K, Cout)) return false;
    s6_run<true, false>(X, W3, bias, nullptr, R, Y, M, K, Cout, 1, act, g, res_first ? 1 : 0, s);
    return true;
}

}  // namespace dfd
