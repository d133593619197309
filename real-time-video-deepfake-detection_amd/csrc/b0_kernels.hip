// EfficientNet-B0 kernels for gfx950 (CDNA4, wave64).  NHWC fp32 activations.
//
//  stem_kernel      3x3 s2 conv from the NCHW network input, folded BN + swish
//  pw_kernel<NT>    1x1 conv = GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain),
//                   W tile double-buffered in LDS, X fragments straight to VGPRs,
//                   epilogue: bias (+swish|relu) (+residual); optional SE gate folded
//                   into the X operand load
//  dw_kernel<...>   depthwise kxk conv: halo tile staged in LDS with 16-byte coalesced
//                   channel vectors, folded BN + swish, per-tile channel sums for the
//                   squeeze-excite pool reduced with wave shuffles
//  se_kernel        pool finish + two tiny FCs -> per-(image,channel) gate
//  avgpool_kernel   global average pool of the head conv output
//
// Semantics follow reference model.py:63-72 (forward = EfficientNet-B0 + MLP head); the
// arithmetic of each layer is checked against oracle/b0_ref.py by tests/test_b0_gpu.py.
#include "b0_kernels.h"
#include "kernel_util.h"

#include <cstdlib>

namespace dfd {

#ifdef MB_TRACE
// cycle trace of one thread of one mbconv block (build with EXTRA=-DMB_TRACE; profiles/micro/mb_trace.py)
__device__ long long g_mb_trace[256];
#define MB_TP(id)                                                                                             \
    do {                                                                                                      \
        if (H == MB_TRACE_H && S == MB_TRACE_S && bid.x == 5 && bid.n == 3 && threadIdx.x == 0 && mtp < 250) { \
            g_mb_trace[mtp++] = (long long)(id);                                                              \
            g_mb_trace[mtp++] = (long long)__builtin_amdgcn_s_memtime();                                      \
        }                                                                                                     \
    } while (0)
#ifndef MB_TRACE_H
#define MB_TRACE_H 56
#endif
#ifndef MB_TRACE_S
#define MB_TRACE_S 1
#endif
extern "C" int dfd_debug_mb_trace(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mb_trace), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#else
#define MB_TP(id) do { } while (0)
#endif

// ------------------------------------------------------------------------------ stem
// One thread = one output pixel x 4 output channels; 8 consecutive lanes share a pixel, so
// the 27 input taps are wave-broadcast loads and the 128-byte NHWC output row is one
// coalesced store per 8 lanes.  TF-SAME for 224 -> 112 at k3 s2 pads one row/col at the
// high side only (reference dependency efficientnet_pytorch Conv2dStaticSamePadding).
template <typename XT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b,
                                                   XT* __restrict__ y, int n_img) {
    __shared__ float ws[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i] = w[i];
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid & 7);
    const long long pix = gid >> 3;
    if (pix >= (long long)n_img * 112 * 112) return;
    const int ox = (int)(pix % 112);
    const int oy = (int)((pix / 112) % 112);
    const int n = (int)(pix / (112 * 112));
    v4f acc = ldg4(b + 4 * cg);
    const float* xb = x + (size_t)n * 3 * 224 * 224;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky;
        if (iy >= 224) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox + kx;
            if (ix >= 224) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = xb[(size_t)ci * 224 * 224 + iy * 224 + ix];
                const v4f wv = *reinterpret_cast<const v4f*>(&ws[((ky * 3 + kx) * 3 + ci) * 32 + 4 * cg]);
                acc += v * wv;
            }
        }
    }
    st4(y + (size_t)pix * 32 + 4 * cg, swish4(acc));
}

template <typename XT>
void launch_stem(const float* x, const float* w, const float* b, XT* y, int n, hipStream_t s) {
    const long long threads = (long long)n * 112 * 112 * 8;
    const int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(stem_kernel<XT>, dim3(grid), dim3(256), 0, s, x, w, b, y, n);
}
template void launch_stem<float>(const float*, const float*, const float*, float*, int, hipStream_t);
template void launch_stem<bf16_t>(const float*, const float*, const float*, bf16_t*, int, hipStream_t);

// ------------------------------------------------------------------------ pointwise GEMM
// D = A*B with v_mfma_f32_16x16x4_f32: A[i][k] on lane (i = l&15, k = l>>4), B[k][j] on lane
// (j = l&15, k = l>>4), D[i][j] on lane l register r with j = l&15, i = 4*(l>>4)+r.
// Here i = output channel, j = pixel: a lane ends with 4 consecutive channels of one pixel,
// i.e. one 16-byte NHWC store.  Each lane fetches 4 consecutive k (one b128) per 16-wide K
// chunk and feeds element s to MFMA s, so MFMA s sums k = {s, 4+s, 8+s, 12+s}: the same
// permutation on both operands, which is all the contraction needs.
constexpr int PW_BK = 32;          // K per LDS stage
constexpr int PW_BKP = PW_BK + 8;  // +8 floats: conflict-free ds_read_b128 of 16 rows x 4 k-quads

// CONV = true turns the same kernel into an implicit-GEMM k x k convolution (SSD detector):
// row m is output pixel (n, oy, ox), K = taps * C_in with the taps outermost, so a 32-wide K
// stage (C_in % 32 == 0) lies inside one tap and the X fragment is still one 16-byte load per
// lane from the NHWC input, zero outside the image.  res_first: add R before the activation
// (ResNet basic block) instead of after it (MBConv skip).
template <int NT, bool CONV, int MT, bool GATE>
__global__ __launch_bounds__(256) void pw_kernel(const float* __restrict__ X,
                                                 const float* __restrict__ W,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ gate,
                                                 const float* __restrict__ R,
                                                 float* __restrict__ Y, int M, int K, int N,
                                                 int HW, int act, int mblocks, int nblocks,
                                                 ConvGeom cg, int res_first) {
    constexpr int BK = PW_BK, BKP = PW_BKP;
    constexpr int BN = NT * 16, BM = 4 * MT * 16;
    constexpr int WLOADS = (BN * (BK / 4) + 255) / 256;
    __shared__ __attribute__((aligned(16))) float ws[2][BN * BKP];

    // XCD-aware order: blocks b and b+8 share an XCD (and its L2); give each XCD a run of
    // m-blocks and walk the n-blocks of one m-block back to back so X is re-read from L2.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int n0 = nblk * BN;

    int m[MT];
    size_t gbase[MT];
    int iy0[MT], ix0[MT];       // CONV: top-left input coordinate of the pixel's receptive field
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

    // Operand loads are UNCONDITIONAL (addresses clamped into the tensor) and nothing touches a loaded
    // value until the step that consumes it: a load under a branch, or an ALU op right behind it (the SE
    // gate multiply used to sit here), makes hipcc wait vmcnt(0) on the spot, and the "prefetch" of the next
    // K-step then overlaps nothing.  Rows >= M are clamped to M-1 (never stored); k >= K is clamped to K-4
    // and meets zero weights (the W tile is zero-filled there at store time).
    v4f wreg[WLOADS];
    v4f xcur[MT][2], xnext[MT][2];
    v4f gcur[MT][2], gnext[MT][2];      // raw SE gate fragments (GATE only), multiplied in at use
    int mclamp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mclamp[mt] = m[mt] < M ? m[mt] : M - 1;

    auto w_ok = [&](int t, int kc) {
        const int e = tid + t * 256;
        return e < BN * (BK / 4) && n0 + (e >> 3) < N && kc + 4 * (e & 7) < K;
    };
    auto load_w = [&](int kc) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) {
            const int e = tid + t * 256;
            int n = n0 + (e >> 3), k = kc + 4 * (e & 7);
            n = n < N ? n : N - 1;
            k = k < K ? k : K - 4;
            wreg[t] = ldg4(W + (size_t)n * K + k);
        }
    };
    auto store_w = [&](int buf, int kc) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) {
            const int e = tid + t * 256;
            const int row = e >> 3, c4 = e & 7;
            if (e < BN * (BK / 4))
                *reinterpret_cast<v4f*>(&ws[buf][row * BKP + 4 * c4]) = w_ok(t, kc) ? wreg[t] : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto load_x = [&](int kc, v4f (&xf)[MT][2], v4f (&gf)[MT][2]) {
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // wave-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int iy = iy0[mt] + ky * cg.dil, ix = ix0[mt] + kx * cg.dil;
                const bool ok = m[mt] < M && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const float* p = X + gbase[mt] + ((size_t)(ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin + ci0 + 4 * q;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    xf[mt][cc] = ldg4(p + cc * 16);
                    gf[mt][cc] = ok ? (v4f){1.f, 1.f, 1.f, 1.f} : (v4f){0.f, 0.f, 0.f, 0.f};   // zero padding, applied at use
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    int k = kc + cc * 16 + 4 * q;
                    k = k < K ? k : K - 4;
                    xf[mt][cc] = ldg4(X + (size_t)mclamp[mt] * K + k);
                    if constexpr (GATE) gf[mt][cc] = ldg4(gate + gbase[mt] + k);
                }
        }
    };

    const int nk = (K + BK - 1) / BK;
    load_w(0);
    load_x(0, xcur, gcur);
    store_w(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        // branch-free body: the last iteration re-loads its own step (unused).  An `if (more)` around the
        // prefetch makes a join point where hipcc must assume nothing newer is in flight and emits
        // vmcnt(0) before the MFMAs - which on the common path drains the prefetch it just issued.
        const int kn = (kt + 1 < nk ? kt + 1 : kt) * BK;
        load_w(kn);
        load_x(kn, xnext, gnext);
        if constexpr (GATE || CONV) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xcur[mt][0] *= gcur[mt][0];
                xcur[mt][1] *= gcur[mt][1];
            }
        }
        const float* wb = ws[kt & 1];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            // all NT weight fragments of this 16-wide K chunk are read before the first MFMA
            v4f wfa[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wfa[nt] = *reinterpret_cast<const v4f*>(&wb[(nt * 16 + j) * BKP + cc * 16 + 4 * q]);
            // every accumulator gets one MFMA per k-quad e before any gets its next: a dependent accumulate is
            // MT*NT issues behind its producer (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle latency)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfa[nt][e], xcur[mt][cc][e], acc[mt][nt], 0, 0, 0);
        }
        store_w((kt + 1) & 1, kn);              // the other buffer: nobody reads it after the last step
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xcur[mt][0] = xnext[mt][0];
            xcur[mt][1] = xnext[mt][1];
            if constexpr (GATE || CONV) {
                gcur[mt][0] = gnext[mt][0];
                gcur[mt][1] = gnext[mt][1];
            }
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

// Tile choice: a wave computes MT 16-pixel tiles x NT 16-channel tiles; a block is 4 waves (64*MT pixels).
// Late layers have few pixels (batch*49 or batch*196), so the largest tile starves the 256 CUs: prefer the
// biggest per-wave tile (MFMAs per operand load) among the choices that still launch >= 3 blocks per CU.
struct PwTile { int mt, nt, mblocks, nblocks; };
static PwTile pick_tile(int M, int N) {
    const int tiles = (N + 15) / 16;
    PwTile best{2, 1, 0, 0};
    double best_score = -1.0;
    static const double fill_target = getenv("DFD_PW_FILL") ? atof(getenv("DFD_PW_FILL")) : 768.0;
    static const double one_nb_bonus = getenv("DFD_PW_BONUS") ? atof(getenv("DFD_PW_BONUS")) : 1.15;
    static const int mt_max = getenv("DFD_PW_MTMAX") ? atoi(getenv("DFD_PW_MTMAX")) : 2;
    for (int mt = 1; mt <= mt_max; ++mt)
        for (int nt = 1; nt <= 10; ++nt) {
            const int mb = (M + 64 * mt - 1) / (64 * mt), nb = (tiles + nt - 1) / nt;
            const double blocks = (double)mb * nb;
            const double useful = (double)tiles / ((double)nb * nt);          // padding waste of the last n-block
            // work per wave-step, discounted when the grid cannot fill the chip (768 = 3 blocks per CU)
            const double fill = blocks >= fill_target ? 1.0 : blocks / fill_target;
            const double score = mt * nt * useful * fill * (nb == 1 ? one_nb_bonus : 1.0);   // one n-block: X read once
            if (score > best_score) { best_score = score; best = PwTile{mt, nt, mb, nb}; }
        }
    return best;
}

#define DFD_PW_NT_CASES(OP) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10)

template <bool CONV, bool GATE>
static void pw_dispatch(const PwTile& t, const float* X, const float* W, const float* bias, const float* gate,
                        const float* R, float* Y, int M, int K, int N, int HW, int act, const ConvGeom& g,
                        int res_first, hipStream_t s) {
    const int grid = ((t.mblocks + 7) / 8) * 8 * t.nblocks;
#define DFD_PW_CASE(NTV)                                                                                             \
    case NTV:                                                                                                        \
        if (t.mt == 2)                                                                                               \
            hipLaunchKernelGGL((pw_kernel<NTV, CONV, 2, GATE>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, R, Y, \
                               M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first);                                \
        else                                                                                                         \
            hipLaunchKernelGGL((pw_kernel<NTV, CONV, 1, GATE>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, R, Y, \
                               M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first);                                \
        break;
    switch (t.nt) { DFD_PW_NT_CASES(DFD_PW_CASE) }
#undef DFD_PW_CASE
}

void launch_pointwise(const float* X, const float* W, const float* bias, const float* gate,
                      const float* R, float* Y, int M, int K, int N, int HW, int act,
                      hipStream_t s) {
    const ConvGeom none{};
    if (gate) pw_dispatch<false, true>(pick_tile(M, N), X, W, bias, gate, R, Y, M, K, N, HW, act, none, 0, s);
    else pw_dispatch<false, false>(pick_tile(M, N), X, W, bias, nullptr, R, Y, M, K, N, HW, act, none, 0, s);
}

bool launch_conv_gemm(const float* X, const float* W, const float* bias, const float* R, float* Y, int n_img,
                      const ConvGeom& g, int Cout, int act, bool res_first, hipStream_t s) {
    if (g.Cin % PW_BK != 0) return false;                  // a K stage must not straddle two taps
    const int M = n_img * g.Ho * g.Wo, K = g.ksize * g.ksize * g.Cin;
    pw_dispatch<true, false>(pick_tile(M, Cout), X, W, bias, nullptr, R, Y, M, K, Cout, 1, act, g, res_first ? 1 : 0, s);
    return true;
}

// --------------------------------------------------------------------------- depthwise
// XCD-aware, image-major block order for the depthwise family (1-D grids of bpi * n_img blocks).  The dispatcher
// deals consecutive workgroup ids round-robin over the 8 XCDs (MI355X_MICROARCH.md, Workgroup dispatch): with the
// natural order the 15 channel-chunk blocks of ONE image landed on all eight L2s and each fetched the image's input
// again (PMC, round 2: block 4 read 8.4x its input, block 5 4.2x, the stem 3.0x).  Here ids that share an XCD
// (id % 8) walk a contiguous run of the image-major sequence, so every block of an image - all its channel chunks
// and spatial tiles with their overlapping halos - runs on one XCD close in time and the input crosses HBM once.
// Bijective for any grid size (guide T1); placement only affects speed, never results.
struct BlkId { int x, n; };
__device__ __forceinline__ BlkId blk_image_major(int bpi) {
    const int L = blockIdx.x, total = gridDim.x;
    const int xcd = L & 7, slot = L >> 3, q = total >> 3, r = total & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int n = Lp / bpi;
    return BlkId{Lp - n * bpi, n};
}

// Block = one image x one CB-channel chunk x one TH x TW output tile.  The (TH-1)S+K by
// (TW-1)S+K input halo tile is staged in LDS as float4 channel vectors (zero-filled outside
// the image: TF-SAME padding), weights too.  Thread = (channel quad, strip of RP outputs
// along W); per kernel row it pulls the (RP-1)S+K input vectors of the strip into
// registers once and reuses them across the K taps.
// LDS tile layout shared by the three producers (halo staging, fused expand, fused stem) and dw_compute: pixel p,
// channel quad c -> 16-byte unit p * CG + (c ^ swz(p)).  The MFMA expand hands a lane (pixel = lane & 15, quad
// = lane >> 4): 8 consecutive lanes of a ds_write_b128 are 8 consecutive pixels of ONE quad, which in the plain
// p * CG + c layout is the same bank group 8 times (CG = 8) or 4 times (CG = 4) - 67 % / 41-51 % of all LDS
// cycles of those kernels were conflict cycles, on an LDS that was busy 50-77 % of the time.  XOR-ing the quad
// with low pixel bits spreads them over all banks; within a pixel it is a permutation, so producers that write
// quad-fastest and the reads (one pixel per lane quad-group) keep their bank pattern.
// Measured: it pays where the writes were 8-way conflicted (CG = 8 fused kernel, block 1: 0.263 -> 0.250 ms); at
// CG = 4 and in the kernels whose producers were conflict-free anyway the per-tap index arithmetic of the reads
// (no more base + immediate offsets) costs more than the conflicts did (+8 %), so SWZ is on for that one only.
// ---- squeeze-excite tail ----------------------------------------------------------------------------------------------
// Called by every thread of a depthwise-family block after its pool partials P[n][tile][c0 ..] are stored (by lanes of
// wave 0, with agent-scope write-through stores).  Hand-off (MI355X_MICROARCH.md, Workgroup dispatch / Valid forms):
//   producer  sc1 stores of the partials -> the storing wave's s_waitcnt vmcnt(0) -> ONE lane's agent-scope atomic add on the
//             image's counter (same wave: the add comes after the wait);
//   consumer  the block whose add returned bpi - 1 is the last of the image: workgroup barrier -> agent-scope ACQUIRE by
//             every wave -> s_waitcnt vmcnt(0) -> barrier -> sc1 loads of all partials (several blocks per CU here, so the
//             acquire stays although stores and loads are both sc1).
// The last block resets the counter (the next launch on the stream starts from 0), reduces the tiles in index order
// (four interleaved partial sums, then a fixed fold: the result does not depend on which block came last), and runs the
// two FCs with the block's NT threads.  `lds`: >= C + 64 floats of the block's LDS that nobody reads any more.
template <int NT>
__device__ __forceinline__ void se_tail(const SeTail& T, const float* __restrict__ P, int n, int C, int bpi, int tiles, float* lds) {
    if (!T.counter) return;                                            // kernel argument: uniform
    __shared__ int se_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // wave 0's partial-sum stores have been acknowledged
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(T.counter + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            se_last = old == (unsigned)(bpi - 1) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!se_last) return;
    if (tid == 0) __hip_atomic_store(T.counter + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* mean = lds;
    float* z = lds + ((C + 15) & ~15);
    const unsigned* p = reinterpret_cast<const unsigned*>(P + (size_t)n * tiles * C);
    for (int c = tid; c < C; c += NT) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int t = 0;
        for (; t + 3 < tiles; t += 4) {
            s0 += __uint_as_float(__hip_atomic_load(p + (size_t)t * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            s1 += __uint_as_float(__hip_atomic_load(p + (size_t)(t + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            s2 += __uint_as_float(__hip_atomic_load(p + (size_t)(t + 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            s3 += __uint_as_float(__hip_atomic_load(p + (size_t)(t + 3) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        for (; t < tiles; ++t) s0 += __uint_as_float(__hip_atomic_load(p + (size_t)t * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        mean[c] = ((s0 + s1) + (s2 + s3)) * T.inv_hw;
    }
    __syncthreads();
    // FC1 + swish: wave w takes outputs w, w + NW, ...; the lanes stride over the channels, butterfly fold
    for (int o = wave; o < T.c_se; o += NW) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += mean[c] * T.w1[(size_t)o * C + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) z[o] = swish1(s + T.b1[o]);
    }
    __syncthreads();
    // FC2 + sigmoid: a thread per channel
    for (int c = tid; c < C; c += NT) {
        float s = T.b2[c];
        for (int o = 0; o < T.c_se; ++o) s += z[o] * T.w2t[(size_t)o * C + c];
        T.gate[(size_t)n * C + c] = sigmoid1(s);
    }
}

// pool partials leave the CU with agent-scope write-through stores (sc1): whichever block of the image runs the
// squeeze-excite tail reads them with sc1 loads
__device__ __forceinline__ void stg4_agent(float* p, v4f v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    __hip_atomic_store(u, __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 1, __float_as_uint(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 2, __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 3, __float_as_uint(v.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS unit (16 bytes) of channel quad c of tile pixel p.  SWZ 0: plain; 1: XOR swizzle (the expand epilogue stores 16
// pixels x one quad per instruction: plain, all 16 land in the same four banks); 2: pixel stride CG + 1 units - the same
// spread (stride 36 dwords), and the depthwise phase's reads keep compile-time offsets (the swizzle costs ~3 VALU
// instructions of address arithmetic per read: a third of that phase's instructions at 14 x 14, RP = 7)
template <int CG, int SWZ>
__device__ __forceinline__ int tile_unit(int p, int c) {
    if constexpr (SWZ == 1 && CG >= 8) return p * CG + (c ^ (p & 7));
    else if constexpr (SWZ == 2) return p * (CG + 1) + c;
    else return p * CG + c;
}

template <int K, int S, int CB, int TH, int TW>
struct DwShape {
    static constexpr int CG = CB / 4;
    static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
};

// depthwise conv of the LDS tile + folded BN + swish + store + per-tile SE partial sums.
// Ends with a barrier-protected write of P; callers that reuse tile/wl/red afterwards must
// __syncthreads() first.
template <int K, int S, int CB, int TH, int TW, int RP, int SWZ = 0, typename XT = float, int ABL = 0, int NT = 256>
__device__ __forceinline__ void dw_compute(const v4f* tile, const v4f* wl, v4f* red,
                                           const v4f bv, XT* __restrict__ Y,
                                           float* __restrict__ P, int n, int Ho, int C, int c0, int ty0,
                                           int tx0, int t, int tiles_sp) {
    constexpr int CG = CB / 4;
    constexpr int IW = (TW - 1) * S + K;
    constexpr int SX = TW / RP, NSTRIP = TH * SX, NSLOT = NT / CG, NW = NT / 64;
    constexpr int NIN = (RP - 1) * S + K;
    constexpr bool POW2 = (CG & (CG - 1)) == 0;
    static_assert(TW % RP == 0 && CG <= 64 && (NW == 4 || NW == 8), "tile shape");
    const int tid = threadIdx.x;
    const int cg = tid % CG, slot = tid / CG;
    v4f psum = (v4f){0.f, 0.f, 0.f, 0.f};
    XT* yb = Y + (size_t)n * Ho * Ho * C + c0 + 4 * cg;
    if (POW2 || slot < NSLOT) {                          // CG not a power of two: the last NT % CG threads idle
        for (int strip = slot; strip < NSTRIP; strip += NSLOT) {
            const int oy = strip / SX, ox0 = (strip % SX) * RP;
            v4f acc[RP];
#pragma unroll
            for (int p = 0; p < RP; ++p) acc[p] = bv;
            auto tap_row = [&](int ky) {
                const int p0 = (oy * S + ky) * IW + ox0 * S;
                v4f in[NIN];
#pragma unroll
                for (int i = 0; i < NIN; ++i) in[i] = tile[tile_unit<CG, SWZ>(p0 + i, cg)];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const v4f w = wl[(ky * K + kx) * CG + cg];
#pragma unroll
                    for (int p = 0; p < RP; ++p) acc[p] += in[p * S + kx] * w;
                }
            };
            // rolled: one kernel row (K weight vectors) live at a time instead of all K * K
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) tap_row(ky);
            const int gy = ty0 + oy;
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const int gx = tx0 + ox0 + p;
                if (gy < Ho && gx < Ho) {
                    const v4f v = ABL == 4 ? acc[p] : swish4(acc[p]);
                    if (ABL != 5) st4(yb + ((size_t)gy * Ho + gx) * C, v);
                    psum += v;
                }
            }
        }
    }
    if constexpr (POW2) {
        // squeeze-excite pool: lanes whose ids differ by a multiple of CG hold the same channels
#pragma unroll
        for (int off = CG; off < 64; off <<= 1) {
            psum.x += __shfl_xor(psum.x, off);
            psum.y += __shfl_xor(psum.y, off);
            psum.z += __shfl_xor(psum.z, off);
            psum.w += __shfl_xor(psum.w, off);
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < CG) red[wave * CG + lane] = psum;
        __syncthreads();
        if (tid < CG) {
            v4f v = (red[tid] + red[CG + tid]) + (red[2 * CG + tid] + red[3 * CG + tid]);
            if constexpr (NW == 8) v += (red[4 * CG + tid] + red[5 * CG + tid]) + (red[6 * CG + tid] + red[7 * CG + tid]);
            stg4_agent(P + ((size_t)n * tiles_sp + t) * C + c0 + 4 * tid, v);
        }
    } else {
        // any CG: every (slot, channel quad) partial through LDS, folded in slot order (a fixed order)
        if (slot < NSLOT) red[slot * CG + cg] = psum;
        __syncthreads();
        if (tid < CG) {
            v4f v = red[tid];
            for (int sl = 1; sl < NSLOT; ++sl) v += red[sl * CG + tid];
            stg4_agent(P + ((size_t)n * tiles_sp + t) * C + c0 + 4 * tid, v);
        }
    }
}

template <int K, int S, int CB, int TH, int TW, int RP, typename XT>
__global__ __launch_bounds__(256, (RP >= 4 ? 2 : 4)) void dw_kernel(const XT* __restrict__ X,
                                                 const float* __restrict__ Wt,
                                                 const float* __restrict__ bias,
                                                 XT* __restrict__ Y, float* __restrict__ P,
                                                 int H, int Ho, int C, int pad_lo, int tiles_x,
                                                 int tiles_sp, SeTail se) {
    using Sh = DwShape<K, S, CB, TH, TW>;
    constexpr int CG = Sh::CG, IH = Sh::IH, IW = Sh::IW;
    __shared__ v4f tile[IH * IW * CG];
    __shared__ v4f wl[K * K * CG];
    __shared__ v4f red[4 * CG];
    const int tid = threadIdx.x;
    const BlkId bid = blk_image_major(tiles_sp * (C / CB));
    const int n = bid.n;
    const int t = bid.x % tiles_sp, chunk = bid.x / tiles_sp;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW, c0 = chunk * CB;
    const v4f bv = ldg4(bias + c0 + 4 * (tid % CG));      // issued with the tile loads, used after the barrier
    for (int i = tid; i < K * K * CG; i += 256)
        wl[i] = ldg4(Wt + (size_t)(i / CG) * C + c0 + 4 * (i % CG));
    const int iy0 = ty0 * S - pad_lo, ix0 = tx0 * S - pad_lo;
    const XT* xb = X + (size_t)n * H * H * C + c0;
    // Halo tile -> LDS.  Every load is unconditional (address clamped into the image, VALUE masked) and all of a
    // thread's loads are issued before the first LDS store: a load under `if (inside)` makes hipcc branch and
    // wait per element - the s_memtime trace of the fused stem kernel showed 57 % of a block's time in ten
    // serialized round trips of exactly this loop shape.  Threads past the last element repeat it (same value,
    // same address), so the stores are unconditional too.
    constexpr int NEL = IH * IW * CG, NLD = (NEL + 255) / 256;
    v4f stg[NLD];
    bool stg_ok[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        const int cg = i % CG, pix = i / CG;
        const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
        const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;
        stg[k] = ld4(xb + ((size_t)(inside ? iy : 0) * H + (inside ? ix : 0)) * C + 4 * cg);
        stg_ok[k] = inside;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * 256 < NEL ? tid + k * 256 : NEL - 1;
        tile[i] = stg_ok[k] ? stg[k] : (v4f){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    dw_compute<K, S, CB, TH, TW, RP, false, XT>(tile, wl, red, bv, Y, P, n, Ho, C, c0, ty0, tx0, t, tiles_sp);
    se_tail<256>(se, P, n, C, tiles_sp * (C / CB), tiles_sp, reinterpret_cast<float*>(tile));
}

// MBConv front half in ONE kernel (blocks 1-5): X is the block INPUT [n][H][H][Cin]; for every
// pixel of the halo tile the 1x1 expand conv (+ folded BN + swish) runs on
// v_mfma_f32_16x16x4_f32 with pixels as MFMA columns, so a lane ends with the 4 consecutive
// channels of one pixel = exactly one float4 slot of the LDS tile; pixels outside the image stay
// zero (the padding applies to the expanded tensor).  The 6x-expanded activation never touches
// HBM.  A block walks NSUB channel chunks of CB channels over the same spatial tile: the input
// fragments of all its pixels are loaded ONCE, up front (one batch of loads in flight), and reused
// for every chunk.  Price: the halo's expand FLOPs are recomputed (1.1-1.65x).
// (round 3) fp32 only, exactly Cin / 4 MFMAs per pixel tile: lane k-group q supplies the NM = Cin / 4 consecutive
// channels q * NM .. of its pixel (see mbconv2_kernel); KC = 16-byte registers per tile and lane = ceil(NM / 4).
template <int K, int S, int CB, int TH, int TW, int RP, int KC, int NSUB, typename XT, int CI>
__global__ __launch_bounds__(256, ((S == 1 && (((TH - 1) * S + K) * ((TW - 1) * S + K) + 63) / 64 * KC > 12) ? 2 : 3)) void mbconv_kernel(const XT* __restrict__ X,
                                                     const float* __restrict__ We,
                                                     const float* __restrict__ be,
                                                     const float* __restrict__ Wt,
                                                     const float* __restrict__ bias,
                                                     XT* __restrict__ Y, float* __restrict__ P,
                                                     int H, int Ho, int C, int Cin, int pad_lo,
                                                     int tiles_x, int tiles_sp, SeTail se) {
    using Sh = DwShape<K, S, CB, TH, TW>;
    constexpr int CG = Sh::CG, IH = Sh::IH, IW = Sh::IW;
    constexpr int NTB = CB / 16;                        // 16-channel MFMA row tiles per chunk
    constexpr int NP = IH * IW, NMT = (NP + 15) / 16, NIT = (NMT + 3) / 4;
    constexpr int NM = CI / 4;
    static_assert(sizeof(XT) == 4 && CI % 8 == 0 && KC == (NM + 3) / 4, "fp32 activations, compile-time Cin");
    __shared__ v4f tile[IH * IW * CG];
    __shared__ v4f wl[K * K * CG];
    __shared__ v4f red[4 * CG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const BlkId bid = blk_image_major(tiles_sp * (C / (CB * NSUB)));
    const int n = bid.n;
    const int t = bid.x % tiles_sp, group = bid.x / tiles_sp;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const int iy0 = ty0 * S - pad_lo, ix0 = tx0 * S - pad_lo;
#ifdef MB_TRACE
    int mtp = 0;
#endif
    MB_TP(0);

    // B operand for all of this wave's pixel tiles: lane (pixel j, k-quad q); loaded once
    const XT* xb = X + (size_t)n * H * H * Cin;
    v4f xf[NIT][KC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = (wave + 4 * it) * 16 + j;
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        const bool inside = p < NP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;
        const float* px = reinterpret_cast<const float*>(xb) + ((size_t)(inside ? iy : 0) * H + (inside ? ix : 0)) * CI + q * NM;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            v4f v;                                               // unconditional load, masked value
            if (NM % 4 == 0 || kk < KC - 1) v = ldg4u(px + 4 * kk);
            else { const v2f t2 = ldg2(px + 4 * kk); v = (v4f){t2.x, t2.y, 0.f, 0.f}; }
            xf[it][kk] = inside ? v : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }

    // Per-chunk weights (depthwise taps, biases, the chunk's expand rows) are requested one chunk ahead into
    // registers: loaded at the top of their own chunk they cost an exposed L2 round trip per chunk (600-1600 of
    // ~10k cycles in the s_memtime trace).  All loads unconditional (clamped), see dw_kernel.
    static_assert(K * K * CG <= 256, "one depthwise weight vector per thread");
    struct ChunkW { v4f wl, bv, wf[NTB][KC], bex[NTB]; };
    auto load_chunk = [&](int sub, ChunkW& w) {
        const int c0 = (group * NSUB + sub) * CB;
        const int i = tid < K * K * CG ? tid : 0;
        w.wl = ldg4(Wt + (size_t)(i / CG) * C + c0 + 4 * (i % CG));
        w.bv = ldg4(bias + c0 + 4 * (tid % CG));
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
            w.bex[nt] = ldg4(be + c0 + nt * 16 + 4 * q);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const float* pw = We + (size_t)(c0 + nt * 16 + j) * CI + q * NM + 4 * kk;
                if (NM % 4 == 0 || kk < KC - 1) w.wf[nt][kk] = ldg4u(pw);
                else { const v2f t2 = ldg2(pw); w.wf[nt][kk] = (v4f){t2.x, t2.y, 0.f, 0.f}; }
            }
        }
    };
    ChunkW cw[2];
    load_chunk(0, cw[0]);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int c0 = (group * NSUB + sub) * CB;
        ChunkW& cur = cw[sub & 1];
        MB_TP(1);
        if (sub > 0) __syncthreads();                            // previous chunk done with tile/wl/red
        MB_TP(2);
        if (tid < K * K * CG) wl[tid] = cur.wl;
        load_chunk(sub + 1 < NSUB ? sub + 1 : NSUB - 1, cw[(sub + 1) & 1]);      // lands during this chunk
        const v4f bv = cur.bv;
        v4f (&wf)[NTB][KC] = cur.wf;
        v4f (&bex)[NTB] = cur.bex;
        MB_TP(3);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int mt = wave + 4 * it;
            if (mt < NMT) {                                      // wave-uniform
                v4f acc[NTB];                                    // starts at the folded-BN bias: no add in the epilogue
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) acc[nt] = bex[nt];
                if constexpr (NTB == 1) {                        // two independent MFMA chains over the K steps (see mbconv2_kernel)
                    v4f acc1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][i / 4][i % 4], xf[it][i / 4][i % 4], acc1, 0, 0, 0);
                        else acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][i / 4][i % 4], xf[it][i / 4][i % 4], acc[0], 0, 0, 0);
                    }
                    acc[0] += acc1;
                } else {
#pragma unroll
                    for (int i = 0; i < NM; ++i)
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][i / 4][i % 4], xf[it][i / 4][i % 4], acc[nt], 0, 0, 0);
                }
                const int p = mt * 16 + j;
                const int iy = iy0 + p / IW, ix = ix0 + p % IW;
                const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;
                if (p < NP) {
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt)
                        tile[tile_unit<CG, (CG >= 8)>(p, nt * 4 + q)] = inside ? swish4(acc[nt]) : (v4f){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        MB_TP(4);
        __syncthreads();
        MB_TP(5);
        dw_compute<K, S, CB, TH, TW, RP, (CG >= 8), XT>(tile, wl, red, bv, Y, P, n, Ho, C, c0, ty0, tx0, t, tiles_sp);
        MB_TP(6);
    }
    se_tail<256>(se, P, n, C, tiles_sp * (C / (CB * NSUB)), tiles_sp, reinterpret_cast<float*>(tile));
#ifdef MB_TRACE
    if (H == MB_TRACE_H && S == MB_TRACE_S && bid.x == 5 && bid.n == 3 && threadIdx.x == 0) g_mb_trace[255] = mtp;
#endif
}

// ---- MBConv front half, second generation (blocks 1-5) ---------------------------------------------------
// What the first version spent its time on (s_memtime trace + ISA): per 16-pixel tile ONE dependent chain of
// up to 12 v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency: s_nops between them), then the
// swish epilogue, then the next tile - nothing overlapped, under exec-masked branches per tile; and the
// halo's expand FLOPs were recomputed up to 1.65x because tiles had to stay small (X fragments of ALL tiles in
// registers for the chunk loop).  Here:
//   * the 1x1 expand runs on v_mfma_f32_16x16x32_bf16 with split-precision operands exactly as gemm_split: the
//     weights are the handle's three bf16 planes (split_weights), an fp32 activation is split in registers into
//     three bf16 terms (six products, fp32-exact), a bf16 activation is the operand as loaded (three products);
//     16 instead of 32 cycles per MFMA and K = 32 per instruction: 96-192 instead of 128-384 cycles per tile;
//   * a block owns ONE channel chunk of CB = 16 * NTB channels (any multiple of 16: C = 144 / 240 take 48 / 80)
//     and keeps NTB independent accumulators per pixel tile in flight;
//   * the pixel-tile loop is branch-free (clamped addresses, masked values) and rolled, with the next tile's
//     activations requested before the current tile's MFMAs: registers stay bounded for any tile size, so tiles
//     grow until LDS says stop and the halo recompute shrinks;
//   * dw_compute takes any CG (pool partials through LDS when CG is not a power of two).
//   * (round 3) NT = threads per block: 512 doubles the waves that share one LDS tile (block 4's whole-image tile is
//     64 KB: two blocks per CU were two waves per SIMD);
//   * (round 3) INS: the pixel tiles enumerate only the pixels of the halo tile that lie INSIDE the image (its clipped
//     rectangle, row-major): the zero border of the TF-SAME padding is written directly, not computed - block 4's
//     32 x 32 halo tile holds 28 x 28 = 49 tiles of real pixels, not 64.
//   * (round 3) fp32 expand with EXACTLY Cin / 4 MFMAs per pixel tile: lane k-group q supplies the NM = Cin / 4
//     consecutive channels q * NM .. q * NM + NM - 1 of its pixel (one contiguous 4 * NM-byte piece of the NHWC record;
//     MFMA e contracts k = {e, NM + e, 2 NM + e, 3 NM + e} - the same permutation on both operands).  The 16-channel
//     steps it replaces issued 16 MFMAs for Cin = 40 (10 needed) and 8 for Cin = 24 (6 needed);
//   * (round 3) the activation ring is refilled AFTER the tile's MFMAs have issued: refilled before them the slot's
//     old value had to be copied aside, the copy landed in the loop latch behind `s_waitcnt vmcnt(0)` (ISA), and every
//     4-tile iteration drained the whole ring - an s_memtime trace put 64 % of block 4's time in that loop.
template <int K, int S, int CB, int TH, int TW, int RP, int NK, typename XT, int ABL = 0, int NT = 256, bool INS = false,
          int CI = 0, int MINB = (NT == 256 ? 2 : 4), bool SKEW = false>
__global__ __launch_bounds__(NT, MINB) void mbconv2_kernel(const XT* __restrict__ X,
                                                      const unsigned short* __restrict__ We3, int plane, int Kp,
                                                      const float* __restrict__ Wef,
                                                      const float* __restrict__ be,
                                                      const float* __restrict__ Wt,
                                                      const float* __restrict__ bias,
                                                      XT* __restrict__ Y, float* __restrict__ P,
                                                      int H, int Ho, int C, int Cin, int pad_lo,
                                                      int tiles_x, int tiles_sp, SeTail se) {
    using Sh = DwShape<K, S, CB, TH, TW>;
    constexpr int CG = Sh::CG, IH = Sh::IH, IW = Sh::IW;
    constexpr int NTB = CB / 16;                        // 16-channel MFMA row tiles of the chunk
    constexpr int NW = NT / 64;
    constexpr int NPX = IH * IW, NMT = (NPX + 15) / 16, NIT = (NMT + NW - 1) / NW;
    constexpr int ESZ = (int)sizeof(XT);
    // fp32 activations: the expand runs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, KC = ceil(Cin / 16) steps of
    // four MFMAs, lane = k-quad q).  These kernels are VALU-bound (SQ_ACTIVE_INST_VALU 80-90 % of the launch) and
    // the matrix pipe is ~20 % busy: splitting every activation into three bf16 terms costs ~45 VALU instructions
    // per 8 values - on the bottleneck - to save time on a unit that is idle anyway.  bf16 activations are the
    // bf16 MFMA's operand as loaded (three products against the three weight planes).
    constexpr int NM = CI / 4;                                         // fp32: MFMAs per pixel tile = floats per lane and tile
    static_assert(ESZ != 4 || (CI > 0 && CI % 8 == 0 && (CI + 31) / 32 == NK), "fp32 path: compile-time Cin, a multiple of 8");
    constexpr int KC = ESZ == 4 ? (NM + 3) / 4 : NK;                   // 16-byte registers per pixel tile and lane
    constexpr bool SWZ = (CG & (CG - 1)) == 0 && CG >= 8;
    constexpr int NSLOT = NT / CG;
    static_assert(CB % 16 == 0, "channel chunk = whole MFMA row tiles");
    __shared__ v4f tile[IH * IW * CG];
    __shared__ v4f wl[K * K * CG];
    __shared__ v4f red[(CG & (CG - 1)) == 0 ? NW * CG : NSLOT * CG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const BlkId bid = blk_image_major(tiles_sp * (C / CB));
    const int n = bid.n;
    const int t = bid.x % tiles_sp, chunk = bid.x / tiles_sp;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW, c0 = chunk * CB;
    const int iy0 = ty0 * S - pad_lo, ix0 = tx0 * S - pad_lo;
#ifdef MB_TRACE
    int mtp = 0;
#endif
    MB_TP(0);

    // activations of pixel tile `mt` for this lane: pixel p = mt * 16 + j, K-step ks: the 8 channels ks*32 + 8q ..
    // (channels >= Cin: the address falls back to channel 0 - the weight planes are zero there)
    const XT* xb = X + (size_t)n * H * H * Cin;
    // INS: the halo tile's rectangle inside the image, in tile coordinates: rows [ry0, ry0 + RH), cols [rx0, rx0 + RW)
    const int ry0 = iy0 < 0 ? -iy0 : 0, rx0 = ix0 < 0 ? -ix0 : 0;
    const int RH = (H - iy0 < IH ? H - iy0 : IH) - ry0, RW = (H - ix0 < IW ? H - ix0 : IW) - rx0;
    const int npin = RH * RW;
    const int nmt = INS ? (npin + 15) >> 4 : NMT;           // pixel tiles that hold work (block-uniform)
    const float inv_rw = 1.0f / (float)RW;
    // pixel index of the enumeration -> (row, col) in tile coordinates
    auto tile_rc = [&](int pidx, int& r, int& c) {
        if constexpr (INS) {
            const int pc = pidx < npin ? pidx : npin - 1;
            const int rr = (int)(((float)pc + 0.5f) * inv_rw);           // exact: pc < 2^12, RW <= 64
            r = ry0 + rr;
            c = rx0 + pc - rr * RW;
        } else {
            const int pc = pidx < NPX ? pidx : NPX - 1;
            r = pc / IW;
            c = pc - r * IW;
        }
    };
    struct XTile { v4f raw[KC]; };
    auto load_tile = [&](int mt, XTile& xt) {
        int r, c;
        tile_rc(mt * 16 + j, r, c);
        const int iy = iy0 + r, ix = ix0 + c;
        const bool inside = INS || ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H);
        const XT* px = xb + ((size_t)(inside ? iy : 0) * H + (inside ? ix : 0)) * Cin;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            if constexpr (ESZ == 4) {
                const float* pf = reinterpret_cast<const float*>(px) + q * NM + 4 * kk;
                if constexpr (NM % 4 == 0) xt.raw[kk] = ldg4u(pf);
                else if (kk < KC - 1) xt.raw[kk] = ldg4u(pf);
                else { const v2f t = ldg2(pf); xt.raw[kk] = (v4f){t.x, t.y, 0.f, 0.f}; }      // NM % 4 == 2
            } else {
                const int k = kk * 32 + 8 * q;
                xt.raw[kk] = *reinterpret_cast<const v4f*>(px + (k < Cin ? k : 0));          // 8 bf16
            }
        }
    };
    // Activation tiles are requested RD tiles ahead (register ring, the tile loop is unrolled by RD): with one tile
    // ahead every iteration ended in an exposed L2 round trip - the ablation runs put the expand phase of block 2
    // at 155 us of a 232 us launch with MFMAs and swish together accounting for 44 of them.
    constexpr int RDMAX = MINB >= 4 && NT == 256 ? 3 : 64 / (4 * KC);       // tighter register budget: a shorter ring
    constexpr int RD = NIT < RDMAX ? NIT : RDMAX;
    XTile ring[RD];
    // Request order = wait order.  The block's constants (expand rows as MFMA A operands: lane = channel row j,
    // k-group q; folded-BN biases; depthwise taps and bias) go FIRST, the activation ring after them: vmcnt retires in
    // order, so "the constants and the ring's oldest tile have landed" is the same count (3 * (RD - 1) loads still in
    // flight) on the loop's entry edge and on its back edge.  Requested after the ring (rounds 1-2) the entry edge
    // needed vmcnt(1), hipcc put the stricter count at the loop header, and every iteration drained the whole ring;
    // the depthwise taps went global -> LDS in the prologue, a vmcnt(0) before the first MFMA (6k of a block's 52k
    // cycles in the s_memtime trace) - they now wait in a register until the expand loop is done.
    constexpr int WREG = ESZ == 4 ? KC : NK * 3;             // 16-byte weight registers per 16-channel tile
    static_assert(K * K * CG <= NT, "one depthwise weight vector per thread");
    v4f wfr[NTB][WREG];
    v4f bex[NTB];
    const int wli = tid < K * K * CG ? tid : 0;
    const v4f wl_v = ldg4(Wt + (size_t)(wli / CG) * C + c0 + 4 * (wli % CG));
    const v4f bv = ldg4(bias + c0 + 4 * (tid % CG));
    const unsigned short* wrow = We3 + (size_t)(c0 + j) * Kp + 8 * q;
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
        bex[nt] = ldg4(be + c0 + nt * 16 + 4 * q);
        if constexpr (ESZ == 4) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const float* pw = Wef + (size_t)(c0 + nt * 16 + j) * CI + q * NM + 4 * kk;
                if constexpr (NM % 4 == 0) wfr[nt][kk] = ldg4u(pw);
                else if (kk < KC - 1) wfr[nt][kk] = ldg4u(pw);
                else { const v2f t = ldg2(pw); wfr[nt][kk] = (v4f){t.x, t.y, 0.f, 0.f}; }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wfr[nt][ks * 3 + pl] = *reinterpret_cast<const v4f*>(wrow + (size_t)pl * plane + (size_t)nt * 16 * Kp + ks * 32);
        }
    }
#pragma unroll
    for (int d = 0; d < RD; ++d) load_tile(wave + NW * d < nmt ? wave + NW * d : nmt - 1, ring[d]);
    if constexpr (INS) {
        // the padding border of the tile: written, not computed (disjoint from the units the expand writes below)
        for (int i = tid; i < NPX * CG; i += NT) {
            const int pp = i / CG, r = pp / IW, c = pp - r * IW;
            if ((unsigned)(r - ry0) >= (unsigned)RH || (unsigned)(c - rx0) >= (unsigned)RW)
                tile[tile_unit<CG, SWZ>(pp, i - pp * CG)] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    MB_TP(1);

    // SKEW (off): the MFMAs of tile t are issued, then the epilogue of tile t - 1 (swish + LDS store) while they sit in the
    // matrix pipe.  Measured again in round 3, after the wait-count fixes: block 2 186.6 vs 185.4 us, block 4 119.3 vs
    // 117.6, bf16 the same - within noise in both directions; four waves per SIMD already fill each other's gaps.
    auto epilogue = [&](int mt, const v4f (&acc)[NTB]) {
        int pr, pcol;
        tile_rc(mt * 16 + j, pr, pcol);
        const int p = pr * IW + pcol;                               // unit of the LDS tile
        const int iy = iy0 + pr, ix = ix0 + pcol;
        const bool inside = INS || ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H);
        if (mt < nmt && mt * 16 + j < (INS ? npin : NPX)) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
                tile[tile_unit<CG, SWZ>(p, nt * 4 + q)] = inside ? (ABL == 2 ? acc[nt] : swish4(acc[nt])) : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    v4f prev[NTB];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) prev[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
    int prev_mt = nmt;                                              // no tile yet: the first epilogue stores nothing
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += RD) {
#pragma unroll
      for (int d = 0; d < RD; ++d) {
        const int it = it0 + d;
        const int mt = wave + NW * it;
        XTile& xa = ring[d];
        v4f acc[NTB];                                            // starts at the folded-BN bias: no add in the epilogue
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) acc[nt] = bex[nt];
        v4f acc1 = (v4f){0.f, 0.f, 0.f, 0.f};
        // the NTB accumulators interleaved (independent chains)
        if constexpr (ABL == 1) {
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) acc[nt] = xa.raw[0] + wfr[nt][0];
        } else if constexpr (ESZ == 4) {
            if constexpr (NTB == 1) {
                // one 16-channel tile: the K steps alternate between two accumulators (two independent MFMA chains instead
                // of one dependent chain of NM), folded once at the end
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[0][i / 4][i % 4], xa.raw[i / 4][i % 4], acc1, 0, 0, 0);
                    else acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[0][i / 4][i % 4], xa.raw[i / 4][i % 4], acc[0], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NM; ++i)
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[nt][i / 4][i % 4], xa.raw[i / 4][i % 4], acc[nt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)                      // smallest weight terms first
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wfr[nt][ks * 3 + pl]),
                                                                          __builtin_bit_cast(bf8, xa.raw[ks]), acc[nt], 0, 0, 0);
        }
        {   // tile it + RD into the slot whose MFMAs have just issued (past the end: re-requests the last tile, never used)
            const int mn = mt + NW * RD;
            load_tile(mn < nmt ? mn : nmt - 1, ring[d]);
        }
        if constexpr (SKEW) {
            epilogue(prev_mt, prev);                                // the previous tile, while this tile's MFMAs run
            if constexpr (ESZ == 4 && NTB == 1) acc[0] += acc1;
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) prev[nt] = acc[nt];
            prev_mt = mt;
        } else {
            if constexpr (ESZ == 4 && NTB == 1) acc[0] += acc1;
            epilogue(mt, acc);
        }
      }
    }
    if constexpr (SKEW) epilogue(prev_mt, prev);
    if (tid < K * K * CG) wl[tid] = wl_v;
    MB_TP(4);
    __syncthreads();
    MB_TP(5);
    if constexpr (ABL == 3) {
        if (tid < CG) stg4(P + ((size_t)n * tiles_sp + t) * C + c0 + 4 * tid, tile[tid] + bv);
    } else {
        dw_compute<K, S, CB, TH, TW, RP, SWZ, XT, ABL, NT>(tile, wl, red, bv, Y, P, n, Ho, C, c0, ty0, tx0, t, tiles_sp);
        se_tail<NT>(se, P, n, C, tiles_sp * (C / CB), tiles_sp, reinterpret_cast<float*>(tile));
    }
    MB_TP(6);
#ifdef MB_TRACE
    if (H == MB_TRACE_H && S == MB_TRACE_S && bid.x == 5 && bid.n == 3 && threadIdx.x == 0) g_mb_trace[255] = mtp;
#endif
}

// Stem + block-0 depthwise in ONE kernel.  Block = image x 8x16 output tile x all 32 channels.  The
// 21x37x3 input patch (NCHW network input) is staged in LDS with row-contiguous loads, the 3x3 s2 stem
// conv (+BN+swish) of the 10x18 halo tile is computed from it straight into the depthwise LDS tile
// (zero outside the 112x112 stem output = the depthwise padding), then dw_compute runs as usual.
// The 112x112x32 stem activation (1.6 MB per crop, read back with a 1.4x halo) never touches HBM.
template <typename XT>
__global__ __launch_bounds__(256, 3) void stem_dw_kernel(const float* __restrict__ x, const unsigned short* __restrict__ ws3,
                                                      int plane, int Kp,
                                                      const float* __restrict__ bs, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, XT* __restrict__ Y,
                                                      float* __restrict__ P, XT* __restrict__ stem_out,
                                                      int tiles_x, int tiles_sp, SeTail se) {
    constexpr int K = 3, S = 1, CB = 32, TH = 8, TW = 16, RP = 4, CG = 8;
    constexpr int IH = TH + 2, IW = TW + 2;                 // 10 x 18 stem pixels
    constexpr int PH = (IH - 1) * 2 + 3, PW = (IW - 1) * 2 + 3, PWP = PW + 1;   // 21 x 37 input patch
    constexpr int NPX = IH * IW, NMT = (NPX + 15) / 16, NIT = (NMT + 3) / 4;    // 180 pixels = 12 MFMA column tiles
    __shared__ v4f tile[IH * IW * CG];
    __shared__ v4f wl[K * K * CG];
    __shared__ v4f red[4 * CG];
    __shared__ float patch[3 * PH * PWP];
    const BlkId bid = blk_image_major(tiles_sp);
    const int tid = threadIdx.x, n = bid.n, t = bid.x;
    const int lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
#ifdef MB_TRACE
    int mtp = 0;
    const int H = 224;
#endif
    MB_TP(0);
    // The 3x3x3 -> 32 stem convolution on the bf16 MFMA, operands split exactly as in gemm_split: A = the two
    // 16-channel row tiles of the weights [32][27 -> 32] (three bf16 planes, requested here, with the patch), B = for
    // pixel j the 8 patch values k = 8q .. 8q + 7 (k = (ky * 3 + kx) * 3 + ci) gathered from the LDS patch and split
    // in registers.  As 27 FMAs per pixel and channel quad this loop was the bulk of a kernel whose VALU was busy
    // 89 % of the time (profiles/r02_sq_counters.md); as 12 MFMAs per 16 pixels the matrix pipe does it.
    bf8 wfr[2][3];
    {
        const unsigned short* wrow = ws3 + (size_t)j * Kp + 8 * q;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wfr[nt][pl] = *reinterpret_cast<const bf8*>(wrow + (size_t)pl * plane + (size_t)nt * 16 * Kp);
    }
    const int wi = tid < K * K * CG ? tid : 0;
    const v4f wl_v = ldg4(Wt + (size_t)(wi / CG) * 32 + 4 * (wi % CG));
    const v4f bv = ldg4(bias + 4 * (tid % CG));
    const v4f bs0 = ldg4(bs + 4 * q), bs1 = ldg4(bs + 16 + 4 * q);
    // stem pixel (sy, sx) = (ty0 - 1 + py, tx0 - 1 + px) reads input rows 2*sy .. 2*sy+2 (TF-SAME: pad high only)
    const int r0 = 2 * (ty0 - 1), c0 = 2 * (tx0 - 1);
    const float* xb = x + (size_t)n * 3 * 224 * 224;
    // input patch -> LDS: unconditional clamped loads, all in flight before the first store (see dw_kernel)
    constexpr int NPE = 3 * PH * PW, NPL = (NPE + 255) / 256;
    float pv[NPL];
    bool pok[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int i = tid + k * 256 < NPE ? tid + k * 256 : NPE - 1;
        const int ci = i / (PH * PW), r = (i / PW) % PH, c = i % PW;
        const int iy = r0 + r, ix = c0 + c;
        const bool inside = (unsigned)iy < 224u && (unsigned)ix < 224u;
        pv[k] = xb[(size_t)ci * 224 * 224 + (inside ? iy : 0) * 224 + (inside ? ix : 0)];
        pok[k] = inside;
    }
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int i = tid + k * 256 < NPE ? tid + k * 256 : NPE - 1;
        const int ci = i / (PH * PW), r = (i / PW) % PH, c = i % PW;
        patch[(ci * PH + r) * PWP + c] = pok[k] ? pv[k] : 0.f;
    }
    if (tid < K * K * CG) wl[tid] = wl_v;
    MB_TP(1);
    __syncthreads();
    MB_TP(2);
    // patch offsets of this lane's 8 k values (relative to the pixel's top-left input sample); k >= 27: offset 0, the
    // weight planes are zero there
    int koff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * q + e, kc = k < 27 ? k : 0;
        const int tap = kc / 3, ci = kc - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
        koff[e] = (ci * PH + ky) * PWP + kx;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int mt = wave + 4 * it;                       // wave-uniform
        if (mt < NMT) {
            const int p = mt * 16 + j, pc = p < NPX ? p : NPX - 1;
            const int py = pc / IW, px = pc - py * IW;
            const float* pp = &patch[(2 * py) * PWP + 2 * px];
            v4f lo, hi;
            lo.x = pp[koff[0]]; lo.y = pp[koff[1]]; lo.z = pp[koff[2]]; lo.w = pp[koff[3]];
            hi.x = pp[koff[4]]; hi.y = pp[koff[5]]; hi.z = pp[koff[6]]; hi.w = pp[koff[7]];
            bf8 x0, x1, x2;
            split8(lo, hi, x0, x1, x2);
            v4f acc[2] = {(v4f){0.f, 0.f, 0.f, 0.f}, (v4f){0.f, 0.f, 0.f, 0.f}};
            const bf8* xs[3] = {&x0, &x1, &x2};
            const int wsel[6] = {2, 1, 0, 1, 0, 0}, xsel[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first
#pragma unroll
            for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[nt][wsel[p6]], *xs[xsel[p6]], acc[nt], 0, 0, 0);
            const int sy = ty0 - 1 + py, sx = tx0 - 1 + px;
            const bool inside = (unsigned)sy < 112u && (unsigned)sx < 112u;
            if (p < NPX) {
                const v4f v0 = inside ? swish4(acc[0] + bs0) : (v4f){0.f, 0.f, 0.f, 0.f};
                const v4f v1 = inside ? swish4(acc[1] + bs1) : (v4f){0.f, 0.f, 0.f, 0.f};
                tile[p * CG + q] = v0;
                tile[p * CG + 4 + q] = v1;
                // optional copy of the stem activation (parity taps only): interior pixels of this tile
                if (stem_out && inside && py >= 1 && py <= TH && px >= 1 && px <= TW) {
                    XT* so = stem_out + (((size_t)n * 112 + sy) * 112 + sx) * 32;
                    st4(so + 4 * q, v0);
                    st4(so + 16 + 4 * q, v1);
                }
            }
        }
    }
    MB_TP(4);
    __syncthreads();
    MB_TP(5);
    dw_compute<K, S, CB, TH, TW, RP, false, XT>(tile, wl, red, bv, Y, P, n, 112, 32, 0, ty0, tx0, t, tiles_sp);
    se_tail<256>(se, P, n, 32, tiles_sp, tiles_sp, reinterpret_cast<float*>(tile));
    MB_TP(6);
#ifdef MB_TRACE
    if (bid.x == 5 && bid.n == 3 && threadIdx.x == 0) g_mb_trace[255] = mtp;
#endif
}

template <typename XT>
void launch_stem_dw(const float* x, const unsigned short* ws3, int plane, int Kp, const float* bs, const float* Wd,
                    const float* bd, XT* Y, float* P, XT* stem_out, int n, int* tiles, hipStream_t s, const SeTail& se) {
    const int tx = 112 / 16, ty = 112 / 8;
    *tiles = tx * ty;
    hipLaunchKernelGGL(stem_dw_kernel<XT>, dim3(tx * ty * n), dim3(256), 0, s, x, ws3, plane, Kp, bs, Wd, bd, Y, P, stem_out, tx,
                       tx * ty, se);
}
template void launch_stem_dw<float>(const float*, const unsigned short*, int, int, const float*, const float*, const float*, float*, float*, float*, int, int*, hipStream_t, const SeTail&);
template void launch_stem_dw<bf16_t>(const float*, const unsigned short*, int, int, const float*, const float*, const float*, bf16_t*, float*, bf16_t*, int, int*, hipStream_t, const SeTail&);

template <int K, int S, int CB, int TH, int TW, int RP, typename XT>
static void dw_launch(const XT* X, const float* W, const float* b, XT* Y, float* P, int n,
                      int H, int C, int pad_lo, int* tiles, hipStream_t s, const SeTail& se) {
    const int Ho = (H + S - 1) / S;
    const int tx = (Ho + TW - 1) / TW, ty = (Ho + TH - 1) / TH;
    const int tiles_sp = tx * ty;
    *tiles = tiles_sp;
    hipLaunchKernelGGL((dw_kernel<K, S, CB, TH, TW, RP, XT>), dim3(tiles_sp * (C / CB) * n), dim3(256), 0,
                       s, X, W, b, Y, P, H, Ho, C, pad_lo, tx, tiles_sp, se);
}

template <int K, int S, int CB, int TH, int TW, int RP, int KC, int NSUB, typename XT, int CI>
static void mb_launch(const XT* X, int Cin, const float* We, const float* be, const float* W, const float* b,
                      XT* Y, float* P, int n, int H, int C, int pad_lo, int* tiles, hipStream_t s, const SeTail& se) {
    const int Ho = (H + S - 1) / S;
    const int tx = (Ho + TW - 1) / TW, ty = (Ho + TH - 1) / TH;
    const int tiles_sp = tx * ty;
    *tiles = tiles_sp;
    hipLaunchKernelGGL((mbconv_kernel<K, S, CB, TH, TW, RP, KC, NSUB, XT, CI>), dim3(tiles_sp * (C / (CB * NSUB)) * n),
                       dim3(256), 0, s, X, We, be, W, b, Y, P, H, Ho, C, Cin, pad_lo, tx, tiles_sp, se);
}


// ---- 7 x 7 layers (blocks 12-15): a row per thread, no LDS tile ------------------------------------------------
// The LDS-tile kernel runs these layers with RP = 1 (a 7 x 7 tile has no strips to give 256 threads): every output
// costs K*K tile reads + K*K weight reads from LDS - 50 ds_read_b128 per output quad at k = 5, 2.9 GB of LDS traffic
// per 256 crops = the whole launch (39.7 us) at LDS peak.  Here a thread owns one output ROW of CPT channels (CPT = 4:
// 16 B of fp32; 8: 16 B of bf16, so a half-wave still moves 512 B per load): per kernel row it loads the 7 input
// pixels and the K weight vectors (wave-contiguous 16-byte loads, all independent; the 5 threads that share an input
// row sit in the same block and meet in L1 / L2) and keeps the 7 accumulators in registers.  Block = 8 rows (7 used)
// x 32 channel groups = 256 threads; grid = images x (C / (32 * CPT)), image-major on one XCD like the others.
template <int K, int CPT, typename XT>
__global__ __launch_bounds__(256) void dw_rows7_kernel(const XT* __restrict__ X, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, XT* __restrict__ Y,
                                                       float* __restrict__ P, int C, int groups, SeTail se) {
    constexpr int H = 7, PAD = (K - 1) / 2, NV = CPT / 4;
    __shared__ v4f red[8 * 32 * NV > 320 ? 8 * 32 * NV : 320];             // >= 1280 floats: the squeeze-excite tail's scratch
    const BlkId bid = blk_image_major(groups);
    const int tid = threadIdx.x, row = tid >> 5, cl = tid & 31;
    const int c = (bid.x * 32 + cl) * CPT;
    const bool live = row < H && c < C;
    const int cc = c < C ? c : 0, r = row < H ? row : 0;
    const XT* xb = X + (size_t)bid.n * H * H * C + cc;
    v4f acc[H][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const v4f bv = ldg4(bias + cc + 4 * v);
#pragma unroll
        for (int ox = 0; ox < H; ++ox) acc[ox][v] = bv;
    }
    // kernel rows rolled two at a time at most: fully unrolled, hipcc hoists all K * (7 + K) loads to the top (256
    // VGPRs at k = 5: one wave per SIMD); the loads of a row pair are independent and issue back to back
    constexpr int KY_UNROLL = K == 3 && CPT == 4 ? 3 : (CPT == 4 ? 2 : 1);
#pragma unroll KY_UNROLL
    for (int ky = 0; ky < K; ++ky) {
        const int iy = r + ky - PAD;
        const bool ok = (unsigned)iy < (unsigned)H;
        const XT* xr = xb + (size_t)(ok ? iy : 0) * H * C;
        v4f in[H][NV], w[K][NV];
#pragma unroll
        for (int ix = 0; ix < H; ++ix)
#pragma unroll
            for (int v = 0; v < NV; ++v) in[ix][v] = ld4(xr + (size_t)ix * C + 4 * v);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const v4f wv = ldg4(Wt + (size_t)(ky * K + kx) * C + cc + 4 * v);
                w[kx][v] = ok ? wv : (v4f){0.f, 0.f, 0.f, 0.f};         // rows outside the image: TF-SAME zero padding
            }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int ox = 0; ox < H; ++ox) {
                const int ix = ox + kx - PAD;                            // compile-time after unrolling
                if (ix >= 0 && ix < H)
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[ox][v] += in[ix][v] * w[kx][v];
            }
    }
    v4f psum[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) psum[v] = (v4f){0.f, 0.f, 0.f, 0.f};
    XT* yb = Y + ((size_t)bid.n * H * H + (size_t)r * H) * C + cc;
#pragma unroll
    for (int ox = 0; ox < H; ++ox)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const v4f o = swish4(acc[ox][v]);
            if (live) st4(yb + (size_t)ox * C + 4 * v, o);
            psum[v] += o;
        }
    // squeeze-excite pool: the 7 row sums of a channel group folded in row order (a fixed order)
#pragma unroll
    for (int v = 0; v < NV; ++v) red[(row * 32 + cl) * NV + v] = psum[v];
    __syncthreads();
    if (tid < 32 && c < C) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            v4f t = red[tid * NV + v];
#pragma unroll
            for (int rr = 1; rr < H; ++rr) t += red[(rr * 32 + tid) * NV + v];
            stg4_agent(P + (size_t)bid.n * C + c + 4 * v, t);
        }
    }
    __syncthreads();                                                   // red[] is the tail's scratch from here on
    se_tail<256>(se, P, bid.n, C, groups, 1, reinterpret_cast<float*>(red));
}

template <int K, typename XT>
static void dw_rows7_launch(const XT* X, const float* W, const float* b, XT* Y, float* P, int n, int C, int* tiles, hipStream_t s,
                            const SeTail& se) {
    *tiles = 1;
    if constexpr (sizeof(XT) == 2) {
        static const int cpt = getenv("DFD_ROWS7_CPT") ? atoi(getenv("DFD_ROWS7_CPT")) : (K == 5 ? 4 : 8);
        if (cpt == 8) {
            const int groups = (C + 255) / 256;
            hipLaunchKernelGGL((dw_rows7_kernel<K, 8, XT>), dim3(groups * n), dim3(256), 0, s, X, W, b, Y, P, C, groups, se);
            return;
        }
    }
    const int groups = (C + 127) / 128;
    hipLaunchKernelGGL((dw_rows7_kernel<K, 4, XT>), dim3(groups * n), dim3(256), 0, s, X, W, b, Y, P, C, groups, se);
}

// tile shapes per B0 depthwise layer class: (k, stride, H_in, C) -> <K,S,CB,TH,TW,RP>
#define DFD_DW_TABLE(OP)                                  \
    OP(3, 1, 112, 32, 32, 8, 16, 4)   /* block 0      */  \
    OP(3, 2, 112, 96, 32, 8, 8, 2)    /* block 1      */  \
    OP(3, 1, 56, 144, 16, 8, 14, 2)   /* block 2      */  \
    OP(5, 2, 56, 144, 16, 7, 14, 2)   /* block 3      */  \
    OP(5, 1, 28, 240, 16, 14, 14, 2)  /* block 4      */  \
    OP(3, 2, 28, 240, 16, 7, 14, 2)   /* block 5      */  \
    OP(3, 1, 14, 480, 32, 14, 14, 7)  /* blocks 6,7   */  \
    OP(5, 1, 14, 480, 32, 14, 14, 7)  /* block 8      */  \
    OP(5, 1, 14, 672, 32, 14, 14, 7)  /* blocks 9,10  */  \
    OP(5, 2, 14, 672, 32, 7, 7, 1)    /* block 11     */  \
    OP(5, 1, 7, 1152, 32, 7, 7, 1)    /* blocks 12-14 */  \
    OP(3, 1, 7, 1152, 32, 7, 7, 1)    /* block 15     */

template <typename XT>
bool launch_depthwise(const XT* X, const float* W, const float* bias, XT* Y, float* P, int n,
                      int H, int C, int k, int stride, int pad_lo, int* tiles, hipStream_t s, const SeTail& se) {
#define DFD_DW_DISPATCH(KK, SS, HH, CC, CB, TH, TW, RP)                                \
    if (k == KK && stride == SS && H == HH && C == CC) {                               \
        dw_launch<KK, SS, CB, TH, TW, RP, XT>(X, W, bias, Y, P, n, H, C, pad_lo, tiles, s, se); \
        return true;                                                                   \
    }
    // 7 x 7 stride-1 layers: the row-per-thread kernel (DFD_DW_ROWS7=0: the LDS-tile kernel, for A/B runs)
    static const bool rows7 = !(getenv("DFD_DW_ROWS7") && atoi(getenv("DFD_DW_ROWS7")) == 0);
    if (rows7 && H == 7 && stride == 1 && C % 8 == 0 && (k == 3 || k == 5) && pad_lo == (k - 1) / 2) {
        if (k == 3) dw_rows7_launch<3, XT>(X, W, bias, Y, P, n, C, tiles, s, se);
        else dw_rows7_launch<5, XT>(X, W, bias, Y, P, n, C, tiles, s, se);
        return true;
    }
    DFD_DW_TABLE(DFD_DW_DISPATCH)
#undef DFD_DW_DISPATCH
    return false;
}
template bool launch_depthwise<float>(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int*, hipStream_t, const SeTail&);
template bool launch_depthwise<bf16_t>(const bf16_t*, const float*, const float*, bf16_t*, float*, int, int, int, int, int, int, int*, hipStream_t, const SeTail&);

// ---- MBConv front half of the 14 x 14 and 7 x 7 blocks (6-15) in ONE launch (round 3) ---------------------------------
// As separate launches these blocks spent 523 us in the expand GEMMs and 411 us in the depthwise kernels per 256 crops:
// the expanded tensor (C_exp = 480-1152 channels) went to memory and came straight back, and the depthwise kernels,
// two blocks per CU, sat in the exposed latency of their tile loads.  Here a block owns 196 pixels - one 14 x 14 image
// or four 7 x 7 images, contiguous NHWC rows - and CB = 32 expanded channels:
//   phase 1  E[32][196] = swish(We[32][Cin] * X[196][Cin]^T + b) on v_mfma_f32_16x16x32_bf16, operands as in gemm_split
//            (weights = the handle's three bf16 planes, an fp32 activation split into three exact bf16 terms in
//            registers: six products, fp32-exact; a bf16 activation is the operand as loaded: three products).
//            K-steps outermost: a wave keeps the accumulators of its (up to) four pixel tiles live, holds the weight
//            fragments of ONE K-step (6 registers of 16 bytes) and has the next K-step's fragments and activations in
//            flight while the current one is split and multiplied.  Channels are MFMA rows, pixels MFMA columns, so a
//            lane ends with four consecutive channels of one pixel = one 16-byte unit of the LDS tile.
//   phase 2  the depthwise conv + folded BN + swish + squeeze-excite pool from that tile: 14 x 14 through dw_compute
//            (zero border written next to the interior), 7 x 7 with an output row per thread (the dw_rows7 scheme, fed
//            from LDS).  A block holds whole images, so the pool sums it writes are final (one "tile" per image).
// The activation tile is read once per block (C_exp / 32 blocks share an image through L2: image-major block order on
// one XCD), the weights once per wave.
template <int K, int S, int HW, int CIN, typename XT>
__global__ __launch_bounds__(256, 2) void mbconv_late_kernel(const XT* __restrict__ X,
                                                             const unsigned short* __restrict__ We3, int plane, int Kp,
                                                             const float* __restrict__ be, const float* __restrict__ Wt,
                                                             const float* __restrict__ bias, XT* __restrict__ Y,
                                                             float* __restrict__ P, int n_img, int C, int pad_lo) {
    constexpr int CB = 32, CG = 8, NTB = 2, NPX = 196, NMT = 13, MTW = 4;
    constexpr int NK = (CIN + 31) / 32, ESZ = (int)sizeof(XT), NB = ESZ == 4 ? 2 : 1;
    constexpr int G = HW == 14 ? 1 : 4;                              // images per block
    constexpr int TH = HW == 14 ? 14 / S : 7;                        // depthwise output tile = the whole image
    constexpr int IW = HW == 14 ? (TH - 1) * S + K : 7;              // LDS tile width (7 x 7: no border, rows clip in registers)
    constexpr int SWZ = 2, UP = CG + 1;                              // padded pixel stride (tile_unit)
    constexpr int NPIX = HW == 14 ? IW * IW : NPX, NUNIT = NPIX * UP;
    static_assert(HW == 14 || (HW == 7 && S == 1), "14 x 14 (stride 1 or 2) or 7 x 7 stride 1");
    static_assert(CIN % 8 == 0 && K * K * CG <= 256, "shape");
    __shared__ v4f tile[NUNIT];
    __shared__ v4f wl[K * K * CG];
    __shared__ v4f red[HW == 14 ? 4 * CG : G * 7 * CG];
    // the block's 32 weight rows, three bf16 planes, rows padded by 16 bytes (consecutive rows start 4 banks apart): read
    // from global memory by every wave (rounds of this kernel's first version) they were half of its vector-memory
    // traffic - 98 of 186 KB per block through a 64 B/clk L1 that the s_memtime trace showed 75 % busy in this phase
    constexpr int KROW = NK * 32 + 8;
    __shared__ __attribute__((aligned(16))) unsigned short wlds[3 * CB * KROW];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const BlkId bid = blk_image_major(C / CB);
    // 13 pixel tiles over 4 waves: one wave carries 4, the others 3.  Which one rotates with the block id - wave w of every
    // block sits on SIMD w, and with a fixed assignment SIMD 0 carried the long wave of both resident blocks
    const int wave = ((tid >> 6) + (int)((blockIdx.x >> 3) + (blockIdx.x >> 8))) & 3;   // (first two rounds of an XCD: +0, +1)
    const int grp = bid.n, c0 = bid.x * CB;
    const int rows_valid = HW == 14 ? NPX : ((n_img - grp * G < G ? n_img - grp * G : G) * 49);
    const XT* xb = X + (size_t)grp * NPX * CIN;
#ifdef MB_TRACE
    int mtp = 0;
    const int H = HW;
#endif
    MB_TP(0);

    // request order = wait order (see mbconv2_kernel): constants, the weight rows (LDS before the first MFMA), then the
    // activation ring
    const int wli = tid < K * K * CG ? tid : 0;
    const v4f wl_v = ldg4(Wt + (size_t)(wli / CG) * C + c0 + 4 * (wli % CG));
    const v4f bv = ldg4(bias + c0 + 4 * (tid % CG));
    v4f bex[NTB];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) bex[nt] = ldg4(be + c0 + nt * 16 + 4 * q);
    unsigned xoff[MTW];                                               // element offset of this lane's pixel row, per pixel tile
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int p = (wave + 4 * i) * 16 + j;
        xoff[i] = (unsigned)(p < rows_valid ? p : 0) * CIN;
    }
    struct Stage { v4f b[MTW][NB]; };
    auto load_stage = [&](int ks, Stage& st) {
        const int k = ks * 32 + 8 * q, kk = k < CIN ? k : 0;          // channels >= Cin: any address, the planes are zero there
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            if constexpr (ESZ == 4) {
                const float* px = reinterpret_cast<const float*>(xb) + xoff[i] + kk;
                st.b[i][0] = ldg4(px);
                st.b[i][NB - 1] = ldg4(px + 4);
            } else {
                st.b[i][0] = *reinterpret_cast<const v4f*>(xb + xoff[i] + kk);   // 8 bf16
            }
        }
    };
    // activation fragments are requested RD K-steps ahead: with one step ahead every K-step lasted exactly one L2 round
    // trip (s_memtime: 2.2k cycles per step for 0.8k cycles of MFMAs)
    constexpr int CPR = NK * 4, NCH = 3 * CB * CPR, NLD = (NCH + 255) / 256;      // weight rows: 16-byte chunks per row / in all
    v4f wv[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ch = tid + u * 256 < NCH ? tid + u * 256 : NCH - 1;
        const int row = ch / CPR, kc = ch - row * CPR, pl = row / CB, r = row - pl * CB;
        wv[u] = *reinterpret_cast<const v4f*>(We3 + (size_t)pl * plane + (size_t)(c0 + r) * Kp + kc * 8);
    }
    constexpr int RD = NK < 3 ? NK : 3;
    Stage st[RD];
#pragma unroll
    for (int d = 0; d < RD; ++d) load_stage(d, st[d]);
    if constexpr (HW == 14) {
        // the zero border of the TF-SAME padding: written, not computed (disjoint from the units phase 1 writes)
        for (int i = tid; i < NPIX * CG; i += 256) {
            const int pp = i / CG, r = pp / IW - pad_lo, c = pp % IW - pad_lo;
            if ((unsigned)r >= 14u || (unsigned)c >= 14u) tile[tile_unit<CG, SWZ>(pp, i % CG)] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ch = tid + u * 256 < NCH ? tid + u * 256 : NCH - 1;
        const int row = ch / CPR, kc = ch - row * CPR;
        *reinterpret_cast<v4f*>(&wlds[row * KROW + kc * 8]) = wv[u];
    }
    __syncthreads();                                                  // weight rows in LDS
    MB_TP(1);
    v4f acc[MTW][NTB];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) acc[i][nt] = bex[nt];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const Stage& cur = st[ks % RD];
        v4f wa[NTB][3];
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wa[nt][pl] = *reinterpret_cast<const v4f*>(&wlds[((pl * CB) + nt * 16 + j) * KROW + ks * 32 + 8 * q]);
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            if (wave + 4 * i < NMT) {                                 // wave-uniform
                if constexpr (ESZ == 4) {
                    bf8 x0, x1, x2;
                    split8(cur.b[i][0], cur.b[i][NB - 1], x0, x1, x2);
                    const bf8* xs[3] = {&x0, &x1, &x2};
                    const int wsel[6] = {2, 1, 0, 1, 0, 0}, xsel[6] = {0, 1, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
                    for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wa[nt][wsel[p6]]),
                                                                                *xs[xsel[p6]], acc[i][nt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wa[nt][pl]),
                                                                                __builtin_bit_cast(bf8, cur.b[i][0]), acc[i][nt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + RD < NK) load_stage(ks + RD, st[ks % RD]);           // refill the slot whose MFMAs have just issued
        __builtin_amdgcn_sched_barrier(0);
        MB_TP(10 + ks);
    }
    // swish -> LDS tile
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int p = (wave + 4 * i) * 16 + j;
        if (p < NPX) {
            int unit_p;
            if constexpr (HW == 14) {
                const int y = p / 14, x = p - y * 14;
                unit_p = (y + pad_lo) * IW + x + pad_lo;
            } else {
                unit_p = p;
            }
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) tile[tile_unit<CG, SWZ>(unit_p, nt * 4 + q)] = swish4(acc[i][nt]);
        }
    }
    if (tid < K * K * CG) wl[tid] = wl_v;
    MB_TP(4);
    __syncthreads();
    MB_TP(5);
    if constexpr (HW == 14) {
        constexpr int RP = S == 1 ? 7 : 1;
        dw_compute<K, S, CB, TH, TH, RP, SWZ, XT, 0, 256>(tile, wl, red, bv, Y, P, grp, TH, C, c0, 0, 0, 0, 1);
    } else {
        // an output row (7 pixels x 4 channels) per thread: 4 images x 7 rows x 8 channel quads = 224 threads
        constexpr int H = 7, PAD = (K - 1) / 2;
        const int cg = tid & 7, rr = (tid >> 3) % H, sub = (tid >> 3) / H;
        const bool live = tid < G * H * CG;
        const int img = grp * G + sub;
        v4f psum = (v4f){0.f, 0.f, 0.f, 0.f};
        if (live) {
            v4f o[H];
#pragma unroll
            for (int ox = 0; ox < H; ++ox) o[ox] = bv;
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                const int iy = rr + ky - PAD;
                const bool ok = (unsigned)iy < (unsigned)H;
                const int pb = sub * 49 + (ok ? iy : 0) * H;
                v4f in[H], w[K];
#pragma unroll
                for (int ix = 0; ix < H; ++ix) in[ix] = tile[tile_unit<CG, SWZ>(pb + ix, cg)];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const v4f wv = wl[(ky * K + kx) * CG + cg];
                    w[kx] = ok ? wv : (v4f){0.f, 0.f, 0.f, 0.f};      // rows outside the image: TF-SAME zero padding
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int ox = 0; ox < H; ++ox) {
                        const int ix = ox + kx - PAD;                 // compile-time after unrolling
                        if (ix >= 0 && ix < H) o[ox] += in[ix] * w[kx];
                    }
            }
            XT* yb = Y + ((size_t)img * H * H + (size_t)rr * H) * C + c0 + 4 * cg;
#pragma unroll
            for (int ox = 0; ox < H; ++ox) {
                const v4f v = swish4(o[ox]);
                if (img < n_img) st4(yb + (size_t)ox * C, v);
                psum += v;
            }
            red[(sub * H + rr) * CG + cg] = psum;
        }
        __syncthreads();
        if (tid < G * CG) {                                            // the 7 row sums of a channel quad folded in row order
            const int s2 = tid >> 3, c2 = tid & 7;
            v4f t = red[(s2 * H) * CG + c2];
#pragma unroll
            for (int r2 = 1; r2 < H; ++r2) t += red[(s2 * H + r2) * CG + c2];
            if (grp * G + s2 < n_img) stg4(P + (size_t)(grp * G + s2) * C + c0 + 4 * c2, t);
        }
    }
    MB_TP(6);
#ifdef MB_TRACE
    if (H == MB_TRACE_H && S == MB_TRACE_S && bid.x == 5 && bid.n == 3 && threadIdx.x == 0) g_mb_trace[255] = mtp;
#endif
}

template <int K, int S, int HW, int CIN, typename XT>
static void mb_late_launch(const XT* X, const unsigned short* We3, int plane, int Kp, const float* be, const float* W, const float* b,
                           XT* Y, float* P, int n, int C, int pad_lo, int* tiles, hipStream_t s) {
    *tiles = 1;
    const int groups = HW == 14 ? n : (n + 3) / 4;
    hipLaunchKernelGGL((mbconv_late_kernel<K, S, HW, CIN, XT>), dim3(groups * (C / 32)), dim3(256), 0, s, X, We3, plane, Kp, be, W, b,
                       Y, P, n, C, pad_lo);
}

// (k, stride, H, C, Cin) of blocks 6-10 and 12-15 (block 11, 14 -> 7 at stride 2, measured 115 us in this form against
// 110 us as GEMM + depthwise: left out)
#define DFD_MB_LATE_TABLE(OP)  \
    OP(3, 1, 14, 480, 80)      \
    OP(5, 1, 14, 480, 80)      \
    OP(5, 1, 14, 672, 112)     \
    OP(5, 1, 7, 1152, 192)     \
    OP(3, 1, 7, 1152, 192)

// expand (1x1 + BN + swish) fused into the depthwise kernel; only the five large-spatial MBConv
// blocks (1..5) are instantiated: there the expanded tensor dominates HBM traffic and C_in <= 48.
// (k, stride, H, C, Cin) -> <K,S,CB,TH,TW,RP, NK = ceil(Cin/32)>; several variants per block, the first is the
// default, DFD_MB_VARIANT_<H>_<stride>=i selects another (kernel experiments; the tile counts follow).
#define DFD_MB2_TABLE(OP)                                  \
    OP(0, 3, 2, 112, 96, 16, 32, 8, 8, 2, 1)               \
    OP(1, 3, 2, 112, 96, 16, 16, 8, 8, 2, 1)               \
    OP(0, 3, 1, 56, 144, 24, 16, 14, 28, 7, 1)             \
    OP(1, 3, 1, 56, 144, 24, 16, 14, 28, 2, 1)             \
    OP(0, 5, 2, 56, 144, 24, 16, 7, 14, 7, 1)              \
    OP(1, 5, 2, 56, 144, 24, 16, 7, 14, 2, 1)              \
    OP(0, 5, 1, 28, 240, 40, 16, 28, 28, 7, 2)             \
    OP(1, 5, 1, 28, 240, 40, 16, 14, 28, 4, 2)             \
    OP(0, 3, 2, 28, 240, 40, 16, 7, 14, 2, 2)              \
    OP(1, 3, 2, 28, 240, 40, 16, 7, 14, 7, 2)

// round 3 variants: + NT (threads per block), INS (pixel tiles enumerate only in-image pixels)
#define DFD_MB3_TABLE(OP)                                          \
    OP(6, 3, 1, 56, 144, 24, 16, 14, 28, 7, 1, 256, true)          \
    OP(7, 3, 1, 56, 144, 24, 16, 28, 28, 7, 1, 512, true)          \
    OP(8, 3, 1, 56, 144, 24, 16, 14, 28, 7, 1, 512, true)          \
    OP(10, 3, 1, 56, 144, 24, 16, 28, 28, 7, 1, 512, false)        \
    OP(6, 5, 2, 56, 144, 24, 16, 7, 14, 7, 1, 256, true)           \
    OP(7, 5, 2, 56, 144, 24, 16, 14, 14, 7, 1, 512, true)          \
    OP(8, 5, 2, 56, 144, 24, 16, 14, 14, 2, 1, 512, true)          \
    OP(6, 5, 1, 28, 240, 40, 16, 28, 28, 7, 2, 256, true)          \
    OP(7, 5, 1, 28, 240, 40, 16, 28, 28, 7, 2, 512, false)         \
    OP(8, 5, 1, 28, 240, 40, 16, 28, 28, 7, 2, 512, true)          \
    OP(9, 5, 1, 28, 240, 40, 16, 14, 28, 7, 2, 256, true)          \
    OP(10, 5, 1, 28, 240, 40, 16, 14, 28, 7, 2, 512, true)         \
    OP(11, 5, 1, 28, 240, 40, 16, 14, 28, 7, 2, 512, false)        \
    OP(12, 5, 1, 28, 240, 40, 16, 14, 28, 7, 2, 256, false)        \
    OP(6, 3, 2, 28, 240, 40, 16, 7, 14, 2, 2, 256, true)           \
    OP(7, 3, 2, 28, 240, 40, 16, 14, 14, 2, 2, 512, true)          \
    OP(8, 3, 2, 28, 240, 40, 16, 14, 14, 7, 2, 512, true)          \
    OP(9, 3, 2, 28, 240, 40, 16, 14, 14, 7, 2, 256, true)

// first-generation instances still used with fp32 activations where they measure faster (blocks 1, 3, 5:
// 248 / 183 / 85 us against 329 / 185 / 89 us of the second generation at batch 256; blocks 2 and 4 run the second
// generation: 214 / 148 us against 263 / 164 us).  With bf16 activations the second generation wins everywhere
// (229 / 128 / 124 / 99 / 53 us against 263 / 179 / 152 / 151 / 62 us).
// (k, stride, H, C, Cin) -> <K,S,CB,TH,TW,RP, KC = ceil(Cin/16), NSUB = channel chunks per block>
#define DFD_MB1_TABLE(OP)                               \
    OP(-1, 3, 2, 112, 96, 16, 32, 8, 8, 2, 1, 3)        \
    OP(-1, 5, 2, 56, 144, 24, 16, 7, 14, 2, 2, 3)       \
    OP(-3, 5, 2, 56, 144, 24, 16, 7, 14, 2, 2, 9)       \
    OP(-1, 3, 2, 28, 240, 40, 16, 7, 14, 2, 3, 1)       \
    OP(-3, 3, 2, 28, 240, 40, 16, 7, 14, 2, 3, 3)       \
    OP(-4, 3, 2, 28, 240, 40, 16, 7, 14, 2, 3, 5)       \
    OP(-5, 3, 2, 28, 240, 40, 16, 7, 14, 7, 3, 1)       \
    OP(-1, 5, 1, 28, 240, 40, 16, 14, 28, 7, 3, 3)      \
    OP(-3, 5, 1, 28, 240, 40, 16, 14, 28, 7, 3, 5)      \
    OP(-4, 5, 1, 28, 240, 40, 16, 7, 28, 7, 3, 5)       \
    OP(-5, 5, 1, 28, 240, 40, 16, 14, 14, 7, 3, 5)      \
    OP(-1, 3, 1, 56, 144, 24, 16, 14, 28, 7, 2, 3)      \
    OP(-3, 3, 1, 56, 144, 24, 16, 14, 28, 7, 2, 9)      \
    OP(-4, 3, 1, 56, 144, 24, 16, 14, 14, 7, 2, 9)

// ablation builds of the default tiles of blocks 2 and 4 (variant 20 + ABL): where does the time go?
//   1 no MFMAs, 2 no swish on the expanded tile, 3 no depthwise phase, 4 no swish after the depthwise conv, 5 no stores
#define DFD_MB2_ABL(OP) OP(1) OP(2) OP(3) OP(4) OP(5)

// DFD_MB_VARIANT_<H>_<stride>=i: kernel experiments (profiles/mb_variants.py); -1 = first generation where built
static int mb_variant(int H, int stride) {
    char name[48];
    snprintf(name, sizeof name, "DFD_MB_VARIANT_%d_%d", H, stride);
    const char* e = getenv(name);
    return e ? atoi(e) : -2;
}

template <int K, int S, int CB, int TH, int TW, int RP, int NK, typename XT, int NT = 256, bool INS = false, int CI = 0,
          int MINB = (NT == 256 ? 2 : 4), bool SKEW = false>
static void mb2_launch(const XT* X, int Cin, const unsigned short* We3, int plane, int Kp, const float* Wef, const float* be,
                       const float* W, const float* b, XT* Y, float* P, int n, int H, int C, int pad_lo, int* tiles, hipStream_t s,
                       const SeTail& se) {
    const int Ho = (H + S - 1) / S;
    const int tx = (Ho + TW - 1) / TW, ty = (Ho + TH - 1) / TH;
    const int tiles_sp = tx * ty;
    *tiles = tiles_sp;
    hipLaunchKernelGGL((mbconv2_kernel<K, S, CB, TH, TW, RP, NK, XT, 0, NT, INS, CI, MINB, SKEW>), dim3(tiles_sp * (C / CB) * n), dim3(NT), 0, s, X,
                       We3, plane, Kp, Wef, be, W, b, Y, P, H, Ho, C, Cin, pad_lo, tx, tiles_sp, se);
}

template <typename XT>
bool launch_mbconv_front(const XT* Xin, int Cin, const unsigned short* We3, int plane, int Kp, const float* Wef, const float* be,
                         const float* Wd, const float* bd, XT* Y, float* P, int n, int H, int C, int k, int stride,
                         int pad_lo, int* tiles, hipStream_t s, const SeTail& se, bool late) {
    if (H <= 14) {
        // option "fuse_late"; the squeeze-excite tail counts blocks per image, a late block holds up to four: not combined
        if (!late || se.gate) return false;
#define DFD_MB_LATE_DISPATCH(KK, SS, HH, CC, CI)                                                                     \
    if (k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                                               \
        mb_late_launch<KK, SS, HH, CI, XT>(Xin, We3, plane, Kp, be, Wd, bd, Y, P, n, C, pad_lo, tiles, s);          \
        return true;                                                                                                \
    }
        DFD_MB_LATE_TABLE(DFD_MB_LATE_DISPATCH)
#undef DFD_MB_LATE_DISPATCH
        return false;
    }
    int var = mb_variant(H, stride);
    if (var == -2) {
        // defaults by measurement at batch 256 (profiles/mb_variants.py, round 3, us fp32 / bf16):
        //   block 1 (112, s2): first generation 237 (second: 261) / variant 0 203
        //   block 2 (56, s1):  variant 6 (INS) 190 (variant 0: 203) / variant 0 127
        //   block 3 (56, s2):  first generation 158 (second: 178-254) / variant 0 117
        //   block 4 (28, s1):  variant 13 (14 x 28 tiles, four blocks per CU: 100 VGPRs, ring of 3) 115 (three blocks: 121;
        //                      whole image: 132) / variant 7 (whole image, 512 threads) 83 (variant 0: 92)
        //   block 5 (28, s2):  first generation, 3 channel chunks per block 69 (1 chunk: 78; second generation: 93) / variant 0 51
        if (sizeof(XT) == 4) var = stride == 2 ? (H == 28 ? -3 : -1) : (H == 56 ? 6 : 13);
        else var = (H == 28 && stride == 1) ? 7 : 0;
    }
    if constexpr (sizeof(XT) == 4) {
        if (var <= -1) {
#define DFD_MB1_DISPATCH(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, KC, NSUB)                                            \
    if (var == VV && k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                                  \
        mb_launch<KK, SS, CB, TH, TW, RP, KC, NSUB, XT, CI>(Xin, Cin, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);  \
        return true;                                                                                                \
    }
            DFD_MB1_TABLE(DFD_MB1_DISPATCH)
#undef DFD_MB1_DISPATCH
        }
    }
    if (var < 0) var = 0;
#define DFD_MB2_DISPATCH(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, NK)                                                 \
    if (var == VV && k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                                  \
        mb2_launch<KK, SS, CB, TH, TW, RP, NK, XT, 256, false, CI>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se); \
        return true;                                                                                                \
    }
    DFD_MB2_TABLE(DFD_MB2_DISPATCH)
#undef DFD_MB2_DISPATCH
#define DFD_MB3_DISPATCH(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, NK, NT, INS)                                       \
    if (var == VV && k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                                  \
        mb2_launch<KK, SS, CB, TH, TW, RP, NK, XT, NT, INS, CI>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se); \
        return true;                                                                                                \
    }
    DFD_MB3_TABLE(DFD_MB3_DISPATCH)
#undef DFD_MB3_DISPATCH
    // four blocks per CU (128 VGPRs, ring of 3): more waves in different phases on a SIMD
    if (var == 13 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {
        mb2_launch<5, 1, 16, 14, 28, 7, 2, XT, 256, false, 40, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 18 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {             // variant 13 with the one-tile skew
        mb2_launch<5, 1, 16, 14, 28, 7, 2, XT, 256, false, 40, 4, true>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 18 && k == 3 && stride == 1 && H == 56 && C == 144 && Cin == 24) {             // variant 6 with the one-tile skew
        mb2_launch<3, 1, 16, 14, 28, 7, 1, XT, 256, true, 24, 2, true>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 15 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {
        mb2_launch<5, 1, 16, 14, 28, 7, 2, XT, 256, true, 40, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 16 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {
        mb2_launch<5, 1, 16, 14, 28, 14, 2, XT, 256, false, 40, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 17 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {
        mb2_launch<5, 1, 16, 14, 28, 4, 2, XT, 256, false, 40, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 14 && k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40) {
        mb2_launch<5, 1, 16, 7, 28, 7, 2, XT, 256, false, 40, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
    if (var == 13 && k == 3 && stride == 1 && H == 56 && C == 144 && Cin == 24) {
        mb2_launch<3, 1, 16, 14, 28, 7, 1, XT, 256, true, 24, 4>(Xin, Cin, We3, plane, Kp, Wef, be, Wd, bd, Y, P, n, H, C, pad_lo, tiles, s, se);
        return true;
    }
#ifdef DFD_MB_ABLATION
#define DFD_ABL_CASE(A)                                                                                              \
    if (var == 20 + A && H == 56 && stride == 1) {                                                                  \
        const int tx = 2, ty = 4;                                                                                   \
        *tiles = tx * ty;                                                                                           \
        hipLaunchKernelGGL((mbconv2_kernel<3, 1, 16, 14, 28, 7, 1, XT, A, 256, false, 24>), dim3(tx * ty * (C / 16) * n), dim3(256), 0, s, Xin, \
                           We3, plane, Kp, Wef, be, Wd, bd, Y, P, H, 56, C, Cin, pad_lo, tx, tx * ty, se);          \
        return true;                                                                                                \
    }                                                                                                               \
    if (var == 20 + A && H == 28 && stride == 1) {                                                                  \
        *tiles = 1;                                                                                                 \
        hipLaunchKernelGGL((mbconv2_kernel<5, 1, 16, 28, 28, 7, 2, XT, A, 256, false, 40>), dim3((C / 16) * n), dim3(256), 0, s, Xin,   \
                           We3, plane, Kp, Wef, be, Wd, bd, Y, P, H, 28, C, Cin, pad_lo, 1, 1, se);                 \
        return true;                                                                                                \
    }
    DFD_MB2_ABL(DFD_ABL_CASE)
#undef DFD_ABL_CASE
#endif
    return false;
}
template bool launch_mbconv_front<float>(const float*, int, const unsigned short*, int, int, const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int*, hipStream_t, const SeTail&, bool);
template bool launch_mbconv_front<bf16_t>(const bf16_t*, int, const unsigned short*, int, int, const float*, const float*, const float*, const float*, bf16_t*, float*, int, int, int, int, int, int, int*, hipStream_t, const SeTail&, bool);

// SE pool partial-sum tiles of a fused launch (the workspace is sized for the largest count over all variants)
int mbconv_tiles(int H, int C, int k, int stride, int Cin, bool late) {
    const int Ho = (H + stride - 1) / stride;
    int best = -1;
#define DFD_MB2_TILES(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, NK)                    \
    if (k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                \
        const int tl = ((Ho + TW - 1) / TW) * ((Ho + TH - 1) / TH);                  \
        if (tl > best) best = tl;                                                    \
    }
    DFD_MB2_TABLE(DFD_MB2_TILES)
#define DFD_MB3_TILES(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, NK, NT, INS) DFD_MB2_TILES(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, NK)
    DFD_MB3_TABLE(DFD_MB3_TILES)
#undef DFD_MB3_TILES
#undef DFD_MB2_TILES
#define DFD_MB1_TILES(VV, KK, SS, HH, CC, CI, CB, TH, TW, RP, KC, NSUB)              \
    if (k == KK && stride == SS && H == HH && C == CC && Cin == CI) {                \
        const int tl = ((Ho + TW - 1) / TW) * ((Ho + TH - 1) / TH);                  \
        if (tl > best) best = tl;                                                    \
    }
    DFD_MB1_TABLE(DFD_MB1_TILES)
#undef DFD_MB1_TILES
    if (k == 5 && stride == 1 && H == 28 && C == 240 && Cin == 40 && best < 4) best = 4;      // variant 14: 7 x 28 tiles
#define DFD_MB_LATE_TILES(KK, SS, HH, CC, CI) \
    if (late && k == KK && stride == SS && H == HH && C == CC && Cin == CI && best < 1) best = 1;
    DFD_MB_LATE_TABLE(DFD_MB_LATE_TILES)
#undef DFD_MB_LATE_TILES
    return best;
}

int depthwise_tiles(int H, int C, int k, int stride) {
    const int Ho = (H + stride - 1) / stride;
#define DFD_DW_TILES(KK, SS, HH, CC, CB, TH, TW, RP) \
    if (k == KK && stride == SS && H == HH && C == CC) return ((Ho + TW - 1) / TW) * ((Ho + TH - 1) / TH);
    DFD_DW_TABLE(DFD_DW_TILES)
#undef DFD_DW_TILES
    return -1;
}

// ---------------------------------------------------------------------- squeeze-excite
// One 1024-thread block per image.  mean[c] = sum over tiles of P / (H*W); z = swish(W1 mean
// + b1); gate = sigmoid(W2 z + b2)  (efficientnet_pytorch MBConvBlock SE branch).  The work is
// tiny (<= 2*1152*48 MACs) and purely latency-bound, so every loop has a compile-time trip
// count (C <= 18*64, c_se <= 16*3) and its loads are issued back to back.
constexpr int SE_MAXC = 1152, SE_MAXSE = 48;
__global__ __launch_bounds__(1024) void se_kernel(const float* __restrict__ P, int tiles, float inv_hw,
                                                  const float* __restrict__ w1,
                                                  const float* __restrict__ b1,
                                                  const float* __restrict__ w2t,
                                                  const float* __restrict__ b2,
                                                  float* __restrict__ gate, int C, int c_se) {
    __shared__ float mean[SE_MAXC];
    __shared__ float z[SE_MAXSE];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = P + (size_t)n * tiles * C;
    if (C <= 512 && tiles > 4) {
        // few channels, many tiles (blocks 0-3: 32-144 channels x 98-8 tiles): the tile sum is split over
        // 1024 / C thread groups (group g takes tiles g, g + parts, ...), then folded in group order - a fixed
        // order for a given layer, so results stay run-to-run and batch invariant.  One thread per channel
        // walking all tiles was 25 dependent L2 round trips for block 0 (20 us for a 32-channel mean).
        __shared__ float part_sum[1024];
        const int parts = 1024 / C;
        const int c = tid % C, g = tid / C;
        if (g < parts) {
            float sacc = 0.f;
            for (int t = g; t < tiles; t += parts) sacc += p[(size_t)t * C + c];
            part_sum[g * C + c] = sacc;
        }
        __syncthreads();
        if (tid < C) {
            float sacc = 0.f;
            for (int g2 = 0; g2 < parts; ++g2) sacc += part_sum[g2 * C + tid];
            mean[tid] = sacc * inv_hw;
        }
    } else {
        for (int c = tid; c < C; c += 1024) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int t = 0;
            for (; t + 3 < tiles; t += 4) {
                s0 += p[(size_t)t * C + c];
                s1 += p[(size_t)(t + 1) * C + c];
                s2 += p[(size_t)(t + 2) * C + c];
                s3 += p[(size_t)(t + 3) * C + c];
            }
            for (; t < tiles; ++t) s0 += p[(size_t)t * C + c];
            mean[c] = ((s0 + s1) + (s2 + s3)) * inv_hw;
        }
    }
    __syncthreads();
    {   // FC1: wave w owns outputs w, w+16, w+32
        float s[3] = {0.f, 0.f, 0.f};
        // every load is unconditional (indices clamped, the VALUE is masked): a load under a
        // runtime condition makes hipcc branch around it and wait per element (guide section 5, trap (c))
#pragma unroll
        for (int i = 0; i < SE_MAXC / 64; ++i) {
            const int c = lane + 64 * i;
            const int cc = c < C ? c : 0;
            const float mv = c < C ? mean[cc] : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int o = wave + 16 * k;
                const int oo = o < c_se ? o : c_se - 1;
                s[k] += mv * w1[(size_t)oo * C + cc];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = s[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int o = wave + 16 * k;
            if (lane == 0 && o < c_se) z[o] = swish1(v + b1[o]);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        float s = b2[c];
        float wv[SE_MAXSE];                     // all 48 loads issued before the first use
#pragma unroll
        for (int o = 0; o < SE_MAXSE; ++o) wv[o] = w2t[(size_t)(o < c_se ? o : 0) * C + c];
#pragma unroll
        for (int o = 0; o < SE_MAXSE; ++o) s += (o < c_se ? z[o < c_se ? o : 0] : 0.f) * wv[o];
        gate[(size_t)n * C + c] = sigmoid1(s);
    }
}

void launch_se(const float* P, int tiles, float inv_hw, const float* w1, const float* b1,
               const float* w2t, const float* b2, float* gate, int n, int C, int c_se, hipStream_t s) {
    hipLaunchKernelGGL(se_kernel, dim3(n), dim3(1024), 0, s, P, tiles, inv_hw, w1, b1, w2t, b2, gate, C, c_se);
}

// ------------------------------------------------------------------- global average pool
// lane = (row part p = lane >> 3, channel quad lane & 7): part p sums rows p, p + 8, ...; the eight partial sums
// are folded with a butterfly over the lane bits 3-5 (a fixed order).  One thread per channel quad walking all 49
// rows left the chip with 1.25 waves per SIMD and read the 64 MB head activation at 2 TB/s.
template <typename XT>
__global__ __launch_bounds__(256) void avgpool_kernel(const XT* __restrict__ X,
                                                      float* __restrict__ Y, int n_img, int hw, int C) {
    const int c4 = C / 4;
    const long long gid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;          // wave index
    const int lane = threadIdx.x & 63, part = lane >> 3;
    const long long item = gid * 8 + (lane & 7);                                      // (image, channel quad)
    const bool live = item < (long long)n_img * c4;
    const long long it = live ? item : 0;
    const int n = (int)(it / c4), c = (int)(it % c4) * 4;
    const XT* p = X + (size_t)n * hw * C + c;
    v4f s = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < hw; i += 8) s += ld4(p + (size_t)i * C);
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        s.x += __shfl_xor(s.x, off);
        s.y += __shfl_xor(s.y, off);
        s.z += __shfl_xor(s.z, off);
        s.w += __shfl_xor(s.w, off);
    }
    if (live && part == 0) stg4(Y + (size_t)n * C + c, s * (1.0f / (float)hw));
}

template <typename XT>
void launch_avgpool(const XT* X, float* Y, int n, int hw, int C, hipStream_t s) {
    const long long waves = ((long long)n * (C / 4) + 7) / 8;
    hipLaunchKernelGGL(avgpool_kernel<XT>, dim3((int)((waves * 64 + 255) / 256)), dim3(256), 0, s, X, Y, n, hw, C);
}
template void launch_avgpool<float>(const float*, float*, int, int, int, hipStream_t);
template void launch_avgpool<bf16_t>(const bf16_t*, float*, int, int, int, hipStream_t);

// bf16 activation buffer -> fp32 (parity taps only)
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}
void launch_bf16_to_f32(const bf16_t* x, float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
}

}  // namespace dfd
