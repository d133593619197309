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
#pragma once
#include "b0_kernels.h"
#include "kernel_util.h"

#include <type_traits>

namespace dfd {


#ifdef S6_TRACE
// cycle trace of one wave (build with EXTRA=-DS6_TRACE; profiles/micro/s6_trace.py reads it): s_memtime at fixed
// points of the first stages of block S6_TRACE_BLOCK of the launch with K == S6_TRACE_K and N == S6_TRACE_N
#ifndef S6_TRACE_K
#define S6_TRACE_K 1152
#endif
#ifndef S6_TRACE_N
#define S6_TRACE_N 192
#endif
__device__ long long g_s6_trace[1024];
#define S6_TP(id)                                                                                          \
    do {                                                                                                   \
        if (K == S6_TRACE_K && N == S6_TRACE_N && blockIdx.x == 8 && threadIdx.x == 0 && tp < 1000) {      \
            g_s6_trace[tp++] = (long long)(id);                                                            \
            g_s6_trace[tp++] = (long long)__builtin_amdgcn_s_memtime();                                    \
        }                                                                                                  \
    } while (0)
#else
#define S6_TP(id) do { } while (0)
#endif

#ifndef S6_ROW_STORES
#define S6_ROW_STORES 0                   // 1: pw6 / pw7 epilogues store in row order through LDS (s6_epilogue_rows) - measured, no gain (below)
#endif
constexpr int S6_BK = 32;                 // K of one MFMA = one K-step
constexpr int S6_KPAD = 64;               // weight planes are zero-padded in K to two K-steps (pw6 with KS = 2)
#ifndef S6_XD
#define S6_XD 3                           // pw6: K-steps of activation prefetch in flight (1 or 3)
#endif
#ifndef S6_WD
#define S6_WD 2                           // K-steps of weight prefetch in flight (<= S6_XD in pw6, 1 or 2 in pw7)
#endif
// rows of a zero-padded weight plane: the last n-block of any tile (block width <= 192) stays inside it
__host__ __device__ constexpr int s6_np(int N) { return ((N + 15) / 16 + 11) * 16; }
constexpr int S6_ROWB = 3 * 64;           // bytes per weight row per stage: 3 planes x 32 bf16 = twelve 16-byte chunks
// LDS image of a row: chunk c (= plane * 4 + k-octet) sits at chunk position (c + 6 * ((row >> 2) & 1)) % 12.
// ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) with bank = dword % 64
// (MI355X_MICROARCH.md, LDS table); with 192-byte rows this rotation gives every lane of a group its own four
// banks for all three plane reads (checked exhaustively; the unrotated image is 2-way conflicted: 42 % extra LDS
// cycles measured).  No padding, so two buffers of the widest block are 48 KB: three blocks per CU.
__host__ __device__ constexpr int s6_chunk_pos(int row, int c) { return (c + 6 * ((row >> 2) & 1)) % 12; }
// bf16 activations (8 consecutive k as loaded: one 16-byte register quad) -> MFMA operand; with GATE the
// squeeze-excite gate (fp32) is multiplied in and the product rounded to bf16 (round to nearest even)
template <bool GATE>
__device__ __forceinline__ bf8 bf16x8_gate(const v4f raw, const v4f g0, const v4f g1) {
    bf8 x = __builtin_bit_cast(bf8, raw);
    if constexpr (GATE) {
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (__bf16)((float)x[i] * g[i]);
    }
    return x;
}

// Squeeze-excite gate of the images a block's rows belong to (SeFuse, b0_kernels.h), before the K loop.  se_kernel's
// arithmetic operation by operation: mean = P * inv_hw; FC1 output o by ONE wave (lane l takes channels l, l + 64, ...
// as a chain of fmas, then the xor-shuffle tree 32 .. 1), + bias, swish; FC2 channel c by one thread, the c_se terms in
// order as fmas from the bias; sigmoid.  Which wave or thread takes an output never enters the arithmetic, so every
// tile and both kernels produce the same gate bits, equal to se_kernel's.  zbuf: 4 x 48 floats of LDS that the K loop
// overwrites afterwards.  The closing barrier is a workgroup release / acquire: the block's own gate rows (global, L2)
// are visible to its buffer loads below; other blocks that share an image write the same values.
template <int NTHR>
__device__ __forceinline__ void s6_se_gate(const SeFuse& se, float* __restrict__ gate, int C, int HW, int M, int m_first,
                                           int BM, float* __restrict__ zbuf) {
    // Everything here is latency: the loops have compile-time trip counts and their loads are issued in batches before
    // the first use (clamped index, masked VALUE - a load under a runtime condition is issued alone behind its own wait).
    constexpr int NW = NTHR / 64, NI = SE_FUSE_MAX_IMG, STEPS = 1152 / 64, OG = 6;      // C <= 1152; OG outputs per batch
    constexpr int OPW = (SE_FUSE_MAX_SE + NW - 1) / NW;                               // FC1 outputs per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m_last = m_first + BM - 1 < M - 1 ? m_first + BM - 1 : M - 1;
    const int img0 = m_first / HW, nimg = m_last / HW - img0 + 1;          // <= NI (host: se_fuse_supported, BM <= 128)
    const float* pimg[NI];
#pragma unroll
    for (int im = 0; im < NI; ++im) pimg[im] = se.P + (size_t)(img0 + (im < nimg ? im : nimg - 1)) * C;
    // means of the lane's channels (lane, lane + 64, ...) for every image of the block
    float mean[NI][STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int c = lane + 64 * i, cc = c < C ? c : 0;
#pragma unroll
        for (int im = 0; im < NI; ++im) mean[im][i] = pimg[im][cc];
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
        for (int im = 0; im < NI; ++im) mean[im][i] = lane + 64 * i < C ? mean[im][i] * se.inv_hw : 0.f;
    // FC1: wave w owns outputs w, w + NW, ...; OG of them per batch of loads
#pragma unroll
    for (int g0 = 0; g0 < OPW; g0 += OG) {
        float wv[OG][STEPS];
#pragma unroll
        for (int k = 0; k < OG; ++k) {
            const int o = wave + NW * (g0 + k), oo = o < se.c_se ? o : se.c_se - 1;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int c = lane + 64 * i;
                wv[k][i] = se.w1[(size_t)oo * C + (c < C ? c : 0)];
            }
        }
#pragma unroll
        for (int k = 0; k < OG; ++k) {
            if (g0 + k >= OPW) continue;
            const int o = wave + NW * (g0 + k);
            const float bo = se.b1[o < se.c_se ? o : 0];
#pragma unroll
            for (int im = 0; im < NI; ++im) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < STEPS; ++i) v = __builtin_fmaf(mean[im][i], wv[k][i], v);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0 && o < se.c_se) zbuf[im * SE_FUSE_MAX_SE + o] = swish1(v + bo);
            }
        }
    }
    __syncthreads();
    // FC2: channel c by one thread, all c_se weights of a channel requested before the first use
    constexpr int CPT = (1152 + NTHR - 1) / NTHR;
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int c = tid + NTHR * r, cc = c < C ? c : 0;
        float wv[SE_FUSE_MAX_SE];
#pragma unroll
        for (int o = 0; o < SE_FUSE_MAX_SE; ++o) wv[o] = se.w2t[(size_t)(o < se.c_se ? o : 0) * C + cc];
        const float bc = se.b2[cc];
#pragma unroll
        for (int im = 0; im < NI; ++im) {
            float sacc = bc;
#pragma unroll
            for (int o = 0; o < SE_FUSE_MAX_SE; ++o) sacc = __builtin_fmaf(o < se.c_se ? zbuf[im * SE_FUSE_MAX_SE + (o < se.c_se ? o : 0)] : 0.f, wv[o], sacc);
            if (im < nimg && c < C) gate[(size_t)(img0 + im) * C + c] = sigmoid1(sacc);
        }
    }
    __syncthreads();
}

// The products of one 16x16x32 block, smallest terms first; ONE definition shared by pw6 and pw7 so that every
// tile of either kernel accumulates each output in the same order (bit-identical results across tiles).
//   fp32 activations (NXS = 3 terms) x 3 weight planes: the six products with i + j <= 2
//   bf16 activations (NXS = 1)       x NP weight planes: w2*x, w1*x, w0*x (NP = 3) or w0*x (NP = 1)
template <int MT, int NXS, int NP, int NT>
__device__ __forceinline__ void s6_products(v4f (&acc)[MT][NT], const bf8 (&xs)[MT][NXS], const bf8 (&f)[NP], int nt) {
    if constexpr (NXS == 3) {
        const bf8 w0 = f[0], w1 = f[1], w2 = f[2];
        const bf8* wsel[6] = {&w2, &w1, &w0, &w1, &w0, &w0};
        const int xsel[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*wsel[p6], xs[mt][xsel[p6]], acc[mt][nt], 0, 0, 0);
    } else {
#pragma unroll
        for (int p = NP - 1; p >= 0; --p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[p], xs[mt][0], acc[mt][nt], 0, 0, 0);
    }
}

// epilogue shared by both kernels: the lane holds Y[m[mt]][n .. n+3] for n = nbase + 16 * nt.
// When N is a multiple of 4 (every layer of B0 and of the detector) the bias and residual fragments are requested
// up front, unconditionally (clamped indices): under the per-tile `continue`s below hipcc issued them one by one,
// each behind its own wait - the s_memtime trace showed 9,000 cycles of epilogue for MT x NT = 6 residual loads.
template <int MT, int NT, typename XT>
__device__ __forceinline__ void s6_epilogue(const v4f (&acc)[MT][NT], const int (&m)[MT], int nbase,
                                            const float* __restrict__ bias, const XT* __restrict__ R,
                                            XT* __restrict__ Y, int M, int N, int act, int res_first) {
    if ((N & 3) == 0) {                                  // uniform
        v4f bv[NT], rv[MT][NT], sv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nbase + nt * 16;
            const int nc = n < N ? n : 0;
            bv[nt] = ldg4(bias + nc);
            sv[nt] = act == ACT_PRELU ? ldg4(bias + N + nc) : (v4f){0.f, 0.f, 0.f, 0.f};      // slopes follow the bias
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int mc = m[mt] < M ? m[mt] : M - 1;
                rv[mt][nt] = R ? ld4(R + (size_t)mc * N + nc) : (v4f){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nbase + nt * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                v4f v = acc[mt][nt] + bv[nt];
                if (res_first) v += rv[mt][nt];
                if (act == ACT_SWISH) v = swish4(v);
                else if (act == ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (act == ACT_PRELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * sv[nt][r];
                }
                if (!res_first) v += rv[mt][nt];
                if (n < N && m[mt] < M) st4(Y + (size_t)m[mt] * N + n, v);
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 16;
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
                if (vec) rv = ld4(R + (size_t)m[mt] * N + n);
                else
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) rv[r] = (float)R[(size_t)m[mt] * N + n + r];
            }
            if (res_first) v += rv;
            if (act == ACT_SWISH) v = swish4(v);
            else if (act == ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (act == ACT_PRELU) {
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) v[r] = v[r] >= 0.f ? v[r] : v[r] * bias[N + n + r];
            }
            if (!res_first) v += rv;
            XT* yp = Y + (size_t)m[mt] * N + n;
            if (vec) st4(yp, v);
            else
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) yp[r] = (XT)v[r];
        }
    }
}

// The same epilogue with ROW-ORDER stores (round 4).  An MFMA accumulator leaves lanes 0-15 with 16 different rows, so
// a store instruction of s6_epilogue writes 16 bytes per lane at a row stride; measured on the narrow projections those
// stores cost 2.3-4.1 TB/s-equivalent next to loads that stream at 6 TB/s (pw8_kernel below).  Here a wave turns each of
// its 16-row tiles through `stage` (its own 16 x NT*16 floats of LDS: the block's operand buffers, free after the K
// loop) and every instruction stores - and reads the residual as - runs of NT*64 contiguous bytes per row with
// consecutive lanes on consecutive 16 bytes.  Bias, residual and activation are applied on that side with the
// operations of s6_epilogue in its order: identical bits.  N % 4 == 0 only (callers keep s6_epilogue otherwise).
// MEASURED (batch 256, all tile / oracle tests green): the projections of blocks 4-15 and the head within +-1 us, block
// 11's expand (135 MB out) 69 -> 76 us.  In pw6 / pw7 a store instruction already covers 64 contiguous bytes per row and
// the stores sit behind a K loop that is the launch's time; compiled out (S6_ROW_STORES 0).  pw8 keeps its own form of
// it: there the stores are a third of the bytes of a launch that is nothing but a stream (-5 %).
template <int MT, int NT, typename XT>
__device__ __forceinline__ void s6_epilogue_rows(const v4f (&acc)[MT][NT], float* __restrict__ stage, int mrow0, int nbase0,
                                                 int lane, const float* __restrict__ bias, const XT* __restrict__ R,
                                                 XT* __restrict__ Y, int M, int N, int act, int res_first) {
    constexpr int UPR = 4 * NT;                           // 4-channel units per row of the wave's tile
    const int j = lane & 15, q = lane >> 4;
    int row[NT], n[NT];
    v4f bv[NT], sv[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int u = 64 * p + lane;
        row[p] = u / UPR;
        n[p] = nbase0 + 4 * (u - row[p] * UPR);
        const int nc = n[p] < N ? n[p] : 0;
        bv[p] = ldg4(bias + nc);
        sv[p] = act == ACT_PRELU ? ldg4(bias + N + nc) : (v4f){0.f, 0.f, 0.f, 0.f};      // slopes follow the bias
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        v4f rv[NT];
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int m = mrow0 + mt * 16 + row[p], mc = m < M ? m : M - 1, nc = n[p] < N ? n[p] : 0;
            rv[p] = R ? ld4(R + (size_t)mc * N + nc) : (v4f){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) stg4(stage + j * (NT * 16) + nt * 16 + 4 * q, acc[mt][nt]);
        // LDS operations of one wave execute in issue order; the compiler only has to keep the two groups apart
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int m = mrow0 + mt * 16 + row[p];
            v4f v = ldg4(stage + 4 * (64 * p + lane)) + bv[p];
            if (res_first) v += rv[p];
            if (act == ACT_SWISH) v = swish4(v);
            else if (act == ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (act == ACT_PRELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * sv[p][r];
            }
            if (!res_first) v += rv[p];
            if (n[p] < N && m < M) st4(Y + (size_t)m * N + n[p], v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// KS = K-steps (MFMA K = 32 each) per LDS stage and barrier: 1, or 2 for half as many handoffs per K
// PIPE (KS = 1 only): the split of the NEXT K-step's activations is issued between the MFMAs of the current one
// (sched_group_barrier pattern 1 MFMA : 2 VALU) instead of in front of them.  The s_memtime trace of the plain
// pipeline (profiles/micro/s6_trace.py) shows a wave spending load issue + split VALU + MFMAs back to back
// (400 + 500 + 1150 cycles per two K-steps at MT = 1, NT = 6): with 1.5 waves per SIMD nothing else fills the
// MFMA pipe while a wave converts.
// NW = waves per block (4 or 8): the block's weight tile is pulled through L2 -> L1 -> LDS once per block and
// K-step, so rows per block (NW * MT * 16) set the L2 read traffic for the weights, (M / rows) * N * K * 6 bytes -
// 260 MB for M = 12544, N = 192, K = 1152 at 64 rows, against 58 MB of activations: every inner-loop variant of
// that layer lands on the same 45-55 us, which is that traffic.  Eight waves share the tile among twice the rows.
// (launch bound: two blocks per CU where the LDS tile allows it - 2 x KS x NT x 3 KB of 160 KB - else one)
template <int NT, bool CONV, int MT, bool GATE, int KS, int NW, typename XT, int NP>
__global__ __launch_bounds__(NW * 64, (NW == 4 && 2 * KS * NT * 16 * S6_ROWB <= 80 * 1024) ? 2 : 1) void pw6_kernel(const XT* __restrict__ X,
                                                     const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gate,
                                                     const XT* __restrict__ R,
                                                     XT* __restrict__ Y, int M, int K, int N,
                                                     int HW, int act, int mblocks, int nblocks,
                                                     ConvGeom cg, int res_first, unsigned xbytes, unsigned gbytes, SeFuse se) {
    constexpr int BK = S6_BK;
    constexpr int BN = NT * 16, BM = NW * MT * 16, NTHR = NW * 64;
    constexpr bool PIPE = false;
    constexpr int ESZ = (int)sizeof(XT);                  // activation element: 4 (fp32, split in registers) or 2 (bf16)
    constexpr int XL = ESZ == 4 ? 2 : 1;                  // 16-byte loads per 8 consecutive k
    constexpr int NXS = ESZ == 4 ? 3 : 1;                 // bf16 terms per activation
    static_assert(NP == 3 || (NP == 1 && ESZ == 2), "weight planes: 3 (fp32-exact), or 1 with bf16 activations");
    constexpr int CHUNKS = BN * 4 * NP * KS;              // 16-byte chunks per stage: row x plane x k-octet
    constexpr int WLOADS = (CHUNKS + NTHR - 1) / NTHR;
    // (the row-order epilogue turns a 16-row tile per wave through this storage: NW x 16 x BN floats)
    constexpr int WS_STAGE = BN * S6_ROWB, WS_MIN = (NW * 16 * BN * 4 + 2 * KS - 1) / (2 * KS);
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][KS][WS_STAGE > WS_MIN ? WS_STAGE : (WS_MIN + 15) / 16 * 16];

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int n0 = nblk * BN;
#ifdef S6_TRACE
    int tp = 0;
#endif
    if constexpr (GATE && !CONV) {
        if (se.P) s6_se_gate<NTHR>(se, const_cast<float*>(gate), K, HW, M, mblk * BM, BM, reinterpret_cast<float*>(&ws[0][0][0]));
    }

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

    // Operands come through buffer loads: a per-lane byte offset fixed for the whole K loop (VGPR) plus the K
    // position as the scalar offset, so a K-step costs no vector address arithmetic and no clamps.
    //
    // weight chunk t of this thread: (row, plane, k-octet) -> fixed global / LDS offsets.  The planes are
    // zero-padded to [Np][Kp], so loads and LDS stores are unconditional and select-free (k >= K meets zero
    // weights, whatever the X load returned); threads past the last chunk repeat the last chunk (same value to
    // the same address).  A store under a branch makes hipcc sink the global load into that branch with a
    // vmcnt(0) behind it: one exposed memory latency per K-step.
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W3), 0, 6 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(X), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(GATE ? (void*)const_cast<float*>(gate) : (void*)const_cast<XT*>(X), 0, GATE ? gbytes : xbytes, 0x00020000);
    // With KS = 2 a row's 4 * KS consecutive octets of one plane are 128 contiguous bytes = one cache line per 8
    // lanes: the 64-byte pieces of a single K-step use half of every line they pull through the L1, and at small
    // tiles the L1 (64 B/clk per CU, 3 * BN + 2 * BM line-cycles per K-step against 0.094 * BM * BN MFMA cycles)
    // is what a K-step waits for.
    int wvo[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * NTHR < CHUNKS ? tid + t * NTHR : CHUNKS - 1;
        const int row = e / (4 * NP * KS), rem = e - row * (4 * NP * KS), pl = rem / (4 * KS), c = rem - pl * (4 * KS);
        wvo[t] = 2 * (pl * plane + (n0 + row) * Kp + 8 * c);
        wlds[t] = (c >> 2) * (BN * S6_ROWB) + row * S6_ROWB + s6_chunk_pos(row, pl * 4 + (c & 3)) * 16;
    }

    // Register rings: the activation stream comes from HBM / Infinity Cache (1-2 us under load, several
    // K-steps of MFMA work at these tile sizes) and is prefetched XD steps ahead; weights and gates are L2
    // hits and stay one step ahead.  The K loop is unrolled by the ring size U = XD + 1, so every ring slot is
    // a fixed register set and nothing is ever copied into place.
    // (ring slots hold a whole stage; the deep rings of MT = 2 or KS = 2 do not fit 256 VGPRs)
    constexpr int XD = MT == 1 && KS == 1 && !PIPE ? S6_XD : 1, U = XD + 1;
    static_assert(U % 2 == 0, "the LDS / gate ping-pong needs an even unroll");
    constexpr int WD = MT == 1 && KS == 1 && !PIPE ? S6_WD : 1;
    static_assert(WD >= 1 && WD <= XD, "weight prefetch distance");
    u4 wr[U][WLOADS];      // weights in flight (a whole stage): slot = stage % U (WD slots live at a time)
    v4f xr[U][KS][MT][XL];
    v4f gr[2][KS][MT][2];  // GATE: raw squeeze-excite gate fragments, multiplied in at use
    bool okr[U][KS][MT];   // CONV: tap inside the image (zero padding applied at use)
    const int nk = (K + BK - 1) / BK, nst = (nk + KS - 1) / KS;      // K-steps, stages
    // X / gate rows are not padded: in the last K-step of a K that is not a multiple of 32, lanes past the row
    // end re-read its last 8 values instead (they meet zero weights)
    int xvo[MT], xvo_last[MT], gvo[MT], gvo_last[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mc = m[mt] < M ? m[mt] : M - 1;
        const int over = (nk - 1) * BK + 8 * q - (K - 8);
        xvo[mt] = ESZ * (mc * K + 8 * q);
        xvo_last[mt] = xvo[mt] - ESZ * (over > 0 ? over : 0);
        gvo[mt] = 4 * ((int)gbase[mt] + 8 * q);
        gvo_last[mt] = gvo[mt] - 4 * (over > 0 ? over : 0);
    }
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int vo, int so) {
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    };
    auto load_w = [&](int st, u4 (&w)[WLOADS]) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[t], 2 * st * KS * BK, 0);
    };
    auto load_g = [&](int kstep, v4f (&g)[MT][2]) {
        if constexpr (GATE) {
            const bool last = kstep == nk - 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int go = last ? gvo_last[mt] : gvo[mt];
                g[mt][0] = ld(rg, go, 4 * kstep * BK);
                g[mt][1] = ld(rg, go + 16, 4 * kstep * BK);
            }
        }
    };
    auto load_x = [&](int kstep, v4f (&x)[MT][XL], bool (&okf)[MT]) {
        const int kc = kstep * BK;
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // wave-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int iy = iy0[mt] + ky * cg.dil, ix = ix0[mt] + kx * cg.dil;
                const bool ok = m[mt] < M && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const int vo = ESZ * ((int)gbase[mt] + ((ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin + 8 * q);
                x[mt][0] = ld(rx, vo, ESZ * ci0);
                if constexpr (XL == 2) x[mt][1] = ld(rx, vo + 16, ESZ * ci0);
                okf[mt] = ok;
            }
        } else {
            const bool last = kstep == nk - 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int vo = last ? xvo_last[mt] : xvo[mt];
                x[mt][0] = ld(rx, vo, ESZ * kc);
                if constexpr (XL == 2) x[mt][1] = ld(rx, vo + 16, ESZ * kc);
                okf[mt] = true;
            }
        }
    };
    auto store_w = [&](const u4 (&w)[WLOADS], unsigned char* dst) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) *reinterpret_cast<u4*>(dst + wlds[t]) = w[t];
    };

    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto split_x = [&](const v4f (&x)[MT][XL], const v4f (&g)[MT][2], const bool (&okf)[MT], bf8 (&xs)[MT][NXS]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (ESZ == 4) {
                v4f lo = x[mt][0], hi = x[mt][XL - 1];
                if constexpr (GATE) { lo *= g[mt][0]; hi *= g[mt][1]; }
                if constexpr (CONV) {
                    if (!okf[mt]) { lo = (v4f){0.f, 0.f, 0.f, 0.f}; hi = lo; }
                }
                split8(lo, hi, xs[mt][0], xs[mt][NXS > 1 ? 1 : 0], xs[mt][NXS > 2 ? 2 : 0]);
            } else {
                // bf16 activations are the MFMA operand as loaded; a squeeze-excite gate is multiplied in in fp32
                // and the product rounded back to bf16 (the gated activation is itself a bf16 activation)
                xs[mt][0] = bf16x8_gate<GATE>(x[mt][0], g[mt][0], g[mt][1]);
                if constexpr (CONV) {
                    if (!okf[mt]) xs[mt][0] = __builtin_bit_cast(bf8, (v4f){0.f, 0.f, 0.f, 0.f});
                }
            }
        }
    };
    auto mfma_tile = [&](const bf8 (&xs)[MT][NXS], const bf8 (&f)[NP], int nt) {
        s6_products<MT, NXS, NP>(acc, xs, f, nt);
    };
    auto compute = [&](const v4f (&x)[MT][XL], const v4f (&g)[MT][2], const bool (&okf)[MT], const unsigned char* wb) {
        bf8 xs[MT][NXS];
        split_x(x, g, okf, xs);
        // The fragments of tile nt + 1 are requested before the MFMAs of tile nt are issued (two register sets,
        // order pinned): reading them right before use, as hipcc schedules it on its own, leaves the LDS latency
        // (~150 cycles) exposed NT times per K-step - as long as the MFMAs themselves at one wave per SIMD.
        bf8 wf[2][NP];
        auto read_w = [&](int nt, bf8 (&f)[NP]) {
            const unsigned char* wp = wb + (nt * 16 + j) * S6_ROWB;       // (nt * 16 + j) >> 2 has the parity of j >> 2
            f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
            if constexpr (NP == 3) {
                f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
                f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
            }
        };
        read_w(0, wf[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + 1 < NT) read_w(nt + 1, wf[(nt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(xs, wf[nt & 1], nt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // One pipelined stage (KS K-steps), ring slot u = stage % U: issue the loads (gate of stage + 1, weights of
    // stage + WD, activations of stage + XD, clamped to the last stage: the repeats are never consumed), compute
    // the stage, then hand the weights of stage + 1 to the other LDS buffer.  Every operand needs more than one
    // K-step to arrive (an L2 hit is ~1 us under this load, a K-step of MFMAs 0.3-0.6 us), hence the rings.
    // A half-stage past the last K-step (odd step count, KS = 2) reads the zero padding of the weight planes
    // (K padded to 64) against re-read activations.  The sched_barriers keep the loads at the top (a whole stage
    // to land) and their first consumers at the bottom; left alone, hipcc sinks the loads to the end of the
    // stage and waits for them at once.
    auto stage = [&](auto uc, int st) {
        constexpr int u = decltype(uc)::value;
        S6_TP(1);
        const int s1 = st + 1 < nst ? st + 1 : nst - 1, sw = st + WD < nst ? st + WD : nst - 1,
                  sd = st + XD < nst ? st + XD : nst - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kg = s1 * KS + ks, kx = sd * KS + ks;
            load_g(kg < nk ? kg : nk - 1, gr[(u + 1) & 1][ks]);
            if (ks == 0) load_w(sw, wr[(u + WD) % U]);
            load_x(kx < nk ? kx : nk - 1, xr[(u + XD) % U][ks], okr[(u + XD) % U][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        S6_TP(2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) compute(xr[u][ks], gr[u & 1][ks], okr[u][ks], ws[u & 1][ks]);
        __builtin_amdgcn_sched_barrier(0);
        S6_TP(3);
        store_w(wr[(u + 1) % U], ws[(u + 1) & 1][0]);
        S6_TP(4);
        __syncthreads();
        S6_TP(5);
    };

    S6_TP(0);
    load_w(0, wr[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) load_g(ks < nk ? ks : nk - 1, gr[0][ks]);
#pragma unroll
    for (int d = 1; d < WD; ++d) load_w(d < nst ? d : nst - 1, wr[d]);
#pragma unroll
    for (int d = 0; d < XD; ++d)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kx = (d < nst ? d : nst - 1) * KS + ks;
            load_x(kx < nk ? kx : nk - 1, xr[d][ks], okr[d][ks]);
        }
    store_w(wr[0], ws[0][0]);
    __syncthreads();
    int kt = 0;
    for (; kt + U <= nst; kt += U) {
        stage(std::integral_constant<int, 0>{}, kt);
        stage(std::integral_constant<int, 1>{}, kt + 1);
        if constexpr (U == 4) {
            stage(std::integral_constant<int, 2>{}, kt + 2);
            stage(std::integral_constant<int, 3>{}, kt + 3);
        }
    }
    // remainder (kt is a multiple of U here, so slot u = stage - kt)
    if (kt < nst) stage(std::integral_constant<int, 0>{}, kt);
    if constexpr (U == 4) {
        if (kt + 1 < nst) stage(std::integral_constant<int, 1>{}, kt + 1);
        if (kt + 2 < nst) stage(std::integral_constant<int, 2>{}, kt + 2);
    }

    S6_TP(8);
    if (S6_ROW_STORES && (N & 3) == 0) {                  // every layer of B0 and of the detectors
        // (the last stage ended with a barrier: no wave still reads the weight buffers)
        float* stage = reinterpret_cast<float*>(&ws[0][0][0]) + wave * (16 * BN);
        s6_epilogue_rows<MT, NT, XT>(acc, stage, mblk * BM + wave * (MT * 16), n0, lane, bias, R, Y, M, N, act, res_first);
    } else {
        s6_epilogue<MT, NT, XT>(acc, m, n0 + 4 * q, bias, R, Y, M, N, act, res_first);
    }
    S6_TP(9);
#ifdef S6_TRACE
    if (K == S6_TRACE_K && N == S6_TRACE_N && blockIdx.x == 8 && threadIdx.x == 0) g_s6_trace[1023] = tp;
#endif
}

// ------------------------------------------------------------------------------------------------------
// pw7_kernel: both operands through LDS.  In pw6 every wave owns its rows and splits its own activation
// fragments: ~45 VALU instructions per 16 rows x 32 k, repeated by every n-block, next to only 6 * NT MFMAs -
// with few rows (batch * 49 or * 196) there are too few waves to hide that.  Here the 4 waves form a WM x WN
// grid over a (WM*MT*16) x (WN*NT*16) block tile: the activation tile is split ONCE per block, cooperatively
// (each thread 8 values of one row per 64 rows), written to LDS as three bf16 planes in the same rotated row
// image as the weights, and every wave reads the MT fragments it needs.  MFMAs per split instruction go up by
// WN * NT / (pw6's NT): the kernel for small M and for wide N.
// The raw activations are prefetched two K-steps ahead (register ring of two, the K loop is unrolled by two),
// the weights one step ahead.
template <int WM, int WN, int MT, int NT, bool CONV, bool GATE, typename XT, int NP>
__global__ __launch_bounds__(256, (2 * (WM * MT + WN * NT) * 16 * S6_ROWB <= 80 * 1024) ? 2 : 1) void pw7_kernel(const XT* __restrict__ X,
                                                     const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gate,
                                                     const XT* __restrict__ R,
                                                     XT* __restrict__ Y, int M, int K, int N,
                                                     int HW, int act, int mblocks, int nblocks,
                                                     ConvGeom cg, int res_first, unsigned xbytes, unsigned gbytes, SeFuse se) {
    static_assert(WM * WN == 4, "four waves per block");
    constexpr int BK = S6_BK;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int ESZ = (int)sizeof(XT), XL = ESZ == 4 ? 2 : 1, NXS = ESZ == 4 ? 3 : 1;      // see pw6
    static_assert(NP == 3 || (NP == 1 && ESZ == 2), "weight planes: 3 (fp32-exact), or 1 with bf16 activations");
    constexpr int WCHUNKS = BN * 4 * NP, WLOADS = (WCHUNKS + 255) / 256;
    constexpr int XUNITS = BM * 4, XL7 = (XUNITS + 255) / 256;      // unit = 8 consecutive k of one row
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][BN * S6_ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char xsm[2][BM * S6_ROWB];

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mblk * BM, n0 = nblk * BN;
    if constexpr (GATE && !CONV) {
        if (se.P) s6_se_gate<256>(se, const_cast<float*>(gate), K, HW, M, m0, BM, reinterpret_cast<float*>(&ws[0][0]));
    }

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W3), 0, 6 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(X), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(GATE ? (void*)const_cast<float*>(gate) : (void*)const_cast<XT*>(X), 0, GATE ? gbytes : xbytes, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int vo, int so) {
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    };
    const int nk = (K + BK - 1) / BK;

    // weight chunks of this thread (as in pw6)
    int wvo[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * 256 < WCHUNKS ? tid + t * 256 : WCHUNKS - 1;
        const int row = e / (4 * NP), rem = e - row * (4 * NP), pl = rem >> 2, c = rem & 3;
        wvo[t] = 2 * (pl * plane + (n0 + row) * Kp + 8 * c);
        wlds[t] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + c) * 16;
    }
    // activation units of this thread: row = unit / 4, k-octet = unit % 4 (4 lanes = 128 contiguous bytes)
    int xvo[XL7], xvo_last[XL7], gvo[XL7], gvo_last[XL7], xlds[XL7][3];
    int ubase[XL7], uy0[XL7], ux0[XL7];
    bool uvalid[XL7];
#pragma unroll
    for (int t = 0; t < XL7; ++t) {
        const int e = tid + t * 256 < XUNITS ? tid + t * 256 : XUNITS - 1;
        const int row = e >> 2, ko = e & 3;
        const int mm = m0 + row;
        const int mc = mm < M ? mm : M - 1;
        uvalid[t] = mm < M;
        const int over = (nk - 1) * BK + 8 * ko - (K - 8);
        if constexpr (CONV) {
            const int img = mc / (cg.Ho * cg.Wo), r = mc - img * (cg.Ho * cg.Wo);
            const int oy = r / cg.Wo, ox = r - oy * cg.Wo;
            ubase[t] = img * cg.H * cg.W * cg.Cin + 8 * ko;
            uy0[t] = oy * cg.stride - cg.pad;
            ux0[t] = ox * cg.stride - cg.pad;
            xvo[t] = xvo_last[t] = gvo[t] = gvo_last[t] = 0;
        } else {
            ubase[t] = uy0[t] = ux0[t] = 0;
            xvo[t] = ESZ * (mc * K + 8 * ko);
            xvo_last[t] = xvo[t] - ESZ * (over > 0 ? over : 0);
            gvo[t] = GATE ? 4 * ((mc / HW) * K + 8 * ko) : 0;
            gvo_last[t] = gvo[t] - 4 * (over > 0 ? over : 0);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xlds[t][pl] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + ko) * 16;
    }

    u4 wr[2][WLOADS];      // weights, ring of two K-steps
    v4f xr[2][XL7][XL];    // raw activations, ring of two K-steps
    v4f gr[2][XL7][2];      // GATE: raw squeeze-excite gate
    bool okr[2][XL7];       // CONV: tap inside the image
    auto load_w = [&](int kstep, u4 (&w)[WLOADS]) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[t], 2 * kstep * BK, 0);
    };
    auto load_x = [&](int kstep, v4f (&x)[XL7][XL], v4f (&g)[XL7][2], bool (&okf)[XL7]) {
        const int kc = kstep * BK;
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // block-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int t = 0; t < XL7; ++t) {
                const int iy = uy0[t] + ky * cg.dil, ix = ux0[t] + kx * cg.dil;
                const bool ok = uvalid[t] && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const int vo = ESZ * (ubase[t] + ((ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin);
                x[t][0] = ld(rx, vo, ESZ * ci0);
                if constexpr (XL == 2) x[t][1] = ld(rx, vo + 16, ESZ * ci0);
                okf[t] = ok;
            }
        } else {
            const bool last = kstep == nk - 1;
#pragma unroll
            for (int t = 0; t < XL7; ++t) {
                const int vo = last ? xvo_last[t] : xvo[t];
                x[t][0] = ld(rx, vo, ESZ * kc);
                if constexpr (XL == 2) x[t][1] = ld(rx, vo + 16, ESZ * kc);
                if constexpr (GATE) {
                    const int go = last ? gvo_last[t] : gvo[t];
                    g[t][0] = ld(rg, go, 4 * kc);
                    g[t][1] = ld(rg, go + 16, 4 * kc);
                }
                okf[t] = true;
            }
        }
    };
    auto store_w = [&](const u4 (&w)[WLOADS], int buf) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) *reinterpret_cast<u4*>(&ws[buf][wlds[t]]) = w[t];
    };
    // split this thread's units and write the three planes into the block's activation tile
    auto store_x = [&](const v4f (&x)[XL7][XL], const v4f (&g)[XL7][2], const bool (&okf)[XL7], int buf) {
#pragma unroll
        for (int t = 0; t < XL7; ++t) {
            if constexpr (ESZ == 4) {
                v4f lo = x[t][0], hi = x[t][XL - 1];
                if constexpr (GATE) { lo *= g[t][0]; hi *= g[t][1]; }
                if constexpr (CONV) {
                    if (!okf[t]) { lo = (v4f){0.f, 0.f, 0.f, 0.f}; hi = lo; }
                }
                bf8 s0, s1, s2;
                split8(lo, hi, s0, s1, s2);
                *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][0]]) = s0;
                *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][1]]) = s1;
                *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][2]]) = s2;
            } else {
                bf8 s0 = bf16x8_gate<GATE>(x[t][0], g[t][0], g[t][1]);      // as pw6: plane 0 only
                if constexpr (CONV) {
                    if (!okf[t]) s0 = __builtin_bit_cast(bf8, (v4f){0.f, 0.f, 0.f, 0.f});
                }
                *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][0]]) = s0;
            }
        }
    };

    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto compute = [&](int buf) {
        bf8 xs[MT][NXS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned char* xp = xsm[buf] + ((wm * MT + mt) * 16 + j) * S6_ROWB;
            xs[mt][0] = *reinterpret_cast<const bf8*>(xp + rd0);
            if constexpr (NXS == 3) {
                xs[mt][1] = *reinterpret_cast<const bf8*>(xp + rd1);
                xs[mt][2] = *reinterpret_cast<const bf8*>(xp + rd2);
            }
        }
        bf8 wf[2][NP];                        // next tile's weight fragments in flight during this tile's MFMAs (see pw6)
        auto read_w = [&](int nt, bf8 (&f)[NP]) {
            const unsigned char* wp = ws[buf] + ((wn * NT + nt) * 16 + j) * S6_ROWB;
            f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
            if constexpr (NP == 3) {
                f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
                f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
            }
        };
        read_w(0, wf[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + 1 < NT) read_w(nt + 1, wf[(nt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            // the same products in the same order as pw6: a result never depends on the kernel or tile chosen
            s6_products<MT, NXS, NP>(acc, xs, wf[nt & 1], nt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // step kstep, ring slot u = kstep & 1: loads for step + 2 (into the slots whose contents went to LDS at the
    // bottom of the previous step); MFMAs of this step; then the tiles of step + 1 go to the other LDS buffers
    // (weights as loaded, activations split) from the other ring slot.
    auto step = [&](auto uc, int kstep) {
        constexpr int u = decltype(uc)::value;
        const int k2 = kstep + 2 < nk ? kstep + 2 : nk - 1;
        load_w(k2, wr[u]);
        load_x(k2, xr[u], gr[u], okr[u]);
        __builtin_amdgcn_sched_barrier(0);
        compute(u);
        __builtin_amdgcn_sched_barrier(0);
        store_w(wr[u ^ 1], u ^ 1);
        store_x(xr[u ^ 1], gr[u ^ 1], okr[u ^ 1], u ^ 1);
        __syncthreads();
    };

    load_w(0, wr[0]);
    load_x(0, xr[0], gr[0], okr[0]);
    load_w(nk > 1 ? 1 : 0, wr[1]);
    load_x(nk > 1 ? 1 : 0, xr[1], gr[1], okr[1]);
    store_w(wr[0], 0);
    store_x(xr[0], gr[0], okr[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < nk) step(std::integral_constant<int, 0>{}, kt);

    if (S6_ROW_STORES && (N & 3) == 0) {
        // (the last step ended with a barrier: the operand tiles are free; a wave needs 16 x NT*16 floats, the weight
        // buffers hold 2 x WN*NT*16 rows of 192 bytes)
        float* stage = reinterpret_cast<float*>(&ws[0][0]) + wave * (16 * NT * 16);
        s6_epilogue_rows<MT, NT, XT>(acc, stage, m0 + wm * MT * 16, n0 + wn * NT * 16, lane, bias, R, Y, M, N, act, res_first);
    } else {
        int m[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) m[mt] = m0 + (wm * MT + mt) * 16 + j;
        s6_epilogue<MT, NT, XT>(acc, m, n0 + wn * NT * 16 + 4 * q, bias, R, Y, M, N, act, res_first);
    }
}

// ------------------------------------------------------------------------------------------------------
// pw8_kernel (round 4): the narrow projections of the large maps (blocks 0-3: K = 32 .. 144, N = 16 .. 40, 0.2 - 3.2 M
// rows).  These layers are streams - 385-616 MB in and out per 256 crops against 60 MFMAs per 16 rows - and pw6 runs
// them at 0.45-0.55 of the HBM rate: a block of 64-128 rows pulls the whole weight matrix through L2 -> LDS again,
// crosses a barrier per K-step for it, keeps one K-step of activations in flight per wave and ends.  Here the weight
// matrix (all K-steps, three planes, pw6's rotated row image) and the image's squeeze-excite gate are staged in LDS
// ONCE per block; a wave then walks its 16-row tiles with NO barrier: the whole next tile (every K-step of its rows
// and its residual fragment) is requested before the current tile is split and multiplied, so a wave has 2-9 KB in
// flight all the time.  Products, their order and the epilogue arithmetic are pw6's (s6_products, the operations of
// s6_epilogue): the output bits equal those of every other tile (tests/test_gemm_tiles_gpu.py walks it as a candidate).
// STORES: an MFMA accumulator leaves lanes 0-15 with 16 different rows, so a store instruction of the plain epilogue
// writes 16 B per lane at a row stride - measured, those stores ran at 2.3-4.1 TB/s-equivalent while the loads of the
// same kernel stream at 6 TB/s (no-store build: 92 vs 125 us for K = 144, N = 24).  Here the finished tile goes through a
// per-wave LDS image of the output rows and leaves as whole 1-KB pieces, 16 consecutive bytes per consecutive lane; the
// residual is read the same way and added on that side (the last operation of the epilogue either way).
//   A block takes `tpb` consecutive 16-row tiles of the whole matrix (equal shares: the grid is a small multiple of the
//   resident blocks, not a function of the image size), wave w the tiles w, w + NW, ... of them (neighbouring waves
//   stream neighbouring rows).  GATE: HW % 16 == 0, so a tile never straddles two images and its gate row is
//   wave-uniform; tpb <= tiles per image (tpg), so a block meets at most two images and stages both gate rows.
template <int NT, int NK, bool GATE, int NW, typename XT, int NP>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 1 : NT * NK <= 10 ? 3 : 2)) void pw8_kernel(const XT* __restrict__ X, const unsigned short* __restrict__ W3, int plane,
                                                      int Kp, const float* __restrict__ bias, const float* __restrict__ gate,
                                                      const XT* __restrict__ R, XT* __restrict__ Y, int M, int K, int N,
                                                      int tpg, int tpb, int act, int res_first, unsigned xbytes, SeFuse se) {
    constexpr int BK = S6_BK, BN = NT * 16, NTHR = NW * 64;
    constexpr int ESZ = (int)sizeof(XT), XL = ESZ == 4 ? 2 : 1, NXS = ESZ == 4 ? 3 : 1;
    static_assert(NP == 3 || (NP == 1 && ESZ == 2), "weight planes: 3 (fp32-exact), or 1 with bf16 activations");
    constexpr int CHUNKS = NK * BN * 4 * NP, WLOADS = (CHUNKS + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char ws[NK][BN * S6_ROWB];
    __shared__ __attribute__((aligned(16))) float gs[2][NK * BK];
    __shared__ __attribute__((aligned(16))) float os[NW][16 * BN];           // the wave's output tile (fp32), rows N elements apart

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int ttot = (M + 15) >> 4;
    const int t0 = blockIdx.x * tpb, t1 = t0 + tpb < ttot ? t0 + tpb : ttot; // this block's tiles
    const int img0 = GATE ? t0 / tpg : 0;                                    // first image the block meets
    const int tsplit = GATE ? (img0 + 1) * tpg : 1 << 30;                    // first tile of the block's second image
    const int rows_end = M;
    const int cnt = t0 + wave < t1 ? (t1 - t0 - wave + NW - 1) / NW : 0;     // tiles of this wave

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(X), 0, xbytes, 0x00020000);
    auto ld = [&](int vo, int so) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, so, 0)); };
    // in the last K-step of a K that is not a multiple of 32, lanes past the row end re-read its last 8 values (zero weights)
    const int over = (NK - 1) * BK + 8 * q - (K - 8);
    const int koff = ESZ * 8 * q, koff_last = koff - ESZ * (over > 0 ? over : 0);
    const int nq = 4 * q;
    auto tile_row = [&](int i) {                                             // row of this lane in the wave's i-th tile (clamped)
        const int ii = i < cnt ? i : (cnt > 0 ? cnt - 1 : 0);
        return (t0 + wave + ii * NW) * 16 + j;
    };
    // output side: piece p of a tile = its 4-channel units 64 p + lane (row-major over 16 rows x N / 4 units)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)((size_t)M * N * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(R ? R : X), 0, R ? (unsigned)((size_t)M * N * ESZ) : 0u, 0x00020000);
    auto ld_unit = [&](const __amdgpu_buffer_rsrc_t& rs, int vo, int so) {
        if constexpr (ESZ == 4) return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
        else return bf4_to_f4(__builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0)));
    };
    v4f xr[2][NK][XL], rr[2][NT];
    auto load_tile = [&](int i, v4f (&x)[NK][XL], v4f (&r)[NT]) {
        const int m = tile_row(i), mc = m < rows_end ? m : rows_end - 1;
        const int vb = ESZ * mc * K;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int vo = vb + (ks == NK - 1 ? koff_last : koff);
            x[ks][0] = ld(vo, ESZ * BK * ks);
            if constexpr (XL == 2) x[ks][1] = ld(vo + 16, ESZ * BK * ks);
        }
        const int ob = ESZ * (m - j) * N;                                     // byte offset of the tile's output rows (wave-uniform)
#pragma unroll
        for (int p = 0; p < NT; ++p)                                         // rows past M: the buffer bound returns zeros
            r[p] = R ? ld_unit(rres, 4 * ESZ * (64 * p + lane), ob) : (v4f){0.f, 0.f, 0.f, 0.f};
    };
    if (cnt > 0) load_tile(0, xr[0], rr[0]);
    // the weight matrix and the gate row -> LDS (threads past the last chunk repeat it: same value, same address)
    {
        u4 wv[WLOADS];
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) {
            const int e = tid + t * NTHR < CHUNKS ? tid + t * NTHR : CHUNKS - 1;
            const int ks = e / (BN * 4 * NP), e2 = e - ks * (BN * 4 * NP);
            const int row = e2 / (4 * NP), rem = e2 - row * (4 * NP), pl = rem >> 2, c = rem & 3;
            wv[t] = *reinterpret_cast<const u4*>(W3 + (size_t)pl * plane + (size_t)row * Kp + ks * BK + 8 * c);
        }
        constexpr int GL = (2 * NK * BK + NTHR - 1) / NTHR;
        const int img_last = GATE ? (M - 1) / (tpg * 16) : 0;
        const bool own_gate = GATE && se.P != nullptr;               // this block evaluates its images' gates itself (below)
        float gv[GL];
#pragma unroll
        for (int t = 0; t < GL; ++t) {
            const int e = tid + t * NTHR, im = e >= NK * BK ? 1 : 0, k = e - im * NK * BK;
            const int img = img0 + im < img_last ? img0 + im : img_last;
            gv[t] = GATE && !own_gate ? gate[(size_t)img * K + (k < K ? k : 0)] : 1.f;
        }
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) {
            const int e = tid + t * NTHR < CHUNKS ? tid + t * NTHR : CHUNKS - 1;
            const int ks = e / (BN * 4 * NP), e2 = e - ks * (BN * 4 * NP);
            const int row = e2 / (4 * NP), rem = e2 - row * (4 * NP), pl = rem >> 2, c = rem & 3;
            *reinterpret_cast<u4*>(&ws[ks][row * S6_ROWB + s6_chunk_pos(row, pl * 4 + c) * 16]) = wv[t];
        }
#pragma unroll
        for (int t = 0; t < GL; ++t) {
            const int e = tid + t * NTHR, im = e >= NK * BK ? 1 : 0, k = e - im * NK * BK;
            if (e < 2 * NK * BK) gs[im][k] = k < K ? gv[t] : 0.f;
        }
    }
    if constexpr (GATE) {
        // Squeeze-excite of the block's (at most two) images by the block itself: on these layers (C <= 256, c_se <= 16) the
        // gate is ~5 k MACs per image and the separate se_kernel launch 10-13 us of launch + three dependent round trips on an
        // otherwise idle chip.  MEASURED (option "se_thin", batch 256): the five projections 105 / 69 / 134 / 34 / 58 -> 130 / 82 /
        // 159 / 41 / 73 us - +85 us for 54 us of se_kernel launches: every block re-reads its images' pool partials (98 x 32
        // floats per image in block 0) behind two barriers before its first tile, and the tuner answers with fewer, longer
        // blocks.  Third form of "squeeze-excite inside a neighbouring launch" that loses (fuse_se, se_in_proj): off.  Fixed
        // summation orders (tile sums as four strided chains folded pairwise, FC1 in 8 lanes per output + xor tree, FC2 as a
        // chain from the bias): every block that meets an image writes the same gate bits.
        if (se.P != nullptr) {
            float* mean_s = reinterpret_cast<float*>(&os[0][0]);     // [2][SE_THIN_MAX_C] (the output image is not in use yet)
            float* z_s = mean_s + 2 * SE_THIN_MAX_C;                 // [2][SE_THIN_MAX_SE]
            const int img_last = (M - 1) / (tpg * 16), C = K;
            for (int e = tid; e < 2 * C; e += NTHR) {
                const int im = e >= C ? 1 : 0, c = e - im * C;
                const int img = img0 + im < img_last ? img0 + im : img_last;
                const float* p = se.P + (size_t)img * se.tiles * C + c;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int t = 0;
                for (; t + 3 < se.tiles; t += 4) {
                    s0 += p[(size_t)t * C];
                    s1 += p[(size_t)(t + 1) * C];
                    s2 += p[(size_t)(t + 2) * C];
                    s3 += p[(size_t)(t + 3) * C];
                }
                for (; t < se.tiles; ++t) s0 += p[(size_t)t * C];
                mean_s[im * SE_THIN_MAX_C + c] = ((s0 + s1) + (s2 + s3)) * se.inv_hw;
            }
            __syncthreads();
            for (int e = tid; e < 2 * SE_THIN_MAX_SE * 8; e += NTHR) {      // (image, output, part of 8): 256 work items
                const int im = e >> 7, o = (e >> 3) & (SE_THIN_MAX_SE - 1), part = e & 7;
                const int oo = o < se.c_se ? o : se.c_se - 1;
                float v = 0.f;
                for (int c = part; c < C; c += 8) v = __builtin_fmaf(mean_s[im * SE_THIN_MAX_C + c], se.w1[(size_t)oo * C + c], v);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (part == 0 && o < se.c_se) z_s[im * SE_THIN_MAX_SE + o] = swish1(v + se.b1[o]);
            }
            __syncthreads();
            for (int e = tid; e < 2 * NK * BK; e += NTHR) {
                const int im = e >= NK * BK ? 1 : 0, c = e - im * NK * BK;
                float g = 0.f;
                if (c < C) {
                    float sacc = se.b2[c];
                    for (int o = 0; o < se.c_se; ++o) sacc = __builtin_fmaf(z_s[im * SE_THIN_MAX_SE + o], se.w2t[(size_t)o * C + c], sacc);
                    g = sigmoid1(sacc);
                    const int img = img0 + im < img_last ? img0 + im : img_last;
                    if (img0 + im <= img_last) const_cast<float*>(gate)[(size_t)img * C + c] = g;      // for taps; every writer agrees
                }
                gs[im][c] = g;
            }
        }
    }
    __syncthreads();
    if (cnt == 0) return;

    v4f bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + nq;
        bv[nt] = ldg4(bias + (n < N ? n : 0));
    }
    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto read_w = [&](int ks, int nt, bf8 (&f)[NP]) {
        const unsigned char* wp = &ws[ks][(nt * 16 + j) * S6_ROWB];
        f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
        if constexpr (NP == 3) {
            f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
            f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
        }
    };
    auto tile = [&](auto bc, int i) {
        constexpr int b = decltype(bc)::value;
        load_tile(i + 1, xr[b ^ 1], rr[b ^ 1]);                              // the whole next tile (clamped: the repeat is never used)
        __builtin_amdgcn_sched_barrier(0);
        v4f acc[1][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        const float* gw = gs[GATE && t0 + wave + i * NW >= tsplit ? 1 : 0];            // wave-uniform
        bf8 wf[2][NP];
        read_w(0, 0, wf[0]);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            bf8 xs[1][NXS];
            if constexpr (ESZ == 4) {
                v4f lo = xr[b][ks][0], hi = xr[b][ks][XL - 1];
                if constexpr (GATE) {
                    lo *= *reinterpret_cast<const v4f*>(&gw[ks * BK + 8 * q]);
                    hi *= *reinterpret_cast<const v4f*>(&gw[ks * BK + 8 * q + 4]);
                }
                split8(lo, hi, xs[0][0], xs[0][NXS > 1 ? 1 : 0], xs[0][NXS > 2 ? 2 : 0]);
            } else {
                const v4f g0 = GATE ? *reinterpret_cast<const v4f*>(&gw[ks * BK + 8 * q]) : (v4f){1.f, 1.f, 1.f, 1.f};
                const v4f g1 = GATE ? *reinterpret_cast<const v4f*>(&gw[ks * BK + 8 * q + 4]) : (v4f){1.f, 1.f, 1.f, 1.f};
                xs[0][0] = bf16x8_gate<GATE>(xr[b][ks][0], g0, g1);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nx = ks * NT + nt + 1;                             // next (K-step, tile) fragment in flight during these MFMAs
                if (nx < NK * NT) read_w(nx / NT, nx % NT, wf[nx & 1]);
                __builtin_amdgcn_sched_barrier(0);
                s6_products<1, NXS, NP, NT>(acc, xs, wf[(ks * NT + nt) & 1], nt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: the operations of s6_epilogue (N % 4 == 0; bias, activation, then the residual - the host sends no
        // residual-before-activation call here), the tile turned into row order through the wave's LDS image on the way
        float* ow = os[wave];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + nq;
            v4f v = acc[0][nt] + bv[nt];
            if (act == ACT_SWISH) v = swish4(v);
            else if (act == ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (n < N) stg4(ow + j * N + n, v);
        }
        // LDS operations of one wave execute in issue order; the compiler only has to keep the two groups apart
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int ob = ESZ * (tile_row(i) - j) * N;
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int u = 64 * p + lane;                                     // 4-channel unit of the tile
            if (u < 4 * N) {
                v4f v = ldg4(ow + 4 * u) + rr[b][p];
                if constexpr (ESZ == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, 16 * u, ob, 0);
                else {
                    const bf4v h = __builtin_convertvector(v, bf4v);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, h), ry, 8 * u, ob, 0);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    int i = 0;
    for (; i + 2 <= cnt; i += 2) {
        tile(std::integral_constant<int, 0>{}, i);
        tile(std::integral_constant<int, 1>{}, i + 1);
    }
    if (i < cnt) tile(std::integral_constant<int, 0>{}, i);
}

// shapes pw8 is instantiated for: all of N in NT 16-column tiles (N <= 48, N % 4 == 0), K in NK K-steps exactly
// ------------------------------------------------------------------------------------------------------
// pw9_kernel (round 4): the wide expansions that still run as their own launch (blocks 8, 9, 11: K = 80 / 112, N = 480 /
// 672 over 50,176 rows).  With K this short a pw6 block is a prologue, three or four K-steps and an epilogue of twelve
// tiles, and every one of the N / 96 blocks of a row range loads and splits the same activations again (MFMA busy 0.30 of
// the launch).  Here a block keeps its 128 rows' split activations IN REGISTERS (MT = 2 row tiles x NKC K-steps x three
// bf16 planes per wave) and walks ALL column blocks itself: the weight stream (the only thing that changes) runs through
// the double-buffered LDS stage continuously across column blocks, the split happens once.  Products and epilogue are
// pw6's (s6_products / s6_epilogue): identical bits, a candidate of the tile tests.  Measured (tuner, batch 256): K = 80,
// N = 480 40.2 -> 38.7 us, K = 112, N = 672 65.8 -> 60.9 us - the launch writes 96 / 135 MB and that is most of its time
// (2.2 TB/s; MFMA work 18 us), the redundant splits were the smaller part.
template <int NT, int NKC, bool PIPE, typename XT, int NP>
__global__ __launch_bounds__(256, 2) void pw9_kernel(const XT* __restrict__ X, const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias, XT* __restrict__ Y, int M, int K, int N, int act,
                                                     unsigned xbytes) {
    static_assert(sizeof(XT) == 4 && NP == 3, "fp32 activations, three weight planes");
    constexpr int BK = S6_BK, MT = 2, BN = NT * 16, BM = 4 * MT * 16;
    constexpr int CHUNKS = BN * 12, WLOADS = (CHUNKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][BN * S6_ROWB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM, nblocks = (N + BN - 1) / BN;                             // (K + 31) / 32 == NKC (host)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<XT*>(X), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W3), 0, 6 * plane, 0x00020000);
    int m[MT];
    // weight chunk t of this thread, as in pw6: (row, plane, k-octet) -> global / LDS offsets
    int wvo[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * 256 < CHUNKS ? tid + t * 256 : CHUNKS - 1;
        const int row = e / 12, rem = e - row * 12, pl = rem >> 2, c = rem & 3;
        wvo[t] = 2 * (pl * plane + row * Kp + 8 * c);
        wlds[t] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + c) * 16;
    }
    u4 wr[WLOADS];
    auto load_w = [&](int nb, int ks) {                              // stage (nb, ks): rows nb * BN .., k = 32 ks ..
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) wr[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[t], 2 * (nb * BN * Kp + ks * BK), 0);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) *reinterpret_cast<u4*>(&ws[buf][wlds[t]]) = wr[t];
    };
    load_w(0, 0);
    // the block's activations: loaded and split once
    bf8 xs[NKC][MT][3];
    {
        const int over = (NKC - 1) * BK + 8 * q - (K - 8);
        int vb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            m[mt] = m0 + wave * (MT * 16) + mt * 16 + j;
            vb[mt] = 4 * ((m[mt] < M ? m[mt] : M - 1) * K + 8 * q);
        }
        // (two K-steps at a time: the raw fragments of all four next to their twelve planes do not fit 256 registers)
#pragma unroll
        for (int k0 = 0; k0 < NKC; k0 += 2) {
            v4f xr[2][MT][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = k0 + kk < NKC ? k0 + kk : NKC - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int vo = vb[mt] - (ks == NKC - 1 && over > 0 ? 4 * over : 0);
                    xr[kk][mt][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 4 * BK * ks, 0));
                    xr[kk][mt][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + 16, 4 * BK * ks, 0));
                }
            }
            if (k0 == 0) store_w(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (k0 + kk < NKC) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        split8(xr[kk][mt][0], xr[kk][mt][1], xs[k0 + kk][mt][0], xs[k0 + kk][mt][1], xs[k0 + kk][mt][2]);
                }
        }
    }
    __syncthreads();
    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto read_w = [&](const unsigned char* wb, int nt, bf8 (&f)[3]) {
        const unsigned char* wp = wb + (nt * 16 + j) * S6_ROWB;
        f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
        f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
        f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
    };
    // finishing one tile of a column block: s6_epilogue's operations without a residual (the host sends none here)
    auto finish_tile = [&](const v4f& a, const v4f& bv, int mt, int n) {
        v4f v = a + bv;
        if (act == ACT_SWISH) v = swish4(v);
        else if (act == ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (n < N && m[mt] < M) st4(Y + (size_t)m[mt] * N + n, v);      // (non-temporal stores: measured, 1-2 us slower)
    };
    auto load_bias = [&](int nb, v4f (&bv)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nb * BN + nt * 16 + 4 * q;
            bv[nt] = ldg4(bias + (n < N ? n : 0));
        }
    };
    int stage = 0;
    if constexpr (!PIPE) {
        for (int nb = 0; nb < nblocks; ++nb) {
            v4f acc[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKC; ++ks, ++stage) {
                // the next stage's weights (next K-step, or the first of the next column block; past the end: a repeat nobody reads)
                const int nks = ks + 1 < NKC ? ks + 1 : 0, nnb = ks + 1 < NKC ? nb : (nb + 1 < nblocks ? nb + 1 : nb);
                load_w(nnb, nks);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* wb = ws[stage & 1];
                bf8 wf[2][3];
                read_w(wb, 0, wf[0]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt + 1 < NT) read_w(wb, nt + 1, wf[(nt + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    s6_products<MT, 3, 3, NT>(acc, xs[ks], wf[nt & 1], nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                store_w((stage + 1) & 1);
                __syncthreads();
            }
            v4f bv[NT];
            load_bias(nb, bv);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) finish_tile(acc[mt][nt], bv[nt], mt, nb * BN + nt * 16 + 4 * q);
        }
    } else {
        // PIPE: two accumulator sets.  While the MFMAs of column block nb run (the matrix pipe works for 16 cycles per issue),
        // the wave finishes the tiles of column block nb - 1 between them - bias, swish (~48 VALU instructions per tile) and
        // the store: the epilogue that followed every K loop as a phase of its own rides inside the next one.  MEASURED (tuner,
        // batch 256, NT = 4): 40.0 against 40.4 us (K = 80), 62.8 against 61.4 us (K = 112) - nothing: the launch is bound by its
        // 96 / 135 MB of stores (2.3 TB/s), not by the serial VALU / MFMA phases.  Stays a candidate (never picked so far).
        constexpr int TILES = MT * NT, TPS = (TILES + NKC - 1) / NKC;      // tiles of the previous block finished per K-step
        v4f accs[2][MT][NT];
        v4f bvp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bvp[nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        auto body = [&](auto cur_c, int nb, bool have_prev) {
            constexpr int cur = decltype(cur_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accs[cur][mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKC; ++ks, ++stage) {
                const int nks = ks + 1 < NKC ? ks + 1 : 0, nnb = ks + 1 < NKC ? nb : (nb + 1 < nblocks ? nb + 1 : nb);
                load_w(nnb, nks);
                const unsigned char* wb = ws[stage & 1];
                bf8 wf[2][3];
                read_w(wb, 0, wf[0]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt + 1 < NT) read_w(wb, nt + 1, wf[(nt + 1) & 1]);
                    s6_products<MT, 3, 3, NT>(accs[cur], xs[ks], wf[nt & 1], nt);
                    // tiles ks * TPS .. of the previous block, spread over this K-step's MFMA groups
#pragma unroll
                    for (int i = 0; i < TPS; ++i) {
                        if (((i + 1) * NT + TPS - 1) / TPS - 1 == nt) {
                            const int e = ks * TPS + i;                // tile index: nt-major, mt inner (s6_epilogue's store order)
                            if (e < TILES && have_prev) {
                                const int pn = e / MT, pm = e - pn * MT;
                                finish_tile(accs[cur ^ 1][pm][pn], bvp[pn], pm, (nb - 1) * BN + pn * 16 + 4 * q);
                            }
                        }
                    }
                }
                store_w((stage + 1) & 1);
                __syncthreads();
            }
            load_bias(nb, bvp);                                          // for this block's tiles, finished inside the next one
        };
        int nb = 0;
        for (; nb + 2 <= nblocks; nb += 2) {
            body(std::integral_constant<int, 0>{}, nb, nb > 0);
            body(std::integral_constant<int, 1>{}, nb + 1, true);
        }
        if (nb < nblocks) {
            body(std::integral_constant<int, 0>{}, nb, nb > 0);
            ++nb;
        }
        // the last block's tiles
        const int last = (nblocks - 1) & 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (last == 0) finish_tile(accs[0][mt][nt], bvp[nt], mt, (nblocks - 1) * BN + nt * 16 + 4 * q);
                else finish_tile(accs[1][mt][nt], bvp[nt], mt, (nblocks - 1) * BN + nt * 16 + 4 * q);
            }
    }
}

// shapes pw9 takes: plain (not gated, not convolved) fp32 1x1 convs with a short K and many column blocks
#define DFD_S9_NK_CASES(OP) OP(3) OP(4)
inline bool s9_supports(int K, int N) {
    const int nk = (K + S6_BK - 1) / S6_BK;
    return K % 8 == 0 && (nk == 3 || nk == 4) && N >= 192 && N % 4 == 0;
}

// (NK, waves per block): K = 240 (block 4) keeps two tiles of 8 K-steps in registers and 74 KB of weights in LDS - one
// block of eight waves per CU
#define DFD_S8_NK_CASES(OP) OP(1, 4) OP(3, 4) OP(5, 4) OP(8, 8)
inline int s8_waves(int K) { return (K + S6_BK - 1) / S6_BK == 8 ? 8 : 4; }
inline bool s8_supports(int K, int N) {
    const int nk = (K + S6_BK - 1) / S6_BK;
    return N % 4 == 0 && N <= 48 && K % 8 == 0 && (nk == 1 || nk == 3 || nk == 5 || nk == 8);
}

// kind 0: pw6 (block = NW waves x MT*16 rows, NT*16 columns; wm = NW); kind 1: pw7 (WM x WN waves of MT x NT tiles);
// kind 2: pw8 (wm = waves per block, ks = tiles per wave, nt = all of N; mblocks = groups x parts)
struct S6Tile { int kind, wm, wn, mt, nt, ks, mblocks, nblocks; bool measured; };      // ks: K-steps per stage (pw6)
inline S6Tile make_tile(int M, int N, int kind, int wm, int wn, int mt, int nt, int ks = 1) {
    const int bm = wm * mt * 16, bn = wn * nt * 16;
    return S6Tile{kind, wm, wn, mt, nt, ks, (M + bm - 1) / bm, (N + bn - 1) / bn, false};
}
inline S6Tile make_tile6(int M, int N, int mt, int nt, int ks = 1) { return make_tile(M, N, 0, 4, 1, mt, nt, ks); }

#define DFD_S6_NT_CASES(OP) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8)
// pw7 instances: (WM, WN, MT, NT)
#define DFD_S7_CONFIGS(OP)                                                                       \
    OP(1, 4, 2, 1) OP(1, 4, 2, 2) OP(1, 4, 2, 3) OP(1, 4, 4, 1) OP(1, 4, 4, 2) OP(1, 4, 4, 3)     \
    OP(2, 2, 1, 2) OP(2, 2, 1, 3) OP(2, 2, 1, 4) OP(2, 2, 1, 6) OP(2, 2, 2, 2) OP(2, 2, 2, 3)     \
    OP(2, 2, 2, 4) OP(2, 2, 2, 6) OP(2, 2, 4, 2) OP(2, 2, 4, 3)

// one launch of tile `t`; the caller keeps every call below 2^31 bytes of activations (the kernels address X /
// gate with 32-bit buffer offsets).  XT = activation storage (float / bf16_t), NP = weight planes used.
template <bool CONV, bool GATE, typename XT, int NP>
void s6_dispatch(const S6Tile& t, const XT* X, const unsigned short* W3, const float* bias,
                 const float* gate, const XT* R, XT* Y, int M, int K, int N, int HW, int act,
                 const ConvGeom& g, int res_first, hipStream_t s, const SeFuse& se) {
    const int grid = ((t.mblocks + 7) / 8) * 8 * t.nblocks;
    const int Kp = (K + S6_KPAD - 1) / S6_KPAD * S6_KPAD, plane = s6_np(N) * Kp;
    const unsigned xbytes = CONV ? (unsigned)((size_t)(M / (g.Ho * g.Wo)) * g.H * g.W * g.Cin * sizeof(XT))
                                 : (unsigned)((size_t)M * K * sizeof(XT));
    const unsigned gbytes = GATE ? (unsigned)((size_t)((M + HW - 1) / HW) * K * 4) : 0u;
    if (t.kind == 3) {
        if constexpr (!CONV && !GATE && sizeof(XT) == 4 && NP == 3) {
            const int nk = (K + S6_BK - 1) / S6_BK, mb = (M + 127) / 128;
#define DFD_S9_CASE(NKV)                                                                                                     \
    if (nk == NKV) {                                                                                                         \
        if (t.nt == 6 && t.ks == 1) hipLaunchKernelGGL((pw9_kernel<6, NKV, false, XT, NP>), dim3(mb), dim3(256), 0, s, X, W3, plane, Kp, \
                                                       bias, Y, M, K, N, act, xbytes);                                       \
        else if (t.ks == 1) hipLaunchKernelGGL((pw9_kernel<4, NKV, false, XT, NP>), dim3(mb), dim3(256), 0, s, X, W3, plane, Kp, bias, \
                                               Y, M, K, N, act, xbytes);                                                     \
        else hipLaunchKernelGGL((pw9_kernel<4, NKV, true, XT, NP>), dim3(mb), dim3(256), 0, s, X, W3, plane, Kp, bias, Y, M, K, \
                                N, act, xbytes);                                                                             \
    }
            DFD_S9_NK_CASES(DFD_S9_CASE)
#undef DFD_S9_CASE
        }
        return;
    }
    if (t.kind == 2) {
        if constexpr (!CONV) {
            // t.ks = blocks per CU the grid aims at; equal shares of the matrix' tiles, at most one image's worth (a block
            // stages two gate rows) and at least one tile per wave
            const int tpg = GATE ? HW / 16 : 1 << 30, ttot = (M + 15) / 16, nk = (K + S6_BK - 1) / S6_BK;
            int tpb = (ttot + 256 * t.ks - 1) / (256 * t.ks);
            tpb = tpb > tpg ? tpg : tpb < t.wm ? t.wm : tpb;
            const int nblk = (ttot + tpb - 1) / tpb;
#define DFD_S8_LAUNCH(NTV, NKV, NWV)                                                                                 \
    hipLaunchKernelGGL((pw8_kernel<NTV, NKV, GATE, NWV, XT, NP>), dim3(nblk), dim3(NWV * 64), 0, s, X, W3, plane, Kp, bias, \
                       gate, R, Y, M, K, N, tpg, tpb, act, res_first, xbytes, se)
#define DFD_S8_CASE(NKV, NWV)                            \
    if (nk == NKV) {                                     \
        if (t.nt == 1) DFD_S8_LAUNCH(1, NKV, NWV);       \
        else if (t.nt == 2) DFD_S8_LAUNCH(2, NKV, NWV);  \
        else DFD_S8_LAUNCH(3, NKV, NWV);                 \
    }
            DFD_S8_NK_CASES(DFD_S8_CASE)
#undef DFD_S8_CASE
#undef DFD_S8_LAUNCH
        }
        return;
    }
    if (t.kind == 1) {
#define DFD_S7_CASE(WMV, WNV, MTV, NTV)                                                                              \
    if (t.wm == WMV && t.wn == WNV && t.mt == MTV && t.nt == NTV) {                                                  \
        hipLaunchKernelGGL((pw7_kernel<WMV, WNV, MTV, NTV, CONV, GATE, XT, NP>), dim3(grid), dim3(256), 0, s, X, W3, plane,  \
                           Kp, bias, gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes, se); \
        return;                                                                                                      \
    }
        DFD_S7_CONFIGS(DFD_S7_CASE)
#undef DFD_S7_CASE
        return;
    }
#define DFD_S6_LAUNCH(NTV, MTV, KSV)                                                                                 \
    hipLaunchKernelGGL((pw6_kernel<NTV, CONV, MTV, GATE, KSV, 4, XT, NP>), dim3(grid), dim3(256), 0, s, X, W3, plane, Kp, bias, gate, \
                       R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes, se)
#define DFD_S6_LAUNCH8(NTV, KSV)                                                                                     \
    hipLaunchKernelGGL((pw6_kernel<NTV, CONV, 1, GATE, KSV, 8, XT, NP>), dim3(grid), dim3(512), 0, s, X, W3, plane, Kp, bias, \
                       gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes, se)
#define DFD_S6_CASE(NTV)                                        \
    case NTV:                                                   \
        if (t.wm == 8 && t.ks == 2) DFD_S6_LAUNCH8(NTV, 2);     \
        else if (t.wm == 8) DFD_S6_LAUNCH8(NTV, 1);             \
        else if (t.mt == 2 && t.ks == 2) DFD_S6_LAUNCH(NTV, 2, 2);   \
        else if (t.mt == 2) DFD_S6_LAUNCH(NTV, 2, 1);           \
        else if (t.ks == 2) DFD_S6_LAUNCH(NTV, 1, 2);           \
        else DFD_S6_LAUNCH(NTV, 1, 1);                          \
        break;
    switch (t.nt) { DFD_S6_NT_CASES(DFD_S6_CASE) }
#undef DFD_S6_LAUNCH
#undef DFD_S6_LAUNCH8
#undef DFD_S6_CASE
}

// Host entry used by the tuner / launchers in gemm_split.hip; explicitly instantiated per (XT, NP) in its own
// translation unit (gemm_split.hip: <float, 3>; gemm_split_bf16.hip: <bf16_t, 3> and <bf16_t, 1>) so that the
// ~100 kernel instances of each flavour compile in parallel.
template <typename XT, int NP>
void s6_dispatch_any(bool conv, bool gated, const S6Tile& t, const XT* X, const unsigned short* W3, const float* bias,
                     const float* gate, const XT* R, XT* Y, int M, int K, int N, int HW, int act, const ConvGeom& g,
                     int res_first, hipStream_t s, const SeFuse& se);

#define DFD_S6_INSTANTIATE(XT, NP)                                                                                      \
    template <>                                                                                                         \
    void s6_dispatch_any<XT, NP>(bool conv, bool gated, const S6Tile& t, const XT* X, const unsigned short* W3,         \
                                 const float* bias, const float* gate, const XT* R, XT* Y, int M, int K, int N, int HW, \
                                 int act, const ConvGeom& g, int res_first, hipStream_t s, const SeFuse& se) {          \
        if (conv) s6_dispatch<true, false, XT, NP>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);  \
        else if (gated) s6_dispatch<false, true, XT, NP>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se); \
        else s6_dispatch<false, false, XT, NP>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s, se);      \
    }

}  // namespace dfd
