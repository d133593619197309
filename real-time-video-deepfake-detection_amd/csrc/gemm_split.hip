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
#include <algorithm>
#include <new>
#include <tuple>
#include <type_traits>
#include <vector>

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

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 struct does not always leave the stack)

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

// W [N][K] fp32 -> three planes [Np][Kp] bf16, zero outside N x K (Kp = K rounded up to 64, Np = s6_np(N)):
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
    return (size_t)s6_np(N) * ((K + S6_KPAD - 1) / S6_KPAD * S6_KPAD);
}

void launch_split_weights(const float* W, unsigned short* out, int N, int K, hipStream_t s) {
    const int Np = s6_np(N), Kp = (K + S6_KPAD - 1) / S6_KPAD * S6_KPAD;
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

// epilogue shared by both kernels: the lane holds Y[m[mt]][n .. n+3] for n = nbase + 16 * nt.
// When N is a multiple of 4 (every layer of B0 and of the detector) the bias and residual fragments are requested
// up front, unconditionally (clamped indices): under the per-tile `continue`s below hipcc issued them one by one,
// each behind its own wait - the s_memtime trace showed 9,000 cycles of epilogue for MT x NT = 6 residual loads.
template <int MT, int NT>
__device__ __forceinline__ void s6_epilogue(const v4f (&acc)[MT][NT], const int (&m)[MT], int nbase,
                                            const float* __restrict__ bias, const float* __restrict__ R,
                                            float* __restrict__ Y, int M, int N, int act, int res_first) {
    if ((N & 3) == 0) {                                  // uniform
        v4f bv[NT], rv[MT][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nbase + nt * 16;
            const int nc = n < N ? n : 0;
            bv[nt] = ldg4(bias + nc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int mc = m[mt] < M ? m[mt] : M - 1;
                rv[mt][nt] = R ? ldg4(R + (size_t)mc * N + nc) : (v4f){0.f, 0.f, 0.f, 0.f};
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
                }
                if (!res_first) v += rv[mt][nt];
                if (n < N && m[mt] < M) stg4(Y + (size_t)m[mt] * N + n, v);
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
template <int NT, bool CONV, int MT, bool GATE, int KS, bool PIPE = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, (NW == 4 && 2 * KS * NT * 16 * S6_ROWB <= 80 * 1024) ? 2 : 1) void pw6_kernel(const float* __restrict__ X,
                                                     const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gate,
                                                     const float* __restrict__ R,
                                                     float* __restrict__ Y, int M, int K, int N,
                                                     int HW, int act, int mblocks, int nblocks,
                                                     ConvGeom cg, int res_first, unsigned xbytes, unsigned gbytes) {
    constexpr int BK = S6_BK;
    constexpr int BN = NT * 16, BM = NW * MT * 16, NTHR = NW * 64;
    constexpr int CHUNKS = BN * 12 * KS;                  // 16-byte chunks per stage: row x plane x k-octet
    constexpr int WLOADS = (CHUNKS + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][KS][BN * S6_ROWB];

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int n0 = nblk * BN;
#ifdef S6_TRACE
    int tp = 0;
#endif

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
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? gate : X), 0, GATE ? gbytes : xbytes, 0x00020000);
    // With KS = 2 a row's 4 * KS consecutive octets of one plane are 128 contiguous bytes = one cache line per 8
    // lanes: the 64-byte pieces of a single K-step use half of every line they pull through the L1, and at small
    // tiles the L1 (64 B/clk per CU, 3 * BN + 2 * BM line-cycles per K-step against 0.094 * BM * BN MFMA cycles)
    // is what a K-step waits for.
    int wvo[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * NTHR < CHUNKS ? tid + t * NTHR : CHUNKS - 1;
        const int row = e / (12 * KS), rem = e - row * (12 * KS), pl = rem / (4 * KS), c = rem - pl * (4 * KS);
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
    v4f xr[U][KS][MT][2];
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
        xvo[mt] = 4 * (mc * K + 8 * q);
        xvo_last[mt] = xvo[mt] - 4 * (over > 0 ? over : 0);
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
    auto load_x = [&](int kstep, v4f (&x)[MT][2], bool (&okf)[MT]) {
        const int kc = kstep * BK;
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // wave-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int iy = iy0[mt] + ky * cg.dil, ix = ix0[mt] + kx * cg.dil;
                const bool ok = m[mt] < M && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const int vo = 4 * ((int)gbase[mt] + ((ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin + 8 * q);
                x[mt][0] = ld(rx, vo, 4 * ci0);
                x[mt][1] = ld(rx, vo + 16, 4 * ci0);
                okf[mt] = ok;
            }
        } else {
            const bool last = kstep == nk - 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int vo = last ? xvo_last[mt] : xvo[mt];
                x[mt][0] = ld(rx, vo, 4 * kc);
                x[mt][1] = ld(rx, vo + 16, 4 * kc);
                okf[mt] = true;
            }
        }
    };
    auto store_w = [&](const u4 (&w)[WLOADS], unsigned char* dst) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) *reinterpret_cast<u4*>(dst + wlds[t]) = w[t];
    };

    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto split_x = [&](const v4f (&x)[MT][2], const v4f (&g)[MT][2], const bool (&okf)[MT], bf8 (&xs)[MT][3]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            v4f lo = x[mt][0], hi = x[mt][1];
            if constexpr (GATE) { lo *= g[mt][0]; hi *= g[mt][1]; }
            if constexpr (CONV) {
                if (!okf[mt]) { lo = (v4f){0.f, 0.f, 0.f, 0.f}; hi = lo; }
            }
            split8(lo, hi, xs[mt][0], xs[mt][1], xs[mt][2]);
        }
    };
    auto mfma_tile = [&](const bf8 (&xs)[MT][3], const bf8 (&f)[3], int nt, bool interleave) {
        const bf8 w0 = f[0], w1 = f[1], w2 = f[2];
        const bf8* wsel[6] = {&w2, &w1, &w0, &w1, &w0, &w0};      // smallest terms first
        const int xsel[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*wsel[p6], xs[mt][xsel[p6]], acc[mt][nt], 0, 0, 0);
                if (interleave) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // ... then two VALU in its shadow
                }
            }
    };
    auto compute = [&](const v4f (&x)[MT][2], const v4f (&g)[MT][2], const bool (&okf)[MT], const unsigned char* wb) {
        bf8 xs[MT][3];
        split_x(x, g, okf, xs);
        // The fragments of tile nt + 1 are requested before the MFMAs of tile nt are issued (two register sets,
        // order pinned): reading them right before use, as hipcc schedules it on its own, leaves the LDS latency
        // (~150 cycles) exposed NT times per K-step - as long as the MFMAs themselves at one wave per SIMD.
        bf8 wf[2][3];
        auto read_w = [&](int nt, bf8 (&f)[3]) {
            const unsigned char* wp = wb + (nt * 16 + j) * S6_ROWB;       // (nt * 16 + j) >> 2 has the parity of j >> 2
            f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
            f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
            f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
        };
        read_w(0, wf[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + 1 < NT) read_w(nt + 1, wf[(nt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(xs, wf[nt & 1], nt, false);
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

    if constexpr (PIPE) {
        static_assert(!PIPE || KS == 1, "PIPE is built for one K-step per stage");
        // slot u = step & 1.  At step s: raw X / gate of step s + 2 and the weights of step s + 2 are requested into
        // slot u (its previous contents, step s, were split / stored during step s - 1); X(s + 1) in slot u ^ 1 is
        // split into xs[u ^ 1] between the MFMAs of step s, which read xs[u] and LDS buffer u; the weights of
        // step s + 1 go to LDS buffer u ^ 1 at the bottom.
        bf8 xs[2][MT][3];
        auto pstep = [&](auto uc, int st) {
            constexpr int u = decltype(uc)::value;
            const int s2 = st + 2 < nk ? st + 2 : nk - 1;
            load_w(s2, wr[u]);
            load_g(s2, gr[u][0]);
            load_x(s2, xr[u][0], okr[u][0]);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* wb = ws[u][0];
            bf8 wf[2][3];
            auto read_w = [&](int nt, bf8 (&f)[3]) {
                const unsigned char* wp = wb + (nt * 16 + j) * S6_ROWB;
                f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
                f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
                f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
            };
            read_w(0, wf[0]);
            split_x(xr[u ^ 1][0], gr[u ^ 1][0], okr[u ^ 1][0], xs[u ^ 1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);              // first fragments
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);              // conversions while they arrive
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt + 1 < NT) {
                    read_w(nt + 1, wf[(nt + 1) & 1]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                }
                mfma_tile(xs[u], wf[nt & 1], nt, true);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_w(wr[u ^ 1], ws[u ^ 1][0]);
            __syncthreads();
        };
        load_w(0, wr[0]);
        load_g(0, gr[0][0]);
        load_x(0, xr[0][0], okr[0][0]);
        const int one = nk > 1 ? 1 : 0;
        load_w(one, wr[1]);
        load_g(one, gr[1][0]);
        load_x(one, xr[1][0], okr[1][0]);
        store_w(wr[0], ws[0][0]);
        split_x(xr[0][0], gr[0][0], okr[0][0], xs[0]);
        __syncthreads();
        int ps = 0;
        for (; ps + 2 <= nk; ps += 2) {
            pstep(std::integral_constant<int, 0>{}, ps);
            pstep(std::integral_constant<int, 1>{}, ps + 1);
        }
        if (ps < nk) pstep(std::integral_constant<int, 0>{}, ps);
        s6_epilogue<MT, NT>(acc, m, n0 + 4 * q, bias, R, Y, M, N, act, res_first);
        return;
    }
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
    s6_epilogue<MT, NT>(acc, m, n0 + 4 * q, bias, R, Y, M, N, act, res_first);
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
template <int WM, int WN, int MT, int NT, bool CONV, bool GATE>
__global__ __launch_bounds__(256, (2 * (WM * MT + WN * NT) * 16 * S6_ROWB <= 80 * 1024) ? 2 : 1) void pw7_kernel(const float* __restrict__ X,
                                                     const unsigned short* __restrict__ W3, int plane, int Kp,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ gate,
                                                     const float* __restrict__ R,
                                                     float* __restrict__ Y, int M, int K, int N,
                                                     int HW, int act, int mblocks, int nblocks,
                                                     ConvGeom cg, int res_first, unsigned xbytes, unsigned gbytes) {
    static_assert(WM * WN == 4, "four waves per block");
    constexpr int BK = S6_BK;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int WCHUNKS = BN * 12, WLOADS = (WCHUNKS + 255) / 256;
    constexpr int XUNITS = BM * 4, XL = (XUNITS + 255) / 256;      // unit = 8 consecutive k of one row
    __shared__ __attribute__((aligned(16))) unsigned char ws[2][BN * S6_ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char xsm[2][BM * S6_ROWB];

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mblk = (idx / nblocks) * 8 + xcd, nblk = idx % nblocks;
    if (mblk >= mblocks) return;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mblk * BM, n0 = nblk * BN;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W3), 0, 6 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? gate : X), 0, GATE ? gbytes : xbytes, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int vo, int so) {
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    };
    const int nk = (K + BK - 1) / BK;

    // weight chunks of this thread (as in pw6)
    int wvo[WLOADS], wlds[WLOADS];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t) {
        const int e = tid + t * 256 < WCHUNKS ? tid + t * 256 : WCHUNKS - 1;
        const int row = e / 12, rem = e - row * 12, pl = rem >> 2, c = rem & 3;
        wvo[t] = 2 * (pl * plane + (n0 + row) * Kp + 8 * c);
        wlds[t] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + c) * 16;
    }
    // activation units of this thread: row = unit / 4, k-octet = unit % 4 (4 lanes = 128 contiguous bytes)
    int xvo[XL], xvo_last[XL], gvo[XL], gvo_last[XL], xlds[XL][3];
    int ubase[XL], uy0[XL], ux0[XL];
    bool uvalid[XL];
#pragma unroll
    for (int t = 0; t < XL; ++t) {
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
            xvo[t] = 4 * (mc * K + 8 * ko);
            xvo_last[t] = xvo[t] - 4 * (over > 0 ? over : 0);
            gvo[t] = GATE ? 4 * ((mc / HW) * K + 8 * ko) : 0;
            gvo_last[t] = gvo[t] - 4 * (over > 0 ? over : 0);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xlds[t][pl] = row * S6_ROWB + s6_chunk_pos(row, pl * 4 + ko) * 16;
    }

    u4 wr[2][WLOADS];      // weights, ring of two K-steps
    v4f xr[2][XL][2];      // raw activations, ring of two K-steps
    v4f gr[2][XL][2];      // GATE: raw squeeze-excite gate
    bool okr[2][XL];       // CONV: tap inside the image
    auto load_w = [&](int kstep, u4 (&w)[WLOADS]) {
#pragma unroll
        for (int t = 0; t < WLOADS; ++t) w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[t], 2 * kstep * BK, 0);
    };
    auto load_x = [&](int kstep, v4f (&x)[XL][2], v4f (&g)[XL][2], bool (&okf)[XL]) {
        const int kc = kstep * BK;
        if constexpr (CONV) {
            const int tap = kc / cg.Cin, ci0 = kc - tap * cg.Cin;      // block-uniform
            const int ky = tap / cg.ksize, kx = tap - ky * cg.ksize;
#pragma unroll
            for (int t = 0; t < XL; ++t) {
                const int iy = uy0[t] + ky * cg.dil, ix = ux0[t] + kx * cg.dil;
                const bool ok = uvalid[t] && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
                const int vo = 4 * (ubase[t] + ((ok ? iy : 0) * cg.W + (ok ? ix : 0)) * cg.Cin);
                x[t][0] = ld(rx, vo, 4 * ci0);
                x[t][1] = ld(rx, vo + 16, 4 * ci0);
                okf[t] = ok;
            }
        } else {
            const bool last = kstep == nk - 1;
#pragma unroll
            for (int t = 0; t < XL; ++t) {
                const int vo = last ? xvo_last[t] : xvo[t];
                x[t][0] = ld(rx, vo, 4 * kc);
                x[t][1] = ld(rx, vo + 16, 4 * kc);
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
    auto store_x = [&](const v4f (&x)[XL][2], const v4f (&g)[XL][2], const bool (&okf)[XL], int buf) {
#pragma unroll
        for (int t = 0; t < XL; ++t) {
            v4f lo = x[t][0], hi = x[t][1];
            if constexpr (GATE) { lo *= g[t][0]; hi *= g[t][1]; }
            if constexpr (CONV) {
                if (!okf[t]) { lo = (v4f){0.f, 0.f, 0.f, 0.f}; hi = lo; }
            }
            bf8 s0, s1, s2;
            split8(lo, hi, s0, s1, s2);
            *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][0]]) = s0;
            *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][1]]) = s1;
            *reinterpret_cast<bf8*>(&xsm[buf][xlds[t][2]]) = s2;
        }
    };

    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int rd0 = s6_chunk_pos(j, q) * 16, rd1 = s6_chunk_pos(j, 4 + q) * 16, rd2 = s6_chunk_pos(j, 8 + q) * 16;
    auto compute = [&](int buf) {
        bf8 xs[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned char* xp = xsm[buf] + ((wm * MT + mt) * 16 + j) * S6_ROWB;
            xs[mt][0] = *reinterpret_cast<const bf8*>(xp + rd0);
            xs[mt][1] = *reinterpret_cast<const bf8*>(xp + rd1);
            xs[mt][2] = *reinterpret_cast<const bf8*>(xp + rd2);
        }
        bf8 wf[2][3];                         // next tile's weight fragments in flight during this tile's MFMAs (see pw6)
        auto read_w = [&](int nt, bf8 (&f)[3]) {
            const unsigned char* wp = ws[buf] + ((wn * NT + nt) * 16 + j) * S6_ROWB;
            f[0] = *reinterpret_cast<const bf8*>(wp + rd0);
            f[1] = *reinterpret_cast<const bf8*>(wp + rd1);
            f[2] = *reinterpret_cast<const bf8*>(wp + rd2);
        };
        read_w(0, wf[0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + 1 < NT) read_w(nt + 1, wf[(nt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const bf8 w0 = wf[nt & 1][0], w1 = wf[nt & 1][1], w2 = wf[nt & 1][2];
            // the same six products in the same order as pw6: a result never depends on the kernel or tile chosen
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

    int m[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) m[mt] = m0 + (wm * MT + mt) * 16 + j;
    s6_epilogue<MT, NT>(acc, m, n0 + wn * NT * 16 + 4 * q, bias, R, Y, M, N, act, res_first);
}

// kind 0: pw6 (block = NW waves x MT*16 rows, NT*16 columns; wm = NW); kind 1: pw7 (WM x WN waves of MT x NT tiles)
struct S6Tile { int kind, wm, wn, mt, nt, ks, mblocks, nblocks; bool measured; };      // ks: K-steps per stage (pw6)
static S6Tile make_tile(int M, int N, int kind, int wm, int wn, int mt, int nt, int ks = 1) {
    const int bm = wm * mt * 16, bn = wn * nt * 16;
    return S6Tile{kind, wm, wn, mt, nt, ks, (M + bm - 1) / bm, (N + bn - 1) / bn, false};
}
static S6Tile make_tile6(int M, int N, int mt, int nt, int ks = 1) { return make_tile(M, N, 0, 4, 1, mt, nt, ks); }

// Heuristic tile (shapes nobody warmed up): the biggest per-wave pw6 tile that still fills the chip.
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
    return best;
}

#define DFD_S6_NT_CASES(OP) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8)
// pw7 instances: (WM, WN, MT, NT)
#define DFD_S7_CONFIGS(OP)                                                                       \
    OP(1, 4, 2, 1) OP(1, 4, 2, 2) OP(1, 4, 2, 3) OP(1, 4, 4, 1) OP(1, 4, 4, 2) OP(1, 4, 4, 3)     \
    OP(2, 2, 1, 2) OP(2, 2, 1, 3) OP(2, 2, 1, 4) OP(2, 2, 1, 6) OP(2, 2, 2, 2) OP(2, 2, 2, 3)     \
    OP(2, 2, 2, 4) OP(2, 2, 2, 6) OP(2, 2, 4, 2) OP(2, 2, 4, 3)

// Every instance the library can launch for a shape, in a fixed order (what the tuner measures and what
// dfd_set_option(h, "gemm_tile", i) indexes): pw6 with 4 waves (KS x MT x NT), pw6 with 8 waves (KS x NT), pw7.
static std::vector<S6Tile> s6_candidates(int M, int K, int N) {
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
    return out;
}

template <bool CONV, bool GATE>
static void s6_dispatch(const S6Tile& t, const float* X, const unsigned short* W3, const float* bias,
                        const float* gate, const float* R, float* Y, int M, int K, int N, int HW, int act,
                        const ConvGeom& g, int res_first, hipStream_t s) {
    const int grid = ((t.mblocks + 7) / 8) * 8 * t.nblocks;
    const int Kp = (K + S6_KPAD - 1) / S6_KPAD * S6_KPAD, plane = s6_np(N) * Kp;
    // the caller (s6_run) keeps every call below 2^31 bytes of activations: the kernels address X / gate with
    // 32-bit buffer offsets
    const unsigned xbytes = CONV ? (unsigned)((size_t)(M / (g.Ho * g.Wo)) * g.H * g.W * g.Cin * 4) : (unsigned)((size_t)M * K * 4);
    const unsigned gbytes = GATE ? (unsigned)((size_t)((M + HW - 1) / HW) * K * 4) : 0u;
    if (t.kind == 1) {
#define DFD_S7_CASE(WMV, WNV, MTV, NTV)                                                                              \
    if (t.wm == WMV && t.wn == WNV && t.mt == MTV && t.nt == NTV) {                                                  \
        hipLaunchKernelGGL((pw7_kernel<WMV, WNV, MTV, NTV, CONV, GATE>), dim3(grid), dim3(256), 0, s, X, W3, plane,  \
                           Kp, bias, gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes); \
        return;                                                                                                      \
    }
        DFD_S7_CONFIGS(DFD_S7_CASE)
#undef DFD_S7_CASE
        return;
    }
#define DFD_S6_LAUNCH(NTV, MTV, KSV)                                                                                 \
    hipLaunchKernelGGL((pw6_kernel<NTV, CONV, MTV, GATE, KSV>), dim3(grid), dim3(256), 0, s, X, W3, plane, Kp, bias, gate, \
                       R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes)
#define DFD_S6_LAUNCH8(NTV, KSV)                                                                                     \
    hipLaunchKernelGGL((pw6_kernel<NTV, CONV, 1, GATE, KSV, false, 8>), dim3(grid), dim3(512), 0, s, X, W3, plane, Kp, bias, \
                       gate, R, Y, M, K, N, HW, act, t.mblocks, t.nblocks, g, res_first, xbytes, gbytes)
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

template <bool CONV, bool GATE>
static void s6_measure(const std::vector<S6Tile>& cands, S6Tile* tile, const S6Key& key, const float* X,
                       const unsigned short* W3, const float* bias, const float* gate, const float* R, float* Y, int M,
                       int K, int N, int HW, int act, const ConvGeom& g, int res_first, hipStream_t s) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return; }
    float best_ms = 1e30f;
    for (const S6Tile& t : cands) {
        s6_dispatch<CONV, GATE>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
        float ms = 1e30f;
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {          // best of two groups of three: robust to a stray hiccup
            hipEventRecord(e0, s);
            for (int r = 0; r < 3; ++r)
                s6_dispatch<CONV, GATE>(t, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
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

template <bool CONV, bool GATE>
static void s6_run_one(S6Table* tab, const float* X, const unsigned short* W3, const float* bias, const float* gate,
                       const float* R, float* Y, int M, int K, int N, int HW, int act, const ConvGeom& g, int res_first,
                       hipStream_t s) {
    static const bool tune_env = !(getenv("DFD_S6_TUNE") && atoi(getenv("DFD_S6_TUNE")) == 0);
    S6Tile tile;
    if (tab && tab->force >= 0) {
        const std::vector<S6Tile> cands = s6_candidates(M, K, N);
        tile = cands[(size_t)tab->force % cands.size()];
    } else {
        // M in 8 buckets per octave: data-dependent row counts (the MTCNN candidate windows) share an entry
        int mkey = M;
        if (M > 64) {
            int sh = 0;
            while ((M >> sh) > 15) ++sh;
            mkey = ((M + (1 << sh) - 1) >> sh) << sh;
        }
        const S6Key key{mkey, K, N, (CONV ? 1 : 0) | (GATE ? 2 : 0) | (CONV ? (g.ksize << 8) | (g.stride << 4) : 0)};
        const bool tuning = tab && tab->tuning && tune_env;
        auto it = tab ? tab->tiles.find(key) : std::map<S6Key, S6Tile>::iterator();
        if (tab && it != tab->tiles.end() && (it->second.measured || !tuning)) {
            tile = it->second;
        } else {
            tile = pick_tile6(M, N);
            if (tuning) s6_measure<CONV, GATE>(s6_candidates(M, K, N), &tile, key, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
            if (tab) tab->tiles[key] = tile;
        }
        tile = make_tile(M, N, tile.kind, tile.wm, tile.wn, tile.mt, tile.nt, tile.ks);      // block counts for this call's M
    }
    s6_dispatch<CONV, GATE>(tile, X, W3, bias, gate, R, Y, M, K, N, HW, act, g, res_first, s);
}

bool split_gemm_supports(int K, int N) { return K % 8 == 0 && K >= 16 && split_weights_count(N, K) * 6 < (1ull << 31); }

bool launch_pointwise_split(S6Table* tab, const float* X, const unsigned short* W3, const float* bias, const float* gate,
                            const float* R, float* Y, int M, int K, int N, int HW, int act, hipStream_t s) {
    const ConvGeom none{};
    if (HW <= 0) HW = 1;
    const long long chunk = s6_chunk_rows(M, (long long)K * 4, gate ? HW : 1);
    if (chunk <= 0) return false;
    for (long long m0 = 0; m0 < M; m0 += chunk) {
        const int mc = (int)std::min<long long>(chunk, M - m0);
        const float* xc = X + (size_t)m0 * K;
        const float* rc = R ? R + (size_t)m0 * N : nullptr;
        float* yc = Y + (size_t)m0 * N;
        if (gate) s6_run_one<false, true>(tab, xc, W3, bias, gate + (size_t)(m0 / HW) * K, rc, yc, mc, K, N, HW, act, none, 0, s);
        else s6_run_one<false, false>(tab, xc, W3, bias, nullptr, rc, yc, mc, K, N, HW, act, none, 0, s);
    }
    return true;
}

bool launch_conv_gemm_split(S6Table* tab, const float* X, const unsigned short* W3, const float* bias, const float* R,
                            float* Y, int n_img, const ConvGeom& g, int Cout, int act, bool res_first, hipStream_t s) {
    if (g.Cin % S6_BK != 0) return false;                  // a K stage must not straddle two taps
    const int K = g.ksize * g.ksize * g.Cin;
    if (!split_gemm_supports(K, Cout)) return false;
    const long long in_img = (long long)g.H * g.W * g.Cin * 4, out_rows = (long long)g.Ho * g.Wo;
    const long long imgs = s6_chunk_rows(n_img, in_img, 1);        // "rows" = images of in_img bytes
    if (imgs <= 0) return false;
    for (long long i0 = 0; i0 < n_img; i0 += imgs) {
        const int ni = (int)std::min<long long>(imgs, n_img - i0);
        s6_run_one<true, false>(tab, X + (size_t)i0 * g.H * g.W * g.Cin, W3, bias, nullptr,
                                R ? R + (size_t)i0 * out_rows * Cout : nullptr, Y + (size_t)i0 * out_rows * Cout,
                                (int)(ni * out_rows), K, Cout, 1, act, g, res_first ? 1 : 0, s);
    }
    return true;
}

#ifdef S6_TRACE
extern "C" int dfd_debug_s6_trace(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s6_trace), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

}  // namespace dfd
