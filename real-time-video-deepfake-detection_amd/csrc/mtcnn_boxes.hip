// The box bookkeeping of the MTCNN cascade on the device (reference deepfake_detection.py:24-28,
// 376-380 -> facenet-pytorch detect_face / select_boxes / extract_face): what mtcnn_api.hip's host path does between the
// three networks - candidate boxes from the P-Net cells, NMS 0.5 per pyramid level and 0.7 across levels, regression,
// squaring, window clipping, R-/O-Net thresholds, NMS 0.7 (IoU) / 0.7 (Min), selection by probability, extract_face
// geometry and Pillow's bilinear coefficient tables - as one block per crop, so that a cascade step reads back two
// window counts and one result row per crop instead of candidate lists, and no host thread touches a box.
//
// Every formula is the host path's (which restates oracle/mtcnn_ref.py line by line) in the same float32 / float64
// operation order; this file is compiled with -ffp-contract=off so that no product is fused into a following add.
//   order      greedy NMS visits boxes by descending score; ties: torchvision's stable descending sort takes the earlier
//              box first (stages 1, 2), nms_numpy's reversed ascending argsort the later one (stage 3).  The sort key is
//              (~score bits) << 32 | tie-break index, sorted ascending by a bitonic network in LDS.
//   stage 1    one walk does both passes: box i that is not suppressed inside its level suppresses same-level boxes at
//              IoU > 0.5; if it is also alive across levels it is kept and suppresses any later box at IoU > 0.7 (a box
//              suppressed inside its level never takes part in the cross-level pass, as in the two separate passes).
//   walk       the boxes a thread owns (j = tid + k * 1024) stay in registers, suppression flags are bit words in LDS
//              (atomic OR), the walk jumps over suppressed boxes a word at a time; one barrier per visited box.
//   capacity   kCap1 candidates / kCap2 windows per crop (LDS); a crop beyond that raises the overflow flag and the
//              caller runs the whole step on the host path (exact either way).
#include <hip/hip_runtime.h>

#include "mtcnn_kernels.h"

namespace dfd {

namespace {

typedef unsigned long long u64;

constexpr int kNT = 1024;

__device__ __forceinline__ float smax(float a, float b) { return a < b ? b : a; }      // std::max / std::min (NaN: first operand)
__device__ __forceinline__ float smin(float a, float b) { return b < a ? b : a; }
// (int) of a float as x86 converts it: out of range and NaN give INT_MIN
__device__ __forceinline__ int f2i(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000; }

// slot of a flagged thread among the block's flagged threads in thread order; *total = their number (block-uniform)
__device__ __forceinline__ int block_ordered_slot(bool flag, int* wave_cnt, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 m = __ballot(flag);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kNT / 64; ++w) {
        const int c = wave_cnt[w];
        before += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return flag ? before + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// unordered append of a flagged thread's key (one LDS atomic per wave)
__device__ __forceinline__ void append_key(bool flag, u64 key, u64* keys, int cap, int* count) {
    const int lane = threadIdx.x & 63;
    const u64 m = __ballot(flag);
    if (!m) return;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (flag && slot < cap) keys[slot] = key;
}

// ascending bitonic sort of keys[0, np2) (np2 a power of two, padded with ~0)
__device__ void bitonic_sort(u64* keys, int np2) {
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (np2 >> 1); t += kNT) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const u64 a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    __syncthreads();
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// Greedy NMS over boxes already in visiting order.  MODE 0: stage 1 (levels: IoU > 0.5 inside a level, > 0.7 across);
// 1: IoU > thr (torchvision.ops.nms, areas without +1); 2: inter / min(area) with +1 on every extent, suppress unless
// <= thr (nms_numpy "Min").  get(i) -> (x1, y1, x2, y2); level(i) for MODE 0.  kept[0, return) = kept indices in order.
template <int R, int MODE, typename GetBox, typename GetLevel>
__device__ int nms_walk(int n, GetBox get, GetLevel level, float thr, u64* dead1, u64* dead2, unsigned short* kept) {
    const int tid = threadIdx.x;
    const float one = MODE == 2 ? 1.f : 0.f;
    float x1[R], y1[R], x2[R], y2[R], ar[R];
    int lv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = tid + k * kNT;
        x1[k] = y1[k] = x2[k] = y2[k] = ar[k] = 0.f;
        lv[k] = 0;
        if (j < n) {
            const float4 b = get(j);
            x1[k] = b.x; y1[k] = b.y; x2[k] = b.z; y2[k] = b.w;
            ar[k] = (b.z - b.x + one) * (b.w - b.y + one);
            lv[k] = MODE == 0 ? level(j) : 0;
        }
    }
    for (int w = tid; w < (n + 63) / 64; w += kNT) dead1[w] = dead2[w] = 0;
    __syncthreads();
    const u64* skip = MODE == 0 ? dead1 : dead2;
    unsigned my1 = 0, my2 = 0;
    int nk = 0, i = 0;
    while (true) {
        while (i < n) {                                              // next box not suppressed (inside its level)
            const u64 w = ~skip[i >> 6] & (~0ull << (i & 63));
            if (w) { i = (i & ~63) + __ffsll((long long)w) - 1; break; }
            i = (i & ~63) + 64;
        }
        if (i >= n) break;
        const bool k2 = MODE != 0 || !((dead2[i >> 6] >> (i & 63)) & 1ull);
        if (k2) {
            if (tid == 0) kept[nk] = (unsigned short)i;
            ++nk;
        }
        const float4 bi = get(i);
        const float ai = (bi.z - bi.x + one) * (bi.w - bi.y + one);
        const int li = MODE == 0 ? level(i) : 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int j = tid + k * kNT;
            if (j > i && j < n) {
                const float xx1 = smax(bi.x, x1[k]), yy1 = smax(bi.y, y1[k]);
                const float xx2 = smin(bi.z, x2[k]), yy2 = smin(bi.w, y2[k]);
                const float w = smax(0.f, xx2 - xx1 + one), hgt = smax(0.f, yy2 - yy1 + one);
                const float inter = w * hgt;
                bool kill1 = false, kill2;
                if (MODE == 2) {
                    const float o = inter / smin(ai, ar[k]);
                    kill2 = !(o <= thr);
                } else {
                    const float o = inter / (ai + ar[k] - inter);
                    if (MODE == 0) {
                        kill1 = lv[k] == li && o > 0.5f;
                        kill2 = k2 && o > thr;
                    } else {
                        kill2 = o > thr;
                    }
                }
                if (kill1 && !((my1 >> k) & 1u)) { my1 |= 1u << k; atomicOr(&dead1[j >> 6], 1ull << (j & 63)); }
                if (kill2 && !((my2 >> k) & 1u)) { my2 |= 1u << k; atomicOr(&dead2[j >> 6], 1ull << (j & 63)); }
            }
        }
        __syncthreads();
        ++i;
    }
    __syncthreads();
    return nk;
}

__device__ __forceinline__ void rerec(float& x1, float& y1, float& x2, float& y2) {
    const float hgt = y2 - y1, w = x2 - x1;
    const float l = smax(w, hgt);
    x1 = x1 + w * 0.5f - l * 0.5f;
    y1 = y1 + hgt * 0.5f - l * 0.5f;
    x2 = x1 + l;
    y2 = y1 + l;
}

__device__ __forceinline__ void bbreg(float& x1, float& y1, float& x2, float& y2, const float4 r) {
    const float w = x2 - x1 + 1.f, hgt = y2 - y1 + 1.f;
    const float a = x1 + r.x * w, b = y1 + r.y * hgt, c = x2 + r.z * w, d = y2 + r.w * hgt;
    x1 = a; y1 = b; x2 = c; y2 = d;
}

// pad(): truncate, clip to the image; the 1-based (y, ey, x, ex) become the 0-based window [y-1, ey) x [x-1, ex)
__device__ __forceinline__ bool window_of(float bx1, float by1, float bx2, float by2, int w, int hgt, MtSrcWindow* out) {
    int x = f2i(truncf(bx1)), y = f2i(truncf(by1)), ex = f2i(truncf(bx2)), ey = f2i(truncf(by2));
    if (x < 1) x = 1;
    if (y < 1) y = 1;
    if (ex > w) ex = w;
    if (ey > hgt) ey = hgt;
    if (!(ey > y - 1 && ex > x - 1)) return false;
    out->x = x - 1; out->y = y - 1; out->w = ex - (x - 1); out->h = ey - (y - 1);
    return true;
}

// ---- stage 1: the crop's P-Net cells at or above the threshold -> boxes for R-Net
__global__ __launch_bounds__(kNT) void mt_stage1_boxes_kernel(const MtCropGeo* __restrict__ crops, const MtLevelGeo* __restrict__ levels,
                                                              const float* __restrict__ prob, const float* __restrict__ reg, float thr,
                                                              MtRow* __restrict__ rows_seg, MtSrcWindow* __restrict__ wins_seg,
                                                              int* __restrict__ counts, int* __restrict__ meta, MtRow* __restrict__ tap_rows) {
    __shared__ u64 keys[kMtCap1];
    __shared__ ushort4 bx[kMtCap1];
    __shared__ unsigned short kept[kMtCap1];
    __shared__ u64 dead1[kMtCap1 / 64], dead2[kMtCap1 / 64];
    __shared__ int wave_cnt[kNT / 64];
    __shared__ int s_n, s_bad;
    const int tid = threadIdx.x, c = blockIdx.x;
    const MtCropGeo cg = crops[c];
    if (tid == 0) { s_n = 0; s_bad = 0; }
    __syncthreads();
    for (int l = 0; l < cg.nlevels; ++l) {
        const MtLevelGeo L = levels[cg.level0 + l];
        const int cells = L.oh > 0 && L.ow > 0 ? L.oh * L.ow : 0;
        // four strides of the level per step: the four loads are in flight together (one load, one ballot + LDS atomic per
        // step left the scan of a crop's ~26k cells as 26 serialized memory round trips - most of this kernel's 31 us)
        for (int base = 0; base < cells; base += 4 * kNT) {
            float p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * kNT + tid;
                p[k] = prob[L.cell_off + (i < cells ? i : cells - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * kNT + tid;
                const bool pass = i < cells && p[k] >= thr;
                append_key(pass, ((u64)(~__float_as_uint(p[k])) << 32) | ((unsigned)l << 27) | (unsigned)i, keys, kMtCap1, &s_n);
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > kMtCap1) {                                               // more candidates than the block holds: the host path takes the step
        if (tid == 0) { atomicOr(&meta[1], 1); counts[c] = 0; }
        return;
    }
    const int np2 = next_pow2(n > 1 ? n : 1);
    for (int k = n + tid; k < np2; k += kNT) keys[k] = ~0ull;
    bitonic_sort(keys, np2);
    // generateBoundingBox: cell (y, x) of a level at `scale` -> floor((2 x + 1) / scale) .. floor((2 x + 12) / scale)
    for (int k = tid; k < n; k += kNT) {
        const unsigned lc = (unsigned)keys[k], l = lc >> 27, i = lc & 0x7FFFFFFu;
        const MtLevelGeo L = levels[cg.level0 + l];
        const unsigned y = i / (unsigned)L.ow, x = i - y * (unsigned)L.ow;
        const float fs = L.scale;
        const float fx1 = floorf((2.f * (float)x + 1.f) / fs), fy1 = floorf((2.f * (float)y + 1.f) / fs);
        const float fx2 = floorf((2.f * (float)x + 12.f) / fs), fy2 = floorf((2.f * (float)y + 12.f) / fs);
        if (!(fx2 < 65536.f && fy2 < 65536.f)) s_bad = 1;             // (coordinates are kept as exact 16-bit integers)
        bx[k] = make_ushort4((unsigned short)fx1, (unsigned short)fy1, (unsigned short)fx2, (unsigned short)fy2);
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { atomicOr(&meta[1], 1); counts[c] = 0; }
        return;
    }
    auto get = [&](int i) { const ushort4 b = bx[i]; return make_float4((float)b.x, (float)b.y, (float)b.z, (float)b.w); };
    auto lvl = [&](int i) { return (int)(((unsigned)keys[i]) >> 27); };
    const int nk = nms_walk<kMtCap1 / kNT, 0>(n, get, lvl, 0.7f, dead1, dead2, kept);
    // regression (on the un-incremented extents, as the package's first stage does), squaring, window for R-Net
    int nlive = 0;
    for (int r0 = 0; r0 < nk; r0 += kNT) {
        const int idx = r0 + tid;
        bool valid = false;
        MtRow row{};
        MtSrcWindow win{};
        if (idx < nk) {
            const int i = kept[idx];
            const u64 key = keys[i];
            const unsigned lc = (unsigned)key, l = lc >> 27, cell = lc & 0x7FFFFFFu;
            const float4 b = get(i);
            const float4 r = *reinterpret_cast<const float4*>(reg + (levels[cg.level0 + l].cell_off + cell) * 4);
            const float regw = b.z - b.x, regh = b.w - b.y;
            float nx1 = b.x + r.x * regw, ny1 = b.y + r.y * regh, nx2 = b.z + r.z * regw, ny2 = b.w + r.w * regh;
            rerec(nx1, ny1, nx2, ny2);
            row = MtRow{nx1, ny1, nx2, ny2, __uint_as_float(~(unsigned)(key >> 32))};
            if (tap_rows && c == 0) tap_rows[idx] = row;
            win.src = cg.src; win.stride = cg.stride;
            valid = window_of(nx1, ny1, nx2, ny2, cg.w, cg.h, &win);
        }
        int tot;
        const int slot = block_ordered_slot(valid, wave_cnt, &tot);
        if (valid && nlive + slot < kMtCap2) {
            rows_seg[cg.seg_off + nlive + slot] = row;
            wins_seg[cg.seg_off + nlive + slot] = win;
        }
        nlive += tot;
    }
    if (tid == 0) {
        if (nlive > kMtCap2) { atomicOr(&meta[1], 1); nlive = 0; }
        counts[c] = nlive;
        if (c == 0) meta[2] = nk;
    }
}

// per-crop segments -> one list in crop order; first_out[c] = number of windows of the crops before c, meta[0] = total.
// Segment of crop c: at the crop's own offset (stage 1) or at seg_first[c] (stage 2 writes where the crop's input began).
__global__ __launch_bounds__(256) void mt_compact_kernel(const int* __restrict__ counts, int n, const MtCropGeo* __restrict__ crops,
                                                         const int* __restrict__ seg_first, const MtRow* __restrict__ rows_seg,
                                                         const MtSrcWindow* __restrict__ wins_seg, MtRow* __restrict__ rows_out,
                                                         MtSrcWindow* __restrict__ wins_out, int* __restrict__ first_out,
                                                         int* __restrict__ meta) {
    __shared__ int part[4];
    const int tid = threadIdx.x, c = blockIdx.x;
    int s = 0;
    for (int k = tid; k < c; k += 256) s += counts[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const int off = part[0] + part[1] + part[2] + part[3];
    const int m = counts[c];
    const long long seg = seg_first ? (long long)seg_first[c] : crops[c].seg_off;
    for (int k = tid; k < m; k += 256) {
        rows_out[off + k] = rows_seg[seg + k];
        wins_out[off + k] = wins_seg[seg + k];
    }
    if (tid == 0) {
        first_out[c] = off;
        if (c == 0) meta[3] = m;
        if (c == n - 1) {
            first_out[n] = off + m;
            meta[0] = off + m;
        }
    }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (src/libImaging/Resample.c), output index
// xx of a 160-wide axis read from `in_size` pixels; coeff [160][ksize], bounds [160][2]
__device__ void pil_axis(int in_size, int xx, int ksize, int* __restrict__ coeff, int* __restrict__ bounds) {
    const double scale = (double)in_size / 160;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fs;
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
        ww += w;
    }
    for (int x = 0; x < ksize; ++x) {
        int cv = 0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            const double v = ww != 0.0 ? w / ww : w;
            cv = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
        }
        coeff[(size_t)xx * ksize + x] = cv;
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int pil_ksize(int in_size) {
    const double scale = (double)in_size / 160;
    const double fs = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(1.0 * fs) * 2 + 1;
}

// ---- stages 2 / 3: network outputs of the crop's windows -> boxes for O-Net (2) / the selected face and its
// extract_face job (3)
template <int STAGE>
__global__ __launch_bounds__(kNT) void mt_refine_boxes_kernel(const MtCropGeo* __restrict__ crops, const int* __restrict__ first,
                                                              const MtRow* __restrict__ rows_in, const float* __restrict__ prob,
                                                              const float* __restrict__ reg, float thr_p, float thr_nms,
                                                              MtRow* __restrict__ rows_seg, MtSrcWindow* __restrict__ wins_seg,
                                                              int* __restrict__ counts, MtFaceJob* __restrict__ jobs,
                                                              float* __restrict__ results, int* __restrict__ tables,
                                                              MtRow* __restrict__ tap_rows, int* __restrict__ meta) {
    __shared__ u64 keys[kMtCap2];
    __shared__ float4 bx[kMtCap2];
    __shared__ unsigned short kept[kMtCap2];
    __shared__ u64 dead1[1], dead2[kMtCap2 / 64];
    __shared__ int wave_cnt[kNT / 64];
    __shared__ int s_n;
    __shared__ MtFaceJob s_job;
    const int tid = threadIdx.x, c = blockIdx.x;
    const MtCropGeo cg = crops[c];
    const int lo = first[c], m = first[c + 1] - lo;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int base = 0; base < m; base += kNT) {
        const int k = base + tid;
        const float p = k < m ? prob[lo + k] : 0.f;
        const bool pass = k < m && p > thr_p;                        // thresholds[1], thresholds[2]: strict
        append_key(pass, ((u64)(~__float_as_uint(p)) << 32) | (STAGE == 2 ? (unsigned)k : 0xFFFFFFFFu - (unsigned)k), keys, kMtCap2, &s_n);
    }
    __syncthreads();
    const int n = s_n;                                               // <= m <= kMtCap2 (stage 1 checked)
    const int np2 = next_pow2(n > 1 ? n : 1);
    for (int k = n + tid; k < np2; k += kNT) keys[k] = ~0ull;
    bitonic_sort(keys, np2);
    auto src_of = [&](int i) { const unsigned t = (unsigned)keys[i]; return (int)(STAGE == 2 ? t : 0xFFFFFFFFu - t); };
    for (int i = tid; i < n; i += kNT) {
        const int k = src_of(i);
        const MtRow r = rows_in[lo + k];
        float x1 = r.x1, y1 = r.y1, x2 = r.x2, y2 = r.y2;
        if (STAGE == 3) bbreg(x1, y1, x2, y2, *reinterpret_cast<const float4*>(reg + (size_t)(lo + k) * 4));     // before the NMS
        bx[i] = make_float4(x1, y1, x2, y2);
    }
    __syncthreads();
    auto get = [&](int i) { return bx[i]; };
    auto lvl = [&](int) { return 0; };
    const int nk = nms_walk<kMtCap2 / kNT, STAGE == 2 ? 1 : 2>(n, get, lvl, thr_nms, dead1, dead2, kept);
    if (STAGE == 2) {
        int nlive = 0;
        for (int r0 = 0; r0 < nk; r0 += kNT) {
            const int idx = r0 + tid;
            bool valid = false;
            MtRow row{};
            MtSrcWindow win{};
            if (idx < nk) {
                const int i = kept[idx], k = src_of(i);
                const float4 b = bx[i];
                float x1 = b.x, y1 = b.y, x2 = b.z, y2 = b.w;
                bbreg(x1, y1, x2, y2, *reinterpret_cast<const float4*>(reg + (size_t)(lo + k) * 4));
                rerec(x1, y1, x2, y2);
                row = MtRow{x1, y1, x2, y2, __uint_as_float(~(unsigned)(keys[i] >> 32))};
                if (tap_rows && c == 0) tap_rows[idx] = row;
                win.src = cg.src; win.stride = cg.stride;
                valid = window_of(x1, y1, x2, y2, cg.w, cg.h, &win);
            }
            int tot;
            const int slot = block_ordered_slot(valid, wave_cnt, &tot);
            if (valid) {
                rows_seg[lo + nlive + slot] = row;
                wins_seg[lo + nlive + slot] = win;
            }
            nlive += tot;
        }
        if (tid == 0) {
            counts[c] = nlive;
            if (c == 0) meta[2] = nk;
        }
        return;
    }
    // stage 3: rows in kept order; select_boxes(method="probability"): np.argsort(probs)[::-1][0] = the LAST row of the
    // highest probability; extract_face(margin 0) geometry; resize tables
    if (tap_rows && c == 0)
        for (int idx = tid; idx < nk; idx += kNT) {
            const int i = kept[idx];
            const float4 b = bx[i];
            tap_rows[idx] = MtRow{b.x, b.y, b.z, b.w, __uint_as_float(~(unsigned)(keys[i] >> 32))};
        }
    if (tid == 0) {
        if (c == 0) meta[2] = nk;
        MtFaceJob j{cg.src, cg.stride, 0, 0, 160, 160, 0, 0, 0, 0, 0, 0, cg.tmp_off, 0};
        float* res = results + (size_t)c * 8;
        for (int k = 0; k < 8; ++k) res[k] = 0.f;
        if (nk > 0) {
            int best = 0;
            const unsigned top = (unsigned)(keys[kept[0]] >> 32);
            while (best + 1 < nk && (unsigned)(keys[kept[best + 1]] >> 32) == top) ++best;
            const int i = kept[best];
            const float4 b = bx[i];
            res[1] = 1.f;
            res[2] = b.x; res[3] = b.y; res[4] = b.z; res[5] = b.w; res[6] = __uint_as_float(~top);
            const int x1 = f2i(smax(b.x, 0.f)), y1 = f2i(smax(b.y, 0.f));
            const int x2 = f2i(smin(b.z, (float)cg.w)), y2 = f2i(smin(b.w, (float)cg.h));
            if (!(x2 <= x1 || y2 <= y1)) {
                j.x1 = x1; j.y1 = y1; j.cw = x2 - x1; j.ch = y2 - y1;
                const int kmx = pil_ksize(cg.w), kmy = pil_ksize(cg.h);             // table capacities (cw <= w, ch <= h)
                j.cx = cg.tab_off; j.bx = j.cx + 160 * kmx; j.cy = j.bx + 320; j.by = j.cy + 160 * kmy;
                j.kx = j.cw != 160 ? pil_ksize(j.cw) : 0;
                j.ky = j.ch != 160 ? pil_ksize(j.ch) : 0;
                j.found = 1;
                res[0] = 1.f;
            }
        }
        s_job = j;
        jobs[c] = j;
    }
    __syncthreads();
    const MtFaceJob j = s_job;
    if (!j.found) return;
    if (tid < 160) {
        if (j.cw != 160) pil_axis(j.cw, tid, j.kx, tables + j.cx, tables + j.bx);
    } else if (tid < 320) {
        if (j.ch != 160) pil_axis(j.ch, tid - 160, j.ky, tables + j.cy, tables + j.by);
    }
}

}  // namespace

void launch_mt_stage1_boxes(const MtCropGeo* crops, const MtLevelGeo* levels, int n, const float* prob, const float* reg, float thr,
                            MtRow* rows_seg, MtSrcWindow* wins_seg, int* counts, int* meta, MtRow* tap_rows, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mt_stage1_boxes_kernel, dim3(n), dim3(kNT), 0, s, crops, levels, prob, reg, thr, rows_seg, wins_seg, counts, meta,
                       tap_rows);
}

void launch_mt_compact(const int* counts, int n, const MtCropGeo* crops, const int* seg_first, const MtRow* rows_seg,
                       const MtSrcWindow* wins_seg, MtRow* rows_out, MtSrcWindow* wins_out, int* first_out, int* meta, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mt_compact_kernel, dim3(n), dim3(256), 0, s, counts, n, crops, seg_first, rows_seg, wins_seg, rows_out, wins_out,
                       first_out, meta);
}

void launch_mt_refine_boxes(int stage, const MtCropGeo* crops, const int* first, int n, const MtRow* rows_in, const float* prob,
                            const float* reg, float thr_p, float thr_nms, MtRow* rows_seg, MtSrcWindow* wins_seg, int* counts,
                            MtFaceJob* jobs, float* results, int* tables, MtRow* tap_rows, int* meta, hipStream_t s) {
    if (n <= 0) return;
    if (stage == 2)
        hipLaunchKernelGGL(mt_refine_boxes_kernel<2>, dim3(n), dim3(kNT), 0, s, crops, first, rows_in, prob, reg, thr_p, thr_nms, rows_seg,
                           wins_seg, counts, jobs, results, tables, tap_rows, meta);
    else
        hipLaunchKernelGGL(mt_refine_boxes_kernel<3>, dim3(n), dim3(kNT), 0, s, crops, first, rows_in, prob, reg, thr_p, thr_nms, rows_seg,
                           wins_seg, counts, jobs, results, tables, tap_rows, meta);
}

}  // namespace dfd
