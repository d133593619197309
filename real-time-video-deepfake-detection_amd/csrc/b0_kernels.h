// Launchers for the EfficientNet-B0 kernels (gfx950).  All activations are NHWC fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace dfd {

// activation storage: float, or bf16_t when the handle runs with "bf16_activations" (kernel_util.h)
typedef __bf16 bf16_t;

enum Act { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_PRELU = 3 };   // PRELU: bias points at [N bias][N slope]

// Squeeze-excite finished inside the depthwise-family launch: the block that completes an image's pool partials last
// (an agent-scope counter per image) computes mean -> FC(c_se) + swish -> FC(C) + sigmoid and writes gate[n][C] - the
// separate se_kernel launch (16 per forward, ~9 us each plus a launch gap on either side) goes away.
// counter == null: off (launch_se does it) - the default: measured at batch 256 the fused form is SLOWER (3.59 vs 3.04 ms
// per step).  The gate is three dependent L2 round trips (pool partials -> FC1 -> FC2) wherever it runs; inside the
// depthwise launch it adds the agent-scope acquire (~2-7 us on a CU that holds several blocks) in front of them and runs
// on 4-8 waves instead of 16, and the launch cannot end before the last image's tail has.  Option "fuse_se" keeps it testable.
struct SeTail {
    const float *w1 = nullptr, *b1 = nullptr, *w2t = nullptr, *b2 = nullptr;
    float* gate = nullptr;
    unsigned* counter = nullptr;     // [n images], zero between launches (the last block of an image resets its entry)
    float inv_hw = 0.f;
    int c_se = 0;
};

// stem: 3x3 stride-2 conv, NCHW (n,3,224,224) -> NHWC (n,112,112,32), folded BN + swish.
template <typename XT>
void launch_stem(const float* x_nchw, const float* w /*[3][3][3][32]*/, const float* b,
                 XT* y, int n, hipStream_t s);

// stem + block-0 depthwise fused (the stem activation stays in LDS); stem_out may be null, or a buffer
// [n][112][112][32] that receives a copy of the stem activation for parity taps.  The tile count (98) is
// that of launch_depthwise for block 0.
// (ws3: the stem weights [32][27] as three bf16 planes, `plane` elements apart, rows Kp long: the 3x3x3 -> 32 stem conv
// runs on the bf16 MFMA with split-precision operands, K = 27 padded to 32)
template <typename XT>
void launch_stem_dw(const float* x_nchw, const unsigned short* ws3, int plane, int Kp, const float* bs, const float* Wd,
                    const float* bd, XT* Y, float* P, XT* stem_out, int n, int* tiles, hipStream_t s, const SeTail& se = SeTail());

// pointwise conv as GEMM: Y[m][o] = act( sum_k X[m][k]*gate[m/HW][k] * W[o][k] + b[o] ) + R[m][o]
// gate / R may be null.  X rows have stride K, Y/R rows stride N.
void launch_pointwise(const float* X, const float* W, const float* bias, const float* gate,
                      const float* R, float* Y, int M, int K, int N, int HW, int act,
                      hipStream_t s);

// geometry of an implicit-GEMM convolution on NHWC fp32 (square kernel, same stride/pad/dilation
// in both directions)
struct ConvGeom {
    int H = 0, W = 0, Ho = 0, Wo = 0, Cin = 0, ksize = 1, stride = 1, pad = 0, dil = 1;
};

// k x k convolution as implicit GEMM on the same MFMA kernel (C_in % 32 == 0; returns false
// otherwise): Y[n][oy][ox][o] = act( conv(X, W[o][ky][kx][ci]) + b[o] (+ R before act if res_first) )
bool launch_conv_gemm(const float* X, const float* W, const float* bias, const float* R, float* Y,
                      int n_img, const ConvGeom& g, int Cout, int act, bool res_first, hipStream_t s);

// Squeeze-excite computed by the projection GEMM itself (option "se_in_proj", round 4): every block of the gated 1x1
// conv first evaluates mean -> FC(c_se) + swish -> FC(C) + sigmoid for the (at most four) images its rows belong to -
// the arithmetic of se_kernel operation by operation, so the gate bits do not depend on who computes them - writes the
// gate rows to `gate` and goes on as before.  The 10-12 us se_kernel launch between the depthwise launch and the
// projection (pure latency: three dependent L2 round trips on an otherwise idle chip, plus a launch gap on either
// side) disappears for every layer whose pool sums are final per image (tiles == 1: the whole-image launches of
// blocks 6-10 / 12-15).  P == nullptr: off (`gate` was written by launch_se).
struct SeFuse {
    const float* P = nullptr;        // [n][C] final per-image channel sums of the depthwise output
    const float *w1 = nullptr, *b1 = nullptr, *w2t = nullptr, *b2 = nullptr;
    float inv_hw = 0.f;
    int c_se = 0;
    int tiles = 1;                   // pool partials per image: P is [n][tiles][C]
    bool thin = false;               // no se_kernel ran: the call must go to pw8_kernel, whose blocks evaluate the gate of the
                                     // (at most two) images they meet in a light prologue (C <= 256, c_se <= 16; any `tiles`)
};
constexpr int SE_THIN_MAX_C = 256, SE_THIN_MAX_SE = 16;
inline bool se_thin_supported(int C, int c_se) { return C <= SE_THIN_MAX_C && c_se >= 1 && c_se <= SE_THIN_MAX_SE; }
constexpr int SE_FUSE_MAX_IMG = 4;   // images one GEMM block (<= 128 rows) can touch when an image has >= 49 rows
constexpr int SE_FUSE_MAX_SE = 48;

// Split-precision variants (gemm_split.hip): same contracts, the weight operand is the three-plane bf16 split
// of W [N][K] written by launch_split_weights (out: 3 * split_weights_count(N, K) bf16, zero-padded planes).
// fp32-exact products, fp32 accumulate.
size_t split_weights_count(int N, int K);
void launch_split_weights(const float* W, unsigned short* out, int N, int K, hipStream_t s, bool transposed = false);
bool split_gemm_supports(int K, int N);
// can a gated 1x1 conv of this shape run on pw8_kernel (weights resident in LDS; images of HW rows, HW % 16 == 0)?
bool split_gemm_thin_supports(int K, int N, int HW);
// `tab` is the handle's tile table (null: heuristic tile, nothing remembered).  A shape the table has no
// measurement for runs the heuristic tile unless the table is in tuning mode (dfd_warmup), where every
// candidate is timed on the caller's operands - the only place these launchers synchronise.  Calls whose
// activations span 2^31 bytes or more are issued as several launches over whole images (32-bit buffer offsets
// inside the kernels); false = shape not supported.
struct S6Table;
S6Table* s6_table_create();
void s6_table_destroy(S6Table* t);
void s6_table_set_force(S6Table* t, int idx);      // >= 0: run candidate idx (mod count) everywhere; -1: normal
void s6_table_set_tuning(S6Table* t, bool on);
int s6_table_measured(const S6Table* t);           // shapes with a measured tile
int s6_max_candidates();                           // upper bound of the candidate count over all shapes
long long s6_chunk_rows(long long M, long long row_bytes, long long HW);
std::string s6_table_export(const S6Table* t);                       // measured entries as text
int s6_table_import(S6Table* t, const char* text, size_t len);       // -> entries accepted
// XT = float: every fp32 activation is split into three bf16 terms in registers (planes must be 3);
// XT = bf16_t: bf16 activation storage in and out, `planes` = 3 (fp32-exact weights) or 1 (bf16 weights).
template <typename XT>
bool launch_pointwise_split(S6Table* tab, const XT* X, const unsigned short* W3, const float* bias, const float* gate,
                            const XT* R, XT* Y, int M, int K, int N, int HW, int act, int planes, hipStream_t s,
                            const SeFuse* se = nullptr);
// se usable with this shape: an image is at least 49 rows, so a block of <= 128 rows touches <= SE_FUSE_MAX_IMG images
inline bool se_fuse_supported(int HW, int c_se) { return HW >= 49 && c_se >= 1 && c_se <= SE_FUSE_MAX_SE; }
template <typename XT>
bool launch_conv_gemm_split(S6Table* tab, const XT* X, const unsigned short* W3, const float* bias, const XT* R,
                            XT* Y, int n_img, const ConvGeom& g, int Cout, int act, bool res_first, int planes, hipStream_t s);

// depthwise kxk conv (k in {3,5}, stride in {1,2}, TF-SAME pad) + folded BN + swish, and
// per-tile channel sums for the squeeze-excite pool: P[n][tile][c].  Returns the tile count
// through *tiles.  Only the 16 shape classes of EfficientNet-B0 at 224x224 are instantiated.
template <typename XT>
bool launch_depthwise(const XT* X, const float* W /*[k][k][C]*/, const float* bias, XT* Y,
                      float* P, int n, int H, int C, int k, int stride, int pad_lo,
                      int* tiles, hipStream_t s, const SeTail& se = SeTail());
int depthwise_tiles(int H, int C, int k, int stride);
// MBConv front half in one kernel: 1x1 expand (+BN+swish) computed per LDS halo tile on the bf16 MFMA with
// split-precision operands (We3 = the three bf16 planes of We [C][Cin] from launch_split_weights, `plane`
// elements apart, rows Kp long), then the depthwise conv as above.  Xin: block input [n][H][H][Cin]; be [C].
// Returns false when no instantiation covers the shape (callers then run the GEMM + launch_depthwise).
template <typename XT>
bool launch_mbconv_front(const XT* Xin, int Cin, const unsigned short* We3, int plane, int Kp, const float* Wef, const float* be,
                         const float* Wd, const float* bd, XT* Y, float* P, int n, int H, int C, int k, int stride,
                         int pad_lo, int* tiles, hipStream_t s, const SeTail& se = SeTail(), bool late = false);
// largest pool-tile count of the fused variants, -1: none.  late: also the whole-image launches of blocks 6-15
// (mbconv_late_kernel, option "fuse_late")
int mbconv_tiles(int H, int C, int k, int stride, int Cin, bool late = false);

// squeeze-excite gate: mean over tiles*pixels -> FC(c_se)+swish -> FC(C)+sigmoid.
void launch_se(const float* P, int tiles, float inv_hw, const float* w1, const float* b1,
               const float* w2t, const float* b2, float* gate, int n, int C, int c_se,
               hipStream_t s);

// global average pool over hw pixels: [n][hw][C] -> [n][C]
template <typename XT>
void launch_avgpool(const XT* X, float* Y, int n, int hw, int C, hipStream_t s);
void launch_bf16_to_f32(const bf16_t* x, float* y, size_t n, hipStream_t s);

}  // namespace dfd
