// Launchers of the MTCNN-stage kernels (mtcnn_kernels.hip).  Activations NHWC fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dfd {

struct MtWindow { int x, y, w, h; };      // source window in pixels
struct MtSrcWindow { const uint8_t* src; long long stride; int x, y, w, h; };     // ... of the image at `src`
// one pyramid level of one crop: source image, output size, offset of its [oh][ow][3] block in the input arena
struct MtLevel { const uint8_t* src; long long stride; int h, w, oh, ow; long long out_off; };
// one image of a ragged layer launch: offsets (floats) into the input / output arenas, input size
struct MtItem { long long in_off, out_off; int ih, iw; };

int mt_pool_out(int in, int k, int st);   // MaxPool2d(k, st, ceil_mode=True) output size
void launch_mt_maxpool(const float* x, float* y, int n, int ih, int iw, int c, int k, int st, hipStream_t s);
// both output heads of R-Net / O-Net in one launch: prob [n] = softmax(f . w1 + b1)[1], reg [n][4] = f . w2 + b2 (w1 [in][2], w2 [in][4])
void launch_mt_heads(const float* f, const float* w1, const float* b1, const float* w2, const float* b2, float* prob, float* reg,
                     int n, int in, hipStream_t s);
// ragged variants: `n` images of different sizes per launch; pre[n + 1] = running total of output elements
void launch_mt_area_resize_ragged(const MtLevel* lv_dev, const long long* pre_dev, int n, long long total, float* dst,
                                  hipStream_t s);
// a P-Net cell whose face probability reaches the stage-1 threshold: appended (unordered, atomic counter) by the conv3
// launch so that the host downloads candidates instead of whole probability / regression maps
struct MtCand { unsigned cell; float p; float r[4]; };
struct MtPnetHeads {
    const float *w41, *b41, *w42, *b42;
    float *prob, *reg;
    MtCand* cand;            // [cand_cap]
    unsigned* cand_count;    // zeroed by the caller; may exceed cand_cap (then only cand_cap records were written)
    unsigned cand_cap;
    float thr;
};
// P-Net conv2 / conv3, register-blocked (a thread owns 4 output pixels, weights in LDS); false = no instance for that
// shape.  With `heads` (conv3) the 1x1 heads + softmax run in the same launch: prob [cell], reg [cell][4], candidates;
// y is not written.
// the same two layers on the bf16 MFMA: w3 = three exact bf16 planes of the weights in the row-contiguous K layout
// [CO][ky * KR + kx * CI + ci] (KR = 32 / 48; "mtcnn.pnet.conv2.wm" / "conv3.wm" of the blob), rows Kp long
bool launch_mt_pnet_mfma(const float* x, const unsigned short* w3, int plane, int Kp, const float* b, const float* slope, float* y,
                         const MtItem* items_dev, const long long* pre_dev, int n, long long total, int ci, int co,
                         const MtPnetHeads* heads, hipStream_t s);
bool launch_mt_convpx_ragged(const float* x, const float* w, const float* b, const float* slope, float* y,
                             const MtItem* items_dev, const long long* pre_dev, int n, long long total, int ci, int co, int k,
                             const MtPnetHeads* heads, hipStream_t s);
// P-Net conv1 (3 -> 10) + PReLU + MaxPool2d(2, 2, ceil_mode=True), ragged; items = {input offset, pooled offset, level h, w},
// pre / total = running totals of pooled elements; w_padded: the [3][3][3][10] weights in a buffer of >= 272 floats
void launch_mt_pnet_conv1_pool(const float* x, const float* w_padded, const float* b, const float* slope, float* y,
                               const MtItem* items_dev, const long long* pre_dev, int n, long long total, hipStream_t s);
// conv 3x3 (3 -> 32) + bias + PReLU + MaxPool2d(3, 2, ceil_mode=True) of `n` maps [ih][iw][3] -> [ph][pw][32]
bool launch_mt_conv1_pool(const float* x, const float* w, const float* b, const float* slope, float* y, int n, int ih, int iw,
                          int co, hipStream_t s);
void launch_mt_area_resize_multi(const MtSrcWindow* win_dev, int n, int oh, int ow, float* dst, hipStream_t s);
// extract_face of every crop of a step: window (x1, y1, cw x ch) of the image at src -> 160 x 160 x 3 (Pillow bilinear,
// horizontal then vertical pass; coefficient / bounds tables at the given int offsets of one table array); found = 0:
// the slot is zero-filled.  tmp_off: byte offset of the crop's [ch][160][3] intermediate.
struct MtFaceJob { const uint8_t* src; long long stride; int x1, y1, cw, ch, kx, ky, cx, bx, cy, by; long long tmp_off; int found; };
void launch_mt_extract_faces(const MtFaceJob* jobs_dev, int n, const int* tables_dev, uint8_t* faces, uint8_t* tmp, hipStream_t s);
void launch_mt_face_chw(const uint8_t* bgr, float* out, int hw, hipStream_t s);

// ---- box bookkeeping on the device (mtcnn_boxes.hip): one block per crop and stage
constexpr int kMtCap1 = 8192;    // P-Net candidates of one crop the stage-1 block holds
constexpr int kMtCap2 = 4096;    // windows of one crop the stage-2 / stage-3 blocks hold
// a crop of the step: image, its run of pyramid levels in the level table, offset of its segment in the stage-1 output
// arenas, offset (ints) of its resize tables, offset (bytes) of its horizontal-pass intermediate
struct MtCropGeo { const uint8_t* src; long long stride; int h, w, level0, nlevels; long long seg_off; long long tmp_off; int tab_off, pad; };
struct MtLevelGeo { long long cell_off; int oh, ow; float scale; int pad; };      // P-Net output grid of a level, first cell, (float) scale
struct MtRow { float x1, y1, x2, y2, score; };
// meta[0] = windows of all crops, meta[3] = windows of crop 0 (compact kernel), meta[1] = overflow flag (zeroed by the
// caller), meta[2] = rows of crop 0
void launch_mt_stage1_boxes(const MtCropGeo* crops, const MtLevelGeo* levels, int n, const float* prob, const float* reg, float thr,
                            MtRow* rows_seg, MtSrcWindow* wins_seg, int* counts, int* meta, MtRow* tap_rows, hipStream_t s);
void launch_mt_compact(const int* counts, int n, const MtCropGeo* crops, const int* seg_first, const MtRow* rows_seg,
                       const MtSrcWindow* wins_seg, MtRow* rows_out, MtSrcWindow* wins_out, int* first_out, int* meta, hipStream_t s);
// stage 2: R-Net prob / reg of the windows first[c] .. first[c + 1] -> rows / windows at first[c] of the segment arenas, counts;
// stage 3: O-Net prob / reg -> jobs[c], results[c][8] = {found, has box, x1, y1, x2, y2, prob, 0}, resize tables
void launch_mt_refine_boxes(int stage, const MtCropGeo* crops, const int* first, int n, const MtRow* rows_in, const float* prob,
                            const float* reg, float thr_p, float thr_nms, MtRow* rows_seg, MtSrcWindow* wins_seg, int* counts,
                            MtFaceJob* jobs, float* results, int* tables, MtRow* tap_rows, int* meta, hipStream_t s);

}  // namespace dfd
