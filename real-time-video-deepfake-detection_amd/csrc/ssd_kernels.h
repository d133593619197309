// Launchers of the SSD detector kernels that are not plain convolutions.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfd {

// the six detection heads and the prior table, passed by value to the decode kernel
struct SsdHeads {
    const float* out[6];       // head conv outputs [img][cell][p*6]  (p*4 loc, then p*2 conf)
    const float* prior_tab;    // [n_priors][4] normalised (xmin,ymin,xmax,ymax)
    int first[7];              // index of each source's first prior; first[6] = n_priors
    int priors[6];             // priors per cell
    int map[6];                // map edge
};

// 7x7 s2 pad 3 conv from the resized u8 BGR image; inside the image a pixel enters as x * in_scale[c] + in_shift[c]
// (blobFromImage's mean subtraction, and a BatchNorm/Scale on the data blob where the prototxt has one), outside
// it as 0 (Caffe pads the transformed blob)
void launch_ssd_conv1(const uint8_t* img, const float* w, const float* b, float* y, int n, const float in_scale[3],
                      const float in_shift[3], bool relu, hipStream_t s);
// the same convolution on the bf16 MFMA against the three exact weight planes [32][Kp] of split_weights(w, 32, 147, transposed)
void launch_ssd_conv1_mfma(const uint8_t* img, const unsigned short* w3, int plane, int Kp, const float* b, float* y, int n,
                           const float in_scale[3], const float in_shift[3], bool relu, hipStream_t s);
// conv1 (as above) + ReLU + the 3x3 stride-2 ceil-mode max pool behind it in one launch: y_pool [n][75][75][32]; the
// 150 x 150 conv map is not materialised
bool launch_ssd_conv1_pool(const uint8_t* img, const unsigned short* w3, int plane, int Kp, const float* b, float* y_pool, int n,
                           const float in_scale[3], const float in_shift[3], bool relu, hipStream_t s);
// y = [relu](x * scale[c] + shift[c]) (scale/shift may be null: 1 / 0) (+ add, before the relu); NHWC fp32, C % 4 == 0:
// a BatchNorm+Scale(+ReLU) that cannot be folded into a convolution (pre-activation ResNet), or an Eltwise SUM
void launch_channel_affine(const float* x, const float* scale, const float* shift, const float* add, float* y,
                           long long npix, int C, bool relu, hipStream_t s);
void launch_maxpool3s2(const float* x, float* y, int n, int H, int Ho, int C, hipStream_t s);
void launch_l2norm128(const float* x, const float* scale, float* y, long long npix, hipStream_t s);
void launch_ssd_decode(const SsdHeads& H, float* boxes, float* prob, int n, int n_priors, float image_size,
                       const float var[4], hipStream_t s);
void launch_ssd_nms(const float* boxes, const float* prob, int n, int n_priors, float conf_thr, double nms_thr,
                    int keep_top_k, float* rows, int* count, hipStream_t s);

}  // namespace dfd
