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

void launch_ssd_conv1(const uint8_t* img, const float* w, const float* b, float* y, int n, const float mean_bgr[3],
                      hipStream_t s);
void launch_maxpool3s2(const float* x, float* y, int n, int H, int Ho, int C, hipStream_t s);
void launch_l2norm128(const float* x, const float* scale, float* y, long long npix, hipStream_t s);
void launch_ssd_decode(const SsdHeads& H, float* boxes, float* prob, int n, int n_priors, float image_size,
                       const float var[4], hipStream_t s);
void launch_ssd_nms(const float* boxes, const float* prob, int n, int n_priors, float conf_thr, double nms_thr,
                    int keep_top_k, float* rows, int* count, hipStream_t s);

}  // namespace dfd
