// Launchers for the 8-bit image kernels (resize, CLAHE path, crop -> network input).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfd {

// Integer LUTs of the 8-bit colour conversions (built by luts.py, uploaded at dfd_create).
struct ColorTables {
    const int *gamma, *cbrt, *L_fy, *L_y, *a_div, *b_div, *ab_xz, *inv_gamma, *hsv_sdiv, *hsv_hdiv;
    int fwd[9];        // RGB->XYZ/white, 12-bit
    long long inv[9];  // XYZ*white->RGB, 12-bit
};

struct CropDesc {
    int x, y, w, h;        // box in the frame
    size_t offset;         // byte offset of this crop's packed w*h*3 region in the scratch buffers
    size_t src_off;        // byte offset of the crop's frame inside the frame buffer (batched frames)
};

void launch_resize_bgr(const uint8_t* src, int n, int sh, int sw, size_t sstride, size_t simg,
                       uint8_t* dst, int dh, int dw, hipStream_t s);
// single-channel cv2.resize(INTER_LINEAR) and u8 -> float (x scale)
// test-time augmentation of a face crop (flip, brightness, small rotation) as the reference builds it with cv2;
// mi = the inverted 2x3 affine matrix (warpAffine's internal form)
void launch_tta_augment(const uint8_t* src, int h, int w, int stride, int flip, float alpha, const double mi[6], uint8_t* dst,
                        hipStream_t s);
void launch_resize_gray(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, hipStream_t s);
void u8_to_float(const uint8_t* src, float* dst, int n, float scale, hipStream_t s);
void launch_clahe(const uint8_t* frame, size_t fstride, const CropDesc* crops_dev, int n, uint8_t* lab,
                  uint8_t* luts, uint8_t* bgr_out, const ColorTables& T, int max_pixels, hipStream_t s);
void launch_crop_norm(const uint8_t* frame, size_t fstride, const uint8_t* scratch, const CropDesc* crops_dev,
                      int n, float* out_nchw, bool from_scratch, hipStream_t s);

}  // namespace dfd
