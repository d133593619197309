// libjpeg's integer arithmetic, restated once for the ELA round trip (forensic_kernels.hip, reference
// frame_analysis.py:233-236) and the JPEG decoder at the HTTP edge (jpeg_decode.hip, reference backend_server.py:139-145):
// RGB <-> YCbCr (jccolor.c / jdcolor.c, 16-bit fixed point), jfdctint.c / jidctint.c ("islow") passes.
#pragma once
#include <hip/hip_runtime.h>

namespace dfd {

#define JFIX(x) ((int)((x) * 65536.0 + 0.5))
__device__ __forceinline__ int ycc_y(int r, int g, int b) { return (JFIX(0.29900) * r + JFIX(0.58700) * g + JFIX(0.11400) * b + 32768) >> 16; }
__device__ __forceinline__ int ycc_cb(int r, int g, int b) { return (-JFIX(0.16874) * r - JFIX(0.33126) * g + JFIX(0.50000) * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ int ycc_cr(int r, int g, int b) { return (JFIX(0.50000) * r - JFIX(0.41869) * g - JFIX(0.08131) * b + (128 << 16) + 32767) >> 16; }

__host__ __device__ __forceinline__ int dsc(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c one pass over 8 values with stride `st`
template <bool FIRST>
__device__ __forceinline__ void fdct8(int* d, int st) {
    const int t0 = d[0] + d[7 * st], t7 = d[0] - d[7 * st], t1 = d[st] + d[6 * st], t6 = d[st] - d[6 * st];
    const int t2 = d[2 * st] + d[5 * st], t5 = d[2 * st] - d[5 * st], t3 = d[3 * st] + d[4 * st], t4 = d[3 * st] - d[4 * st];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int n = FIRST ? 11 : 15;
    d[0] = FIRST ? (t10 + t11) << 2 : dsc(t10 + t11, 2);
    d[4 * st] = FIRST ? (t10 - t11) << 2 : dsc(t10 - t11, 2);
    int z1 = (t12 + t13) * 4433;
    d[2 * st] = dsc(z1 + t13 * 6270, n);
    d[6 * st] = dsc(z1 + t12 * (-15137), n);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
    d[7 * st] = dsc(a4 + z1 + z3, n);
    d[5 * st] = dsc(a5 + z2 + z4, n);
    d[3 * st] = dsc(a6 + z2 + z3, n);
    d[st] = dsc(a7 + z1 + z4, n);
}

// jidctint.c one pass
template <bool FIRST>
__device__ __forceinline__ void idct8(int* v, int st) {
    int z2 = v[2 * st], z3 = v[6 * st];
    int z1 = (z2 + z3) * 4433;
    int t2 = z1 + z3 * (-15137), t3 = z1 + z2 * 6270;
    int t0 = (v[0] + v[4 * st]) << 13, t1 = (v[0] - v[4 * st]) << 13;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = v[7 * st]; t1 = v[5 * st]; t2 = v[3 * st]; t3 = v[st];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    constexpr int n = FIRST ? 11 : 18;
    v[0] = dsc(t10 + t3, n); v[7 * st] = dsc(t10 - t3, n);
    v[st] = dsc(t11 + t2, n); v[6 * st] = dsc(t11 - t2, n);
    v[2 * st] = dsc(t12 + t1, n); v[5 * st] = dsc(t12 - t1, n);
    v[3 * st] = dsc(t13 + t0, n); v[4 * st] = dsc(t13 - t0, n);
}


}  // namespace dfd
