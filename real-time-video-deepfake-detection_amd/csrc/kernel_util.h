// Device helpers shared by the fp32 kernels (b0_kernels.hip) and the split-precision GEMM (gemm_split.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dfd {

typedef float v4f __attribute__((ext_vector_type(4)));

// swish(x) = x * sigmoid(x) as mul + v_exp_f32 + add + v_rcp_f32 + mul.  (`__fdividef` / `/` expand to the full
// IEEE division sequence here - ~10 VALU instructions per element, which made the fused kernels VALU-bound.)
// v_rcp_f32 and v_exp_f32 are accurate to 1 ulp; the 1e-3 logit bar holds with three orders of margin.
__device__ __forceinline__ float sigmoid1(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swish1(float x) { return x * sigmoid1(x); }
__device__ __forceinline__ v4f swish4(v4f v) {
    v4f r;
    r.x = swish1(v.x); r.y = swish1(v.y); r.z = swish1(v.z); r.w = swish1(v.w);
    return r;
}
__device__ __forceinline__ v4f ldg4(const float* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ void stg4(float* p, v4f v) { *reinterpret_cast<v4f*>(p) = v; }

}  // namespace dfd
