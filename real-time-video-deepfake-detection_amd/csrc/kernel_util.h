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
// 16 / 8 bytes from a 4-byte aligned address (global memory takes dword-aligned dwordx4 / dwordx2 loads)
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ v4f ldg4u(const float* p) { return *reinterpret_cast<const v4f_u*>(p); }

// Activation storage type XT = float (fp32 path) or bf16_t ("bf16_activations": every activation tensor that
// reaches HBM is bf16, all arithmetic and every accumulator stays fp32).  ld4 / st4 move 4 consecutive channels
// of one pixel: 16 bytes of fp32 or 8 bytes of bf16 (bf16 -> fp32 is a shift, fp32 -> bf16 rounds to nearest
// even: v_cvt_pk_bf16_f32).
typedef __bf16 bf16_t;
typedef __bf16 bf4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f ld4(const float* p) { return ldg4(p); }
__device__ __forceinline__ void st4(float* p, v4f v) { stg4(p, v); }
__device__ __forceinline__ v4f bf4_to_f4(u2v u) {
    v4f r;
    r.x = __builtin_bit_cast(float, u.x << 16);
    r.y = __builtin_bit_cast(float, u.x & 0xffff0000u);
    r.z = __builtin_bit_cast(float, u.y << 16);
    r.w = __builtin_bit_cast(float, u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ v4f ld4(const bf16_t* p) { return bf4_to_f4(*reinterpret_cast<const u2v*>(p)); }
__device__ __forceinline__ void st4(bf16_t* p, v4f v) {
    const bf4v b = __builtin_convertvector(v, bf4v);
    *reinterpret_cast<u2v*>(p) = __builtin_bit_cast(u2v, b);
}


// ---- split-precision operands (gemm_split_impl.h, and the fused expand of b0_kernels.hip) ----
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 struct does not always leave the stack)

// 8 fp32 values -> three bf16x8 terms whose sum is exact (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1))
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


// ---- scalar operands ----
// Scalar (wave-uniform) operand loads under program control: 16 consecutive floats of a kernel-argument array into
// SGPRs.  Plain C++ reads of uniform addresses also become s_load, but hipcc hoists all of a loop's invariant ones to
// the top and then spills hundreds of SGPR values lane by lane (v_readlane): here the load is issued one group ahead of
// its use and the wait names the registers it releases.
typedef float sf16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f ldg2(const float* p) { return *reinterpret_cast<const v2f_u*>(p); }
template <int BYTE_OFF>
__device__ __forceinline__ sf16 sload16(const float* p) {
    sf16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "n"(BYTE_OFF));
    return v;
}
__device__ __forceinline__ sf16 sload16s(const float* p, int byte_off) {      // offset in an SGPR: usable from an unrolled loop
    sf16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "s"(byte_off));
    return v;
}
__device__ __forceinline__ void swait1(sf16& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void swait2(sf16& a, sf16& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

}  // namespace dfd
