// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 for the accumulate patterns of gemm_split.hip.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
// One wave per SIMD (256-thread blocks, one block per CU).  Reports ns per MFMA per wave; the independent
// pattern is the 16-cycle reference of MI355X_MICROARCH.md.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int CHAIN, int ACCS>
__global__ __launch_bounds__(256) void k(const bf8* in, v4f* out, int iters) {
    bf8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    v4f acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACCS; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    v4f s = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ACCS; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with the three-term split of a fresh operand between chains (the GEMM's VALU work)
template <int ACCS>
__global__ __launch_bounds__(256) void ksplit(const bf8* in, const float* xin, v4f* out, int iters) {
    bf8 a = in[threadIdx.x];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = xin[threadIdx.x * 8 + i];
    v4f acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        bf8 s0, s1, s2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float f = x[i] + (float)it;
            const __bf16 h0 = (__bf16)f;
            const float r1 = f - (float)h0;
            const __bf16 h1 = (__bf16)r1;
            s0[i] = h0; s1[i] = h1; s2[i] = (__bf16)(r1 - (float)h1);
        }
#pragma unroll
        for (int i = 0; i < ACCS; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s2, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, s0, acc[i], 0, 0, 0);
        }
    }
    v4f s = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ACCS; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    bf8* in; float* xin; v4f* out;
    hipMalloc(&in, 512 * sizeof(bf8)); hipMemset(in, 0, 512 * sizeof(bf8));
    hipMalloc(&xin, 256 * 8 * 4); hipMemset(xin, 0, 256 * 8 * 4);
    hipMalloc(&out, 256 * 256 * sizeof(v4f));
    const int iters = 20000;
#define RUN(CH, AC)                                                                                         \
    {                                                                                                       \
        double ms = time_ms([&] { hipLaunchKernelGGL((k<CH, AC>), dim3(256), dim3(256), 0, 0, in, out, iters); }); \
        printf("chain %d x accs %2d : %.2f ns per MFMA\n", CH, AC, ms * 1e6 / ((double)iters * CH * AC));     \
    }
    RUN(1, 16) RUN(1, 4) RUN(1, 1) RUN(6, 1) RUN(6, 2) RUN(6, 6) RUN(2, 6)
#define RUNS(AC)                                                                                            \
    {                                                                                                       \
        double ms = time_ms([&] { hipLaunchKernelGGL((ksplit<AC>), dim3(256), dim3(256), 0, 0, in, xin, out, iters); }); \
        printf("split + 6-chain x accs %2d : %.2f ns per MFMA (%.1f ns per K-step)\n", AC, ms * 1e6 / ((double)iters * 6 * AC), ms * 1e6 / iters); \
    }
    RUNS(1) RUNS(2) RUNS(6) RUNS(12)
    return 0;
}
