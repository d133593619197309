"""Vendor-library reference times for the late-layer GEMM shapes (torch.mm -> rocBLAS / hipBLASLt), to place the
hand-written split-precision kernels: python profiles/micro/vendor_gemm.py"""
import torch

shapes = [(12544, 1152, 192), (12544, 192, 1152), (12544, 1152, 320), (12544, 320, 1280), (50176, 480, 80), (50176, 80, 480),
          (50176, 672, 112), (50176, 112, 672)]
for dt in (torch.float32, torch.bfloat16):
    for M, K, N in shapes:
        a = torch.randn(M, K, device="cuda", dtype=dt)
        b = torch.randn(K, N, device="cuda", dtype=dt)
        for _ in range(5):
            torch.mm(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.mm(a, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{str(dt):16s} M={M:6d} K={K:5d} N={N:5d}: {us:7.1f} us  {2 * M * K * N / us / 1e6:7.1f} TFLOP/s")
