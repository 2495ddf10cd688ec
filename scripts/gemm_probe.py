"""What does the vendor GEMM (torch -> hipBLASLt / rocBLAS) reach on the 1x1 layers' shapes?  out[M, N] = x[M, K] @ w[N, K]^T, fp16 in, fp32 accumulate."""
import torch
shapes = [(204800, 512, 1024), (204800, 512, 512), (819200, 256, 512), (819200, 256, 256), (51200, 768, 1536), (51200, 768, 768), (204800, 256, 1024), (3276800, 128, 256), (12800, 1024, 2048)]
for M, N, K in shapes:
    x = torch.randn((M, K), device="cuda", dtype=torch.float16)
    w = torch.randn((N, K), device="cuda", dtype=torch.float16) / K ** 0.5
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    for name, fn in (("linear", lambda: torch.nn.functional.linear(x, w)), ("linear+bias", lambda: torch.nn.functional.linear(x, w, b)), ("linear+bias+silu", lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, w, b)))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print("M=%-8d N=%-5d K=%-5d %-17s %8.1f us  %7.1f TF/s  %6.0f GB/s" % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6, (M * K + M * N + N * K) * 2 / us / 1e3))
