# what the vendor GEMM (torch -> hipBLASLt / rocBLAS) does on the encoder shapes: calibration for DESIGN.md §4
import torch, time
dev = torch.device("cuda")
shapes = [(12000, 2048, 512, "fc1 x8"), (12000, 1536, 512, "qkv x8"), (12000, 512, 2048, "fc2 x8"), (12000, 512, 512, "o x8"),
          (12000, 6144, 512, "crosskv x8"), (1500, 2048, 512, "fc1 x1"), (1500, 512, 2048, "fc2 x1"), (4096, 4096, 4096, "4096^3")]
for M, N, K, what in shapes:
    a = (torch.rand(M, K, device=dev, dtype=torch.float16) - 0.5)
    w = (torch.rand(N, K, device=dev, dtype=torch.float16) - 0.5) * 0.1
    bias = torch.rand(N, device=dev, dtype=torch.float16)
    for name, fn in (("matmul", lambda: torch.matmul(a, w.t())), ("linear+bias", lambda: torch.nn.functional.linear(a, w, bias))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        print(f"{what:11s} M={M:5d} N={N:5d} K={K:5d} {name:12s} {us:8.2f} us {2.0*M*N*K/us/1e6:8.1f} TF/s")
