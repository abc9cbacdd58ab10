# torch scaled_dot_product_attention on the encoder attention shapes (calibration; the product kernel does an exact two-sweep soft-max)
import torch
dev = torch.device("cuda")
for B in (1, 8):
    q = torch.randn(B, 8, 1500, 64, device=dev, dtype=torch.float16)
    k = torch.randn_like(q); v = torch.randn_like(q)
    f = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 30
    print(f"sdpa B={B} H=8 T=1500 D=64: {us:8.2f} us  ({B*2*2.0*1500*1500*512/us/1e6:6.1f} TF/s on the 2-GEMM count)")
