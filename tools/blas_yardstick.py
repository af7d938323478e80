#!/usr/bin/env python3
"""Yardstick only (not part of the product path): how fast does the vendor fp32 GEMM (torch.mm -> hipBLASLt/rocBLAS) run
the wide layer shapes of config A?  Used to judge how much headroom pgemm_kernel has left on MI355X."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [(163840, 128, 128), (40960, 256, 256), (10240, 128, 256), (10240, 512, 512), (40960, 128, 128), (655360, 128, 128), (163840, 512, 512)]
for (M, K, N) in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); Y = torch.empty(M, N, device="cuda")
    for _ in range(3): torch.mm(A, W, out=Y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.mm(A, W, out=Y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} K={K} N={N}: {ms*1e3:.1f} us  {2.0*M*K*N/ms/1e9:.1f} TF/s", flush=True)
