#!/usr/bin/env python3
"""Diagnostic: TFLOP/s of avd_gemm_bt_bf16 (the GEMM under the shared-set learner) on random operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from avddpg_amd._hip import call, ptr, stream_handle

for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (262144, 1024, 1088), (262144, 1072, 1024), (1072, 1024, 16384)]:
    zeros = os.environ.get("GEMM_ZEROS")  # zero operands draw less power: the chip holds a higher clock (upper bound only)
    A = (torch.zeros if zeros else torch.randn)(M + 256, K, device="cuda").to(torch.bfloat16)
    B = (torch.zeros if zeros else torch.randn)(N + 256, K, device="cuda").to(torch.bfloat16)
    D = torch.empty(M, N, device="cuda")
    f = lambda: call("avd_gemm_bt_bf16", M, N, K, ptr(A), K, ptr(B), K, ptr(D), N, stream_handle())
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms * 1e-9:.0f} TFLOP/s")
    del A, B, D
