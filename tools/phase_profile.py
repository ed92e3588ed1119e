#!/usr/bin/env python3
"""Diagnostic: where learn_kernel spends its cycles. Builds/loads the -DAVD_PHASE_TIMING library
(avddpg_amd/lib/libavddpg_hip_diag.so, `make -C avddpg_amd/csrc diag`), runs the kernel on synthetic
batches and prints the share of workgroup-thread-0 shader cycles per phase. Not part of the product."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from avddpg_amd import _hip

_hip.LIB_PATH = os.environ.get("AVD_PHASE_LIB", os.path.join(ROOT, "avddpg_amd", "lib", "libavddpg_hip_phase.so"))  # tools/build_phase_lib.sh
from avddpg_amd import config, vec

NAMES = {0: "stage batch", 1: "actor L1 (VALU) + bn coefs", 2: "actor L2 GEMM fwd", 3: "actor out layer",
         4: "critic L1 (VALU) + bn coefs", 5: "critic L2 GEMM fwd", 6: "critic out layer",
         7: "critic out-layer bwd (+glue)", 8: "critic col sums", 9: "critic dW2 GEMM", 10: "critic dX GEMM + BN bwd",
         11: "critic L1 grads", 12: "actor-path critic out bwd (+glue)", 13: "actor-path dX (action cols)",
         14: "da reduction", 15: "actor out-layer bwd (+glue)", 16: "actor col sums", 17: "actor dW2 GEMM",
         18: "actor dX GEMM + BN bwd", 19: "actor L1 grads"}

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FUSED = len(sys.argv) > 2 and sys.argv[2] in ("fused", "lean-fused")
LEAN = len(sys.argv) > 2 and sys.argv[2].startswith("lean")  # learn_kernel_l (lean.hip); else learn_kernel_t
os.environ["AVD_LEARN_KERNEL"] = "lean" if LEAN else "fast"
CENTRAL = len(sys.argv) > 2 and sys.argv[2].startswith("centralized")  # S = 20, A = 5, widths x1.2: cen::learn_kernel_c
FUSED = FUSED or (len(sys.argv) > 2 and sys.argv[2].endswith("-fused"))
if len(sys.argv) > 2 and "general" in sys.argv[2]:  # "centralized-general[-fused]": gen::learn_kernel_g on the same shape
    os.environ["AVD_LEARN_GENERAL"] = "1"
conf = config.Config()
S, A = (20, 5) if CENTRAL else (4, 1)
grp = vec.AgentGroup(n, S, A, conf, hidd_mult=1.2 if CENTRAL else 1.0)
f = lambda *s: torch.randn(*s, device="cuda")
s, a, r, s2 = f(n, 64, S), f(n, 64, A), f(n, 64), f(n, 64, S)
lib = _hip.lib()
dbg = lib.avd_debug_phase_cycles_lean if LEAN else lib.avd_debug_phase_cycles
if CENTRAL and "general" not in sys.argv[2]:
    dbg = lib.avd_debug_phase_cycles_cen
dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
gscr = torch.zeros(n, grp.lay.theta_size, device='cuda')
run = (lambda: grp.learn_update(s, a, r, s2, gscr)) if FUSED else (lambda: grp.learn(s, a, r, s2, 0))
run()
torch.cuda.synchronize()
import time
t_warm = time.time()
while time.time() - t_warm < 2.0:  # the chip settles on the clock it holds under this load (DVFS) before anything is counted
    for _ in range(10):
        run()
    torch.cuda.synchronize()
dbg(None, 1)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
dbg(buf, 0)
tot = sum(buf[:24])
if buf[31]:  # learn_kernel_l stamps its lifetime with both counters: the clock the chip held while it ran
    print(f"in-kernel clock = {100e6 * buf[30] / buf[31] / 1e9:.3f} GHz (sum of s_memtime / sum of s_memrealtime x 100 MHz over all workgroups)")
    print(f"workgroup lifetime = {buf[30] / (3 * n):.0f} cyc/tile = {buf[31] / (3 * n) / 100:.1f} us/tile "
          f"(stamped phases below cover {100 * tot / buf[30]:.1f} % of it; the rest is the batch staging before the first stamp and, "
          f"fused, the small-tensor update + next-action epilogue after the last)")
print(f"tiles={3 * n}  cycles/tile={tot / (3 * n):.0f}")
NAMES.update({20: "  dx: issue next-tile loads", 21: "  dx: MFMA loop", 22: "  dx: BN epilogue", 23: "  dx: wait next tile (copy)"})
if LEAN:
    NAMES.update({1: "actor BN coefficient tables", 4: "critic BN coefficient tables", 9: "critic dW2 + dX (+update)",
                  10: "-", 11: "-", 17: "actor dW2 + dX (+update)", 18: "-", 19: "-"})
if CENTRAL:  # the general kernel stamps its first-layer pieces under these ids (and they are NOT part of phases 1 / 4)
    NAMES.update({20: "stage state batch", 21: "actor first layer", 22: "critic first layer (state)", 23: "critic first layer (action)"})
if CENTRAL and "general" not in sys.argv[2]:
    NAMES.update({24: "  fwd GEMMs: to the end of k-block 0", 25: "  fwd GEMMs: k-blocks 1 .. NB/2", 26: "  fwd GEMMs: k-blocks NB/2+1 .. NB-1",
                  27: "  fwd GEMMs: epilogue"})
for i in range(28 if CENTRAL else 24):
    print(f"{i:2d} {NAMES[i]:38s} {buf[i] / (3 * n):9.0f} cyc/tile  {100 * buf[i] / tot:5.1f}%")
