#!/usr/bin/env python3
"""Per-kernel instruction statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only): MFMA / VALU / LDS / VMEM / scratch counts
and where the scratch accesses sit relative to the MFMAs. usage: isa_stats.py file.s [kernel-name substring]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//"))]
    ins = [l for l in lines if not l.endswith(":")]
    cnt = lambda pat: sum(bool(re.match(pat, l)) for l in ins)
    sc = [i for i, l in enumerate(ins) if l.startswith("scratch_")]
    print(f"{name[:90]}\n  instr {len(ins)}  mfma {cnt('v_mfma')}  valu {cnt('v_') - cnt('v_mfma')}  salu {cnt('s_')}  ds {cnt('ds_')}  "
          f"global {cnt('global_|buffer_|flat_')}  scratch {len(sc)}  waitcnt {cnt('s_waitcnt')}  barrier {cnt('s_barrier')}")
    if sc:
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        print("  scratch at instruction index (mfma before it):", [(i, sum(1 for x in mf if x < i)) for i in sc][:60])
