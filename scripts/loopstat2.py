#!/usr/bin/env python3
"""Static instruction mix of the main loop of one k_jacobi_stream instantiation (fp64):
    python3 scripts/loopstat2.py "3, 2, 2, false, 2, true, 3" [rows_per_body] [extra hipcc flags...]
compiles csrc/mg_stream.hip to gfx950 assembly (device side only) and counts the instructions of the largest loop."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigrid_poisson_solver_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "3, 2, 2, false, 2, true, 3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tmp = tempfile.mkdtemp(prefix="loopstat_")
asm = os.path.join(tmp, "k.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "mg_stream.hip"), "-o", asm] + sys.argv[3:], check=True,
               stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2mg1k3f64\w*k_jacobi_stream\w+:", l)]
for n, (i, name) in enumerate(starts):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
    if f"<{want}>" not in dem:
        continue
    end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm")), len(lines))
    # the kernel may have several s_endpgm: take up to the .Lfunc_end label
    end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), end)
    body = lines[i:end]
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    spans = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            spans.append((labels[m.group(1)], k))
    lo, hi = max(spans, key=lambda s: s[1] - s[0])
    loop = [l.strip() for l in body[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    kinds = Counter()
    for ins in loop:
        op = ins.split()[0]
        kinds["VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "other"] += 1
    print(f"k_jacobi_stream<{want}>: loop of {len(loop)} instructions = {rows} row steps; per row step:", {k: round(v / rows, 1) for k, v in kinds.items()})
    cv = Counter(("dpp:" + i.split()[0] if "dpp" in i else i.split()[0]) for i in loop if i.startswith("v_"))
    print("   VALU per row step:", ", ".join(f"{k}:{v / rows:.1f}" for k, v in cv.most_common(24)))
    cs = Counter(i.split()[0] for i in loop if i.startswith("s_"))
    print("   SALU per row step:", ", ".join(f"{k}:{v / rows:.1f}" for k, v in cs.most_common(12)))
    print("   asm:", asm, "lines", i + lo, "-", i + hi)
