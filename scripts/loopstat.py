#!/usr/bin/env python3
"""Static instruction mix of the main loop (2*PF row steps) of the fused node kernels: compiles
csrc/mg_stream.hip to gfx950 assembly and counts VALU / SALU / VMEM instructions per row step.
The small levels of a cycle are bound by the instruction stream of a lone wave, so this count is
what a kernel change there has to move.   python3 scripts/loopstat.py [S]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigrid_poisson_solver_amd", "csrc")
S = sys.argv[1] if len(sys.argv) > 1 else "3"
tmp = tempfile.mkdtemp(prefix="loopstat_")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                "-I" + CSRC, "-x", "hip", "-c", os.path.join(CSRC, "mg_stream.hip"), "-o", os.path.join(tmp, "s.o"), "-save-temps=obj"],
               check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
lines = open(os.path.join(tmp, asm)).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2mg1k3f6415k_jacobi_stream\w+:", l)]
meta = "\n".join(lines)
for n, (i, name) in enumerate(starts):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
    m = re.search(r"<(\d), (\d), (\d), (\w+), (\d)>", dem)
    if not m or m.group(1) != S or m.group(2) != "2":
        continue
    if (m.group(3), m.group(4)) not in (("1", "true"), ("2", "false")):
        continue
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    body = lines[i:end]
    labels = {}
    for k, l in enumerate(body):
        mm = re.match(r"(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = k
    spans = []
    for k, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            spans.append((labels[mm.group(1)], k))
    spans = [s for s in spans if (s[1] - s[0]) < 0.8 * len(body)]
    lo, hi = max(spans, key=lambda s: s[1] - s[0])
    loop = [l.strip() for l in body[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = Counter()
    for ins in loop:
        op = ins.split()[0]
        c["VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else
          "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "other"] += 1
    vg = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", meta, re.S)
    sp = re.search(re.escape(name) + r".*?\.vgpr_spill_count:\s+(\d+)", meta, re.S)
    kind = "`-1` node (zero start, restrict)" if m.group(3) == "1" else "`1` node (prolong)"
    print(f"{kind} S={S}: loop of {len(loop)} instructions = 8 row steps; per row step:",
          {k: round(v / 8, 1) for k, v in c.items()}, "VGPRs", vg.group(1) if vg else "?", "spills", sp.group(1) if sp else "?")
    cv = Counter(("dpp" if "dpp" in i else i.split()[0]) for i in loop if i.startswith("v_"))
    print("   VALU:", ", ".join(f"{k}:{v}" for k, v in cv.most_common(14)))
    cs = Counter(i.split()[0] for i in loop if i.startswith("s_"))
    print("   SALU:", ", ".join(f"{k}:{v}" for k, v in cs.most_common(14)))
    cm = Counter(i.split()[0] for i in loop if i.startswith(("global_", "buffer_", "flat_")))
    print("   VMEM:", ", ".join(f"{k}:{v}" for k, v in cm.most_common(8)))
