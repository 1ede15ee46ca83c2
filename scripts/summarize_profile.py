#!/usr/bin/env python3
"""Summarise scripts/profile.sh output: per-kernel time (rocprofv3 --kernel-trace --stats)
and HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes (gfx950: FETCH_SIZE reports half the
bytes of a wide coalesced read -> x2; WRITE_SIZE exact; both in KiB)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mg::k::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:60]


def grid_of(r):
    gx = r.get("Grid_Size_X") or r.get("Grid_Size", "")
    gy = r.get("Grid_Size_Y", "1")
    return f"{gx}x{gy}" if gy not in ("", "1") else str(gx)


print(f"# rocprofv3 summary `{tag}`\n")
bj = os.path.join(out_dir, "bench_trace.json")
if os.path.exists(bj):
    line = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if line:
        b = json.loads(line[-1])
        print(f"bench under the tracer: {b['value']} {b['unit']}, {b['ms_per_step']} ms/step; workload: {b['config']['workload']}\n")

stats = find("trace", "*kernel_stats.csv")
if stats:
    print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    rows = list(csv.DictReader(open(stats)))
    for r in rows[:16]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    print()

# per-kernel average duration split by grid size needs the trace itself
trace = find("trace", "*kernel_trace.csv")
dur = defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace)):
        key = (short(r["Kernel_Name"]), grid_of(r))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def counter(sub, cname):
    f = find(sub, "*counter_collection.csv")
    acc = defaultdict(list)
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != cname:
            continue
        key = (short(r["Kernel_Name"]), grid_of(r))
        acc[key].append(float(r["Counter_Value"]))
    return acc


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
if fetch or write:
    print("## HBM traffic per launch (separate --pmc passes; FETCH_SIZE x2 on gfx950, KiB -> bytes)\n")
    print("| kernel | grid | launches | avg us | read MB | write MB | total MB | GB/s |\n|---|---|---|---|---|---|---|---|")
    keys = sorted(set(fetch) | set(write), key=lambda k: -sum(dur.get(k, [0])))
    for k in keys[:14]:
        rd = 2.0 * 1024 * sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [1])))
        wr = 1024 * sum(write.get(k, [0])) / max(1, len(write.get(k, [1])))
        d = dur.get(k, [])
        avg = sum(d) / len(d) if d else 0.0
        gbs = (rd + wr) / (avg * 1e-6) / 1e9 if avg else 0.0
        print(f"| {k[0]} | {k[1]} | {len(d)} | {avg:.1f} | {rd/1e6:.1f} | {wr/1e6:.1f} | {(rd+wr)/1e6:.1f} | {gbs:.0f} |")


# ---- per-kernel HBM bytes of the LARGEST launches (the finest level), for bench.py's roofline.traffic
def top_cluster(sub, cname):
    f = find(sub, "*counter_collection.csv")
    acc = defaultdict(list)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == cname:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        m = max(v)
        top = [x for x in v if x > 0.5 * m]
        out[k] = (sum(top) / len(top), len(top))
    return out


rd, wr = top_cluster("fetch", "FETCH_SIZE"), top_cluster("write", "WRITE_SIZE")
traffic = {}
for k in set(rd) | set(wr):
    r = 2.0 * 1024 * rd.get(k, (0, 0))[0]   # gfx950: FETCH_SIZE counts half the bytes of a wide read
    w = 1024 * wr.get(k, (0, 0))[0]
    if r + w > 1e6:
        traffic[k] = {"read_bytes": r, "write_bytes": w, "total_bytes": r + w, "launches": max(rd.get(k, (0, 0))[1], wr.get(k, (0, 0))[1])}
with open(os.path.join(out_dir, "traffic.json"), "w") as f:
    json.dump({"tag": tag, "note": "HBM bytes per launch of the finest-level launches; FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB->B; separate --pmc passes",
               "kernels": traffic}, f, indent=1)
