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

# bench.py warms up with W windows before it times K: the --stats averages above cover all W + K + 1 launches of a kernel,
# the clock ramp of the first ~40 windows included.  The K launches of the TIMED region are the last K dispatches of
# the finest-level kernels (one launch per window each): their average is what bench.py's live hipEvent number must agree with.
trace_csv = find("trace", "*kernel_trace.csv")
if trace_csv and os.path.exists(bj):
    try:
        K = int(b["steps"])
        rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
        by = defaultdict(list)
        for r in rows:
            by[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("## the timed region only (last K dispatches of the kernels launched once per window)\n")
        print("| kernel | launches per run | avg us, all | avg us, last K = %d |\n|---|---|---|---|" % K)
        for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            per_window = len(d) / (2 * K + 1.0)
            if len(d) < 2 * K or abs(per_window - round(per_window)) > 0.02:
                continue
            m = int(round(per_window))
            # m launches per window (the same template serves several levels): take the largest of each window
            tail = d[-K * m:]
            big = [max(tail[i * m:(i + 1) * m]) for i in range(K)]
            allbig = [max(d[i * m:(i + 1) * m]) for i in range(len(d) // m)]
            print(f"| {name} (largest of {m} per window) | {len(d)} | {sum(allbig) / len(allbig):.1f} | {sum(big) / len(big):.1f} |")
        print()
    except Exception as exc:  # never lose the rest of the summary
        print(f"(timed-region table skipped: {exc})\n")

# Per-launch join of the three runs.  The program is deterministic, so the i-th dispatch of a
# kernel name in the trace run is the i-th dispatch of that name in the two counter runs.  The grid
# cannot tell the levels apart (one resident round of workgroups at every size), so the launches of
# one kernel are clustered by the bytes they moved.
def per_dispatch(path, value):
    acc = defaultdict(list)
    if not path:
        return acc
    rows = list(csv.DictReader(open(path)))
    idk = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if idk:
        rows.sort(key=lambda r: int(r[idk]))
    for r in rows:
        v = value(r)
        if v is not None:
            acc[short(r["Kernel_Name"])].append(v)
    return acc


dur = per_dispatch(find("trace", "*kernel_trace.csv"), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
fetch = per_dispatch(find("fetch", "*counter_collection.csv"),
                     lambda r: float(r["Counter_Value"]) if r.get("Counter_Name") == "FETCH_SIZE" else None)
write = per_dispatch(find("write", "*counter_collection.csv"),
                     lambda r: float(r["Counter_Value"]) if r.get("Counter_Name") == "WRITE_SIZE" else None)
if fetch or write:
    print("## HBM traffic per launch (separate --pmc passes joined per dispatch; FETCH_SIZE x2 on gfx950, KiB -> bytes)\n")
    print("| kernel | launches | avg us | read MB | write MB | total MB | GB/s |\n|---|---|---|---|---|---|---|")
    table = []
    for name in set(fetch) | set(write):
        d, f, w = dur.get(name, []), fetch.get(name, []), write.get(name, [])
        n = min(len(d), len(f), len(w))
        if n == 0:
            continue
        rows = sorted(((2.0 * 1024 * f[i], 1024 * w[i], d[i]) for i in range(n)), key=lambda t: -(t[0] + t[1]))
        clusters, cur = [], [rows[0]]
        for t in rows[1:]:
            if (cur[-1][0] + cur[-1][1]) > 1.6 * (t[0] + t[1]) + 1e5:
                clusters.append(cur)
                cur = []
            cur.append(t)
        clusters.append(cur)
        for c in clusters:
            rd = sum(t[0] for t in c) / len(c)
            wr = sum(t[1] for t in c) / len(c)
            avg = sum(t[2] for t in c) / len(c)
            table.append((avg * len(c), name, len(c), avg, rd, wr))
    for _tot, name, n, avg, rd, wr in sorted(table, reverse=True)[:18]:
        gbs = (rd + wr) / (avg * 1e-6) / 1e9 if avg else 0.0
        print(f"| {name} | {n} | {avg:.1f} | {rd/1e6:.1f} | {wr/1e6:.1f} | {(rd+wr)/1e6:.1f} | {gbs:.0f} |")


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
import hashlib
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multigrid_poisson_solver_amd", "lib", "libmgpoisson.so")
h = hashlib.sha256()
with open(lib, "rb") as f:
    for blk in iter(lambda: f.read(1 << 20), b""):
        h.update(blk)
with open(os.path.join(out_dir, "traffic.json"), "w") as f:
    json.dump({"tag": tag, "lib_sha": h.hexdigest()[:16], "note": "HBM bytes per launch of the finest-level launches; FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB->B; separate --pmc passes",
               "kernels": traffic}, f, indent=1)
