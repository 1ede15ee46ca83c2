#!/bin/bash
# SQ counters of every fused node launch of ONE V-cycle at size N (scripts/perf_levels.py under rocprofv3 --pmc,
# separate passes): scripts/pmc_cycle.sh <N> [tag]  ->  gpurun_out/pmc_cycle_<N><tag>.txt   (MG_LIB selects the build)
set -e
N=${1:-8192}
TAG=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_cycle_$N$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $ROOT/scripts/perf_levels.py $N > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$N" <<'PY' > $ROOT/gpurun_out/pmc_cycle_$N$TAG.txt
import csv, glob, sys, collections
out, N = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_jacobi_stream" not in k:
            continue
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
        k = k[k.index("k_jacobi_stream"):][:48] + f" grid={grid}"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:<24} {sum(v)/len(v):16.0f}  (avg of {len(v)} launches)")
PY
head -34 $ROOT/gpurun_out/pmc_cycle_$N$TAG.txt
