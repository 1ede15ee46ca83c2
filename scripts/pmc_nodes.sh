#!/bin/bash
# SQ counters of the two fused node kernels at one size: scripts/pmc_nodes.sh <N>
# (separate --pmc passes, kernel trace only; summary to gpurun_out/pmc_<N>.txt)
set -e
N=${1:-2048}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$N
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_WAVES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $ROOT/scripts/perf_nodes.py $N > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$N" <<'PY' > $ROOT/gpurun_out/pmc_$N.txt
import csv, glob, sys, collections
out, N = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_jacobi_stream" not in k:
            continue
        k = k[k.index("k_jacobi_stream"):][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:<24} {sum(v)/len(v):16.0f}  (avg of {len(v)} launches)")
PY
cat $ROOT/gpurun_out/pmc_$N.txt
