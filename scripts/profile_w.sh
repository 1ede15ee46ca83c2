#!/bin/bash
# kernel-time breakdown of one W(3,3)-cycle (BASELINE.json config 3 shape); run through gpurun
set -e
N=${1:-8192}
OUT=$(pwd)/gpurun_out/prof_w
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 $(pwd)/bench.py --steps 3 --warmup 1 --no-cpu --cycle W --n $N > $OUT/bench.json 2> $OUT/log.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    name = r["Name"].replace("mg::k::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{name:<62} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6}%")
PY
tail -c 400 $OUT/bench.json
