#!/bin/bash
# Register use of every kernel of one compilation unit:  scripts/kernel_regs.sh mg_stream.hip [extra hipcc flags]
# (compiles the device side only, to assembly, and reads the .amdhsa metadata)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/multigrid_poisson_solver_amd/csrc/$1; shift
OUT=$(mktemp -d /tmp/kregs_XXXX)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$ROOT/include -I$ROOT/multigrid_poisson_solver_amd/csrc \
  -x hip --cuda-device-only -S "$SRC" -o $OUT/k.s "$@"
python3 - "$OUT/k.s" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void mg::k::", "", dem)[:110]
    print(f"vgpr {g('next_free_vgpr'):>4} accum_off {g('accum_offset'):>4} sgpr {g('next_free_sgpr'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {dem}")
PY
echo "asm: $OUT/k.s"
