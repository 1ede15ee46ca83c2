#!/bin/bash
# The reference's own main() + linked list driving the engine operator by operator (oracle/_ref/MG_HIP_dropin, the
# INTEGRATION.md port) against the engine's fused driver (bin/MG_HIP) and the reference CPU program, on one generated
# V(3,3) file: the "Time Used" line each program prints (the reference's window, :156..:429).
#   scripts/dropin_compare.sh [N]
N=${1:-4096}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d); cd $D
python3 -c "import sys; sys.path.insert(0,'$ROOT'); import multigrid_poisson_solver_amd as m; m.write_vcycle_file('V.txt',$N,8,3,1e-7)"
for rep in 1 2; do
  echo "drop-in (reference main() on the C ABI, one launch per operator): $($ROOT/oracle/_ref/MG_HIP_dropin 16 V.txt | grep -E 'Time Used|   Error')"
  echo "engine driver (fused nodes):                                     $($ROOT/multigrid_poisson_solver_amd/bin/MG_HIP 16 V.txt | grep -E 'Time Used|   Error')"
done
echo "reference CPU program (-O2, 16 threads):                          $($ROOT/oracle/_ref/MG_CPU_ref 16 V.txt | grep -E 'Time Used|   Error')"
