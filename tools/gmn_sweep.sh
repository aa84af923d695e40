#!/bin/bash
# the WT_* switches exist in the LAB build only
export WAVTOK_HIP_LIB=${WAVTOK_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools/lib/libwavtok_hip_lab.so}
# tile-order sweep for gemm16s: WT_GEMM16S_GM x WT_GEMM16S_GN over the bench step (all GEMM launches take the same setting)
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --repeats 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
run default
for c in "8 6" "15 6" "8 4" "4 6" "16 3" "8 3" "4 4" "2 6" "30 6" "8 2" "4 2"; do set -- $c; WT_GEMM16S_GM=$1 WT_GEMM16S_GN=$2 run "GM=$1,GN=$2"; done
run default
