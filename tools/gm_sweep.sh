#!/bin/bash
# the WT_* switches exist in the LAB build only
export WAVTOK_HIP_LIB=${WAVTOK_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools/lib/libwavtok_hip_lab.so}
# FETCH_SIZE of the two ConvNeXt GEMM shapes for several scheduling group sizes (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for gm in 1 2 4 8 16 64; do
  WT_GEMM_GM=$gm WT_GEMM_TILE=2 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /root/repo/gpurun_out/gm_$gm -- python3 /root/repo/tools/gemm_bench.py > /root/repo/gpurun_out/gm_$gm.log 2>&1
  grep "pwconv\|res k3" /root/repo/gpurun_out/gm_$gm.log | sed "s/^/gm=$gm /"
done
