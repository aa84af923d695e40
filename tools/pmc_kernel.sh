#!/bin/bash
# SQ counters of the bench step's kernels (separate passes per group, --kernel-trace only); output gpurun_out/<tag>_<i>  (tag = $1, default pmck)
tag=${1:-pmck}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_$i.log 2>&1 || echo "pass $i failed"
done
