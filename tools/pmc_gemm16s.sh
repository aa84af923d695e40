#!/bin/bash
# PMC passes over tools/gemm16s_bench.py pmc (separate runs per counter group, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc16s_$i -- python3 $R/tools/gemm16s_bench.py $R/gpurun_out/pmc16s_$i.json pmc > $R/gpurun_out/pmc16s_$i.log 2>&1 || echo "pass $i failed"
done
