#!/bin/bash
# One profiling pass on the GPU box: kernel-trace stats and the two PMC passes (separately, as the
# MI355X guide prescribes) over the same bench command.  Output under gpurun_out/<tag>_*.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
# (20 steps: the first 3-4 steps after an idle period run 10-15 % slower while the clocks ramp up)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --steps 3 --warmup 1 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_write.log 2>&1 || exit 1
grep -h '"metric"' $R/gpurun_out/${tag}_stats.log | cut -c1-200
