#!/bin/bash
# rocprofv3 kernel stats of the other shapes (hop-320 64 x 3 s, hop-600 32 x 30 s, hop-600 B = 1): gpurun_out/<tag>_{h320,30s,b1}
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_h320 -- python3 $R/bench.py --arch hop320 --steps 10 --warmup 3 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_h320.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_30s -- python3 $R/bench.py --clips 32 --clip-seconds 30 --steps 5 --warmup 2 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_30s.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_b1 -- python3 $R/bench.py --clips 1 --steps 50 --warmup 5 --lanes 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/${tag}_b1.log 2>&1 || exit 1
for s in h320 30s b1; do grep -h '"metric"' $R/gpurun_out/${tag}_$s.log | cut -c1-220; done
