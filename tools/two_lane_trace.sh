#!/bin/bash
# rocprofv3 kernel trace of the default bench line's two blocks (one lane, then two lanes): how long do the kernels of the
# headline schedule take beside each other?  Output gpurun_out/<tag>_2lane (tag = $1); summarise with tools/two_lane_trace.py
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_2lane -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --repeats 2 > $R/gpurun_out/${tag}_2lane.log 2>&1 || exit 1
grep -h '"metric"' $R/gpurun_out/${tag}_2lane.log | cut -c1-200
