#!/bin/bash
# pwconv1's duration three ways in one process each: rocprofv3 kernel trace, the device stamps bench.py reports
# (wt_plan_set_timing("@cnx.pwconv1")), and - second run - HIP events bracketing the launch ("cnx.pwconv1").
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for mode in stamps events; do
  if [ $mode = events ]; then export WT_BENCH_TIMING=events; else unset WT_BENCH_TIMING; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tve_$mode -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/tve_$mode.log 2>&1 || exit 1
  python3 - $mode <<'PY'
import csv, glob, json, statistics, sys
mode = sys.argv[1]
f = sorted(glob.glob('/root/repo/gpurun_out/tve_%s/**/*kernel_trace.csv' % mode, recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d, gaps = [], []
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'gemm16s_kernel<128, 192, 4, 2, 3, 2, 1' in r['Kernel_Name']:
        d.append((e - s) / 1e3)
        if prev_end is not None:
            gaps.append((s - prev_end) / 1e3)
    prev_end = e
print("== bench.py timing mode: %s (pwconv1 launches in the trace: %d)" % (mode, len(d)))
for k in range(0, len(d), 12 * 4):
    print("   steps %2d-%2d: rocprofv3 kernel duration %.1f us, idle gap in front of the launch %.1f us" % (k // 12, k // 12 + 3, statistics.mean(d[k:k + 48]), statistics.mean(gaps[k:k + 48])))
print("   timed steps (last 240 launches): rocprofv3 kernel %.2f us; kernel + gap %.2f us" % (statistics.mean(d[-240:]), statistics.mean(d[-240:]) + statistics.mean(gaps[-240:])))
line = [l for l in open('/root/repo/gpurun_out/tve_%s.log' % mode) if l.startswith('{"metric"')][-1]
j = json.loads(line)
print("   bench.py reports: avg_launch_ms %.4f (%s), ms_per_step %.3f" % (j['roofline']['avg_launch_ms'], mode, j['ms_per_step']))
PY
done
