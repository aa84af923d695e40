#!/bin/bash
# the WT_* switches exist in the LAB build only
export WAVTOK_HIP_LIB=${WAVTOK_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools/lib/libwavtok_hip_lab.so}
# HBM-side read traffic (FETCH_SIZE) of the ConvNeXt GEMMs per tile order: WT_GEMM16S_GM x WT_GEMM16S_GN, one rocprofv3 --pmc pass each;
# then the launch time of pwconv1 / pwconv2 per order in the GEMM lab.  Output: gpurun_out/gmn_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
out=$R/gpurun_out/gmn_traffic.txt
: > $out
for c in "0 0" "5 6" "10 3" "4 6" "8 6" "15 3" "6 4"; do
  set -- $c
  tag=gmn_$1_$2
  if [ "$1" = "0" ]; then
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/$tag.log 2>&1 || exit 1
  else
    WT_GEMM16S_GM=$1 WT_GEMM16S_GN=$2 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 1 > $R/gpurun_out/$tag.log 2>&1 || exit 1
  fi
  python3 - "$R/gpurun_out/$tag" "GM=$1 GN=$2" >> $out <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for tag, pat in (("pwconv1", "gemm16s_kernel<128, 192, 4, 2, 3, 2, 1"), ("pwconv2", "gemm16s_kernel<128, 192, 4, 2, 3, 3, 0"), ("res.conv", "gemm16s_kernel<128, 192, 4, 2, 3, 0, 0")):
        if pat in k and r["Counter_Name"] == "FETCH_SIZE":
            acc[tag].append(float(r["Counter_Value"]))
print(sys.argv[2], " ".join("%s read %.1f MB" % (t, 2 * 1024 * sum(v) / len(v) / 1e6) for t, v in sorted(acc.items())))
PY
done
cd $R
for c in "8 0" "5 6" "10 3" "4 6" "8 6" "15 3" "6 4"; do
  set -- $c
  echo "--- lab GM=$1 GN=$2" >> $out
  LAB_GM=$1 LAB_GN=$2 timeout -k 10 120 tools/micro/gemm_lab 12 shipped 2>&1 | grep -E "shipped" >> $out
done
cat $out
