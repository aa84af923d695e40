for i in 1 2 3; do
  WAVTOK_HIP_LIB=$PWD/tools/lib/libwavtok_hip_prev.so timeout -k 10 100 python bench.py --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
done
