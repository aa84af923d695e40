# same-box A/B of an environment toggle: tools/ab_env.sh VAR=VALUE  (runs bench.py with and without it, three times each, alternating)
for i in 1 2 3; do
  env "$1" timeout -k 10 100 python bench.py --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
done
