set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "res_ln" 2>&1 | tail -3
timeout -k 10 400 python -u -m pytest tests/test_gpu_tta.py -x -q --timeout 200 --timeout-method thread -k "packed or tiny or spread or dropin or autograd" 2>&1 | tail -3
for v in 0 32768 0 32768; do
  echo "== NEAT_FUSE_LN_MAX_ROWS=$v"
  NEAT_FUSE_LN_MAX_ROWS=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
