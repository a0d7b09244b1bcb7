set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -2
for v in 0 1 0 1; do
  echo "== NEAT_GEMM_ONE_ROUND=$v"
  NEAT_GEMM_ONE_ROUND=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 0 1; do
  NEAT_GEMM_ONE_ROUND=$v NEAT_GEMM=panel M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -E "out  fwd|proj fwd"
  NEAT_GEMM_ONE_ROUND=$v timeout -k 10 100 python3 bench.py --workload mcq --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mcq', d['value'], d['latency'])"
done
