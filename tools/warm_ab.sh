# A/B of the weight warm-up at kernel start and the epilogue-operand touch (NEAT_GEMM_WARM_W bit 0 / bit 1): GEMM tests
# first, then the packed and the full-length step, then the eight GEMMs of a layer at the packed row counts.
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -3 || exit 1
for v in 1 3 1 3; do
  echo "== packed step, NEAT_GEMM_WARM_W=$v"
  NEAT_GEMM_WARM_W=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 1 3 1 3; do
  echo "== full-length step, NEAT_GEMM_WARM_W=$v"
  NEAT_GEMM_WARM_W=$v timeout -k 10 200 python3 bench.py --no-packed --steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 1 3; do
  NEAT_GEMM_WARM_W=$v M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -9
done
