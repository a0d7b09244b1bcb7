# A/B of the 16-wave split-K form of the 64 x 512 unit (NEAT_GEMM_WIDE_SPLIT=0: the 8-wave kernel)
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -2 || exit 1
for v in 0 1 0 1; do
  echo "== packed step, NEAT_GEMM_WIDE_SPLIT=$v"
  NEAT_GEMM_WIDE_SPLIT=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 0 1; do
  echo "== layer GEMMs at the packed size, NEAT_GEMM_WIDE_SPLIT=$v"
  NEAT_GEMM_WIDE_SPLIT=$v M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -9
done
