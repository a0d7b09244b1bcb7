# A/B of the row tiles per unit of the panel kernel (NEAT_GEMM_RT=8 forces the 128-row unit; unset = the launcher's choice)
set -o pipefail
for rt in 8 7 6 5; do
  NEAT_GEMM_RT=$rt timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" 2>&1 | tail -1 || exit 1
done
for v in 8 0 8 0; do
  echo "== packed step, NEAT_GEMM_RT=$v"
  NEAT_GEMM_RT=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 8 0 8 0; do
  echo "== full-length step, NEAT_GEMM_RT=$v"
  NEAT_GEMM_RT=$v timeout -k 10 200 python3 bench.py --no-packed --steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for v in 8 7 6 5; do
  echo "== layer GEMMs at full length, panel kernel, NEAT_GEMM_RT=$v"
  NEAT_GEMM_RT=$v NEAT_GEMM=panel timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -9
done
for v in 8 7 6 5; do
  echo "== layer GEMMs at the packed size, panel kernel, NEAT_GEMM_WIDE512=0 NEAT_GEMM_RT=$v"
  NEAT_GEMM_WIDE512=0 NEAT_GEMM_RT=$v NEAT_GEMM=panel M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -9
done
