set -o pipefail
for k in panel t256; do
  echo "== NEAT_GEMM=$k packed text shapes"
  NEAT_GEMM=$k M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
for k in panel t256; do
  echo "== NEAT_GEMM=$k gallery 250 (C=768)"
  NEAT_GEMM=$k M4=12500 C=768 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
for k in panel t256; do
  echo "== NEAT_GEMM=$k gallery 1000 (C=768)"
  NEAT_GEMM=$k M4=50000 C=768 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
