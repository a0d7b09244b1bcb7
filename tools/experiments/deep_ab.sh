# correctness, then A/B of gemm_deep_kernel on underfilled shapes (run on the GPU box from the repo root)
set -o pipefail
O=gpurun_out/deep; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for d in 0 125; do
  echo "== packed text shapes, NEAT_GEMM_DEEP_UNITS=$d"
  NEAT_GEMM_DEEP_UNITS=$d NEAT_GEMM=panel M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== 256-image ViT-B/32 gallery batch (C=768), NEAT_GEMM_DEEP_UNITS=$d"
  NEAT_GEMM_DEEP_UNITS=$d NEAT_GEMM=panel M4=12800 C=768 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
