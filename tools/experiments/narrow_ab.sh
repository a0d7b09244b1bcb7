# A/B of the 128 x 128 panel units on underfilled shapes (run on the GPU box from the repo root)
set -o pipefail
O=gpurun_out/narrow; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "gemm" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for fill in 0 80; do
  echo "== packed text shapes, NEAT_GEMM_NARROW_FILL=$fill"
  NEAT_GEMM_NARROW_FILL=$fill NEAT_GEMM=panel M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== gallery batch of 250 (ViT-B/32 vision, C=768), NEAT_GEMM_NARROW_FILL=$fill"
  NEAT_GEMM_NARROW_FILL=$fill NEAT_GEMM=panel M4=12500 C=768 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== library's choice (policy), packed"
M4=14848 M3=11136 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
