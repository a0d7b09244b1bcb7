set -o pipefail
for v in _NO_W_FILL _NO_W_MFMA _NO_ALL; do
  echo "== variant libneat_hip$v.so"
  NEAT_HIP_LIB=$PWD/neat_amd/libneat_hip$v.so NEAT_GEMM=panel M4=14848 M3=11136 ROUNDS=6 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -E "proj fwd|out  fwd|dfc|dqkv|do  "
done
