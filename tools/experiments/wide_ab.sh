set -o pipefail
for v in 0 1 2 0 1 2; do
  echo "== NEAT_GEMM_WIDE512=$v"
  NEAT_GEMM_WIDE512=$v timeout -k 10 200 python3 bench.py --pack-tokens --steps 8 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
