# A/B: de-phased CUs in the 256 x 256 kernel (NEAT_GEMM_STAGGER = sleeps of ~3.4 us for every other workgroup of an XCD)
set -o pipefail
for v in 0 1 2 3 4 0; do
  echo "== NEAT_GEMM_STAGGER=$v"
  NEAT_GEMM_STAGGER=$v timeout -k 10 200 python3 bench.py --no-packed --steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
done
