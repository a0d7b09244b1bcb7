# kernel-trace stats of the packed step only (run on the GPU box from the repo root); output gpurun_out/prof_packed/
# usage: bash tools/profile_packed.sh [extra bench.py arguments, e.g. --model ViT-L-14]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_packed; rm -rf $O; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode --pack-tokens "$@" > $O/bench.log 2>&1 || exit 1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/kt
grep -h '^{' $O/bench.log | cut -c1-200
