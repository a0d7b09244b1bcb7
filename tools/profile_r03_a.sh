# round 3, first look: kernel traces of the packed text path, the gallery encode (two models) and the MCQ scorer
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof3a; rm -rf $O; mkdir -p $O
run() { # tag, args...
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 bench.py "$@" > $O/$tag.log 2>&1 || return 1
  cp $(ls $O/$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $O/$tag
}
run packed --pack-tokens --steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode &&
run gallery_vitb32 --workload gallery --gallery 5000 &&
run gallery_vitl14 --workload gallery --model ViT-L-14 --gallery 1000 &&
run mcq --workload mcq --steps 10 --warmup 2
grep -h '^{' $O/*.log | cut -c1-400
