# Round-3 profile recipe (run on the GPU box from the repo root): kernel-trace stats of the full-length and the packed
# step, one --pmc pass per counter for each, the gallery encode of two models, the MCQ scorer.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
FULL="--steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode --no-packed"
PACK="--steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode --pack-tokens"
kt() { # tag, args...
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 bench.py "$@" > $O/$tag.log 2>&1 || return 1
  cp $(ls $O/$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $O/$tag
}
pmc() { # tag, counters, args...
  tag=$1; ctr=$2; shift; shift
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 bench.py "$@" > $O/pmc_$tag.log 2>&1 || return 1
  python3 tools/summarize_pmc.py $(ls $O/pmc_$tag/*/*counter_collection.csv | head -1) $O/${tag}.csv || return 1
  rm -rf $O/pmc_$tag
}
ONE="--steps 1 --warmup 1 --no-cpu-baseline --skip-gallery-encode"
kt bench $FULL && kt bench_packed $PACK &&
pmc bench_pmc_fetch_size FETCH_SIZE $ONE --no-packed &&
pmc bench_pmc_write_size WRITE_SIZE $ONE --no-packed &&
pmc bench_pmc_mfma_busy "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $ONE --no-packed &&
pmc bench_packed_pmc_fetch_size FETCH_SIZE $ONE --pack-tokens &&
pmc bench_packed_pmc_write_size WRITE_SIZE $ONE --pack-tokens &&
pmc bench_packed_pmc_mfma_busy "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $ONE --pack-tokens &&
kt gallery_vitb32 --workload gallery --gallery 5000 &&
kt gallery_vitl14 --workload gallery --model ViT-L-14 --gallery 1024 &&
kt mcq --workload mcq --steps 10 --warmup 2
grep -h '^{' $O/bench.log $O/bench_packed.log | cut -c1-300
