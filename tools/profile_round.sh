# Round profile recipe (run on the GPU box from the repo root): kernel-trace stats, then one --pmc pass per counter.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --skip-gallery-encode --no-packed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $ARGS > $O/kt.log 2>&1 || exit 1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --skip-gallery-encode --no-packed > $O/pmc_$c.log 2>&1 || exit 1
  python3 tools/summarize_pmc.py $(ls $O/pmc_$c/*/*counter_collection.csv | head -1) $O/bench_pmc_$(echo $c | tr A-Z a-z).csv || exit 1
done
# matrix-pipe utilisation per kernel inside the step
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --skip-gallery-encode --no-packed > $O/pmc_mfma.log 2>&1 || exit 1
python3 tools/summarize_pmc.py $(ls $O/pmc_mfma/*/*counter_collection.csv | head -1) $O/bench_pmc_mfma_busy.csv || exit 1
rm -rf $O/pmc_mfma
timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes_auto.txt
NEAT_GEMM=panel timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes_panel.txt
NEAT_GEMM=t256 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes_t256.txt
# retrieval scoring at config-2 size: per-kernel time and FETCH_SIZE of the top-k passes (25 014 x 5 000, both directions)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tk -- python3 tools/topk_bench.py > $O/topk_bench.txt 2>&1 || exit 1
cp $(ls $O/tk/*/*kernel_stats.csv | head -1) $O/topk_kernel_stats.csv
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_tk -- python3 tools/topk_bench.py > $O/pmc_tk.log 2>&1 || exit 1
python3 tools/summarize_pmc.py $(ls $O/pmc_tk/*/*counter_collection.csv | head -1) $O/topk_pmc_fetch_size.csv || exit 1
# exact-fp32 similarity GEMM: rate per shape, and how busy the matrix pipe is (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))
timeout -k 10 120 python3 tools/f32_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/sim_gemm_f32.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sg -- python3 tools/f32_gemm_bench.py 25014x5000x512 > $O/pmc_sg.log 2>&1 || exit 1
python3 tools/summarize_pmc.py $(ls $O/pmc_sg/*/*counter_collection.csv $O/pmc_sg/*counter_collection.csv 2>/dev/null | head -1) $O/sim_gemm_pmc_mfma_busy.csv || exit 1
rm -rf $O/pmc_sg
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/tk $O/pmc_tk
head -8 $O/bench_kernel_stats.csv | cut -c1-160; head -6 $O/bench_pmc_fetch_size.csv | cut -c1-200; head -6 $O/bench_pmc_write_size.csv | cut -c1-200
