#!/bin/bash
# Round artefacts, run on the GPU box from the repo root (gpurun):  bash tools/profile_round.sh r01
# Writes everything under gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --steps 10 --warmup 3 --all-sites --no-cpu-baseline > $OUT/${TAG}_bench_nocpu.json 2> $OUT/${TAG}_bench_sites.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# PMC passes (counters only with --kernel-trace; one counter group per pass)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$c -- python $R/tools/prof_wgrad.py > /dev/null 2>&1
done
python $R/tools/pmc_to_json.py xf_wgrad wgrad_dma_kernel $(find $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv") > $OUT/${TAG}_pmc_summary.json
python $R/tools/gemm_bench.py > $OUT/${TAG}_gemm_microbench.txt 2>&1
tail -3 $OUT/${TAG}_bench.json; cat $OUT/${TAG}_pmc_summary.json; head -12 $OUT/${TAG}_bench_kernel_stats.csv
