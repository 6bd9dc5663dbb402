#!/bin/bash
# Round artefacts, run on the GPU box from the repo root (gpurun):  bash tools/profile_round.sh r06
# Writes everything under gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --steps 10 --warmup 3 --all-sites --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_nocpu.json 2> $OUT/${TAG}_bench_sites.txt
python $R/bench.py --steps 10 --warmup 3 --all-sites --no-cpu-baseline --no-extras --padded > $OUT/${TAG}_bench_padded.json 2> $OUT/${TAG}_bench_sites_padded.txt
# the reference's own default batch (examples/training/train_grande.py:45): which sites the launch-bound regime spends its time in
python $R/bench.py --batch 160 --steps 20 --warmup 5 --all-sites --no-cpu-baseline --no-extras --no-other-layout > $OUT/${TAG}_bench_b160.json 2> $OUT/${TAG}_bench_sites_b160.txt
# the host feed: worker scaling and where the feed thread's time goes; the fed step loop against replayed / cycled batches
python $R/tools/feed_probe.py 2>&1 | grep -v "^Too\|^tokeni\|^Tokeni\|amdgpu.ids" > $OUT/${TAG}_feed_probe.txt
python $R/tools/feed_e2e_probe.py 2>&1 | grep -v "^Too\|^tokeni\|^Tokeni\|amdgpu.ids" > $OUT/${TAG}_feed_e2e_probe.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-layout --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# HBM traffic from PMC passes over THE BENCH COMMAND itself (counters only with --kernel-trace; one counter per pass): the nominated
# kernels per launch (gemm_ring1_kernel<14> = the ring GEMM with the LayerNorm backward in its write-out: top row of the kernel table;
# the grouped weight gradient) and the whole step (2 x FETCH_SIZE + WRITE_SIZE over every kernel / steps in the trace)
export PMC_LAYOUT=packed
export PMC_COMMAND="rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-layout --no-extras (round $TAG)"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-layout --no-extras > /dev/null 2>&1
done
python $R/tools/pmc_to_json.py "dgrad_lnbwd=gemm_ring1_kernel<14>" "xf_wgrad=wgrad256_table_kernel" -- $(find $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv") > $OUT/${TAG}_pmc_summary.json
cp $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_bench_FETCH_SIZE.csv
cp $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_bench_WRITE_SIZE.csv
# configs[4] shape (d = 512, 16 heads of 32, batch 2048): bf16 and MXFP8 operand lines
python $R/bench.py --config coati2_shape --batch 2048 --steps 5 --warmup 2 --no-cpu-baseline --no-other-layout --all-sites > $OUT/${TAG}_bench_coati2_bf16.json 2> $OUT/${TAG}_bench_coati2_bf16_sites.txt
python $R/bench.py --config coati2_shape --batch 2048 --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-layout --all-sites > $OUT/${TAG}_bench_coati2_fp8.json 2> $OUT/${TAG}_bench_coati2_fp8_sites.txt
python $R/tools/rb16_bench.py 50000 > $OUT/${TAG}_rb16_microbench.txt 2>&1
python $R/tools/gemm_bench.py > $OUT/${TAG}_gemm_microbench.txt 2>&1
python $R/tools/probes/membw.py > $OUT/${TAG}_membw.txt 2>&1
bash $R/tools/pmc_sq.sh > /dev/null 2>&1 && cp $OUT/pmcsq_table.txt $OUT/${TAG}_sq_counters.txt
# matrix-core utilisation per kernel (second SQ pass): SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES
COATI_PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" bash $R/tools/pmc_sq.sh > /dev/null 2>&1 && cp $OUT/pmcsq_table.txt $OUT/${TAG}_sq_mfma.txt
python $R/tools/cpu_baseline_full.py > $OUT/${TAG}_cpu_baseline_full.json 2> /dev/null
tail -3 $OUT/${TAG}_bench.json; cat $OUT/${TAG}_pmc_summary.json; head -12 $OUT/${TAG}_bench_kernel_stats.csv; cat $OUT/${TAG}_cpu_baseline_full.json
