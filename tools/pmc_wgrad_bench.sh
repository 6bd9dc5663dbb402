#!/bin/bash
# L2-miss traffic (FETCH_SIZE) of the grouped weight-gradient kernel inside the bench step.  usage: pmc_wgrad_bench.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmcw_*
for c in FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcw_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
done
python $R/tools/pmc_to_json.py xf_wgrad _table_kernel $(find $OUT/pmcw_FETCH_SIZE -name "*counter_collection.csv") | grep -E "FETCH|read_bytes"
