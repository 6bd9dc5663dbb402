#!/bin/bash
# Where the waves of each kernel of the bench step spend their cycles (SQ counters, one pass).  usage: pmc_sq.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmcsq
# COATI_PMC_SQ: counter list (<= 8 SQ counters per pass); SQ_WAVE_CYCLES first: the table prints fractions of it
CTRS=${COATI_PMC_SQ:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"}
rocprofv3 --pmc $CTRS \
  --kernel-trace --output-format csv -d $OUT/pmcsq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-layout --no-extras "$@" > /dev/null 2>&1
python $R/tools/pmc_sq_table.py $(find $OUT/pmcsq -name "*counter_collection.csv") | tee $OUT/pmcsq_table.txt
