# SQ occupancy / stall counters for selected micro-benchmarks:  bash tools/pmc_sq.sh attn wgrad768 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2 $R/gpurun_out/pmc_sq3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/tools/prof_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python $R/tools/prof_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq3 -- python $R/tools/prof_one.py "$@" > /dev/null 2>&1
cd $R && python tools/pmc_summary.py $(find gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 -name "*counter_collection.csv") | grep -v "at::native" 
