cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -- python $R/tools/prof_one.py wgrad768 wgrad1024 gelu1024 > /dev/null 2>&1
done
cd $R && for d in gpurun_out/pmc_*; do python tools/pmc_summary.py $(find $d -name "*counter_collection.csv"); done
