cd /tmp && export TMPDIR=/tmp
export COATI_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
rocprofv3 --kernel-trace --output-format csv -d /tmp/dptr -o dp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 10 --no-cpu-baseline --no-other-layout > /tmp/dp.json 2>/tmp/dp.err
tail -2 /tmp/dp.err
f=$(find /tmp/dptr -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/step_trace.py $f -v > $GRAFT_REPO_ROOT/gpurun_out/dp_step_trace_v.txt
tail -40 $GRAFT_REPO_ROOT/gpurun_out/dp_step_trace_v.txt | cut -c1-150
