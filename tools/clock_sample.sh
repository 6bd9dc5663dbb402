#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 2500 --warmup 10 --no-cpu-baseline --no-other-layout > /tmp/b.json 2>/dev/null &
BP=$!
sleep 32
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
tail -1 /tmp/b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo idle:
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i power
