"""One training step out of a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python bench.py ...`):
every kernel of the last complete step (adamw .. adamw) with its start offset, duration and queue, the idle gaps of the main
queue, and the time during which kernels of BOTH queues are running.   python tools/step_trace.py DIR/bench_kernel_trace.csv"""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw")]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
qs = collections.Counter(r["Queue_Id"] for r in step)
main_q = qs.most_common(1)[0][0]
print(f"kernels {len(step)}  span {span:.1f} us  queues {dict(qs)}")
per = collections.defaultdict(lambda: [0, 0.0])
gaps, last_end = 0.0, None
side_busy = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0][:70]
    per[(r["Queue_Id"] == main_q, k)][0] += 1
    per[(r["Queue_Id"] == main_q, k)][1] += (e - s) / 1e3
    if r["Queue_Id"] == main_q:
        if last_end is not None and s > last_end:
            gaps += (s - last_end) / 1e3
        last_end = max(last_end or 0, e)
    else:
        side_busy += (e - s) / 1e3
main_busy = sum(v[1] for (m, _), v in per.items() if m)
print(f"main queue: busy {main_busy:.1f} us, idle gaps {gaps:.1f} us;  side queue busy {side_busy:.1f} us")
if "-v" in sys.argv:
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:8.1f} us  q{r['Queue_Id']}  {r['Kernel_Name'][:100]}")
else:
    for (m, k), (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{'main' if m else 'side'}  {n:4d} x  {t / n:8.1f} us = {t:9.1f} us   {k}")
