"""rocprofv3 --pmc SQ_* counter_collection CSV -> one line per kernel (per-launch averages, fractions of SQ_WAVE_CYCLES).
usage: pmc_sq_table.py <csv> [<csv> ...]"""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in acc.items():
    n = max(len(v) for v in c.values())
    avg = {name: sum(v) / len(v) for name, v in c.items()}
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    rows.append((sum(dur[k]) / max(1, len(c)), k, n, avg, wc))
rows.sort(reverse=True)
names = sorted({n for _, _, _, a, _ in rows for n in a if n != "SQ_WAVE_CYCLES"})
print("total_us(sampled)  launches  avg_us  kernel  | " + "  ".join(n.replace("SQ_", "") for n in names) + "   (fractions of SQ_WAVE_CYCLES)")
for tot, k, n, avg, wc in rows[:24]:
    d = sum(dur[k]) / len(dur[k])
    print(f"{tot:10.0f} {n:5d} {d:9.1f}  {k:60s} | " + "  ".join(f"{(avg.get(x, 0.0) / wc if wc else 0.0):6.3f}" for x in names))
