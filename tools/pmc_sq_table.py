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
mf = any("SQ_VALU_MFMA_BUSY_CYCLES" in a for _, _, _, a, _ in rows)
if mf:
    print("# matrix-core utilisation per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs): the counter ticks once per cycle of an")
    print("# executing MFMA, summed over the chip (calibrated: BUSY / SQ_INSTS_MFMA = 32.0 for v_mfma_f32_32x32x16_bf16, 16.0 for 16x16x32); bf16 rate =")
    print("# SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / time (rocprofv3's MfmaFlopsBF16), against the 2500 TFLOP/s dense peak")
print("total_us(sampled)  launches  avg_us  kernel  | " + "  ".join(n.replace("SQ_", "") for n in names) + "   (fractions of SQ_WAVE_CYCLES)")
for tot, k, n, avg, wc in rows[:24]:
    d = sum(dur[k]) / len(dur[k])
    extra = ""
    if mf and avg.get("SQ_BUSY_CYCLES"):
        extra = f"   mfma_util {avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (d * 2400.0 * 1024.0):6.3f}"
        if avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            tf = avg["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (d * 1e-6) / 1e12
            extra += f"   bf16 {tf:7.1f} TFLOP/s = {tf / 2500.0:5.3f} of peak"
    print(f"{tot:10.0f} {n:5d} {d:9.1f}  {k:60s} | " + "  ".join(f"{(avg.get(x, 0.0) / wc if wc else 0.0):6.3f}" for x in names) + extra)
