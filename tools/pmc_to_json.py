"""rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE pass + WRITE_SIZE pass over THE BENCH COMMAND) -> JSON on stdout:
  * one entry per nominated launch site: per-launch HBM bytes of its kernel (bench.py `roofline.traffic`);
  * "_step": the whole training step -- bytes of every kernel of the trace summed and divided by the number of steps in it
    (= adamw_kernel launches), with the per-kernel breakdown of the 12 largest;
usage: pmc_to_json.py <site>=<kernel-substring> [<site>=<kernel-substring> ...] -- <csv> [<csv> ...]   (env PMC_COMMAND, PMC_LAYOUT)
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section): it is doubled here.  WRITE_SIZE is taken as reported."""
import csv, sys, json, collections, os, re
args = sys.argv[1:]
sep = args.index("--")
sites = dict(a.split("=", 1) for a in args[:sep])
acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
steps = collections.Counter()
for f in args[sep + 1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if k.startswith("adamw_kernel"):
            steps[r["Counter_Name"]] += 1
out = {}
for site, sub in sites.items():
    ks = [k for k in acc if sub in k]
    ent = {"kernel": sub, "layout": os.environ.get("PMC_LAYOUT", "packed")}
    if os.environ.get("PMC_COMMAND"):
        ent["command"] = os.environ["PMC_COMMAND"]
    fetch = [v for k in ks for v in acc[k].get("FETCH_SIZE", [])]
    write = [v for k in ks for v in acc[k].get("WRITE_SIZE", [])]
    ent["launches_sampled"] = max(len(fetch), len(write))
    if fetch:
        ent["FETCH_SIZE_avg"] = sum(fetch) / len(fetch)
        ent["hbm_read_bytes_per_launch"] = ent["FETCH_SIZE_avg"] * 1024 * 2
    if write:
        ent["WRITE_SIZE_avg"] = sum(write) / len(write)
        ent["hbm_write_bytes_per_launch"] = ent["WRITE_SIZE_avg"] * 1024
    if fetch and write:
        ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
    out[site] = ent
nf, nw = steps.get("FETCH_SIZE", 0), steps.get("WRITE_SIZE", 0)
if nf and nw:
    per = []
    for k, c in acc.items():
        rd = sum(c.get("FETCH_SIZE", [])) * 1024 * 2 / nf
        wr = sum(c.get("WRITE_SIZE", [])) * 1024 / nw
        per.append((rd + wr, rd, wr, len(c.get("FETCH_SIZE", [])) / nf, k))
    per.sort(reverse=True)
    out["_step"] = {"steps_in_trace": [nf, nw], "hbm_read_GB_per_step": round(sum(p[1] for p in per) / 1e9, 2),
                    "hbm_write_GB_per_step": round(sum(p[2] for p in per) / 1e9, 2),
                    "hbm_GB_per_step": round(sum(p[0] for p in per) / 1e9, 2),
                    "note": "2 x FETCH_SIZE + WRITE_SIZE of every kernel of the trace / steps in the trace (adamw launches)",
                    "largest": [{"kernel": k, "launches_per_step": round(n, 1), "read_GB": round(rd / 1e9, 2), "write_GB": round(wr / 1e9, 2)}
                                for _, rd, wr, n, k in per[:12]]}
print(json.dumps(out, indent=1))
