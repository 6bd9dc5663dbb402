import csv, sys, collections, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}")
