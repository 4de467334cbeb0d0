"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter)."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    if "pba::" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("(")[0][-44:]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    n = max(v[1] for v in cs.values())
    if n < 5:
        continue
    print(k, "(%d launches)" % n)
    for c, v in sorted(cs.items()):
        print("    %-26s %14.0f" % (c, v[0] / v[1]))
