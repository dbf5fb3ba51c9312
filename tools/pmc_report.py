"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_report.py DIR [name-filter]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:70], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        h, m = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
        print("   L2 hit rate %.3f" % (h / (h + m)))
