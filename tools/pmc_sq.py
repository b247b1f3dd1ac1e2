"""per-kernel means of the SQ counters of rocprofv3 --pmc passes.  usage: python tools/pmc_sq.py <filter> a_counter_collection.csv [b_...]"""
import collections, csv, sys
pat = sys.argv[1]
for f in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if pat in k:
            print(k)
            for c, x in sorted(v.items()):
                print("   %-24s n=%d mean=%.4g" % (c, len(x), sum(x) / len(x)))
