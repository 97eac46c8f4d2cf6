"""Per-kernel totals of a rocprofv3 --pmc collection CSV: for every kernel name the number of dispatches and the SUM of each counter over them,
sorted by the first counter given.  usage: python scripts/probes/pmc_by_kernel.py <counter_collection.csv> [sort counter]"""
import collections
import csv
import re
import sys

rows = csv.DictReader(open(sys.argv[1]))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("fn2::", "")[:70]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
names = sorted({c for v in tot.values() for c in v})
key = sys.argv[2] if len(sys.argv) > 2 else names[0]
print("%-72s %6s " % ("kernel", "disp") + " ".join("%14s" % n[:14] for n in names))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get(key, 0)):
    print("%-72s %6d " % (k, len(disp[k])) + " ".join("%14.0f" % v.get(n, 0) for n in names))
