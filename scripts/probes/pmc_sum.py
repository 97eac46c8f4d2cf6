import csv, sys, collections
f = sys.argv[1]; pat = sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
