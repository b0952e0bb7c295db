"""Compact a rocprofv3 counter_collection.csv to one line per (kernel, counter): launches, sum, mean (the raw file of a
whole training step is tens of MB): python tools/pmc_compact.py file.csv"""
import collections, csv, sys
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Counter_Name"])
    n, s = acc.get(k, (0, 0.0))
    acc[k] = (n + 1, s + float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Counter_Name", "launches", "sum", "mean_per_launch"])
for (k, c), (n, s) in acc.items():
    w.writerow([k, c, n, s, s / n])
