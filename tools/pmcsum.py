import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    if not any(t in k.lower() for t in ('attn', 'gemm', 'cijk', 'rel_')): continue
    print(k)
    for c, val in v.items(): print(f"    {c:32s} {val / cnt[(k, c)]:16.1f}  (avg of {cnt[(k, c)]})")
