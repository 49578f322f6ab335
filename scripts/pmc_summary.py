import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[(r['Kernel_Name'][:60], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if any(x in k[0] for x in sys.argv[2:]):
        print(k[0], k[1], {a: '%.3g' % b for a, b in v.items()})
