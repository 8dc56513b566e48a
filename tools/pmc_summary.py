import csv, collections, glob, sys
for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'conv_gemm' in r['Kernel_Name'] or 'wgrad' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: '%.4g' % (sum(v)/len(v)) for k, v in sorted(agg.items())})
