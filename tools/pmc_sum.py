import csv, glob, sys, collections
# sum counters of ctu_search_kernel over all dispatches in rocprofv3 counter_collection csv files
tot = collections.Counter(); nd = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'ctu_search' not in row.get('Kernel_Name', ''): continue
        tot[row['Counter_Name']] += float(row['Counter_Value']); nd[row['Counter_Name']] += 1
nctu = float(sys.argv[2]) if len(sys.argv) > 2 else 1
for k in sorted(tot): print('%-28s %14.4e  per CTU %12.1f  (%d dispatches)' % (k, tot[k], tot[k] / nctu, nd[k]))
