import csv, sys, glob, collections
for tag in sys.argv[1:]:
    files = glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv')
    if not files: print(tag, 'no file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row['Kernel_Name'].split('(')[0][:60]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    print('==', tag)
    for k, d in acc.items():
        if any(x in k for x in ('k_strided','k_c2r','k_collapse')):  # k_c2r_invariants, k_collapse_inv included
            print(k, {c: '%.4g' % v for c, v in d.items()})
