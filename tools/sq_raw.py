"""Mean per-wave value of every counter of every rocprofv3 --pmc pass under a directory, for the step kernels."""
import collections, csv, glob, sys
for fn in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        if 'step_kernel' in r['Kernel_Name']:
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        w = m.get('SQ_WAVES', 1.0)
        print(fn.split('/')[-3], k[:40], ' '.join('%s=%.0f' % (c.replace('SQ_', ''), x / w) for c, x in sorted(m.items()) if c != 'SQ_WAVES'), '(per wave; waves %d)' % w)
