#!/usr/bin/env python3
"""Per-kernel PMC values of the LAST dispatch of each pipeline kernel, from one or more rocprofv3 --pmc output dirs."""
import collections, csv, glob, sys
vals = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            k = (k[5:] if k.startswith('void ') else k).split('(')[0]
            if not k.startswith('gz_'):
                continue
            vals.setdefault(k, collections.OrderedDict())[r['Counter_Name']] = float(r['Counter_Value'])   # later rows win
names = []
for k in vals:
    for c in vals[k]:
        if c not in names:
            names.append(c)
print("%-22s" % "kernel" + "".join("%16s" % c[-15:] for c in names))
for k, v in vals.items():
    print("%-22s" % k[:22] + "".join("%16.4g" % v.get(c, float('nan')) for c in names))
