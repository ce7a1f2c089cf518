#!/usr/bin/env python3
"""Bytes fetched per cell and ring pass from two `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv` runs of
tools/pmc_traffic.py (FETCH_SIZE in KiB, x 2.000: the calibration of profiles/pmc_summary.json), side by side per radius
(timing experiment: profiles/r05_segment_balance.md section 5).

    python tools/experiments/fetch_cmp.py <run A dir> <run B dir> <n of A> <n of B>
"""
import csv,re,collections,glob,sys
def load(d):
    p=glob.glob(d+'/**/*counter_collection.csv',recursive=True)[0]
    out=collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if r['Counter_Name']!='FETCH_SIZE': continue
        m=re.search(r'ring_kernel<float, (\d+), (true|false)',r['Kernel_Name'])
        if m: out[(int(m.group(1)),m.group(2)=='true')]=float(r['Counter_Value'])
    return out
a=load(sys.argv[1]); b=load(sys.argv[2])
na,nb=int(sys.argv[3]),int(sys.argv[4])
print("R pass  B/cell n=%d  n=%d"%(na,nb))
for k in sorted(a):
    if k in b and k[0]%5==0: print(k, round(a[k]*1024*2/na/na,2), round(b[k]*1024*2/nb/nb,2))
