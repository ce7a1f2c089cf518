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
