import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if not any(w in k for w in (sys.argv[2:] or ['fq1'])): continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if (k,r['Dispatch_Id']) not in seen: seen.add((k,r['Dispatch_Id'])); cnt[k]+=1
for k in acc:
    print(k,cnt[k])
    for c,v in sorted(acc[k].items()): print('   %-24s %14.0f'%(c,v/cnt[k]))
