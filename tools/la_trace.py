import csv,collections,sys,glob
f=(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')+glob.glob(sys.argv[1]+'/*kernel_trace.csv'))[0]
rows=list(csv.DictReader(open(f)))
ws=[r for r in rows if r['Kernel_Name'].startswith('k_ldlt_workers')]
w=ws[-1]; t0=int(w['Start_Timestamp']); t1=int(w['End_Timestamp'])
print('workers kernel', (t1-t0)/1e3,'us')
sel=[r for r in rows if int(r['Start_Timestamp'])>=t0-2500000 and int(r['End_Timestamp'])<=t1+1000]
agg=collections.defaultdict(list)
for r in sel:
    agg[r['Kernel_Name'][:30]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print('%-32s n=%4d avg %8.1f us total %9.1f us'%(k,len(v),sum(v)/len(v),sum(v)))
gates=[r for r in sel if r['Kernel_Name'].startswith('k_la_gate')]
print('gate durations:',[round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in gates])
sigs=[r for r in sel if r['Kernel_Name'].startswith('k_la_signal')]
ends=[(int(r['End_Timestamp'])-t0)/1e3 for r in sigs]
print('chain block periods:',[round(b-a) for a,b in zip(ends,ends[1:])])
