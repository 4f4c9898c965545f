import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def sh(n): return n.split('(')[0].replace('void ','').strip()
ks=[(sh(r['Kernel_Name']),int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows]
idx=[i for i,k in enumerate(ks) if k[0].startswith('k_node_update<2>')]
i0=idx[50]-1
t0=ks[i0][1]
for k in ks[i0:i0+12]:
    print('%-34s start %8.2f end %8.2f dur %6.2f'%(k[0],(k[1]-t0)/1e3,(k[2]-t0)/1e3,(k[2]-k[1])/1e3))
