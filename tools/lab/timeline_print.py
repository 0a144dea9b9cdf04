import csv,glob,sys
f=glob.glob('/root/repo/gpurun_out/lazy_prof/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
end=max(int(r['End_Timestamp']) for r in rows)
span=float(sys.argv[1]) if len(sys.argv)>1 else 17.0
rows=[r for r in rows if int(r['Start_Timestamp'])>=end-int(span*1e6)]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if e-s>20000: print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:8.1f} q{r['Queue_Id']:>2} {r['Kernel_Name'][:70]}")
