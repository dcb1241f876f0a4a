#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace: per 25 ms bin, the fraction of the bin with at least one kernel running and
the mean number of kernels in flight (the lanes of bench.py / pipeline.run_stream overlap their kernels).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --cpu-sample 0
    python tools/gpu_busy.py gpurun_out/prof"""
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']))
rows.sort()
t0=rows[0][0]; t1=max(r[1] for r in rows)
B=25_000_000
nb=int((t1-t0)/B)+1
busy=[0]*nb; conc=[0]*nb
ev=[]
for s,e,_ in rows: ev.append((s,1)); ev.append((e,-1))
ev.sort()
cur=0; last=t0
def add(a,b,c):
    while a<b:
        k=int((a-t0)/B); end=min(b,t0+(k+1)*B)
        if c>0: busy[k]+=end-a
        conc[k]+=c*(end-a)
        a=end
for t,d in ev:
    add(last,t,cur); cur+=d; last=t
print(" ".join("%.2f/%.1f"%(busy[k]/B,conc[k]/B) for k in range(nb)))
# what runs in low-concurrency bins? list kernels active around idle bins
