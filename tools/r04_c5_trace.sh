cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-recall --steps 20 --warmup 3 > $O/out.json 2> $O/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c5trace/*kernel_trace.csv')+glob.glob('gpurun_out/c5trace/*/*kernel_trace.csv')
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:int(r['Start_Timestamp']))
sc=[i for i,r in enumerate(rows) if 'coarse_scan_i8' in r['Kernel_Name']]
per=[(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3 for a,b in zip(sc,sc[1:])]
print('periods',[round(x) for x in per])
# pipelined region = where period < 850: print the kernels between two consecutive scans there
best=[j for j,x in enumerate(per) if x<850]
if best:
    j=best[len(best)//2]; a,b=sc[j],sc[j+1]
    t0=int(rows[a]['Start_Timestamp'])
    for r in rows[a-4:b+3]:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:60]}")
PY
