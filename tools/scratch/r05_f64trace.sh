#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f64 -o run -- python3 $R/tools/bench_f64_leg.py > $O/trace_f64.log 2>&1
python3 tools/prof_summary.py $O/trace_f64/run_kernel_stats.csv $O/f64_stats.csv x; head -9 $O/f64_stats.csv | tail -6; grep -E "^(regmean)" $O/trace_f64.log
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/trace_f64/run_kernel_trace.csv")))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if 'gemm_f64_batched' in r['Kernel_Name']:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        gx,gy,gz=int(r['Grid_Size_X'])//256,int(r['Grid_Size_Y']),int(r['Grid_Size_Z'])
        key=('N1tile' if gx==1 else 'wide', gz)
        agg[key][0]+=1; agg[key][1]+=d
for k,v in sorted(agg.items()): print(k, v[0]//2, 'launches/merge %.2f ms/merge avg %.1f us'%(v[1]/2e3, v[1]/v[0]))
PY
rm -rf $O/trace_f64
