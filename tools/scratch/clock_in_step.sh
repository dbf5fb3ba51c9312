#!/bin/bash
# shader clock (GRBM_GUI_ACTIVE per XCD / duration) and MFMA busy per kernel inside the step, for a library build:
#   tools/scratch/clock_in_step.sh [path to libvlm_hip.so]
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export VLM_LIB_PATH=$1
rm -rf /tmp/cs; VLM_WGRAD_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cs -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibrate --no-gemm-timer --no-secondary --no-merge > /tmp/cs.log 2>&1
python3 - <<'PY'
import csv, collections, re
acc = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for r in csv.DictReader(open("/tmp/cs/run_counter_collection.csv")):
    k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[k][0] += float(r["Counter_Value"]) / 8; acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); acc[k][3] += 1
    elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        acc[k][2] += float(r["Counter_Value"]) / 1024
for k, (cyc, ns, busy, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:7]:
    if n: print("%-28s n=%4d  clock %.2f GHz  busy %.3f  avg %.1f us" % (k[:28], n, cyc / ns, busy / cyc if cyc else 0, ns / n / 1e3))
PY
