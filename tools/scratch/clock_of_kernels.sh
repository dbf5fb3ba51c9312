#!/bin/bash
# Shader clock WHILE a kernel runs = GRBM_GUI_ACTIVE (per XCD) / its duration, and MFMA busy, from one rocprofv3 --pmc pass:
#   tools/scratch/clock_of_kernels.sh harness <attn_bench args>     the standalone attention harness (back-to-back launches)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ck; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/ck -o run -- $GRAFT_REPO_ROOT/tools/scratch/attn_bench "$@" > /tmp/ck.log 2>&1
python3 - <<'PY'
import csv, collections, re
acc = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for r in csv.DictReader(open("/tmp/ck/run_counter_collection.csv")):
    k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[k][0] += float(r["Counter_Value"]) / 8; acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); acc[k][3] += 1
    elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        acc[k][2] += float(r["Counter_Value"]) / 1024
for k, (cyc, ns, busy, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
    if n: print("%-28s n=%4d  clock %.2f GHz  busy %.3f  avg %.1f us" % (k[:28], n, cyc / ns, busy / cyc if cyc else 0, ns / n / 1e3))
PY
