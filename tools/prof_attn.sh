#!/bin/bash
# per-kernel times of the attention microbench under rocprofv3: usage tools/prof_attn.sh TAG [B]
TAG=${1:-attn}
B=${2:-88}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/tools/bench_attn.py $B > $OUT/log.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$OUT/trace/run_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %5s avg %9.1f us  total %8.2f ms  %5s%%" % (re.sub(r"[<(].*", "", r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -7 $OUT/log.txt
if [ "$3" == "pmc" ]; then
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- python3 $ROOT/tools/bench_attn.py $B > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o run -- python3 $ROOT/tools/bench_attn.py $B > $OUT/pmc_mfma.log 2>&1
  cd $ROOT
  python3 - <<PY
import collections, csv, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("pmc_fetch", "pmc_mfma"):
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "attn" in k:
        line = "%-28s" % k
        if v.get("FETCH_SIZE"):
            line += " fetch %8.1f MB/launch" % (2 * v["FETCH_SIZE"] / n[(k, "FETCH_SIZE")] * 1024 / 1e6)
        if v.get("GRBM_GUI_ACTIVE"):
            line += "  mfma_busy %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0))
        print(line)
PY
fi
