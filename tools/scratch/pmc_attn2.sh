#!/bin/bash
cd /tmp && export TMPDIR=/tmp
BIN=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn2
mkdir -p $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- $GRAFT_REPO_ROOT/tools/scratch/$BIN "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "attn" in k:
                print(k, {c: "%.4g (n=%d)" % (x, n[(k, c)]) for c, x in v.items()})
PY
tail -3 $OUT/p1.log
