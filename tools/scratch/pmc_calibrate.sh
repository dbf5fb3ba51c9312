#!/bin/bash
# What SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE reads on a kernel that is nothing but back-to-back MFMAs (tools/scratch/mfma_shapes):
# the scale of every "mfma_busy_frac" in profiles/r0*_pmc_mfma_busy.json.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/cal -o run -- $GRAFT_REPO_ROOT/tools/scratch/mfma_shapes > /tmp/cal.log 2>&1
tail -2 /tmp/cal.log
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("/tmp/cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:12]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k, {c: x for c, x in v.items()}, "launches", n[(k, "GRBM_GUI_ACTIVE")])
    if v.get("GRBM_GUI_ACTIVE"):
        print("   busy / 1024 / (gui_active / 8) = %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)))
PY
