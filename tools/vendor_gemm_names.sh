#!/bin/bash
# hipBLASLt's choice per shape: kernel name (the MT... part is the macro tile), grid and workgroup size, average duration
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/vg -o run -- python3 $GRAFT_REPO_ROOT/tools/vendor_gemm_names.py > /tmp/vg.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/vg/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
seen = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" not in n: continue
    key = (n, r["Grid_Size_X"], r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen.setdefault(key, []).append(d)
for (n, g, w), ds in seen.items():
    print("grid %8s wg %4s  n=%d avg %8.1f us  LDS %s  %s" % (g, w, len(ds), sum(ds) / len(ds), "", n[:230]))
PY
