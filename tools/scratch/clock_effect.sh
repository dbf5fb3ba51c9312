#!/bin/bash
# Does an attention kernel change the time of the GEMMs around it?  Serial kernel traces of the step with one kernel exchanged at a time.
cd /tmp && export TMPDIR=/tmp
for cfg in "base" "VLM_ATT_FWD2=0" "VLM_ATT_DQ2=0" "VLM_ATT_BWD_FUSED=0"; do
  rm -rf /tmp/ce
  ( [ "$cfg" != base ] && export $cfg; VLM_BENCH_SETUP_STEPS=0 VLM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ce -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/ce.log 2>&1 )
  python3 - "$cfg" <<'PY'
import csv, re, collections, sys
d = collections.defaultdict(float)
for r in csv.DictReader(open("/tmp/ce/run_kernel_stats.csv")):
    k = re.sub(r"[<(].*", "", r["Name"]).replace("void ", "")
    d[k] += float(r["TotalDurationNs"]) / 6e6
att = {k: v for k, v in d.items() if k.startswith("attn")}
print("%-22s gemm_big %.3f  bigT %.3f  ln_bwd %.3f  total %.2f | %s" % (sys.argv[1], d["vlm_gemm_big_kernel"], d["vlm_gemm_bigT_kernel"], d["ln_bwd_kernel"], sum(d.values()),
      "  ".join("%s %.2f" % (k.replace("attn_", "").replace("_kernel", ""), v) for k, v in sorted(att.items()))))
PY
done
