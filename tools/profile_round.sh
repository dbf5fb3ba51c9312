#!/bin/bash
# Round profile on the GPU box: kernel-trace stats + three separate PMC passes over the SAME bench command, condensed into
# gpurun_out/<tag>/ (copy what should be judged into profiles/).   usage: tools/profile_round.sh r02
TAG=${1:-r06}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VLM_BENCH_SETUP_STEPS=0  # the traces cover exactly warm-up + timed steps (6 steps), as in rounds 1-4
# (--no-merge: the traces cover the training steps only -- round 5's also held the merge bench's set-up, ~770 copyBuffer and ~850
# elementwise launches that are not part of a step; the merge kernel keeps its own line through the PMC passes below)
CMD="python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate --no-secondary --no-merge"
# Two kernel traces of the same command.  (1) VLM_WGRAD_STREAM=0: every launch alone on the chip -- the per-kernel durations
# that bench.py's roofline figure (whose bracketed steps also run without the side stream) has to agree with; the PMC passes
# below run the same way.  (2) the default schedule (weight-gradient GEMMs on a second stream): what the timed steps execute.
export VLM_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- $CMD > $OUT/trace.log 2>&1
unset VLM_WGRAD_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o run -- $CMD > $OUT/trace_overlap.log 2>&1
export VLM_WGRAD_STREAM=0
PMC="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibrate --no-gemm-timer --no-secondary"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o run -- $PMC > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- $PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- $PMC > $OUT/pmc_write.log 2>&1
unset VLM_WGRAD_STREAM
# configs[2] on one GPU (all_moe: grouped expert GEMMs) and the fp64 leg (RegMean + Gram capture), kernel traces only
MOE="python3 $ROOT/bench.py --arch all_moe --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate --no-secondary --no-merge"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_moe -o run -- $MOE > $OUT/trace_moe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f64 -o run -- python3 $ROOT/tools/bench_f64_leg.py > $OUT/trace_f64.log 2>&1
cd $ROOT
python3 tools/prof_summary.py $OUT/trace_moe/run_kernel_stats.csv $OUT/${TAG}_train_all_moe384_b22_kernel_stats.csv "bench.py --arch all_moe --steps 4 --warmup 2 --no-secondary --no-merge under rocprofv3 --kernel-trace --stats; 6 steps"
python3 tools/prof_summary.py $OUT/trace_f64/run_kernel_stats.csv $OUT/${TAG}_f64_leg_kernel_stats.csv "tools/bench_f64_leg.py (RegMean at base size x2, Gram capture D = 768 / 3072 at 54 296 rows x4) under rocprofv3 --kernel-trace --stats"
grep -E "^(regmean|gram capture)" $OUT/trace_f64.log > $OUT/${TAG}_f64_leg_log.txt
grep -E '^\{"metric' $OUT/trace_moe.log | tail -1 | cut -c1-600 > $OUT/${TAG}_bench_line_all_moe_under_trace.txt
python3 tools/prof_summary.py $OUT/trace/run_kernel_stats.csv $OUT/${TAG}_train_ufo384_b22_kernel_stats.csv "VLM_WGRAD_STREAM=0 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate --no-secondary under rocprofv3 --kernel-trace --stats; 6 steps; every launch alone on the chip (the durations the roofline figure agrees with)"
python3 tools/prof_summary.py $OUT/trace_overlap/run_kernel_stats.csv $OUT/${TAG}_train_ufo384_b22_overlap_kernel_stats.csv "bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate --no-secondary under rocprofv3 --kernel-trace --stats; 6 steps; default schedule (weight-gradient GEMMs on a second stream: overlapping kernels stretch each other's durations)"
grep -E '^\{"metric' $OUT/trace_overlap.log | tail -1 | cut -c1-600 > $OUT/${TAG}_bench_line_under_trace_overlap.txt
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json > $OUT/traffic.txt
python3 - <<PY
import collections, csv, glob, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
res = {}
for k, v in acc.items():
    if ("attn" in k or "vlm_gemm" in k or "merge" in k) and v.get("GRBM_GUI_ACTIVE"):
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        res[k] = {"launches": n[(k, "GRBM_GUI_ACTIVE")], "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)}
# attention on ALGORITHMIC flops next to the busy counter (the counter also counts the bias-selection, statistics and row-sum MFMAs):
# per step forward 4 n_q n_k 64 per (sample, head, layer) summed over the passes of configs[1] = 1.427 TFLOP, backward 2 x that;
# kernel times from the serial kernel trace of the same command (6 steps)
alg = {}
try:
    t = {}
    for r in csv.DictReader(open("$OUT/trace/run_kernel_stats.csv")):
        nm = re.sub(r"[<(].*", "", r["Name"]).replace("void ", "")
        t[nm] = t.get(nm, 0.0) + float(r["TotalDurationNs"]) / 6e6  # ms per step
    fwd = sum(v for k, v in t.items() if k.startswith("attn_fwd"))
    bwd = sum(v for k, v in t.items() if k.startswith("attn_bwd_") or k == "attn_dbias_fold_kernel")
    alg = {"fwd_tflop_per_step": 1.427, "fwd_ms_per_step": fwd, "fwd_tflops": 1.427 / fwd * 1e3 if fwd else None,
           "fwd_frac_of_2500": 1.427 / fwd * 1e3 / 2500 if fwd else None,
           "bwd_tflop_per_step": 2.854, "bwd_ms_per_step": bwd, "bwd_tflops": 2.854 / bwd * 1e3 if bwd else None,
           "bwd_frac_of_2500": 2.854 / bwd * 1e3 / 2500 if bwd else None,
           "bwd_kernels_ms_per_step": {k: v for k, v in t.items() if k.startswith("attn_")}}
except Exception as e:
    alg = {"error": repr(e)}
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py --steps 2 --warmup 1: busy cycles summed over the 1024 SIMDs / (1024 x per-XCD active cycles), time-weighted over all launches of a kernel", "kernels": res, "attention_on_algorithmic_flops": alg}, open("$OUT/${TAG}_pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
grep -E '^\{"metric' $OUT/trace.log | tail -1 | cut -c1-600 > $OUT/${TAG}_bench_line_under_trace.txt
head -25 $OUT/${TAG}_train_ufo384_b22_kernel_stats.csv
# launches of one step by origin (this library / torch-native / copies), and the attention kernels standalone: the hand-placed
# streams against the round-3 kernels, with the in-kernel clock (s_memtime / s_memrealtime stamps of the -DVLM_DIAG harness)
python3 tools/count_launches.py > $OUT/${TAG}_launches_per_step.txt 2>&1
{
  for args in "88 0 1 0" "22 0 1 0"; do for v in 0 1; do
    echo -n "forward  hand-placed=$v  B mode bias = $args: "; VLM_ATT_FWD2=$v bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_fwd" | awk '{print $1, $(NF-1), $NF}'
  done; done
  for args in "88 0 1 1 1" "22 0 1 1 1"; do for v in 0 1; do
    echo -n "backward hand-placed dQ=$v  $args: "; VLM_ATT_DQ2=$v bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_bwd" | awk '{printf "%s %s us; ", $1, $(NF-1)} END {print ""}'
  done; done
  for v in 0 1; do echo "stamps forward hand-placed=$v:"; VLM_ATT_FWD2=$v tools/scratch/attn_bench_diag 88 0 1 0 2>&1 | grep -E "wave 0 clock"; done
  for v in 0 1; do echo "stamps backward (dQ kernel) hand-placed=$v:"; VLM_ATT_DQ2=$v tools/scratch/attn_bench_diag 88 0 1 1 1 2>&1 | grep -E "wave 0 clock"; done
} > $OUT/${TAG}_attention_harness.txt 2>&1
cat $OUT/${TAG}_attention_harness.txt
