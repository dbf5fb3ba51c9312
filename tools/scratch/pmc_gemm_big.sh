#!/bin/bash
# FETCH_SIZE of the 256x256 GEMM per shape and raster group height (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_big
rm -rf $OUT; mkdir -p $OUT
for gm in 1 2 4 8; do
  for shape in "54296 3072 768" "54296 2304 768" "54296 768 768" "54296 768 3072" "13574 3072 768"; do
    tag=$(echo $shape | tr ' ' x)_g$gm
    VLM_GEMM_BIG_GROUP_M=$gm rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$tag -o run -- $GRAFT_REPO_ROOT/tools/scratch/gemm_bench_full $shape 0 > $OUT/$tag.log 2>&1
  done
done
python3 - <<PY
import csv, glob, os
for d in sorted(glob.glob("$OUT/*/")):
    tot = n = 0
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "big_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                tot += float(r["Counter_Value"]); n += 1
    log = open(d.rstrip("/") + ".log").read().strip().splitlines()
    t = [l for l in log if l.startswith("M=")]
    print(os.path.basename(d.rstrip("/")), "fetch MB/launch %.1f" % (2 * tot / max(n, 1) * 1024 / 1e6), "|", t[-1][:60] if t else "")
PY
