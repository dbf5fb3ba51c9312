"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md 'HBM': both in KiB,
FETCH_SIZE doubled on gfx950).  python tools/pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json"""
import collections, csv, glob, json, re, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main(fd, wd, out, steps=3):
    f, w = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) & set(w)):
        if not (k.startswith("vlm_") or k.startswith("attn_") or k.endswith("_kernel")) or "at::" in k or "rocprim" in k:
            continue
        nf, sf = f[k]
        nw, sw = w[k]
        res[k] = {"launches": nf, "fetch_bytes_per_launch": 2.0 * sf / nf * 1024, "write_bytes_per_launch": sw / nw * 1024,
                  "hbm_bytes_per_launch": (2.0 * sf / nf + sw / nw) * 1024}
    steps = int(steps)
    # per training step: every kernel of the passes except the merge bench's own (it runs after the steps, not in them)
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res.items() if k != "vlm_merge_kernel") / steps
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 (%d steps with "
                       "VLM_BENCH_SETUP_STEPS=0); KiB units, FETCH_SIZE x2 (gfx950 correction); memory-side (fabric) requests, "
                       "Infinity-Cache hits included; total_bytes_per_step = sum over the listed kernels except vlm_merge_kernel "
                       "(the merge bench runs after the steps)" % steps,
               "steps": steps, "total_bytes_per_step": total, "kernels": res}, open(out, "w"), indent=1)
    print("total HBM-side bytes per step: %.1f GB" % (total / 1e9))
    for k, v in res.items():
        print("%-28s n=%5d  fetch %9.2f MB  write %9.2f MB" % (k, v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
