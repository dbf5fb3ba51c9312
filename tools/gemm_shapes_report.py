"""GEMM launches of a rocprofv3 kernel trace aggregated by (template flags, grid): python tools/gemm_shapes_report.py TRACE.csv STEPS"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "vlm_gemm" not in n:
        continue
    t = re.search(r"<(.*)>", n).group(1).replace("true", "T").replace("false", "F").replace(" ", "")
    k = (t, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("flags = TA,TB,OUT_F32,DMA_A,DMA_B,SPLITK ; total %.2f ms/step" % (tot / steps / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-14s wgs %6d  n/step %5.1f  avg %7.1f us  %6.2f ms/step" % (k[0], k[1], v[0] / steps, v[1] / v[0], v[1] / steps / 1e3))
