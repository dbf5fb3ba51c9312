"""Idle-gap report for one training step out of a rocprofv3 kernel trace (steps are delimited by adamw launches)."""
import collections
import csv
import re
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    bounds = []
    for a, b in zip(ad, ad[1:] + [None]):
        if b is None or int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 20e6:
            bounds.append(a)
    seg = rows[bounds[-2] + 1:bounds[-1] + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("step span %.2f ms, busy %.2f ms, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, len(seg)))
    gaps = []
    for a, b in zip(seg, seg[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        gaps.append((g, (int(a["Start_Timestamp"]) - t0) / 1e6, re.sub(r"\(.*", "", a["Kernel_Name"])[:50],
                     re.sub(r"\(.*", "", b["Kernel_Name"])[:50]))
    print("total gaps %.2f ms" % (sum(g for g, *_ in gaps if g > 0) / 1e6))
    for g in sorted(gaps, reverse=True)[:12]:
        print("  %.0f us at %.1f ms: %s -> %s" % (g[0] / 1e3, g[1], g[2], g[3]))
    h = collections.Counter()
    for g in gaps:
        if g[0] > 0:
            h[min(int(g[0] / 1e3) // 5 * 5, 100)] += g[0]
    print("gap histogram (us bucket -> ms):", sorted((k, round(v / 1e6, 2)) for k, v in h.items()))
    agg = collections.defaultdict(float)
    for r in seg:
        agg[re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "")[:40]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-40s %.2f ms" % (k, v))


if __name__ == "__main__":
    main(sys.argv[1])
