"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short, committed summary (profiles/)."""
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "<...>", name)
    name = name.replace("void ", "").replace("at::native::", "")
    return name[:80]


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
        a[2] = max(a[2], float(r["MaxNs"]))
    total = sum(a[1] for a in agg.values())
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % note)
        f.write("# source: %s ; total kernel time %.3f ms\n" % (src, total / 1e6))
        f.write("kernel,calls,total_ms,avg_us,pct,max_us\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.3f,%.2f,%.2f,%.1f\n" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100 * a[1] / total, a[2] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
