import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
full_a = np.concatenate([a["text"], a["image"]], 1); full_b = np.concatenate([b["text"], b["image"]], 1)
d = np.abs(full_a - full_b).max(-1)   # [B, 617]
for s in range(d.shape[0]):
    rows = np.nonzero(d[s] > thr)[0]
    if len(rows):
        blocks = sorted(set(int(r) // 32 for r in rows))
        print("sample %2d: %3d rows > %.3g, 32-row blocks %s, max %.3f" % (s, len(rows), thr, blocks, d[s].max()))
