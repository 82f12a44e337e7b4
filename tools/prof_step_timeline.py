"""Per-step view of a rocprofv3 --kernel-trace database of the training leg of bench.py: for the timed steps, the
main stream's busy time, its idle time (sum of the gaps between consecutive kernels) and the histogram of those gaps."""
import glob
import sqlite3
import sys

cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
rows = cur.execute("select stream_id, start, end, name from kernels order by start").fetchall()
by = {}
for s, a, b, n in rows:
    by.setdefault(s, []).append((a, b, n))
main = max(by, key=lambda k: len(by[k]))
m = by[main]
# steps end with the fused AdamW kernel
ends = [i for i, (_, _, n) in enumerate(m) if "adamw_kernel" in n]
print("main stream", main, "launches", len(m), "; optimizer launches (steps)", len(ends))
for k in range(max(1, len(ends) - 8), len(ends) - 1):
    seg = m[ends[k] + 1 : ends[k + 1] + 1]
    if not seg or any("spin_kernel" in n for _, _, n in seg):
        continue
    busy = sum(b - a for a, b, _ in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    h = [sum(1 for g in pos if lo <= g < hi) for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 50e3), (50e3, 1e12))]
    print(f"step: {len(seg)} launches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {sum(pos) / 1e6:.2f} ms; "
          f"gaps <2us {h[0]}, 2-5us {h[1]}, 5-10us {h[2]}, 10-50us {h[3]}, >50us {h[4]}")
    side = 0
    for s in by:
        if s != main:
            side += sum(min(b, seg[-1][1]) - max(a, seg[0][0]) for a, b, _ in by[s] if b > seg[0][0] and a < seg[-1][1])
    print(f"      other streams busy within the step: {side / 1e6:.2f} ms")
