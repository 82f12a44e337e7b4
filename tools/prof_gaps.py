"""The largest idle gaps of the main stream inside a training step (rocprofv3 --kernel-trace database of bench.py's training
leg): gap length, the kernel before and the kernel after, averaged over the timed steps by position in the step."""
import collections
import glob
import re
import sqlite3
import sys

cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
rows = cur.execute("select stream_id, start, end, name from kernels order by start").fetchall()
by = {}
for s, a, b, n in rows:
    by.setdefault(s, []).append((a, b, n))
main = max(by, key=lambda k: len(by[k]))
m = by[main]
ends = [i for i, (_, _, n) in enumerate(m) if "adamw_kernel" in n]
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:60]
agg = collections.defaultdict(lambda: [0, 0.0])
nstep = 0
tot_idle = 0.0
for k in range(max(1, len(ends) - 8), len(ends) - 1):
    seg = m[ends[k] + 1: ends[k + 1] + 1]
    if not seg or any("spin_kernel" in n for _, _, n in seg):
        continue
    nstep += 1
    for i in range(len(seg) - 1):
        g = seg[i + 1][0] - seg[i][1]
        if g > 0:
            tot_idle += g
            key = (short(seg[i][2]), short(seg[i + 1][2]))
            agg[key][0] += 1
            agg[key][1] += g
print(f"{nstep} steps, idle {tot_idle / nstep / 1e6:.2f} ms per step on the main stream")
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {t / nstep / 1e3:8.1f} us/step  {c / nstep:6.1f} x {t / c / 1e3:7.1f} us   {a}  ->  {b}")
