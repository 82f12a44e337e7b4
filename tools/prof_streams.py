"""Multi-stream view of a training step (rocprofv3 --kernel-trace database of bench.py's training leg): per step (AdamW to AdamW on
the main stream) the wall time, the time at least one kernel runs (union over streams), the time nothing runs (host-bound or
dependency bubbles), the time >= 2 kernels overlap, and per stream its busy time, launch count and top kernels."""
import collections
import glob
import re
import sqlite3
import sys

cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
rows = cur.execute("select stream_id, start, end, name from kernels order by start").fetchall()
by = collections.defaultdict(list)
for s, a, b, n in rows:
    by[s].append((a, b, n))
main = max(by, key=lambda k: len(by[k]))
ends = [b for a, b, n in by[main] if "adamw_kernel" in n]
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:48]
steps = []
for k in range(max(1, len(ends) - 8), len(ends) - 1):
    lo, hi = ends[k], ends[k + 1]
    seg = [(s, a, b, n) for s, a, b, n in rows if a >= lo and b <= hi]
    if any("spin_kernel" in n for _, _, _, n in seg):
        continue
    steps.append((lo, hi, seg))
print(f"{len(steps)} steps; main stream {main}; streams in use: {len(by)}")
acc = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
cnt_main = collections.Counter()
tw = tu = t2 = 0.0
for lo, hi, seg in steps:
    ev = []
    for s, a, b, n in seg:
        ev.append((a, 1))
        ev.append((b, -1))
        acc[s][0] += b - a
        acc[s][1] += 1
        acc[s][2][short(n)] += b - a
        if s == main:
            cnt_main[short(n)] += 1
    ev.sort()
    depth, last = 0, lo
    for t, d in ev:
        if depth >= 1:
            tu += t - last
        if depth >= 2:
            t2 += t - last
        depth += d
        last = t
    tw += hi - lo
n = len(steps)
print(f"per step: wall {tw / n / 1e6:.2f} ms, some kernel running {tu / n / 1e6:.2f} ms, nothing running {(tw - tu) / n / 1e6:.2f} ms, "
      f">= 2 kernels at once {t2 / n / 1e6:.2f} ms; sum of kernel durations {sum(v[0] for v in acc.values()) / n / 1e6:.2f} ms")
for s, (busy, cnt, top) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  stream {s}: busy {busy / n / 1e6:6.2f} ms, {cnt / n:6.1f} launches per step; " +
          ", ".join(f"{k} {v / n / 1e3:.0f} us" for k, v in top.most_common(14 if s == main else 5)))
print("main stream, launches per step by kernel (count, total us):")
for k, c in cnt_main.most_common(45):
    print(f"  {c / n:6.1f} x  {acc[main][2][k] / n / 1e3:8.1f} us  {k}")
