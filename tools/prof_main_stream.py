"""Per-stream kernel breakdown of one training step from a rocprofv3 --kernel-trace database (mean over the timed steps):
for every stream, launches and busy time per kernel name."""
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
ends = [b for a, b, n in by[main] if "adamw_kernel" in n]
lo, hi = ends[max(0, len(ends) - 9)], ends[-1]
nstep = min(8, len(ends) - 1)
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:110]
for s in sorted(by, key=lambda k: -len(by[k])):
    agg = {}
    for a, b, n in by[s]:
        if lo < b <= hi:
            c = agg.setdefault(short(n), [0, 0])
            c[0] += 1
            c[1] += b - a
    if not agg:
        continue
    tot = sum(v[1] for v in agg.values())
    print(f"== stream {s}{' (main)' if s == main else ''}: {sum(v[0] for v in agg.values()) / nstep:.0f} launches, busy {tot / nstep / 1e6:.2f} ms per step")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: (45 if s == main else 14)]:
        print(f"   {c / nstep:7.1f} x {t / c / 1e3:8.1f} us = {t / nstep / 1e3:8.1f} us  {n}")
