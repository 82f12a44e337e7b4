"""Main-stream busy time of the forward and the backward of a training step from a rocprofv3 --kernel-trace database:
the step is cut at its first backward-only kernel; top kernels of each phase."""
import glob, re, sqlite3, sys
cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
rows = cur.execute("select stream_id, start, end, name from kernels order by start").fetchall()
by = {}
for s, a, b, n in rows:
    by.setdefault(s, []).append((a, b, n))
main = max(by, key=lambda k: len(by[k]))
m = by[main]
ends = [i for i, (_, _, n) in enumerate(m) if "adamw_kernel" in n]
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:100]
BWD = ("_bwd", "wgrad", "grad_sumsq")
agg = {"fwd": {}, "bwd": {}}
tot = {"fwd": 0, "bwd": 0}
nstep = 0
for k in range(max(1, len(ends) - 7), len(ends) - 1):
    seg = m[ends[k] + 1: ends[k + 1] + 1]
    if any("spin_kernel" in n for _, _, n in seg):
        continue
    nstep += 1
    cut = next((i for i, (_, _, n) in enumerate(seg) if any(t in n for t in BWD)), len(seg))
    for i, (a, b, n) in enumerate(seg):
        ph = "fwd" if i < cut else "bwd"
        tot[ph] += b - a
        c = agg[ph].setdefault(short(n), [0, 0])
        c[0] += 1
        c[1] += b - a
for ph in ("fwd", "bwd"):
    print(f"== {ph}: main-stream busy {tot[ph] / nstep / 1e6:.2f} ms per step, {sum(v[0] for v in agg[ph].values()) / nstep:.0f} launches")
    for n, (c, t) in sorted(agg[ph].items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {c / nstep:6.1f} x {t / c / 1e3:7.1f} us = {t / nstep / 1e3:7.1f} us  {n}")
