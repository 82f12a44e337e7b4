"""Which stream does the end of the training step wait for?  From a rocprofv3 --kernel-trace database of bench.py's training leg:
for the instrumented step (everything enqueued behind a spin kernel: the device's own timeline, no host limits) every stream's
busy intervals, merged, between the end of the spin kernel and the optimiser -- start / end relative to the spin kernel's end,
number of launches, busy time, first and last kernel -- plus the waits (> 50 us) of the main stream with what ran elsewhere."""
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
spins = [(a, b) for a, b, n in by[main] if "spin_kernel" in n]
assert spins, "no instrumented step in the trace"
t0 = spins[-1][1]
t1 = next(b for a, b, n in by[main] if a > t0 and "adamw_kernel" in n)
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:60]
print(f"instrumented step: {(t1 - t0) / 1e6:.2f} ms from the end of the spin kernel to the end of the optimiser (traced)")
for s in sorted(by, key=lambda k: -len(by[k])):
    ks = [(a, b, n) for a, b, n in by[s] if a >= t0 and b <= t1]
    if not ks:
        continue
    iv = []
    for a, b, n in ks:
        if iv and a - iv[-1][1] < 30e3:
            iv[-1][1] = max(iv[-1][1], b)
            iv[-1][2] += 1
            iv[-1][3] += b - a
            iv[-1][5] = n
        else:
            iv.append([a, b, 1, b - a, n, n])
    print(f"stream {s}{' (main)' if s == main else ''}: {len(ks)} launches, busy {sum(b - a for a, b, _ in ks) / 1e6:.2f} ms")
    for a, b, c, busy, n0, n1 in iv:
        if b - a > 100e3 or s != main:
            print(f"    {(a - t0) / 1e3:9.1f} .. {(b - t0) / 1e3:9.1f} us  {c:4d} launches, busy {busy / 1e3:8.1f} us   {short(n0)}  ...  {short(n1)}")
m = [(a, b, n) for a, b, n in by[main] if a >= t0 and b <= t1]
print("waits of the main stream (> 50 us):")
for (a0, b0, n0), (a1, b1, n1) in zip(m, m[1:]):
    if a1 - b0 > 50e3:
        other = {s: sum(min(b, a1) - max(a, b0) for a, b, _ in by[s] if b > b0 and a < a1) for s in by if s != main}
        print(f"    at {(b0 - t0) / 1e3:9.1f} us: {(a1 - b0) / 1e3:7.1f} us before {short(n1)}; busy elsewhere: "
              + ", ".join(f"stream {s} {v / 1e3:.0f} us" for s, v in other.items() if v > 0))
