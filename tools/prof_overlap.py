"""Timeline analysis of a rocprofv3 --kernel-trace database: busy time per stream, how much of the side
stream's work overlaps the main stream's, and the largest idle gaps of the main stream (with the kernels
before / after each gap)."""
import glob
import sqlite3
import sys

cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
rows = cur.execute("select stream_id, start, end, name from kernels order by start").fetchall()
by = {}
for s, a, b, n in rows:
    by.setdefault(s, []).append((a, b, n))
main = max(by, key=lambda k: len(by[k]))
others = sorted((k for k in by if k != main), key=lambda k: -sum(b - a for a, b, _ in by[k]))
print("main stream", main, "launches", len(by[main]), "busy %.1f ms" % (sum(b - a for a, b, _ in by[main]) / 1e6))
spin = [(a, b) for a, b, n in by[main] if "spin_kernel" in n]
t_hi = spin[0][0] if spin else by[main][-1][1]  # analyse up to the instrumented step
m = [(a, b, n) for a, b, n in by[main] if b <= t_hi]
for k in others[:3]:
    o = [(a, b) for a, b, _ in by[k] if b <= t_hi]
    if not o:
        continue
    # overlap of o with main intervals (both sorted)
    i = 0
    ov = 0
    for a, b in o:
        while i < len(m) and m[i][1] <= a:
            i += 1
        j = i
        while j < len(m) and m[j][0] < b:
            ov += max(0, min(b, m[j][1]) - max(a, m[j][0]))
            j += 1
    tot = sum(b - a for a, b in o)
    print(f"stream {k}: {len(o)} launches, busy {tot / 1e6:.1f} ms, of which {ov / 1e6:.1f} ms while the main stream also runs a kernel")
gaps = []
for (a0, b0, n0), (a1, b1, n1) in zip(m[:-1], m[1:]):
    if a1 - b0 > 100e3:
        gaps.append((a1 - b0, n0[:60], n1[:60]))
print("main-stream idle gaps > 100 us:", len(gaps), "total %.1f ms" % (sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0] / 1e3:8.1f} us after {g[1]}  before {g[2]}")

# pairwise overlap between the busiest streams (e.g. BigVGAN's three AMP-block streams)
top = sorted(by, key=lambda k: -sum(b - a for a, b, _ in by[k]))[:5]


def overlap(xs, ys):
    i = 0
    ov = 0
    for a, b, _ in xs:
        while i < len(ys) and ys[i][1] <= a:
            i += 1
        j = i
        while j < len(ys) and ys[j][0] < b:
            ov += max(0, min(b, ys[j][1]) - max(a, ys[j][0]))
            j += 1
    return ov


print("pairwise overlap (ms) between the busiest streams:", top)
for i, p in enumerate(top):
    print(f"  stream {p}: busy {sum(b - a for a, b, _ in by[p]) / 1e6:8.1f} |", " ".join(f"{overlap(by[p], by[q]) / 1e6:8.1f}" for q in top))
