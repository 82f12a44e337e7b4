"""Issue-ordered list of ONE training step's main-stream launches from a rocprofv3 --kernel-trace database of bench.py's
training leg (the last complete step): start offset, duration, gap to the previous kernel, grid, name -- run-length
compressed.  Used to decide which launches to fold."""
import glob
import re
import sqlite3
import sys

cur = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = f"select stream_id, start, end, name{', ' + gx if gx else ''} from kernels order by start"
rows = cur.execute(q).fetchall()
by = {}
for r in rows:
    by.setdefault(r[0], []).append(r[1:])
main = max(by, key=lambda k: len(by[k]))
if len(sys.argv) > 3:  # another stream, by rank of its launch count (1 = the busiest after the main stream)
    main_ends = [r[1] for r in by[main] if "adamw_kernel" in r[2]]
    other = sorted((k for k in by if k != main), key=lambda k: -len(by[k]))[int(sys.argv[3]) - 1]
    # the steps' boundaries come from the main stream's optimiser launches
    m = sorted(by[other] + [r for r in by[main] if "adamw_kernel" in r[2] or "spin_kernel" in r[2]], key=lambda r: r[0])
    print(f"(stream {other}, between the main stream's optimiser launches)")
else:
    m = by[main]
ends = [i for i, r in enumerate(m) if "adamw_kernel" in r[2]]
want_spin = len(sys.argv) > 2 and sys.argv[2] == "spin"  # the instrumented step: the host enqueues everything behind a spin kernel
seg = None
for k in range(len(ends) - 2, 0, -1):
    cand = m[ends[k] + 1: ends[k + 1] + 1]
    if any("spin_kernel" in r[2] for r in cand) == want_spin:
        seg = cand
        break
assert seg is not None, "no such step in the trace"
short = lambda n: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:90]
print(f"main stream {main}: {len(seg)} launches in the step, span {(seg[-1][1] - seg[0][0]) / 1e6:.2f} ms")
t0 = seg[0][0]
prev_end = t0
for i, r in enumerate(seg):
    a, b, n = r[0], r[1], r[2]
    g = r[3] if len(r) > 3 else 0
    print(f"{i:4d} {(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:7.1f}  gap {(a - prev_end) / 1e3:6.1f}  grid {g:>8}  {short(n)}")
    prev_end = b
