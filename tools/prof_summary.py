"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as markdown."""
import glob
import sqlite3
import sys

src, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(src + "/*.db")[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as fh:
    fh.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; total kernel time {tot / 1e3:.1f} ms (durations in us)\n\n")
    fh.write("| kernel | calls | total_us | avg_us | pct |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:40]:
        fh.write(f"| `{r[0][:120]}` | {r[1]} | {r[2]:.0f} | {r[3]:.1f} | {r[4]:.2f} |\n")
print(open(out).read()[:6000])
