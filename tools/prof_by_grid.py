"""Per (kernel, grid, workgroup) breakdown of a rocprofv3 --kernel-trace database (rocpd sqlite),
optionally filtered by a substring of the kernel name; also per-stream busy time."""
import glob
import sqlite3
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(glob.glob(src + "/*.db")[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
if not cols:
    print("views:", [r[0] for r in cur.execute("select name from sqlite_master").fetchall()])
    sys.exit(1)
print("columns:", cols)
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
gy, gz = gx.replace("_x", "_y"), gx.replace("_x", "_z")
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
q = f"select name, {gx}*{gy}*{gz}, {wx}, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where name like ? " \
    f"group by name, {gx}*{gy}*{gz}, {wx} order by 5 desc limit {limit}"
for r in cur.execute(q, (f"%{pat}%",)).fetchall():
    print(f"{r[0][:90]:90s} grid={r[1]:>8} wg={r[2]:>5} n={r[3]:>6} total={r[4]:>10.0f}us avg={r[5]:>8.1f}us")
scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
if scol:
    for r in cur.execute(f"select {scol}, count(*), sum(end-start)/1e6 from kernels group by {scol}").fetchall():
        print(f"stream {r[0]}: {r[1]} launches, {r[2]:.1f} ms busy")
