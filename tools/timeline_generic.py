"""Kernels after the last elementwise/copy kernel in a rocprofv3 rocpd database, in order (development aid)."""
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
idx = max(i for i, r in enumerate(rows) if 'elementwise' in r[0].lower())
sel = rows[idx + 1:]
t0 = sel[0][1]
for name, s, e, st, gx in sel:
    print(f"{name.split('(')[0][-44:]:44s} grid {gx:8d} start {1e-3*(s-t0):9.1f} dur {1e-3*(e-s):8.1f} us")
print(f"total {1e-6*(sel[-1][2]-t0):.2f} ms")
