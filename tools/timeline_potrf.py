"""Per-panel timeline of the last factorisation in a rocprofv3 rocpd database (development aid)."""
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
idx = max(i for i, r in enumerate(rows) if 'copy' in r[0].lower() or 'elementwise' in r[0].lower())
sel = rows[idx + 1:]
t0 = sel[0][1]
def short(n):
    if 'potrf_panel' in n: return 'PANEL'
    if 'true, 1>' in n: return 'TRAIL'
    if 'gemm' in n: return 'LA   '
    if 'fill' in n.lower(): return None
    return n.split('(')[0][-20:]
for name, s, e, st, gx in sel:
    k = short(name)
    if k is None: continue
    print(f"{k} stream {st} grid {gx:6d}  start {1e-3*(s-t0):9.1f} us  end {1e-3*(e-t0):9.1f} us  dur {1e-3*(e-s):8.1f} us")
