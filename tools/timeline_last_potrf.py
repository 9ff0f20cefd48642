"""Launch-by-launch timeline of the LAST factorisation in a rocprofv3 rocpd database: from its potrf_zero_flags launch to the last
gpar kernel that follows without a foreign kernel in between (development aid).  python tools/timeline_last_potrf.py <dir with .db>"""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "potrf_zero_flags" in r[0])
sel = []
for r in rows[first:]:
    if "gpar::" not in r[0]:
        break
    sel.append(r)
t0 = sel[0][1]
print(f"{len(sel)} launches, span {1e-3 * (max(r[2] for r in sel) - t0):.1f} us")
print(f"{'start_us':>9s} {'dur_us':>8s} {'end_us':>9s}  q  {'workgroups':>14s}  kernel")
streams = {}
for name, s, e, st, gx, gy, gz, wx in sel:
    q = streams.setdefault(st, len(streams))
    wg = f"{gx // max(wx, 1)}x{gy}x{gz}"
    print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f} {1e-3 * (e - t0):9.1f}  {q}  {wg:>14s}  {name.split('(')[0][-52:]}")
