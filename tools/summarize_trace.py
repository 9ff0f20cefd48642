"""Summarise a rocprofv3 rocpd database: per-kernel totals and the busy/idle structure of the last potrf."""
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
# last potrf: from the last copyBuffer-like kernel before the final diag chain -> take kernels after the last 'copy' kernel
idx = max(i for i, r in enumerate(rows) if 'copy' in r[0].lower() or 'elementwise' in r[0].lower())
sel = rows[idx + 1:]
t0, t1 = sel[0][1], max(r[2] for r in sel)
print(f"kernels in last factorisation: {len(sel)}, wall {1e-6*(t1-t0):.2f} ms")
agg = {}
for name, s, e, st, gx in sel:
    key = (name.split('(')[0][-40:], st)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (e - s) * 1e-6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[0]:42s} stream {k[1]}: n={v[0]:5d} total {v[1]:8.3f} ms avg {1e3*v[1]/v[0]:8.1f} us")
# union busy time
ivs = sorted((s, e) for _, s, e, _, _ in sel)
busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
for s, e in ivs[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union of kernel intervals) {busy*1e-6:.2f} ms, idle gaps {1e-6*(t1-t0-busy):.2f} ms")
