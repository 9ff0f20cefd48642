"""Dev aid: everything one single-stream layer evaluation launches, in order, with the gaps between kernels
(rocprofv3 --kernel-trace database of `bench.py --p 1 --steps 1 --warmup 1 --no-extras --no-cpu` under GPAR_LAYER_PIPELINE=1)."""
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]); cur = db.cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
# the last evaluation: from the last featurize kernel on
idx = max(i for i, r in enumerate(rows) if 'featurize' in r[0])
sel = rows[idx - 3:]
t0 = sel[0][1]; prev_end = t0
other = 0.0
for name, s, e, st, gx in sel:
    short = name.split('(')[0][-40:]
    big = ('gemm_f64' in name) or ('panel' in name)
    if not big or (s - prev_end) > 20000:
        print(f"{short:40s} stream {st} grid {gx:8d} start {1e-3*(s-t0):9.1f} dur {1e-3*(e-s):8.1f} gap_before {1e-3*(s-prev_end):7.1f} us")
    prev_end = max(prev_end, e)
print(f"total {1e-6*(prev_end-t0):.2f} ms")
