"""Kernel timeline of the LAST evaluation in a rocprofv3 kernel trace (csv), one line per kernel with its stream/queue:
   python tools/timeline_last_eval.py <dir> [gap_ms]      (an evaluation = kernels after the last idle gap > gap_ms)"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
start = 0
end_prev = rows[0][1]
for i, r in enumerate(rows):
    if r[0] - end_prev > gap:
        start = i
    end_prev = max(end_prev, r[1])
sel = rows[start:]
t0 = sel[0][0]
for s, e, name, q, st, g in sel:
    short = name.split("(")[0].replace("gpar::", "").replace("void ", "")[-40:]
    print(f"{1e-3 * (s - t0):9.1f} +{1e-3 * (e - s):8.1f} us  q{q:>3} s{st:>3}  grid {g:7d}  {short}")
print(f"total {1e-6 * (max(r[1] for r in sel) - t0):.3f} ms, {len(sel)} kernels")
