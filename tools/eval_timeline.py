"""Kernel-by-kernel timeline of the LAST evaluation in a rocprofv3 kernel trace (the launches after the last idle gap of more
than `gap_us`): start offset, duration, queue, grid, name.   python tools/eval_timeline.py <trace dir> [gap_us]"""
import csv
import sys

from kernel_table import find, short

rows = list(csv.DictReader(open(find(sys.argv[1], "kernel_trace.csv"))))
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 500e3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = 0
end_so_far = 0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if i and s - end_so_far > gap:
        start = i
    end_so_far = max(end_so_far, e)
sel = rows[start:]
t0 = int(sel[0]["Start_Timestamp"])
queues = {}
print(f"{len(sel)} launches, span {(max(int(r['End_Timestamp']) for r in sel) - t0) * 1e-3:.1f} us")
print(f"{'start_us':>9s} {'dur_us':>8s} {'end_us':>9s} {'q':>2s} {'grid':>16s}  kernel")
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    grid = "x".join(str(int(r[k]) // max(int(r[w]), 1)) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")) if k in r)
    print(f"{s * 1e-3:9.1f} {(e - s) * 1e-3:8.1f} {e * 1e-3:9.1f} {q:2d} {grid:>16s}  {short(r['Kernel_Name'], 60)}")
