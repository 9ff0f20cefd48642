"""Text summary of a rocprofv3 `--kernel-trace --stats` run: the per-kernel table (calls, total, average, share) and,
from the trace, the union of all kernel intervals (GPU busy time) and the wall-clock span.

    python tools/kernel_table.py <dir with *kernel_stats.csv and *kernel_trace.csv> [title]
"""
import csv
import glob
import os
import sys


def find(directory, suffix):
    hits = sorted(glob.glob(os.path.join(directory, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def short(name, width=64):
    name = name.split("(")[0]
    for prefix in ("void ",):
        if name.startswith(prefix):
            name = name[len(prefix):]
    return name if len(name) <= width else name[-width:]


def union(intervals):
    intervals = sorted(intervals)
    if not intervals:
        return 0
    busy, cs, ce = 0, intervals[0][0], intervals[0][1]
    for s, e in intervals[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


def main():
    directory = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else directory
    stats = find(directory, "kernel_stats.csv")
    trace = find(directory, "kernel_trace.csv")
    print(title)
    if stats:
        rows = list(csv.DictReader(open(stats)))
        print(f"{'kernel':64s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>10s} {'pct':>6s} {'min_us':>9s} {'max_us':>10s}")
        for r in rows[:14]:
            print(f"{short(r['Name']):64s} {int(r['Calls']):7d} {int(r['TotalDurationNs']) * 1e-6:11.3f} "
                  f"{float(r['AverageNs']) * 1e-3:10.2f} {float(r['Percentage']):6.2f} {int(r['MinNs']) * 1e-3:9.2f} "
                  f"{int(r['MaxNs']) * 1e-3:10.2f}")
    if trace:
        rows = list(csv.DictReader(open(trace)))
        iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
        span = max(e for _, e in iv) - min(s for s, _ in iv)
        total = sum(e - s for s, e in iv)
        busy = union(iv)
        print(f"trace: {len(iv)} launches, sum of durations {total * 1e-6:.3f} ms, union (GPU busy) {busy * 1e-6:.3f} ms, "
              f"first-to-last span {span * 1e-6:.3f} ms")
        per = {}
        for r in rows:
            per.setdefault(short(r["Kernel_Name"], 48), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        for name, ivs in sorted(per.items(), key=lambda kv: -union(kv[1]))[:8]:
            tot = sum(e - s for s, e in ivs)
            u = union(ivs)
            print(f"  union {name:48s} {u * 1e-6:10.3f} ms  (concurrency {tot / max(u, 1):.2f})")


if __name__ == "__main__":
    main()
