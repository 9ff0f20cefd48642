"""Per-kernel totals of a rocprofv3 --kernel-trace CSV plus the UNION of the intervals of one kernel (launches issued
from different streams overlap, so sum of durations / union = average number in flight)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "gemm_f64_kernel<false, true, 1,"   # both tile forms of the trailing update
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"])
tot = sum(e - s for s, e in iv)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"{key}: {len(iv)} launches, sum of durations {tot*1e-6:.3f} ms (avg {tot/len(iv)*1e-3:.2f} us), union of intervals {busy*1e-6:.3f} ms, "
      f"average concurrency {tot/busy:.3f}")
