"""Dev aid: where the sporadic 50-75 ms evaluations of C4 come from (allocator traffic per evaluation)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd.engine import HipEngine, set_engine
from tools.run_config import build

eng = HipEngine(seed=1)
set_engine(eng)
cfg, reg, x, y = build(sys.argv[1] if len(sys.argv) > 1 else "C4", eng)
float(reg.logpdf(x, y))
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    gc.disable()
if len(sys.argv) > 2 and sys.argv[2] == "t1":
    torch.set_num_threads(1)
keys = ["num_device_alloc", "num_device_free", "num_alloc_retries"]
for i in range(24):
    s0 = torch.cuda.memory_stats()
    g0 = [g["collections"] for g in gc.get_stats()]
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = time.process_time()
    v = float(reg.logpdf(x, y))
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    s1 = torch.cuda.memory_stats()
    g1 = [g["collections"] for g in gc.get_stats()]
    print(f"eval {i:2d}: {dt:7.2f} ms  cpu {1e3 * (time.process_time() - c0):7.1f} ms  " + "  ".join(f"{k} +{s1.get(k, 0) - s0.get(k, 0)}" for k in keys) +
          f"  reserved {s1['reserved_bytes.all.current'] / 2**30:.2f} GiB  allocated peak {s1['allocated_bytes.all.peak'] / 2**30:.2f} GiB  gc {[b - a for a, b in zip(g0, g1)]}")

# which threads burnt the CPU time (the container is limited to 16 CPUs per 100 ms period: throttling stalls the host)
import glob
tot = []
for p in glob.glob("/proc/self/task/*/stat"):
    try:
        f = open(p).read()
        comm = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        tot.append((int(rest[11]) + int(rest[12]), comm))
    except Exception:
        pass
tot.sort(reverse=True)
print("threads:", len(tot), "top by CPU ticks (10 ms):", tot[:12], "sum", sum(t for t, _ in tot))
print(open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
