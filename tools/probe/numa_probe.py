import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip())
    except OSError as e:
        print(d, e)
print("nodes", sorted(glob.glob("/sys/devices/system/node/node*")))
for nd in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(nd, open(nd + "/cpulist").read().strip())
print("affinity", len(os.sched_getaffinity(0)))
if len(sys.argv) > 1:
    cpus = set()
    for part in open(sys.argv[1]).read().strip().split(","):
        a, _, b = part.partition("-")
        cpus |= set(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
    print("pinned to", len(cpus), "cpus")
import numpy as np, torch
from baler_amd import hostio
n = 10_000_000
dev = torch.rand((n, 15), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
z = None
for rep in range(4):
    t0 = time.perf_counter(); del z; dtf = time.perf_counter() - t0
    t0 = time.perf_counter(); z = hostio.download_rows(dev); dt = time.perf_counter() - t0
    print(f"free previous {dtf * 1e3:.1f} ms; download_rows fresh: {n * 120 / 1e9 / dt:.1f} GB/s ({dt * 1e3:.1f} ms)")
    t0 = time.perf_counter(); z = hostio.download_rows(dev, out=z); dt = time.perf_counter() - t0
    print(f"download_rows same array again: {n * 120 / 1e9 / dt:.1f} GB/s")
