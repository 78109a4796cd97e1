"""D2H DMA rate by stream / buffer pattern.  gpurun -- python tools/probe/d2h_stream_probe.py"""
import os, sys, time
import torch
n = 10_000_000
dev = torch.rand((n, 15), dtype=torch.float64, device="cuda")
flat = dev.view(-1)
gb = flat.numel() * 8 / 1e9
pins = [torch.empty((1 << 26) // 8, dtype=torch.float64).pin_memory() for _ in range(2)]
side = torch.cuda.Stream()
def run(stream, nbuf, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for k, s in enumerate(range(0, flat.numel(), pins[0].numel())):
            e = min(s + pins[0].numel(), flat.numel())
            pins[k % nbuf][:e - s].copy_(flat[s:e], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {gb / dt:.1f} GB/s")
for _ in range(2):
    run(torch.cuda.current_stream(), 1, "default stream, 1 buffer")
    run(side, 1, "side stream, 1 buffer")
    run(side, 2, "side stream, 2 buffers")
# 2-D slices (rows) like download_rows
rows = (1 << 26) // 120
st2 = [p[:rows * 15].view(rows, 15) for p in pins]
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(side):
    for k, s in enumerate(range(0, n, rows)):
        e = min(s + rows, n)
        st2[k & 1][:e - s].copy_(dev[s:e], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"side stream, row slices: {gb / dt:.1f} GB/s")
# with an event record + host event sync per chunk (the pipeline's structure, no drain copy)
torch.cuda.synchronize(); t0 = time.perf_counter()
prev = None
for k, s in enumerate(range(0, n, rows)):
    e = min(s + rows, n)
    with torch.cuda.stream(side):
        st2[k & 1][:e - s].copy_(dev[s:e], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(side)
    if prev is not None: prev.synchronize()
    prev = ev
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"side stream, events + host waits: {gb / dt:.1f} GB/s")
print("HSA_ENABLE_SDMA", os.environ.get("HSA_ENABLE_SDMA"))
