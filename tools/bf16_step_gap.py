import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from baler_amd import native, synth
from baler_amd.modules import models
raw = torch.as_tensor(synth.cms_rows(1000000)).cuda()
x = native.normalize(raw, native.minmax(raw))
if len(sys.argv) > 1:      # the bench's order: an fp32 handle trains and encodes on the same rows first
    m32 = models.AE(24, 15, mode="fp32").to("cuda:0")
    h32 = m32.handle(); g32 = torch.zeros_like(m32.flat); m0, v0 = torch.zeros_like(g32), torch.zeros_like(g32)
    for i in range(35):
        h32.fwd_bwd(x, g32); h32.adam_step(m32.flat, g32, m0, v0, i + 1, 1e-3)
    for _ in range(10): z = h32.encode(x)
    torch.cuda.synchronize()
m = models.AE(24, 15, mode="bf16").to("cuda:0")
h = m.handle(); f = m.flat.clone(); h.load_params(f)
g, mm, vv = torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f)
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = [0]
def step():
    h.fwd_bwd(x, g); t[0] += 1; h.adam_step(f, g, mm, vv, t[0], 1e-3)
def wall(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for _ in range(3):
    print("fwd_bwd event %.4f ms | step event %.4f ms | step wall(15) %.4f ms | adam only event %.4f ms" % (
        ev(lambda: h.fwd_bwd(x, g), 20), ev(step, 20), wall(step, 15), ev(lambda: h.adam_step(f, g, mm, vv, 5, 1e-3), 20)))
