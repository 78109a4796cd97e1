import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).float().cuda()
h.load_params(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)
x = torch.rand((bs * 200, 24), dtype=torch.float64, device="cuda")
def steps(n0):
    for i in range(200): h.train_step(x[i * bs:(i + 1) * bs], p, m, v, n0 + i + 1, 1e-3)
steps(0); torch.cuda.synchronize()
t0 = time.perf_counter(); steps(200); torch.cuda.synchronize(); print(f"eager: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us/step")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    steps(400); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        steps(600)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); print(f"graph of 200 steps: {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us/step")
