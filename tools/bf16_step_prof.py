import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from baler_amd import native, synth
from baler_amd.modules import models
raw = torch.as_tensor(synth.cms_rows(1_000_000)).cuda()
xd = native.normalize(raw, native.minmax(raw))
m = models.AE(24, 15, mode="bf16").to("cuda:0"); h = m.handle()
g = torch.zeros_like(m.flat); mm, vv = torch.zeros_like(m.flat), torch.zeros_like(m.flat)
for i in range(30):
    h.fwd_bwd(xd, g); h.adam_step(m.flat, g, mm, vv, i + 1, 1e-3)
torch.cuda.synchronize()
