#!/usr/bin/env python3
"""fp32 encode / decode launches at 1M rows for rocprofv3 (no child processes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baler_amd import native, synth
from baler_amd.modules import models
raw = torch.as_tensor(synth.cms_rows(1000000)).cuda()
x = native.normalize(raw, native.minmax(raw))
m = models.AE(24, 15, mode="fp32").to("cuda:0")
h = m.handle()
z = h.encode(x)
for _ in range(40):
    z = h.encode(x)
    y = h.decode(z)
torch.cuda.synchronize()
print("done")
