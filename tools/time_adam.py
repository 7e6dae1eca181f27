"""Time scvae_adam_clip_step on a parameter buffer of the headline model's size.
    python tools/time_adam.py [n] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scvae_amd import _lib
from scvae_amd.engine import current_stream_handle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6651951
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
theta = torch.randn(n, device=dev, generator=g)
grad = torch.randn(n, device=dev, generator=g) * 0.5
m = torch.zeros(n, device=dev)
v = torch.zeros(n, device=dev)


def call():
    lib.scvae_adam_clip_step(theta.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                             1.0, 1e-3, 0.9, 0.999, 1e-8, current_stream_handle(dev))


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(launches):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / launches * 1e3
print("adam_clip_step, {} parameters: {:.1f} us = {:.2f} TB/s; checksum {:.9e} {:.9e}".format(
    n, us, 28.0 * n / us / 1e6, theta.double().sum().item(), v.double().sum().item()))
