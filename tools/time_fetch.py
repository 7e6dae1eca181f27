"""Time the minibatch fetch alone (scvae_csr_minibatch, uint16 and fp32) on bench-shaped data:
python tools/time_fetch.py [rows];  --all <lib.so> ...: one subprocess per library."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--all":
    for lib in sys.argv[2:]:
        env = dict(os.environ, SCVAE_HIP_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip() or out.stderr[-400:], flush=True)
    sys.exit(0)
import torch
from scvae_amd.minibatch import synthetic_count_matrix
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
matrix, _ = synthetic_count_matrix(16384, 32738, density=0.05, seed=60, device=dev)
perm = torch.randperm(16384, device=dev)
res = []
for u16 in (True, False):
    out = (torch.empty(B, matrix.u16_pitch, dtype=torch.uint16, device=dev) if u16
           else torch.empty(B, 32738, device=dev))
    rc = torch.empty(B, device=dev)
    reqs = [matrix.request(perm[i * B:(i + 1) * B], out, rc) for i in range(16384 // B)]
    for r in reqs:
        r.issue()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        for r in reqs:
            r.issue()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / (10 * len(reqs)) * 1e3)
print("fetch of {} cells: uint16 {:.1f} us, fp32 {:.1f} us".format(B, *res))
