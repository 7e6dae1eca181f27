"""Phase probe of the evaluation chain (tile_chain_fwd_kernel, a -DSCVAE_TC_PROBE build of
tilechain.hip loaded through SCVAE_HIP_LIBRARY): s_memtime ticks of workgroup 1 per
(stage, phase), mean per launch.
    scvae_amd/csrc/build_tcprobe.sh
    SCVAE_HIP_LIBRARY=$PWD/scvae_amd/csrc/libscvae_hip_tcprobe.so python tools/tc_probe_eval.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", "50"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_eval.py"),
               run_name="__main__")
from scvae_amd import _lib
lib = _lib.load()
buf = (ctypes.c_ulonglong * 128)()
assert lib.scvae_debug_tc_probe(buf) == 0
names = ["run", "stores done", "issue", "wait", "partials"]
n = buf[63]
print("forward launches", n)
tot = 0
for st in range(12):
    vals = [buf[5 * st + k] / n for k in range(5)]
    if any(vals):
        tot += sum(vals)
        print("  stage {:2d}: ".format(st) + "  ".join(
            "{} {:7.0f}".format(nm, v) for nm, v in zip(names, vals)))
print("  total ticks", round(tot))
inner = ["merge", "normalise + tile -> LDS", "weights -> LDS", "product", "acc -> LDS",
         "bias + store", "tile statistics"]
print("  inside run, all tile stages: " + "  ".join(
    "{} {:.0f}".format(nm, buf[40 + k] / n) for k, nm in enumerate(inner)))
