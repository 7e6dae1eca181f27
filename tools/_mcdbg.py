import ctypes, numpy as np, torch
from scvae_amd.engine import Engine
from scvae_amd import _lib
dev=torch.device("cuda:0")
F,L,H,B=32738,25,(100,100),100
eng=Engine(F,L,H,"negative binomial",batch_norm=True,device=dev,seed=1)
rng=np.random.default_rng(0)
x=torch.from_numpy((rng.poisson(2.,(B,F))*(rng.random((B,F))<0.05)).astype(np.float32)).to(dev)
eps=torch.randn(1,B,L,device=dev)
for _ in range(20):
    eng.step(x,x,eps=eps,training=True,x_counts=True)
torch.cuda.synchronize()
lib=_lib.load()
out=(ctypes.c_longlong*64)()
lib.scvae_mc_debug(out)
a=np.array(out[:10])
print("enc1 fwd:", np.diff(a))
