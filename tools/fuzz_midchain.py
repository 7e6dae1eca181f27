"""Fuzz: small-minibatch steps on the two cooperative launches of midchain.hip against the
chain of launches over random shapes (rows x samples <= 128, widths <= 128, 1-3 layers, all
four count likelihoods): scalars, per-cell log-likelihood, gradients, moving statistics and an
evaluation step within 5e-5 of the tensor's magnitude.
Usage (GPU box): PYTHONPATH=. python tools/fuzz_midchain.py <seed> <configs>"""
import numpy as np, torch, sys
from scvae_amd.engine import Engine
dev=torch.device("cuda:0")
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
bad=0; n=0
LK=["negative binomial","poisson","zero-inflated negative binomial","zero-inflated poisson"]
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 150):
    n_iw=int(rng.integers(1,4)); n_mc=int(rng.integers(1,3))
    S=n_iw*n_mc
    B=int(rng.integers(1,128//S+1))
    nl=int(rng.integers(1,4))
    H=tuple(int(rng.integers(1,129)) for _ in range(nl))
    L=int(rng.integers(1,129))
    F=int(rng.integers(5,400))
    lk=LK[int(rng.integers(0,4))]
    x=torch.from_numpy((rng.poisson(2.,(B,F))*(rng.random((B,F))<0.3)).astype(np.float32)).to(dev)
    eps=torch.from_numpy(rng.standard_normal((S,B,L)).astype(np.float32)).to(dev)
    res=[]
    try:
        for mid in (True,False):
            eng=Engine(F,L,H,lk,batch_norm=True,device=dev,seed=3)
            g=torch.Generator().manual_seed(9)
            for name,p in eng.named_parameters().items():
                if not name.endswith("weights"): p.copy_(torch.randn(p.shape,generator=g)*0.1)
            eng.set_mid_chain(mid)
            ll=torch.zeros(S*B,device=dev)
            s=eng.step(x,x,eps=eps,training=True,n_iw=n_iw,n_mc=n_mc,warm_up_weight=0.6,outputs={"log_p_x_given_z":ll}).clone()
            ev=eng.step(x,x,eps=eps,training=False,n_iw=n_iw,n_mc=n_mc).clone()
            torch.cuda.synchronize()
            res.append([s.cpu(),ll.cpu(),eng.grads.clone().cpu(),eng.moving.clone().cpu(),eng.params.clone().cpu(),ev.cpu()])
    except Exception as e:
        print("EXC",B,H,L,F,lk,n_iw,n_mc,repr(e)[:200]); bad+=1; continue
    n+=1
    for k,(a,b) in enumerate(zip(*res)):
        if not (torch.isfinite(a).all() and torch.isfinite(b).all()):
            # non-finite in both is fine only if identical pattern
            if not torch.equal(torch.isfinite(a),torch.isfinite(b)):
                print("NONFINITE",k,B,H,L,F,lk,n_iw,n_mc); bad+=1; break
            continue
        sc=b.abs().max().item()
        err=(a-b).abs().max().item()
        if err>5e-5*sc+1e-8:
            print("MISMATCH",k,err/(sc+1e-30),B,H,L,F,lk,n_iw,n_mc); bad+=1; break
print("configs",n,"bad",bad)
