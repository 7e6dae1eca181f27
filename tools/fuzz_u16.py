"""Fuzz: a training + evaluation step on the uint16 minibatch against the same step on the
fp32 batch (both on the count kernels) over random shapes -- bit-identical results expected.
Usage (GPU box): PYTHONPATH=. python tools/fuzz_u16.py <seed> <configs>"""
import numpy as np, torch, sys
import scipy.sparse as sp
from scvae_amd.engine import Engine
from scvae_amd.minibatch import DeviceCSR
dev=torch.device("cuda:0")
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
bad=0; n=0; skipped=0
LK=["negative binomial","poisson","zero-inflated negative binomial","zero-inflated poisson"]
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 100):
    n_iw=int(rng.integers(1,3)); n_mc=int(rng.integers(1,3)); S=n_iw*n_mc
    B=int(rng.integers(1,700))
    nl=int(rng.integers(0,3))
    H=tuple(int(rng.integers(1,64))*2 for _ in range(nl))
    L=int(rng.integers(1,60))
    F=int(rng.integers(5,3000))
    lk=LK[int(rng.integers(0,4))]
    N=B+int(rng.integers(0,50))
    xh=(rng.poisson(2.,(N,F))*(rng.random((N,F))<0.1)).astype(np.float32)
    k=max(1,xh.size//300)
    xh.flat[rng.integers(0,xh.size,k)]=rng.integers(256,65536,k).astype(np.float32)
    csr=DeviceCSR.from_scipy(sp.csr_matrix(xh),dev)
    rows=torch.from_numpy(rng.permutation(N)[:B]).to(dev)
    x32=csr.gather_dense(rows); rc=torch.zeros(B,device=dev)
    x16=csr.gather_counts_u16(rows,row_const_out=rc)
    eps=torch.from_numpy(rng.standard_normal((S,B,L)).astype(np.float32)).to(dev)
    res=[]
    try:
        ok=True
        for u16 in (False,True):
            eng=Engine(F,L,H,lk,batch_norm=True,device=dev,seed=3)
            eng.set_count_gemm(True,always=True)
            eng.set_dd_atomics(False)   # (bit identity needs the fixed-order slabs: the plan default since round 5 is atomics)
            if u16 and not eng.accepts_counts_u16(B,True):
                ok=False; break
            x = x16 if u16 else x32
            ll=torch.zeros(S*B,device=dev)
            s=eng.step(x,x,eps=eps,training=True,n_iw=n_iw,n_mc=n_mc,row_const=rc,x_counts=True,outputs={"log_p_x_given_z":ll}).clone()
            ev=eng.step(x,x,eps=eps,training=False,n_iw=n_iw,n_mc=n_mc,row_const=rc,x_counts=True).clone()
            torch.cuda.synchronize()
            res.append([s.cpu(),ll.cpu(),eng.grads.clone().cpu(),eng.moving.clone().cpu(),ev.cpu()])
        if not ok:
            skipped+=1; continue
    except Exception as e:
        print("EXC",B,H,L,F,lk,n_iw,n_mc,repr(e)[:300]); bad+=1; continue
    n+=1
    for k,(a,b) in enumerate(zip(*res)):
        same = torch.equal(a,b) or (torch.equal(torch.isnan(a),torch.isnan(b)) and torch.equal(torch.nan_to_num(a),torch.nan_to_num(b)))
        if not same:
            print("MISMATCH",k,(a-b).abs().max().item(),B,H,L,F,lk,n_iw,n_mc); bad+=1; break
print("configs",n,"skipped",skipped,"bad",bad)
