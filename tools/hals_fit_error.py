import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import cmf_jl_amd as cmf
from oracle import cmf_oracle as oracle
data, _, _ = oracle.c_gen_synthetic(N=120, T=1200, K=3, L=20, seed=1234)
W0, H0 = oracle.c_init_rand(data, L=20, K=8, seed=0)
res = cmf.fit_cnmf(data, L=20, K=8, alg=":hals", max_itr=12, check_convergence=False, W_init=W0, H_init=H0)
Wr, Hr, lr, _ = oracle.c_fit_hals(data, W0, H0, max_itr=12, check_convergence=False)
fr = lambda a,b: np.linalg.norm(a-b)/np.linalg.norm(b)
print("relW", fr(res.W, Wr), "relH", fr(res.H, Hr), "loss", np.max(np.abs(res.loss_hist-lr)/lr))
for name,a,b in (("W",res.W,Wr),("H",res.H,Hr)):
    flip = (a==0) != (b==0)
    print(name, "entries", a.size, "zero-pattern differs at", int(flip.sum()), "frob rel without those", np.linalg.norm((a-b)[~flip])/np.linalg.norm(b), "max abs at flips", float(np.abs(a-b)[flip].max()) if flip.any() else 0.0)
