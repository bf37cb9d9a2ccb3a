import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmf_jl_amd as cmf
N,T,K,L = 2000,50000,32,20
data = cmf.gen_synthetic(N=N,T=T,seed=1234)
W0,H0 = cmf.init_rand(data,L=L,K=K,seed=0)
rule = cmf.HALSUpdate(data,W0,H0)
print("loss0", rule.compute_loss())
for it in range(4):
    t0=time.perf_counter(); rule.update_motifs(); 
    l = rule.compute_loss(); t1=time.perf_counter()
    l2 = rule.update_feature_maps(); t2=time.perf_counter()
    print(it, "W phase(+loss) %.1f ms  H phase %.1f ms  loss after W %.5f after H %.5f" % ((t1-t0)*1e3, (t2-t1)*1e3, l, l2), flush=True)
rule.close()
mu = cmf.MultUpdate(data,W0,H0)
for it in range(4):
    mu.update_motifs(); print("mu", it, mu.update_feature_maps())
