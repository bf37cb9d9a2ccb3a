import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.'))
import numpy as np
import cmf_jl_amd as cmf
t=time.perf_counter(); data = cmf.gen_synthetic(N=2000, T=50000, seed=1234); print('gen_synthetic', round(time.perf_counter()-t,3))
t=time.perf_counter(); W0,H0 = cmf.init_rand(data, L=20, K=32, seed=0); print('init_rand', round(time.perf_counter()-t,3))
t=time.perf_counter(); rule = cmf.MultUpdate(data, W0, H0); print('MultUpdate ctor (upload)', round(time.perf_counter()-t,3))
t=time.perf_counter(); lh,th,_ = rule.fit_native(100, np.inf, False, 3, 1e-4, False); print('cmf_fit 100 iterations', round(time.perf_counter()-t,3))
t=time.perf_counter(); W,H = rule.download(); print('download', round(time.perf_counter()-t,3)); rule.close()
t=time.perf_counter(); res = cmf.fit_cnmf(data, L=20, K=32, alg=":mult", max_itr=100, check_convergence=False, seed=0); print('fit_cnmf total (100 iterations)', round(time.perf_counter()-t,3), res.loss_hist[-1])
