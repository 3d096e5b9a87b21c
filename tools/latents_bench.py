import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from vbq_amd import ChannelwisePriorCDFQuantizer, priors
from bench import LAMBDAS
C=256
rng=np.random.default_rng(0)
q=ChannelwisePriorCDFQuantizer(C,10)
q.build_code_points(priors.FactoredGaussianPrior(np.zeros(C), np.exp(rng.uniform(-1,1,C))))
means=rng.normal(0,1,(1,32,48,C)).astype(np.float32); logvars=rng.normal(-4,1,(1,32,48,C)).astype(np.float32)
class V:
    def encode(self,X): return means, logvars
    def decode(self,Z): return np.zeros((np.asarray(Z).shape[0],512,768,3),np.float32)
lambs=LAMBDAS
q.build_entropy_models(np.zeros((1,512,768,3),np.float32), V(), lambs, add_n_smoothing=1)
for _ in range(3): out=q.compress_latents(means, logvars, lambs)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): out=q.compress_latents(means, logvars, lambs)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print(f"compress_latents 1 image x {len(lambs)} lambdas: {dt*1e3:.2f} ms")
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(10): out=q.compress_latents(means, logvars, lambs)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
