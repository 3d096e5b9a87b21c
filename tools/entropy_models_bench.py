import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from vbq_amd import ChannelwisePriorCDFQuantizer, priors
from bench import LAMBDAS, make_inputs
rows, C = 36864, 256
mu, sg, tab = make_inputs(rows, C, 0)
q = ChannelwisePriorCDFQuantizer(C, 10)
q.build_code_points(priors.FactoredGaussianPrior(np.zeros(C), np.sqrt(np.mean(mu.astype(np.float64)**2, axis=0))))
for _ in range(2): q.build_entropy_models_from_latents(mu, sg, LAMBDAS, add_n_smoothing=1)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): q.build_entropy_models_from_latents(mu, sg, LAMBDAS, add_n_smoothing=1)
torch.cuda.synchronize(); print("build_entropy_models_from_latents Kodak-24 x 32 lambdas:", (time.perf_counter()-t0)/3*1e3, "ms")
import cProfile, pstats
pr=cProfile.Profile(); pr.enable(); q.build_entropy_models_from_latents(mu, sg, LAMBDAS, add_n_smoothing=1); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
