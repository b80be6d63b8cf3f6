import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch; torch.manual_seed(0)
from trained_weights import train_nerf
sd, rep = train_nerf(600)
print(rep)
np.savez("gpurun_out/trained_nerf_600.npz", **sd)
sd, rep = train_nerf(3000)
print(rep)
np.savez("gpurun_out/trained_nerf_3000.npz", **sd)
