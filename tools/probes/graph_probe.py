import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from types import SimpleNamespace
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.renderer import NeRFAll
sd = W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True, rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)
model = NeRFAll(args, sd, precision="f16").eval()
K = W.synthetic_camera(); rays = torch.as_tensor(W.synthetic_rays(100, 4096), device="cuda")
kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=128, N_importance=0, retraw=False)
def step(): return model.render(400, 400, K, rays=rays, **kw)[0]
for _ in range(5): ref = step()
torch.cuda.synchronize()
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  ms/step", timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay(); torch.cuda.synchronize()
print("graph  ms/step", timeit(g.replay), " max diff vs eager", float((out - ref).abs().max()))
