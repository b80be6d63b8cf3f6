import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.nerf import NeRF
from evdeblurnerf_amd.renderer import NeRFAll
dev = "cuda"
R, S = 4096, 128
K = W.synthetic_camera()
rays = torch.as_tensor(W.synthetic_rays(100, R), device=dev)
rb = NeRFAll.ray_batch_train(400, 400, K, rays).contiguous()
z = torch.linspace(0, 1, S, device=dev).expand(R, S).contiguous()
sd = W.make_nerf_state_dict(21)
variants = {"full": sd}
z0 = {k: v.copy() for k, v in sd.items()}
for k in z0:
    if "weight" in k: z0[k] = np.zeros_like(z0[k])
variants["zero weights"] = z0
v = {k: x.copy() for k, x in sd.items()}
for k in v:
    if k.startswith("pts_linears") and k.endswith("weight") and not k.startswith("pts_linears.0"): v[k] = np.zeros_like(v[k])
variants["hidden weights zero (L0 + heads only)"] = v
v = {k: x.copy() for k, x in sd.items()}
v["views_linears.0.weight"] = v["views_linears.0.weight"].copy(); v["views_linears.0.weight"][:, 256:] = 0
variants["no dir columns"] = v
v = {k: x.copy() for k, x in sd.items()}
v["pts_linears.5.weight"] = v["pts_linears.5.weight"].copy(); v["pts_linears.5.weight"][:, :63] = 0
variants["no skip PE columns"] = v
for name, s in variants.items():
    net = NeRF(s)
    ref, _ = net.mlpforward(rb, z, precision="f32")
    raw, _ = net.mlpforward(rb, z, precision="f16c")
    f16, _ = net.mlpforward(rb, z, precision="f16")
    d = (raw - ref).abs().reshape(-1, 4)
    d16 = (f16 - ref).abs().reshape(-1, 4)
    print(f"{name}: f16c max {d.max(0).values.tolist()}  f16 max {d16.max(0).values.tolist()}")
    if name == "full":
        e = d.max(1).values.reshape(-1, 128)      # [workgroup, sample in workgroup]
        print("  by wave:", e.reshape(-1, 4, 32).amax((0, 2)).tolist())
        print("  by lane (n):", [f"{x:.1e}" for x in e.reshape(-1, 4, 32).amax((0, 1)).tolist()])
        print("  frac samples with err > 1e-3:", float((e > 1e-3).float().mean()))
        big = (e > 1e-2).nonzero()[:10].tolist()
        print("  some big:", big)
net = NeRF(variants["hidden weights zero (L0 + heads only)"])
ref, _ = net.mlpforward(rb, z, precision="f32")
bads = []
for rep in range(3):
    raw, _ = net.mlpforward(rb, z, precision="f16c")
    e = (raw - ref).abs().amax(-1)     # [R, S]
    bad = (e.amax(1) > 1e-3).nonzero().flatten().tolist()
    bads.append(bad)
    print(f"run {rep}: {len(bad)} bad rays; first {bad[:12]}; all samples of a bad ray bad: {float((e[bad] > 1e-3).float().mean()):.3f}")
print("same set every run:", bads[0] == bads[1] == bads[2])
b = bads[0][:6]
print("viewdirs of bad rays:", rb[b, 8:11].tolist())
good = [i for i in range(20) if i not in bads[0]][:4]
print("viewdirs of good rays:", rb[good, 8:11].tolist())
print("raw f16c vs ref at bad ray 0 sample 0:", raw[b[0], 0].tolist(), ref[b[0], 0].tolist())
