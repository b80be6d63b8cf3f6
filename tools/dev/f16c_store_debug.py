"""developer script: where the f16c training store differs from the float64 activations (per fragment / lane half / element)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from test_gpu_train_f16c import _level, _slots
from test_gpu_train import vdecode
from torch_restatement import TorchVoxLevel
net, sd, HD, G, FT = _level("fine", "f16c")
R, S = 70, 33
rs = np.random.RandomState(11)
pts = rs.uniform(-1, 1, (R, S, 3)).astype(np.float32)
d = rs.normal(size=(R, 3)); vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
fts = (0.3 * rs.normal(size=(R, S, FT))).astype(np.float32)
T = lambda a: torch.tensor(a, device="cuda")
raw, store, _ = net.mlpforward_train(T(pts), T(vd), T(fts), "f16c")
torch.cuda.synchronize()
n = R * S
sl = _slots(HD, G, FT)
ref = TorchVoxLevel(sd)
keep = {}
with torch.no_grad():
    rraw, rgeo = ref(torch.tensor(pts, dtype=torch.float64).reshape(-1, 3), torch.tensor(np.repeat(vd[:, None], S, 1), dtype=torch.float64).reshape(-1, 3),
                     torch.tensor(fts, dtype=torch.float64).reshape(-1, FT), want_geo=True, keep=keep)
geo = vdecode(store, n, sl["TILE_FRAGS"], sl["GEO"], 8, torch.float16).cpu().double()
err = (geo - rgeo).abs()
print("geo err by channel block of 16:", [f"{err[:, 16*j:16*j+16].max().item():.2e}" for j in range(8)])
bad = (err > 1e-2).nonzero()
print("bad entries:", bad.shape[0], "of", err.numel())
print("bad sample idx mod 32 histogram:", np.bincount((bad[:, 0] % 32).numpy(), minlength=32))
print("bad tile histogram (first 20):", np.bincount((bad[:, 0] // 32).numpy())[:20])
print("bad channel histogram:", np.bincount(bad[:, 1].numpy(), minlength=128))
for k, slot in (("hid", sl["HID"]), ("c0", sl["C0"]), ("c1", sl["C1"])):
    a = vdecode(store, n, sl["TILE_FRAGS"], slot, 16, torch.float16).cpu().double()
    print(k, "err", (a - keep[k].clamp(min=0)).abs().max().item())
