import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gsbp_amd
from gsbp_amd import synthetic as syn
from util import scene_np, to_dev
dev = torch.device("cuda:0")
cfg, sc = scene_np("T1")
d = to_dev(sc, dev)
D = 128
feats = torch.randn(cfg.height, cfg.width, D, generator=torch.Generator().manual_seed(3))
feats[10, 20, :] = float("nan")
feats[50, 100, 7] = float("inf")
feats[100, 150, 64:] = float("-inf")
eng = gsbp_amd.Engine(cfg.n_gaussians, cfg.width, cfg.height, device=dev, group_scatter=True)
view = eng.view(d["vms"][0].cpu(), d["K"].cpu(), cfg.width, cfg.height)
eng.project(view, d["means"], d["quats"], d["scales"], d["opac"]); eng.bin_sort(view); eng.blend_weights(view)
F = torch.zeros(cfg.n_gaussians, D, device=dev)
eng.scatter(view, feats.to(dev), F, None)
eng.set_group_scatter(False)
F2 = torch.zeros_like(F)
eng.scatter(view, feats.to(dev), F2, None)
a, b = F.cpu(), F2.cpu()
print("stats", eng.stats())
na, nb = torch.isnan(a), torch.isnan(b)
ia, ib = torch.isinf(a), torch.isinf(b)
print("nan elems", int(na.sum()), int(nb.sum()), "rows", int(na.any(1).sum()), int(nb.any(1).sum()))
print("inf elems", int(ia.sum()), int(ib.sum()), "rows", int(ia.any(1).sum()), int(ib.any(1).sum()))
diff = (na != nb) | (ia != ib)
rows = torch.nonzero(diff.any(1))[:, 0]
print("rows differing", rows[:10].tolist(), len(rows))
for r in rows[:3].tolist():
    cols = torch.nonzero(diff[r])[:, 0][:6].tolist()
    print(r, cols, [float(a[r, c]) for c in cols], [float(b[r, c]) for c in cols])
gid, pix, w = eng.dump_pairs(view)
gid, pix = gid.cpu().numpy(), pix.cpu().numpy()
for (y, x) in ((10, 20), (50, 100), (100, 150)):
    sel = pix == y * cfg.width + x
    print("pixel", y, x, "gaussians", sorted(set(gid[sel].tolist()))[:10], int(sel.sum()))
