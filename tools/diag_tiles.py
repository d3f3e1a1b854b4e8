import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
d, _ = H.load("kernel_full_ad")
sd = H.full_kernel_sd()
xc = d["x_coords"] - fo.centre_of_mass(d["x_coords"], d["masked"])
S = 2
outs = {}
for p in (1, 3):
    m = H.tw_kernel_model(sd, path=p)
    a, o = m.debug_netblock(7, 0, d["atom_types"].cuda(), xc.cuda(), d["x_velocs"].cuda(), d["masked"].cuda(), d["z_coords"][:S, 0].cuda(), p)
    outs[p] = a.cpu()
ref = outs[1]; got = outs[3]
for stage in (0, 1):
    e = (got[stage] - ref[stage]).abs().reshape(S * 22, 128)   # tokens x features
    scale = ref[stage].abs().max()
    print("stage", stage, "max rel", float(e.max() / scale))
    # per feature tile (16) x token (all)
    ft = e.reshape(S * 22, 8, 16).amax(dim=(0, 2)) / scale
    tk = e.amax(dim=1) / scale
    print("  per feature-tile:", [f"{v:.1e}" for v in ft.tolist()])
    print("  per token:", [f"{v:.0e}" for v in tk.tolist()])
