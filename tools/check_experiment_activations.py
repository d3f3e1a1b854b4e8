import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests import helpers as H
from oracle import flow_oracle as fo
sd = H.full_kernel_sd(calibrated=True)
d,_ = H.load("kernel_full_ad_calibrated")
m = H.tw_kernel_model(sd, path=3); m._defer_range_check += 1
S=1000
g = torch.Generator().manual_seed(5)
zc, zv = fo.draw_latents(sd, S, (1, 22, 3), g)
xc = d["x_coords"] - fo.centre_of_mass(d["x_coords"], d["masked"])
acts, out = m.debug_netblock(3, 0, d["atom_types"].cuda(), xc.cuda(), d["x_velocs"].cuda(), d["masked"].cuda(), zc[:,0].cuda(), 3)
for i in range(acts.shape[0]):
    a = acts[i]
    print("stage", i, "finite frac", float(torch.isfinite(a).float().mean()), "absmax", float(a[torch.isfinite(a)].abs().max()), "std", float(a[torch.isfinite(a)].std()))
print("out finite", float(torch.isfinite(out).float().mean()), float(out[torch.isfinite(out)].abs().max()))
