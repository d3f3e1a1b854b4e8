"""How ill-conditioned is the reverse-move density of BASELINE config 3's test case (60 atoms, un-calibrated weights)?
tests/test_flow_gpu.py::test_full_size_v60_S512_rows_vs_oracle compares log p(x|y) of the GPU's proposals with the oracle's
log p(x|y_oracle): two evaluations at inputs that differ by the proposals' own error.  This script measures, in fp64, what
a perturbation of y by 1 / 2 / 3e-6 of its scale does to that log-density, and how far the fp32 oracle itself sits from the
same computation in fp64.  CPU only.  Output: profiles/r05_v60_conditioning.txt"""
import sys, torch, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
torch.set_num_threads(8)
sd = H.full_kernel_sd()
d, _ = H.load("kernel_full_v60")
V = 60; S = 512
g = torch.Generator().manual_seed(11)
zc, zv = fo.draw_latents(sd, S, (1, V, 3), g)
rows = list(range(0, 4)) + list(range(256, 260)) + list(range(508, 512))
rest = [r for r in torch.randperm(S, generator=g).tolist() if r not in rows][:28]
rows = torch.tensor(sorted(rows + rest))
t = time.time()
ryc, ryv, rlp = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, d["atom_types"], d["x_coords"], d["x_velocs"], d["masked"], zc[rows], zv[rows])
n = len(rows)
def yx(sd_, yc, yv, dt):
    return fo.log_likelihood(sd_, H.FULL_KERNEL_SPEC, d["atom_types"].repeat(n, 1), yc.squeeze(1).to(dt), -yv.squeeze(1).to(dt),
                             d["x_coords"].repeat(n, 1, 1).to(dt), -d["x_velocs"].repeat(n, 1, 1).to(dt), d["masked"].repeat(n, 1))
r32 = yx(sd, ryc, ryv, torch.float32)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
r64 = yx(sd64, ryc, ryv, torch.float64)
print("fp32 oracle vs fp64 at the same inputs: rel", H.rel_err(r32, r64), "elem", H.elem_rel_err(r32, r64))
# sensitivity: perturb y by 2e-6 of its scale
for eps in (1e-6, 2e-6, 3e-6):
    gy = torch.Generator().manual_seed(5)
    pc = ryc + eps * ryc.abs().max() * torch.randn(ryc.shape, generator=gy)
    pv = ryv + eps * ryv.abs().max() * torch.randn(ryv.shape, generator=gy)
    p64 = yx(sd64, pc.double(), pv.double(), torch.float64)
    print(f"fp64, inputs perturbed by {eps:g} of scale: elem", H.elem_rel_err(p64, r64), "rel", H.rel_err(p64, r64))
# the fp64 sampled y vs fp32 oracle's y: the reference's own y noise
eyc, eyv, elp = fo.conditional_sample_with_logp(sd64, H.FULL_KERNEL_SPEC, d["atom_types"], d["x_coords"].double(), d["x_velocs"].double(), d["masked"], zc[rows].double(), zv[rows].double())
print("oracle fp32 y vs fp64 y: coords", H.rel_err(ryc, eyc), "velocs", H.rel_err(ryv, eyv), "logp elem", H.elem_rel_err(rlp, elp))
e64 = yx(sd64, eyc, eyv, torch.float64)
print("fp32 oracle chain (y32 -> p_yx32) vs fp64 chain: elem", H.elem_rel_err(r32, e64), "rel", H.rel_err(r32, e64))
print("time", time.time() - t)
