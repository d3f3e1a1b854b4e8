"""A 600-state Metropolis-Hastings chain on the fast path (TW_PATH_FUSED_H1) next to the same chain on the split-fp16 parity
kernel, same host-drawn noise, weights whose coupling nets act, AMBER energy kernel: how long do the two chains stay
identical (accept indicators), and how far apart are the exponents of the acceptance ratio?  r04 result
(profiles/r04_h1_chain_check.txt): identical for all 601 recorded proposals (27 accepted), max |d exponent| 5.3e-3."""
import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from tests import helpers as H
from timewarp_amd import synthetic
from timewarp_amd.dataloader import single_state_batch
from timewarp_amd.energy import AmberPotentialEnergyTorch
from timewarp_amd.utils.evaluation_utils import sample_with_model
sd = H.mh_state_dict("scaled", True)
types, coords, masses = synthetic.alanine_dipeptide_state()
energy = AmberPotentialEnergyTorch.alanine_dipeptide()
kw = dict(accept=True, num_proposal_steps=64, random_velocs=True, resample_velocs=True)
res = {}
for path in (3, 4):
    model = H.tw_kernel_model(sd, path=path)
    out = sample_with_model(single_state_batch("ad", types, coords), model, torch.device("cuda"), energy, masses, 600, disable_tqdm=True, noise=H.HostNoise(5, "cuda"), **kw)
    c, v, acc, st = out
    res[path] = out
    print("path", path, "states", c.shape[0], "accepted", acc, "mean p_acc", float(st.acceptance.mean()), "demoted", model.demoted)
a, b = res[3], res[4]
n = min(len(a[3].exponent), len(b[3].exponent))
same = (a[3].acceptance_indicator[:n] == b[3].acceptance_indicator[:n])
first = int(np.argmin(same)) if not same.all() else n
print("chains identical for the first", first, "of", n, "recorded proposals; max |dexp| over that prefix", float(np.abs(a[3].exponent[:first]-b[3].exponent[:first]).max()) if first else None)
