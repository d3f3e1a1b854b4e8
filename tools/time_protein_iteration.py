"""Whole MH iterations on the reference's 691-atom test protein (1hgv: topology and a frame from tests/golden/energy_kat_1hgv.npz,
amber99sb-ildn + OBC tables) with the full-size kernel-attention flow on the model's default path there (TW_PATH_SIMPLE_H3):
ms per iteration at S proposals, and the AMBER energy launch on its own.  `python tools/time_protein_iteration.py 16 64`"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from timewarp_amd.dataloader import elements_from_atom_names, single_state_batch
from timewarp_amd.energy import AmberPotentialEnergyTorch
from timewarp_amd.forcefield import ELEMENT_MASSES, amber99sbildn_obc_tables
from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain

z = np.load(H.GOLDEN + "/energy_kat_1hgv.npz")
names = [str(n) for n in z["atom_names"]]
tables = amber99sbildn_obc_tables(names, [str(r) for r in z["residue_names"]], [int(i) for i in z["residue_ids"]], improper_neighbour_order="pyset")
V = len(names)
types = elements_from_atom_names(names)
coords = torch.from_numpy(z["positions"][0].astype(np.float32))
masses = torch.tensor([ELEMENT_MASSES[next(ch for ch in n if ch.isalpha())] for n in names], dtype=torch.float32)
energy = AmberPotentialEnergyTorch(tables)
sd = H.mh_state_dict("scaled", True, out_scale=1e-6, coords_log_scale=-9.5)
dev = torch.device("cuda")
model = H.tw_kernel_model(sd, path=None)
for S in [int(a) for a in sys.argv[1:]] or [16]:
    chain = MetropolisHastingsChain(single_state_batch("1hgv", types, coords), model, dev, energy, masses, accept=True, num_proposal_steps=S,
                                    random_velocs=True, resample_velocs=True)
    with torch.no_grad():
        for _ in range(2):
            chain.step_deferred()
        chain.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 6
        for _ in range(n):
            chain.step_deferred()
        chain.flush()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        x = coords.to(dev)[None].repeat(S + 1, 1, 1).contiguous()
        energy(x); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            energy(x)
        torch.cuda.synchronize()
        de = (time.perf_counter() - t0) / 5 * 1e3
    print(f"1hgv (691 atoms) x {S} proposals, path {model._path_for(V)}: {dt:.2f} ms per MH iteration; energy of {S + 1} conformations alone {de:.3f} ms; "
          f"accepted {chain.accepted} of {chain.proposals // S} iterations", flush=True)

# the force kernel (hybrid moves / Langevin steps run on it): energy + forces of 4 conformations
x4 = coords.to(dev)[None].repeat(4, 1, 1).contiguous()
with torch.no_grad():
    energy.energy_and_forces(x4); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        energy.energy_and_forces(x4)
    torch.cuda.synchronize()
    print(f"1hgv: energy + forces of 4 conformations {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per launch", flush=True)
