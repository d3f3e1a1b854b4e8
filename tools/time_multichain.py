"""Throughput of lock-step chains on one GPU: C chains x S proposals per iteration, same rows per launch as
the bench's 1 x 1000.  Not the BASELINE configuration - an aside for DESIGN.md."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from timewarp_amd import synthetic
from timewarp_amd.dataloader import single_state_batch
from timewarp_amd.energy import AmberPotentialEnergyTorch
from timewarp_amd.utils.multichain import MetropolisHastingsChains

sd = synthetic.synth_state_dict(H.full_kernel_sd(), 0, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
model = H.tw_kernel_model(sd, path=3)
types, coords, masses = synthetic.alanine_dipeptide_state()
energy = AmberPotentialEnergyTorch.alanine_dipeptide()
dev = torch.device("cuda")
for C, S in ((1, 1000), (4, 250), (8, 125), (16, 62), (32, 31)):
    chains = MetropolisHastingsChains([single_state_batch("ad", types, coords) for _ in range(C)], model, dev, energy, masses, S,
                                      random_velocs=True, resample_velocs=True)
    with torch.no_grad():
        for _ in range(3):
            chains.step_deferred()
        chains.flush()
        torch.cuda.synchronize()
        a0 = sum(chains.accepted)
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            chains.step_deferred()
            if (i + 1) % 5 == 0:
                chains.flush()
        chains.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{C:3d} chains x {S:4d} proposals: {dt / n * 1e3:.2f} ms per lock-step iteration, {(sum(chains.accepted) - a0) / dt:.0f} accepted/s, "
          f"{sum(chains.emitted) / C:.0f} states per chain", flush=True)
