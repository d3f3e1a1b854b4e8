"""The bench's MH chains, each configuration and both half-precision modes, run REPS times from the same seed: trajectories
(coordinates, velocities, accept counts) must repeat bit for bit - whole MH iterations (tw_mh_iteration: two flow passes, the
energy launch on its side stream, the accept kernels, deferred read-backs) under the same scrutiny as the single launches of
stress_all.py.  profiles/r04_stress_chain.txt."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

STEPS = int(os.environ.get("STRESS_STEPS", "96"))
REPS = int(os.environ.get("STRESS_REPS", "6"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for config in ("ad", "4aa", "4aa-nnqq", "dense"):
    for pname in ("h3", "h1"):
        first, bad, acc = None, 0, None
        for rep in range(REPS):
            chain, model = bench.build_chain(dev, 1234, bench.CONFIGS[config]["S"], bench.PATHS[pname]["path"], config)
            junk.fill_(float(rep))
            with torch.no_grad():
                for it in range(STEPS):
                    chain.step_deferred()
                    if (it + 1) % 8 == 0:
                        chain.flush()
                chain.flush()
                traj, velocs = chain.trajectory()
            torch.cuda.synchronize()
            sig = (traj.cpu().clone(), velocs.cpu().clone() if velocs is not None else None, chain.accepted, bool(getattr(model, "demoted", False)))
            if first is None:
                first = sig
            else:
                same = torch.equal(sig[0], first[0]) and sig[2] == first[2] and (first[1] is None or torch.equal(sig[1], first[1]))
                bad += not same
            del chain, model
        print(f"{config} {pname}: {bad}/{REPS - 1} repeats differ from the first run; {first[0].shape[0]} states, {first[2]} accepted in {STEPS} iterations, "
              f"range guard fired: {first[3]}", flush=True)
