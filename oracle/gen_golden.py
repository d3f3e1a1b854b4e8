"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference does not exist on the GPU box and never
travels).  The committed fixtures are pure data: inputs, explicit noise and the reference's
outputs.  Full-size weights are NOT stored: both sides regenerate them from the name-seeded
recipe in oracle/flow_oracle.py (synth_state_dict); tiny models store their state_dict.

    python oracle/gen_golden.py            # rewrites tests/golden/

Import recipe: SURVEY.md appendix A (package symlink + 4 stub modules).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    link_dir = "/tmp/tw_oracle_ref"
    os.makedirs(link_dir, exist_ok=True)
    link = os.path.join(link_dir, "timewarp")
    if not os.path.islink(link):
        os.symlink(REF, link)
    sys.path.insert(0, link_dir)
    sys.path.append(REF)  # append: reference/profile.py shadows stdlib `profile`

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            if k[:1].isupper():  # something that may be used as a base class
                c = type(k, (), {"__init__": lambda self, *a, **kw: None})
                setattr(self, k, c)
                return c
            m = _Stub(f"{self.__name__}.{k}")
            setattr(self, k, m)
            return m

        def __call__(self, *a, **k):
            return None

    # third-party packages that are absent here and only imported at module scope by the files
    # we need (utils/evaluation_utils.py:4,13-15,23 for the MH loop).  None
    # of their functionality is exercised: the energy is passed to sample_with_model as a callable.
    for n in ("mdtraj", "pymol2", "torch.utils.tensorboard", "tensorboard", "git", "git.types", "openmm",
              "openmm.app", "openmm.unit", "bgflow", "bgflow.distribution", "bgflow.distribution.energy",
              "bgflow.distribution.energy.openmm", "bgflow.distribution.energy.base", "bgflow.utils",
              "bgflow.utils.types", "matplotlib", "matplotlib.pyplot", "simtk", "simtk.unit", "simtk.openmm",
              "simtk.openmm.app"):
        sys.modules[n] = _Stub(n)


import_reference()
sys.path.insert(0, ROOT)

from timewarp.model_constructor import (  # noqa: E402
    custom_transformer_nvp_constructor,
    transformer_nvp_constructor,
    model_constructor,
)
from timewarp.model_configs import (  # noqa: E402
    CustomAttentionTransformerNVPConfig,
    TransformerNVPConfig,
    ModelConfig,
)
from timewarp.modules.layers.custom_attention_encoder import CustomAttentionEncoderLayerConfig  # noqa: E402
from timewarp.modules.layers.transformer_block import TransformerConfig  # noqa: E402
from timewarp.modules.layers.rff_position_encoder import RFFPositionEncoderConfig  # noqa: E402
from timewarp.modules.layers.kernel_attention import compute_kernel_attention_scores  # noqa: E402

from oracle import flow_oracle as fo  # noqa: E402

AD_NAMES = "1HH3 CH3 2HH3 3HH3 C O N H CA HA CB 1HB 2HB 3HB C O N H CH3 1HH3 2HH3 3HH3".split()
VOCAB = {"C": 0, "H": 1, "N": 2, "O": 3, "S": 4}


def ad_topology():
    """simulation/testdata/alanine-dipeptide.pdb: coordinates (A -> nm) and element ids."""
    coords, types = [], []
    for line in open(os.path.join(REF, "simulation/testdata/alanine-dipeptide.pdb")):
        if line.startswith("ATOM"):
            name = line[12:16].strip()
            el = next(ch for ch in name if ch.isalpha())
            types.append(VOCAB[el])
            coords.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return torch.tensor(coords, dtype=torch.float32) * 0.1, torch.tensor(types, dtype=torch.int64)


def kernel_model(emb, d_model, ff, mlp_hidden, n_coupling, n_layers, lengthscales, attention_type="kernel",
                 cheb_order=None, force_asymptotic_zero=None, normalise=True):
    enc = CustomAttentionEncoderLayerConfig(
        d_model=d_model, dim_feedforward=ff, dropout=0.0, num_heads=len(lengthscales),
        attention_type=attention_type, lengthscales=list(lengthscales), normalise_kernel_values=normalise,
        cheb_order=cheb_order, force_asymptotic_zero=force_asymptotic_zero,
    )
    cfg = CustomAttentionTransformerNVPConfig(
        atom_embedding_dim=emb, latent_mlp_hidden_dims=list(mlp_hidden), num_coupling_layers=n_coupling,
        num_transformer_layers=n_layers, encoder_layer_config=enc,
    )
    return custom_transformer_nvp_constructor(cfg).eval()


def dense_model(emb, d_model, ff, mlp_hidden, n_coupling, n_layers, n_head, rff=None):
    tc = TransformerConfig(n_head=n_head, dim_feedforward=ff, dropout=0.0)
    cfg = TransformerNVPConfig(
        atom_embedding_dim=emb, transformer_hidden_dim=d_model, latent_mlp_hidden_dims=list(mlp_hidden),
        num_coupling_layers=n_coupling, num_transformer_layers=n_layers, transformer_config=tc,
        rff_position_encoder_config=rff,
    )
    return transformer_nvp_constructor(cfg).eval()


def np_sd(sd):
    return {"sd::" + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def run_case(model, atom_types, x_c, x_v, mask, y_c, y_v, S, seed):
    """Reference outputs for the three model calls of one MH iteration."""
    B = x_c.shape[0]
    none = torch.zeros((0, 2), dtype=torch.int64)
    ebi = torch.zeros((0,), dtype=torch.int64)
    out = {}
    with torch.no_grad():
        out["loglik"] = model.log_likelihood(
            atom_types=atom_types, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v,
            adj_list=none, edge_batch_idx=ebi, masked_elements=mask,
        ).numpy()
        if B == 1:
            # explicit latents: re-seed so the reference draws exactly these (flow.py:274-275)
            torch.manual_seed(seed)
            sc = torch.exp(model.coords_prior_log_scale)
            sv = torch.exp(model.velocs_prior_log_scale)
            z_c = torch.distributions.Normal(torch.zeros_like(x_c), sc).rsample((S,))
            z_v = torch.distributions.Normal(torch.zeros_like(x_v), sv).rsample((S,))
            torch.manual_seed(seed)
            yy_c, yy_v, lp = model.conditional_sample_with_logp(
                atom_types=atom_types, x_coords=x_c, x_velocs=x_v, adj_list=none, edge_batch_idx=ebi,
                masked_elements=mask, num_samples=S,
            )
            out.update(z_coords=z_c.numpy(), z_velocs=z_v.numpy(), s_y_coords=yy_c.numpy(),
                       s_y_velocs=yy_v.numpy(), s_logp=lp.numpy())
            # the reverse-move density used by MH (evaluation_utils.py:648-657, velocities negated)
            yc, yv = yy_c.squeeze(1), yy_v.squeeze(1)
            out["logp_yx"] = model.log_likelihood(
                atom_types=atom_types.repeat(S, 1), y_coords=x_c.repeat(S, 1, 1), y_velocs=-x_v.repeat(S, 1, 1),
                x_coords=yc, x_velocs=-yv, adj_list=none, edge_batch_idx=ebi, masked_elements=mask.repeat(S, 1),
            ).numpy()
    return out


def base_inputs(atom_types, x_c, x_v, mask, y_c, y_v):
    return dict(atom_types=atom_types.numpy(), x_coords=x_c.numpy(), x_velocs=x_v.numpy(),
                masked=mask.numpy(), y_coords=y_c.numpy(), y_velocs=y_v.numpy())


def padded_batch(g, B, V, lens):
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.3
    x_v = torch.randn(B, V, 3, generator=g)
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.05
    y_v = torch.randn(B, V, 3, generator=g)
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b, n in enumerate(lens):
        mask[b, n:] = True
    return at, x_c, x_v, mask, y_c, y_v


class SyntheticEnergy:
    """E(x) = k * sum_atoms |x - x_ref|^2 + w * sum_{i<j} exp(-|xi - xj|^2) in kJ/mol, [n,1]; `.kbT`.
    Stands in for OpenmmPotentialEnergyTorch (openmm_bridge.py:252-307) in the MH scenarios: the
    loop only needs a callable with that contract.  Restated in oracle/mh_oracle.py."""

    def __init__(self, x_ref, k=40.0, w=3.0, kbT=2.5):
        self.x_ref, self.k, self.w, self.kbT = x_ref, k, w, kbT

    def __call__(self, coords):
        c = coords.reshape(-1, self.x_ref.shape[-2], 3)
        e = self.k * ((c - self.x_ref) ** 2).sum(dim=(-1, -2))
        d2 = ((c[:, :, None, :] - c[:, None, :, :]) ** 2).sum(-1)
        iu = torch.triu_indices(c.shape[1], c.shape[1], offset=1)
        e = e + self.w * torch.exp(-d2[:, iu[0], iu[1]]).sum(-1)
        return e[:, None]


MH_SCENARIOS = {
    "s10": dict(accept=True, num_proposal_steps=10, num_samples=25),
    "s10_randv": dict(accept=True, num_proposal_steps=10, num_samples=25, random_velocs=True, resample_velocs=True),
    "adaptive": dict(accept=True, num_proposal_steps=10, num_samples=30, adaptive_parallelism=True),
    "noaccept_s1": dict(accept=False, num_proposal_steps=1, num_samples=6),
    "chirality": dict(accept=True, num_proposal_steps=10, num_samples=20, chirality=True),
    # rotate=True is not generated: the reference itself raises there ((Q @ x.T).T on a [1,V,3] tensor,
    # utils/evaluation_utils.py:604-607)
    "init_random": dict(accept=True, num_proposal_steps=4, num_samples=8, initialize_randomly=True),
}
# OpenMM steps inside the chain (evaluation_utils.py:558-565, 594-602, 623-626) with oracle/fake_sim.FakeSimulation in
# the Simulation's place (`fake_sim=True` -> a fresh one per scenario); openmm_on_proposal needs one proposal per
# iteration (openmm_step squeezes dimension 0)
MH_OPENMM_SCENARIOS = {
    "omm_current": dict(accept=True, num_proposal_steps=10, num_samples=20, fake_sim=True, num_openmm_steps=3,
                        openmm_on_current=True),
    "omm_current_randv": dict(accept=True, num_proposal_steps=10, num_samples=20, random_velocs=True, resample_velocs=True,
                              fake_sim=True, num_openmm_steps=2, openmm_on_current=True),
    "omm_proposal": dict(accept=True, num_proposal_steps=1, num_samples=10, fake_sim=True, num_openmm_steps=2,
                         openmm_on_proposal=True),
    "omm_both_noaccept": dict(accept=False, num_proposal_steps=1, num_samples=6, fake_sim=True, num_openmm_steps=1,
                              openmm_on_proposal=True, openmm_on_current=True),
}


def gen_mh_goldens(model, scenarios=None, out_name="mh_tiny.npz"):
    """Run the REAL sample_with_model (utils/evaluation_utils.py:468-745) on CPU with the tiny
    kernel model, a synthetic energy and recorded noise; store inputs, noise and all outputs."""
    from timewarp.utils import evaluation_utils as eu
    from timewarp.dataloader import DenseMolDynBatch
    import timewarp.utils.evaluation_utils as eu_mod

    V = 7
    g = torch.Generator().manual_seed(77)
    at = torch.randint(0, 5, (1, V), generator=g)
    x0 = torch.randn(1, V, 3, generator=g) * 0.3
    v0 = torch.randn(1, V, 3, generator=g)
    mask = torch.zeros(1, V, dtype=torch.bool)
    masses = torch.tensor([12.01, 1.008, 14.01, 16.0, 12.01, 1.008, 1.008])
    batch = DenseMolDynBatch(
        names=["tiny"], atom_types=at, adj_list=torch.zeros((0, 2), dtype=torch.int64),
        edge_batch_idx=torch.zeros((0,), dtype=torch.int64), atom_coords=x0, atom_velocs=v0,
        atom_forces=torch.zeros_like(x0), atom_coord_targets=x0, atom_veloc_targets=v0,
        atom_force_targets=torch.zeros_like(x0), masked_elements=mask)
    energy = SyntheticEnergy(x0.clone())
    centres = torch.tensor([[0, 1, 2, 3], [4, 5, 6, 1]])
    # make proposals small enough that some are accepted: shrink the prior and the last layers
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model.coords_prior_log_scale.fill_(-3.0)
        model.velocs_prior_log_scale.fill_(0.0)
        for k, v in model.state_dict().items():
            if ".out_mlp._layers.2." in k:
                v.mul_(0.002)
    ref_signs = None
    scenarios = MH_SCENARIOS if scenarios is None else scenarios
    out = dict(atom_types=at.numpy(), x0=x0.numpy(), v0=v0.numpy(), masses=masses.numpy(), centres=centres.numpy())
    out.update(np_sd(model.state_dict()))
    for name, kw in scenarios.items():
        kw = dict(kw)
        rec = {"normal": [], "rand": [], "randn_like": [], "rot": []}
        orig_rsample = torch.distributions.Normal.rsample
        orig_rand, orig_randn_like = torch.rand, torch.randn_like
        orig_rot = eu_mod.random_rotation_matrix

        def rsample(self, shape=torch.Size()):
            r = orig_rsample(self, shape)
            rec["normal"].append(r.detach().numpy().copy().reshape(-1))
            return r

        def rand(*a, **k):
            r = orig_rand(*a, **k)
            rec["rand"].append(r.numpy().copy().reshape(-1))
            return r

        def randn_like(t, **k):
            r = orig_randn_like(t, **k)
            rec["randn_like"].append(r.numpy().copy().reshape(-1))
            return r

        def rot(device=None, dtype=None):
            gq = torch.Generator().manual_seed(1000 + len(rec["rot"]))
            q, r_ = torch.linalg.qr(torch.randn(3, 3, generator=gq, dtype=torch.float64))
            q = q * torch.sign(torch.diagonal(r_))
            if torch.det(q) < 0:
                q[:, 0] = -q[:, 0]
            q = q.to(dtype or torch.float32)
            rec["rot"].append(q.numpy().copy().reshape(-1))
            return q

        chir = kw.pop("chirality", False)
        extra = {}
        if kw.pop("fake_sim", False):
            from oracle.fake_sim import FakeSimulation
            extra["sim"] = FakeSimulation()
        if chir:
            from timewarp.utils.chirality import compute_chirality_sign
            extra = dict(chirality_centers=centres, reference_signs=compute_chirality_sign(x0, centres))
            out[name + "/reference_signs"] = extra["reference_signs"].numpy()
        torch.distributions.Normal.rsample = rsample
        torch.rand, torch.randn_like = rand, randn_like
        eu_mod.random_rotation_matrix = rot
        try:
            torch.manual_seed(4242)
            coords, velocs, accepted, stats = eu.sample_with_model(
                batch, model, torch.device("cpu"), energy, masses, disable_tqdm=True, **kw, **extra)
        finally:
            torch.distributions.Normal.rsample = orig_rsample
            torch.rand, torch.randn_like = orig_rand, orig_randn_like
            eu_mod.random_rotation_matrix = orig_rot
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros((0,), np.float32)
        out[name + "/coords"], out[name + "/velocs"] = coords, velocs
        out[name + "/accepted"] = np.array(accepted)
        for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
                  "energies_pot_delta", "energies_kin_delta"):
            out[f"{name}/stats_{f}"] = np.asarray(getattr(stats, f))
        out[name + "/noise_normal"] = cat(rec["normal"])
        out[name + "/noise_normal_sizes"] = np.array([len(a) for a in rec["normal"]])
        out[name + "/noise_rand"] = cat(rec["rand"])
        out[name + "/noise_rand_sizes"] = np.array([len(a) for a in rec["rand"]])
        out[name + "/noise_randn_like"] = cat(rec["randn_like"])
        out[name + "/noise_rot"] = cat(rec["rot"])
        print("mh", name, "states", coords.shape[0], "accepted", accepted, "iters", len(rec["rand"]),
              "mean p_acc", float(np.mean(stats.acceptance)))
    model.load_state_dict(sd)
    np.savez_compressed(os.path.join(OUT, out_name), **out)


def gen_sob_golden(model):
    """Run the REAL sample_on_batches (utils/evaluation_utils.py:190-333) on CPU with the tiny kernel
    model, a synthetic energy and recorded noise over three single-molecule batches."""
    from timewarp.utils import evaluation_utils as eu
    from timewarp.dataloader import DenseMolDynBatch

    V = 7
    g = torch.Generator().manual_seed(91)
    masses = torch.tensor([12.01, 1.008, 14.01, 16.0, 12.01, 1.008, 1.008])
    x_ref = torch.randn(1, V, 3, generator=g) * 0.3
    energy = SyntheticEnergy(x_ref.clone())
    batches = []
    for b in range(3):
        at = torch.randint(0, 5, (1, V), generator=g)
        x0 = x_ref + torch.randn(1, V, 3, generator=g) * 0.05
        v0 = torch.randn(1, V, 3, generator=g)
        y0 = x0 + torch.randn(1, V, 3, generator=g) * 0.05
        w0 = torch.randn(1, V, 3, generator=g)
        batches.append(DenseMolDynBatch(
            names=[f"tiny{b}"], atom_types=at, adj_list=torch.zeros((0, 2), dtype=torch.int64),
            edge_batch_idx=torch.zeros((0,), dtype=torch.int64), atom_coords=x0, atom_velocs=v0,
            atom_forces=torch.zeros_like(x0), atom_coord_targets=y0, atom_veloc_targets=w0,
            atom_force_targets=torch.zeros_like(x0), masked_elements=torch.zeros(1, V, dtype=torch.bool)))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model.coords_prior_log_scale.fill_(-3.0)
        model.velocs_prior_log_scale.fill_(0.0)
        for k, v in model.state_dict().items():
            if ".out_mlp._layers.2." in k:
                v.mul_(0.002)
    out = dict(masses=masses.numpy(), x_ref=x_ref.numpy())
    for b, bt in enumerate(batches):
        out[f"batch{b}/atom_types"] = bt.atom_types.numpy()
        out[f"batch{b}/x"] = bt.atom_coords.numpy()
        out[f"batch{b}/v"] = bt.atom_velocs.numpy()
        out[f"batch{b}/y"] = bt.atom_coord_targets.numpy()
        out[f"batch{b}/w"] = bt.atom_veloc_targets.numpy()
    out.update(np_sd(model.state_dict()))
    names = ("y_coords_model", "y_velocs_model", "traj_coords", "traj_velocs", "traj_coords_conditioning",
             "traj_velocs_conditioning", "ll_reverse", "ll_forward", "ll_reverse_training", "ll_forward_training",
             "acceptance")
    for tag, rv in (("fixedv", False), ("randv", True)):
        rec = {"normal": [], "randn_like": []}
        orig_rsample = torch.distributions.Normal.rsample
        orig_randn_like = torch.randn_like

        def rsample(self, shape=torch.Size()):
            r = orig_rsample(self, shape)
            rec["normal"].append(r.detach().numpy().copy().reshape(-1))
            return r

        def randn_like(t, **k):
            r = orig_randn_like(t, **k)
            rec["randn_like"].append(r.numpy().copy().reshape(-1))
            return r

        torch.distributions.Normal.rsample = rsample
        torch.randn_like = randn_like
        try:
            torch.manual_seed(777)
            res = eu.sample_on_batches(batches, model, torch.device("cpu"), energy, False, masses, random_velocs=rv)
        finally:
            torch.distributions.Normal.rsample = orig_rsample
            torch.randn_like = orig_randn_like
        for n, a in zip(names, res):
            out[f"{tag}/{n}"] = np.asarray(a)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros((0,), np.float32)
        out[f"{tag}/noise_normal"] = cat(rec["normal"])
        out[f"{tag}/noise_normal_sizes"] = np.array([len(a) for a in rec["normal"]])
        out[f"{tag}/noise_randn_like"] = cat(rec["randn_like"])
        print("sob", tag, {n: np.asarray(a).shape for n, a in zip(names, res)}, "mean p_acc", float(np.mean(res[-1])))
    model.load_state_dict(sd)
    np.savez_compressed(os.path.join(OUT, "sob_tiny.npz"), **out)


SOSC_NAMES = ("y_coords_model", "y_velocs_model", "traj_coords", "traj_velocs", "traj_coords_conditioning")


def gen_sosc_golden(model):
    """Run the REAL sample_on_single_conditional (utils/evaluation_utils.py:356-413) on CPU with the tiny kernel model and
    oracle/fake_sim.FakeSimulation as the OpenMM Simulation; store inputs, recorded noise and the five outputs."""
    from timewarp.utils import evaluation_utils as eu
    from timewarp.dataloader import DenseMolDynBatch
    from oracle.fake_sim import FakeSimulation

    V = 7
    g = torch.Generator().manual_seed(191)
    at = torch.randint(0, 5, (1, V), generator=g)
    x0 = torch.randn(1, V, 3, generator=g) * 0.3
    v0 = torch.randn(1, V, 3, generator=g)
    batch = DenseMolDynBatch(
        names=["tiny"], atom_types=at, adj_list=torch.zeros((0, 2), dtype=torch.int64),
        edge_batch_idx=torch.zeros((0,), dtype=torch.int64), atom_coords=x0, atom_velocs=v0,
        atom_forces=torch.zeros_like(x0), atom_coord_targets=x0, atom_veloc_targets=v0,
        atom_force_targets=torch.zeros_like(x0), masked_elements=torch.zeros(1, V, dtype=torch.bool))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model.coords_prior_log_scale.fill_(-3.0)
        model.velocs_prior_log_scale.fill_(0.0)
    out = dict(atom_types=at.numpy(), x0=x0.numpy(), v0=v0.numpy(), num_samples=np.array(5), step_width=np.array(3))
    out.update(np_sd(model.state_dict()))
    for tag, rv in (("fixedv", False), ("randv", True)):
        rec = {"normal": [], "randn_like": []}
        orig_rsample = torch.distributions.Normal.rsample
        orig_randn_like = torch.randn_like

        def rsample(self, shape=torch.Size()):
            r = orig_rsample(self, shape)
            rec["normal"].append(r.detach().numpy().copy().reshape(-1))
            return r

        def randn_like(t, **k):
            r = orig_randn_like(t, **k)
            rec["randn_like"].append(r.numpy().copy().reshape(-1))
            return r

        torch.distributions.Normal.rsample = rsample
        torch.randn_like = randn_like
        try:
            torch.manual_seed(555)
            res = eu.sample_on_single_conditional(batch, model, 5, FakeSimulation(allow_thermal=True), 3, rv, torch.device("cpu"))
        finally:
            torch.distributions.Normal.rsample = orig_rsample
            torch.randn_like = orig_randn_like
        for n, a in zip(SOSC_NAMES, res):
            out[f"{tag}/{n}"] = np.asarray(a)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros((0,), np.float32)
        out[f"{tag}/noise_normal"] = cat(rec["normal"])
        out[f"{tag}/noise_normal_sizes"] = np.array([len(a) for a in rec["normal"]])
        out[f"{tag}/noise_randn_like"] = cat(rec["randn_like"])
        print("sosc", tag, {n: np.asarray(a).shape for n, a in zip(SOSC_NAMES, res)})
    model.load_state_dict(sd)
    np.savez_compressed(os.path.join(OUT, "sosc_tiny.npz"), **out)


def gen_learnable_golden():
    """Tiny learnable-lengthscale model (attention_type "learnable_kernel") with DIFFERENT
    log_lengthscales in every attention layer: pins which layer's lengthscales a flow call uses."""
    torch.manual_seed(4711)
    m = kernel_model(emb=4, d_model=8, ff=16, mlp_hidden=[8], n_coupling=2, n_layers=2, lengthscales=[0.1, 0.5, 1.2],
                     attention_type="learnable_kernel")
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        m.coords_prior_log_scale.fill_(-0.3)
        m.velocs_prior_log_scale.fill_(0.2)
        for k, v in m.state_dict().items():
            if k.endswith("log_lengthscales"):
                v.add_(torch.randn(v.shape, generator=g) * 0.4)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 7, [7, 5, 6])
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(m, at, x_c, x_v, mask, y_c, y_v, 0, 0))
    d.update(np_sd(m.state_dict()))
    at1, x1, v1, m1, yc1, yv1 = padded_batch(g, 1, 7, [5])
    r = run_case(m, at1, x1, v1, m1, yc1, yv1, 4, 97)
    d.update({"b1_" + k: v for k, v in base_inputs(at1, x1, v1, m1, yc1, yv1).items()})
    d.update({"b1_" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "kernel_learnable_tiny.npz"), **d)
    print("kernel_learnable_tiny", {k: v.shape for k, v in d.items() if not k.startswith("sd::")})


def gen_chebyshev_golden():
    """Tiny chebyshev_kernel models (order 6, with and without force_asymptotic_zero), coefficients perturbed
    differently in every attention layer: pins the basis, the per-layer scores and the coefficient centring."""
    for tag, fz in (("kernel_cheb_tiny", False), ("kernel_cheb_zero_tiny", True)):
        torch.manual_seed(815)
        m = kernel_model(emb=4, d_model=8, ff=16, mlp_hidden=[8], n_coupling=2, n_layers=2, lengthscales=[0.1, 0.5, 1.2],
                         attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=fz)
        g = torch.Generator().manual_seed(23)
        with torch.no_grad():
            m.coords_prior_log_scale.fill_(-0.3)
            m.velocs_prior_log_scale.fill_(0.2)
            # the reference builds cheb_coeffs from an expand()ed (stride-0) tensor, which cannot be written in place:
            # give every attention module its own dense, perturbed coefficient matrix
            for name, p_ in list(m.named_parameters()):
                if name.endswith("cheb_coeffs"):
                    p_.data = p_.detach().clone() + torch.randn(p_.shape, generator=g) * 0.05
        at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 7, [7, 5, 6])
        d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
        d.update(run_case(m, at, x_c, x_v, mask, y_c, y_v, 0, 0))
        d.update(np_sd(m.state_dict()))
        at1, x1, v1, m1, yc1, yv1 = padded_batch(g, 1, 7, [5])
        r = run_case(m, at1, x1, v1, m1, yc1, yv1, 4, 96)
        d.update({"b1_" + k: v for k, v in base_inputs(at1, x1, v1, m1, yc1, yv1).items()})
        d.update({"b1_" + k: v for k, v in r.items()})
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **d)
        print(tag, "ok")


def gen_chebyshev_full_golden():
    """Full-size chebyshev_kernel model (order 6) on alanine dipeptide, name-seeded weights (cheb_coeffs included, so
    every attention layer has different coefficients): the case the fused kernels' per-layer score fragments need."""
    ad_x, ad_t = ad_topology()
    m = kernel_model(emb=32, d_model=128, ff=2048, mlp_hidden=[256], n_coupling=8, n_layers=3,
                     lengthscales=[0.1, 0.2, 0.5, 0.7, 1.0, 1.2], attention_type="chebyshev_kernel", cheb_order=6,
                     force_asymptotic_zero=True)
    for name, p_ in list(m.named_parameters()):  # expand()ed parameter -> dense, so load_state_dict can write it
        if name.endswith("cheb_coeffs"):
            p_.data = p_.detach().clone()
    m.load_state_dict(fo.synth_state_dict(m.state_dict(), base_seed=0))
    g = torch.Generator().manual_seed(12)
    x_c = ad_x[None].clone()
    x_v = torch.randn(1, 22, 3, generator=g) * 0.5
    mask = torch.zeros(1, 22, dtype=torch.bool)
    y_c = x_c + torch.randn(1, 22, 3, generator=g) * 0.01
    y_v = torch.randn(1, 22, 3, generator=g) * 0.5
    d = base_inputs(ad_t[None], x_c, x_v, mask, y_c, y_v)
    d.update(run_case(m, ad_t[None], x_c, x_v, mask, y_c, y_v, S_FULL, 2025))
    np.savez_compressed(os.path.join(OUT, "kernel_cheb_full_ad.npz"), **d)
    print("kernel_cheb_full_ad ok")


def gen_energy_kat():
    """(9) the reference's own known-answer data for the energy boundary, as data: the 40 frames of NNQQ that
    simulation/tests/test_md.py:35-83 checks OpenMM against (positions, E_pot/E_kin, forces, velocities) plus the atom /
    residue names of its state0 PDB (amber99sbildn + amber99_obc, preset T1-peptides = the alanine-dipeptide preset)."""
    base = "/root/reference/simulation/testdata/implicit-2olx-traj-cpu-"
    z = np.load(base + "arrays.npz")
    names, res, rid, els = [], [], [], []
    for line in open(base + "state0.pdb"):
        if line.startswith(("ATOM", "HETATM")):
            names.append(line[12:16].strip()); res.append(line[17:20].strip()); rid.append(int(line[22:26]))
            els.append(line[76:78].strip())
    np.savez_compressed(os.path.join(OUT, "energy_kat_2olx.npz"), positions=z["positions"], velocities=z["velocities"],
                        forces=z["forces"], energies=z["energies"], atom_names=np.array(names),
                        residue_names=np.array(res), residue_ids=np.array(rid, dtype=np.int32), elements=np.array(els))


def tiny_kernel_model():
    torch.manual_seed(1234)
    tiny = kernel_model(emb=4, d_model=8, ff=16, mlp_hidden=[8], n_coupling=2, n_layers=2,
                        lengthscales=[0.1, 0.5, 1.2])
    with torch.no_grad():
        tiny.coords_prior_log_scale.fill_(-0.3)
        tiny.velocs_prior_log_scale.fill_(0.2)
    return tiny


S_FULL = 64  # proposals per full-size case: 16 waves / 4 workgroups of the fused kernels per net


def gen_nonorm_golden():
    """Tiny kernel model built with normalise_kernel_values=False.  KernelAttention.forward never passes the flag on
    to compute_kernel_attention_scores (kernel_attention.py:197-206; its default is True, :75), so the reference
    L1-normalises regardless: these vectors pin that the build does the same."""
    torch.manual_seed(1234)
    m = kernel_model(emb=4, d_model=8, ff=16, mlp_hidden=[8], n_coupling=2, n_layers=2, lengthscales=[0.1, 0.5, 1.2],
                     normalise=False)
    with torch.no_grad():
        m.coords_prior_log_scale.fill_(-0.3)
        m.velocs_prior_log_scale.fill_(0.2)
    g = torch.Generator().manual_seed(7)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 7, [7, 5, 6])
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(m, at, x_c, x_v, mask, y_c, y_v, 0, 0))
    d.update(np_sd(m.state_dict()))
    at1, x1, v1, m1, yc1, yv1 = padded_batch(g, 1, 7, [5])
    r = run_case(m, at1, x1, v1, m1, yc1, yv1, 4, 99)
    d.update({"b1_" + k: v for k, v in base_inputs(at1, x1, v1, m1, yc1, yv1).items()})
    d.update({"b1_" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "kernel_nonorm_tiny.npz"), **d)
    print("kernel_nonorm_tiny ok")


def gen_tiny_kernel(tiny):
    # ---- (1) tiny kernel model, reference-initialised weights stored in full ------------------
    g = torch.Generator().manual_seed(7)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 7, [7, 5, 6])
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(tiny, at, x_c, x_v, mask, y_c, y_v, 0, 0))
    d.update(np_sd(tiny.state_dict()))
    # B=1 sampling case with padding (V=7, 5 real atoms)
    at1, x1, v1, m1, yc1, yv1 = padded_batch(g, 1, 7, [5])
    r = run_case(tiny, at1, x1, v1, m1, yc1, yv1, 4, 99)
    d.update({"b1_" + k: v for k, v in base_inputs(at1, x1, v1, m1, yc1, yv1).items()})
    d.update({"b1_" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "kernel_tiny.npz"), **d)
    print("kernel_tiny", {k: v.shape for k, v in d.items() if not k.startswith("sd::")})


def full_kernel_reference_model():
    return kernel_model(emb=32, d_model=128, ff=2048, mlp_hidden=[256], n_coupling=8, n_layers=3,
                        lengthscales=[0.1, 0.2, 0.5, 0.7, 1.0, 1.2])


def gen_full_kernel(full, ad_x, ad_t):
    # ---- (2) full-size kernel model on alanine dipeptide, name-seeded weights -------------------
    for tag, calibrated in (("kernel_full_ad", False), ("kernel_full_ad_calibrated", True)):
        sd = fo.synth_state_dict(full.state_dict(), base_seed=0, calibrated=calibrated)
        full.load_state_dict(sd)
        g = torch.Generator().manual_seed(11)
        x_c = ad_x[None].clone()
        x_v = torch.randn(1, 22, 3, generator=g) * 0.5
        mask = torch.zeros(1, 22, dtype=torch.bool)
        y_c = x_c + torch.randn(1, 22, 3, generator=g) * 0.01
        y_v = torch.randn(1, 22, 3, generator=g) * 0.5
        at = ad_t[None]
        S = S_FULL
        d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
        # layer trace of the first net evaluated in the reverse pass (chain[7].scale_transformer)
        trace = {}

        def saver(key):
            def hook(m, i, o):  # must return None: a returned value would replace the module output
                trace.setdefault(key, o[:2].detach().numpy().copy())
            return hook

        d.update(run_case(full, at, x_c, x_v, mask, y_c, y_v, S, 2024))
        if not calibrated:
            net = full.flow.chain[7].scale_transformer
            hooks = [net.in_mlp.register_forward_hook(saver("tr_in_mlp"))]
            for l in range(3):
                hooks.append(net.encoder_layers[l].register_forward_hook(saver(f"tr_enc{l}")))
            hooks.append(net.out_mlp.register_forward_hook(saver("tr_out_mlp")))
            torch.manual_seed(2024)
            with torch.no_grad():
                full.conditional_sample_with_logp(
                    atom_types=at, x_coords=x_c, x_velocs=x_v, adj_list=torch.zeros((0, 2), dtype=torch.int64),
                    edge_batch_idx=torch.zeros((0,), dtype=torch.int64), masked_elements=mask, num_samples=S)
            for h in hooks:
                h.remove()
            d.update(trace)
            com = x_c.mean(dim=1, keepdim=True)
            ls = torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])
            d["scores"] = compute_kernel_attention_scores(x_c - com, x_c - com, mask, ls).numpy()
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **d)
        print(tag, {k: v.shape for k, v in d.items()})


def gen_v60(full):
    # ---- (3) V=60: pins the cdist matmul branch (kernel_attention.py:98-102) --------------------
    sd = fo.synth_state_dict(full.state_dict(), base_seed=0)
    full.load_state_dict(sd)
    g = torch.Generator().manual_seed(60)
    V = 60
    x_c = torch.randn(1, V, 3, generator=g) * 0.45
    x_v = torch.randn(1, V, 3, generator=g) * 0.5
    at = torch.randint(0, 5, (1, V), generator=g)
    mask = torch.zeros(1, V, dtype=torch.bool)
    y_c = x_c + torch.randn(1, V, 3, generator=g) * 0.01
    y_v = torch.randn(1, V, 3, generator=g) * 0.5
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(full, at, x_c, x_v, mask, y_c, y_v, S_FULL, 606))
    com = x_c.mean(dim=1, keepdim=True)
    d["scores"] = compute_kernel_attention_scores(
        x_c - com, x_c - com, mask, torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])).numpy()
    np.savez_compressed(os.path.join(OUT, "kernel_full_v60.npz"), **d)
    print("kernel_full_v60", {k: v.shape for k, v in d.items()})


def gen_dense_tiny():
    # ---- (4a) dense softmax variant, tiny (stored weights, padding, RFF) --------------------------
    torch.manual_seed(4321)
    dt = dense_model(emb=4, d_model=8, ff=16, mlp_hidden=[8], n_coupling=2, n_layers=2, n_head=2,
                     rff=RFFPositionEncoderConfig(4, 1.0, 1.0))
    g = torch.Generator().manual_seed(8)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 7, [7, 5, 6])
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(dt, at, x_c, x_v, mask, y_c, y_v, 0, 0))
    d.update(np_sd(dt.state_dict()))
    at1, x1, v1, m1, yc1, yv1 = padded_batch(g, 1, 7, [5])
    r = run_case(dt, at1, x1, v1, m1, yc1, yv1, 4, 98)
    d.update({"b1_" + k: v for k, v in base_inputs(at1, x1, v1, m1, yc1, yv1).items()})
    d.update({"b1_" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "dense_tiny.npz"), **d)
    print("dense_tiny ok")


def gen_dense_full(ad_x, ad_t):
    # ---- (4b) dense softmax variant, full size on alanine dipeptide ------------------------------
    dfull = dense_model(emb=32, d_model=128, ff=2048, mlp_hidden=[256], n_coupling=8, n_layers=3, n_head=8)
    dfull.load_state_dict(fo.synth_state_dict(dfull.state_dict(), base_seed=0))
    g = torch.Generator().manual_seed(12)
    x_c = ad_x[None].clone()
    x_v = torch.randn(1, 22, 3, generator=g) * 0.5
    mask = torch.zeros(1, 22, dtype=torch.bool)
    y_c = x_c + torch.randn(1, 22, 3, generator=g) * 0.01
    y_v = torch.randn(1, 22, 3, generator=g) * 0.5
    d = base_inputs(ad_t[None], x_c, x_v, mask, y_c, y_v)
    d.update(run_case(dfull, ad_t[None], x_c, x_v, mask, y_c, y_v, S_FULL, 2025))
    np.savez_compressed(os.path.join(OUT, "dense_full_ad.npz"), **d)
    print("dense_full_ad ok")
    # a padded full-size batch (nn.MultiheadAttention's src_key_padding_mask, transformer_block.py:57-68): log-likelihood
    # of three molecules of 22 / 17 / 20 real atoms
    g = torch.Generator().manual_seed(13)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 3, 22, [22, 17, 20])
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(run_case(dfull, at, x_c, x_v, mask, y_c, y_v, 0, 0))
    np.savez_compressed(os.path.join(OUT, "dense_full_padded.npz"), **d)
    print("dense_full_padded ok")


def gen_dense_posenc_full(ad_x, ad_t):
    """transformer_nvp_posenc.yaml at full size (128 random Fourier features of the conditioning positions,
    rff_position_encoder.py:41-137) on alanine dipeptide.  The Gaussian vectors are buffers drawn when the reference
    builds the model (Gamma-distributed scales: some are large, so cos / sin see arguments of tens of radians); they are
    stored with the vectors, every other weight is regenerated from the name-seeded recipe on both sides."""
    torch.manual_seed(31)
    dfull = dense_model(emb=32, d_model=128, ff=2048, mlp_hidden=[256], n_coupling=8, n_layers=3, n_head=8,
                        rff=RFFPositionEncoderConfig(128, 1.0, 1.0))
    dfull.load_state_dict(fo.synth_state_dict(dfull.state_dict(), base_seed=0))
    g = torch.Generator().manual_seed(14)
    x_c = ad_x[None].clone()
    x_v = torch.randn(1, 22, 3, generator=g) * 0.5
    mask = torch.zeros(1, 22, dtype=torch.bool)
    y_c = x_c + torch.randn(1, 22, 3, generator=g) * 0.01
    y_v = torch.randn(1, 22, 3, generator=g) * 0.5
    d = base_inputs(ad_t[None], x_c, x_v, mask, y_c, y_v)
    d.update(run_case(dfull, ad_t[None], x_c, x_v, mask, y_c, y_v, 16, 2026))
    gv = {k: v for k, v in dfull.state_dict().items() if k.endswith("gaussian_vectors")}
    d.update(np_sd(gv))
    print("dense_posenc_full_ad ok; |G| max", max(float(v.abs().max()) for v in gv.values()))
    np.savez_compressed(os.path.join(OUT, "dense_posenc_full_ad.npz"), **d)


def gen_euler_maruyama():
    # ---- (5) EulerMaruyamaGaussian (cfg 1 plumbing) ----------------------------------------------
    em = model_constructor(ModelConfig(model_type="euler_maruyama_gaussian")).eval()
    g = torch.Generator().manual_seed(5)
    at, x_c, x_v, mask, y_c, y_v = padded_batch(g, 2, 9, [9, 6])
    f = torch.randn(2, 9, 3, generator=g) * 100.0
    with torch.no_grad():
        pc, pv = em._get_y_dist(atom_types=at, x_coords=x_c, x_velocs=x_v, x_forces=f)
        ll = em.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, x_forces=f, y_coords=y_c,
                               y_velocs=y_v, adj_list=None, edge_batch_idx=None, masked_elements=mask)
    d = base_inputs(at, x_c, x_v, mask, y_c, y_v)
    d.update(x_forces=f.numpy(), coord_mean=pc.loc.numpy(), coord_std=pc.scale.numpy(),
             veloc_mean=pv.loc.numpy(), veloc_std=pv.scale.numpy(), loglik=ll.numpy())
    d.update(np_sd(em.state_dict()))
    np.savez_compressed(os.path.join(OUT, "euler_maruyama.npz"), **d)
    print("euler_maruyama ok")


def gen_data_fixtures(ad_x, ad_t):
    # ---- (8) the reference's own smallest test molecule as data (testdata/smallest_molecule: 2 frames x 65 atoms,
    #          elements from PDB columns 77-78) for the config-0 plumbing test
    z = np.load("/root/reference/testdata/smallest_molecule/2olx-traj-arrays.npz")
    els = [l[76:78].strip() for l in open("/root/reference/testdata/smallest_molecule/2olx-traj-state0.pdb")
           if l.startswith(("ATOM", "HETATM"))]
    np.savez_compressed(os.path.join(OUT, "smallest_molecule.npz"), positions=z["positions"], velocities=z["velocities"],
                        forces=z["forces"], elements=np.array(els))
    # ---- (6) alanine-dipeptide topology as data (22 atoms) ---------------------------------------
    np.savez_compressed(os.path.join(OUT, "ad_topology.npz"), coords_nm=ad_x.numpy(), atom_types=ad_t.numpy(),
                        atom_names=np.array(AD_NAMES))


def main():
    """python oracle/gen_golden.py [--only NAME ...]; NAME in SECTIONS; no flag = everything."""
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ad_x, ad_t = ad_topology()
    cache = {}

    def full():
        if "full" not in cache:
            cache["full"] = full_kernel_reference_model()
        return cache["full"]

    sections = {
        "tiny": lambda: gen_tiny_kernel(tiny_kernel_model()),
        "nonorm": gen_nonorm_golden,
        "full": lambda: gen_full_kernel(full(), ad_x, ad_t),
        "v60": lambda: gen_v60(full()),
        "dense-tiny": gen_dense_tiny,
        "dense-full": lambda: gen_dense_full(ad_x, ad_t),
        "dense-posenc": lambda: gen_dense_posenc_full(ad_x, ad_t),
        "em": gen_euler_maruyama,
        "mh": lambda: gen_mh_goldens(tiny_kernel_model()),  # (7) the MH loop itself, driven with a synthetic energy
        "mh-omm": lambda: gen_mh_goldens(tiny_kernel_model(), MH_OPENMM_SCENARIOS, "mh_tiny_openmm.npz"),  # (7b) with OpenMM steps
        "sob": lambda: gen_sob_golden(tiny_kernel_model()),
        "sosc": lambda: gen_sosc_golden(tiny_kernel_model()),  # sample_on_single_conditional with the fake Simulation
        "learnable": gen_learnable_golden,
        "cheb": gen_chebyshev_golden,
        "cheb-full": gen_chebyshev_full_golden,
        "energy-kat": gen_energy_kat,
        "data": lambda: gen_data_fixtures(ad_x, ad_t),
    }
    only = [a for a in sys.argv[1:] if a in sections]
    for name, fn in sections.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
