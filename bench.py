#!/usr/bin/env python
"""Benchmark of the Timewarp sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one Metropolis-Hastings iteration of `kernel_transformer_nvp` on alanine dipeptide
with 1000 parallel proposals (BASELINE.json configs[1]): 1000 proposals + log p(y|x) through the
8-layer flow (reverse pass), potential and kinetic energies, log p(x|y) (forward pass), accept scan.
Inputs are synthetic (no checkpoint/trajectory exists offline): the PDB geometry of
simulation/testdata/alanine-dipeptide.pdb, name-seeded random weights with the SURVEY section 8d
calibration so the acceptance path is non-degenerate, device-generated noise.  Everything is
resident in HBM before the timed region.

N > 1: one independent chain per rank/GPU (no data-path collective), one all-gather of the
trajectories at collection time (inside the timed region), `value` = accepted samples of all
ranks / max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S_PROPOSALS = 1000
V_ATOMS = 22
FLOP_PER_SAMPLE_PASS = 1.612e9  # 16 * V * F_blk(V), V = 22
N_COUPLING = 8
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (~2.5 PF)
# Fallback for roofline.power_bound only: the clock a bare MFMA stream sustains on all 256 CUs is measured by every run
# (tw_probe_mfma_clock, 1.9-2.05 GHz on the boxes seen so far); this constant is used if that call fails.
SUSTAINED_MFMA_CLOCK_GHZ = 1.74


def h3_mfma_per_wave(dims) -> int:
    """MFMA instructions one wave of the 48-token split-fp16 kernel issues per launch - a static property of the generated
    statements, counted from their structure for the model's dimensions (three MFMAs per 16x16x32 product tile and token tile,
    NT = 3): in-MLP 108 per 32-unit chunk, per encoder layer 4 k-steps x (24 windowed mixing + 72 GEMM) per head, 144 per FFN
    chunk and 48 for the transposer, out-MLP 81 per chunk.  36 216 for kernel_transformer_nvp - SQ_INSTS_MFMA reads 36.2 k and
    SQ_VALU_MFMA_BUSY_CYCLES 579.5 k = 16 clocks each (profiles/r05_ad_sq_counters.md)."""
    hid, ff = dims.d_hidden // 32, dims.d_ff // 32
    per_layer = dims.n_heads * 4 * (24 + 72) + 144 * ff + 48
    return 108 * hid + dims.n_layers * per_layer + 81 * hid
# execution paths of the flow (include/timewarp_hip.h): h3 and f32 hold the 1e-5 parity bar.  `kernel` is only the fallback
# name (<NT, ASM, DENSE, WIDE, RFF, ENC, H1, NG6>): a line reports the instantiation tw_last_netblock_kernel names
PATHS = {
    "h3": dict(path=3, dtype="f16x3 (split-fp16 operands, 3 MFMAs per fp32 product, fp32 accumulate)",
               kernel="tw::netblock_h3_kernel<3, true, false, false, false, true, false, false>", peak=F16_MFMA_PEAK_TFLOPS, mfma_per_product=3),
    "f32": dict(path=1, dtype="f32", kernel="tw::netblock_kernel<3>", peak=F32_MFMA_PEAK_TFLOPS, mfma_per_product=1),
    # opt-in fast mode, NOT a parity path and never the headline: fp16 operands (11 significand bits), one MFMA per product
    "h1": dict(path=4, dtype="f16 (fp16 operands, ONE MFMA per product, fp32 accumulate; ~1e-4 relative deviation from the "
                             "reference's fp32 arithmetic - tests/test_flow_h1_gpu.py; opt-in fast mode, not a parity path)",
               kernel="tw::netblock_h3_kernel<3, true, false, false, false, true, true, false>", peak=F16_MFMA_PEAK_TFLOPS, mfma_per_product=1),
}
# Synthetic-weight calibration (SURVEY section 8d idea, tuned so acceptance is non-degenerate against
# the stiff bonded terms): identity flow (last out_mlp layer zeroed), coordinate prior std e^-7 nm,
# velocity prior std 1 with isotropic resampled velocities (the reference's --random-velocities
# --resample-velocities mode), for which the velocity terms of the MH exponent cancel exactly.
CALIBRATION = dict(coords_log_scale=-7.0, velocs_log_scale=0.0)
MH_MODE = dict(accept=True, random_velocs=True, resample_velocs=True)

# --config: the default "ad" is BASELINE.json configs[1] (the headline, what the driver runs); "4aa" and "dense" are
# configs[3] and configs[4], benched by hand for their roofline lines (profiles/r04_bench_{4aa,dense}.json).
F_BLK_KERNEL = lambda V: 4478976 + 4608 * V     # FLOP per token per net-block, kernel attention
F_BLK_DENSE = 3726336                           # dense softmax variant, V = 22
CONFIGS = {
    "ad": dict(V=22, S=1000, model="kernel", flop_sample_pass=FLOP_PER_SAMPLE_PASS, calibration=CALIBRATION,
               workload="kernel_transformer_nvp.yaml, alanine-dipeptide (22 atoms), 1000-proposal parallel MH, "
                        "1 chain per GPU (BASELINE.json configs[1]; configs[2] when n_gpus=8)"),
    "4aa": dict(V=61, S=512, model="kernel", flop_sample_pass=16 * 61 * F_BLK_KERNEL(61),
                calibration=dict(coords_log_scale=-7.5, velocs_log_scale=0.0),
                kernel_note="64-token waves: one molecule per wave, 128 workgroups per net = one round of the chip; both coupling "
                            "nets of one coupling layer, all proposals",
                workload="kernel_transformer_nvp.yaml, 4AA tetrapeptide NAQQ (61 atoms: the reference's OpenMM test peptide NNQQ, "
                         "simulation/testdata/implicit-2olx-*, with its second asparagine cut back to alanine - CG becomes HB1 at "
                         "1.09 A - so that the molecule has BASELINE's '~60 atoms'; amber99sb-ildn + OBC tables pinned by the "
                         "reference's two OpenMM files), 512-proposal parallel MH, 1 chain per GPU (BASELINE.json configs[3]; "
                         "SURVEY 8d's 4.675 TFLOP per iteration is the same formula at V = 60)"),
    "4aa-nnqq": dict(V=65, S=512, model="kernel", flop_sample_pass=16 * 65 * F_BLK_KERNEL(65),
                     calibration=dict(coords_log_scale=-7.5, velocs_log_scale=0.0),
                     kernel_note="wide layout: 2 molecules per workgroup at a slot stride of 96, three-group key windows; both "
                                 "coupling nets of one coupling layer, all proposals",
                     workload="kernel_transformer_nvp.yaml, 4AA tetrapeptide NNQQ (65 atoms: the reference's own OpenMM test molecule, "
                              "one atom above what a 64-token wave holds: the wide layout), 512-proposal parallel MH, 1 chain per GPU"),
    "dense": dict(V=22, S=1000, model="dense", flop_sample_pass=16 * 22 * F_BLK_DENSE, calibration=CALIBRATION,
                  kernel_note="dense-softmax kernel; with --path h1 the MLP sections are single-MFMA, the softmax attention "
                              "block stays split-fp16; both coupling nets of one coupling layer, all proposals",
                  workload="transformer_nvp.yaml (dense softmax attention variant), alanine-dipeptide (22 atoms), 1000-proposal "
                           "parallel MH, 1 chain per GPU (BASELINE.json configs[4])"),
    "1hgv": dict(V=691, S=16, model="kernel", flop_sample_pass=16 * 691 * F_BLK_KERNEL(691),
                 calibration=dict(coords_log_scale=-9.5, velocs_log_scale=0.0),
                 workload="kernel_transformer_nvp.yaml, the reference's 691-atom test protein (testdata/output/1hgv-traj-state0.pdb: topology "
                          "and a frame from the committed known-answer fixture, amber99sb-ildn + OBC tables pinned on its OpenMM energies and "
                          "forces), 16-proposal parallel MH, 1 chain per GPU - above every fused layout: per-op launches on split-fp16 MFMAs"),
}


def molecule(config):
    """(name, atom types, coordinates nm, masses, energy) of a bench configuration."""
    from timewarp_amd import synthetic
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    if CONFIGS[config]["V"] == 22:
        types, coords, masses = synthetic.alanine_dipeptide_state()
        return "alanine-dipeptide", types, coords, masses, AmberPotentialEnergyTorch.alanine_dipeptide()
    # NNQQ: topology, a frame of coordinates and the elements from the reference's known-answer file (data fixture)
    import numpy as np
    from timewarp_amd.forcefield import ELEMENT_MASSES, amber99sbildn_obc_tables

    if config == "1hgv":
        from timewarp_amd.dataloader import elements_from_atom_names

        z = np.load(os.path.join(ROOT, "tests", "golden", "energy_kat_1hgv.npz"))
        names = [str(n) for n in z["atom_names"]]
        tables = amber99sbildn_obc_tables(names, [str(r) for r in z["residue_names"]], [int(i) for i in z["residue_ids"]],
                                          improper_neighbour_order="pyset")
        masses = torch.tensor([ELEMENT_MASSES[next(ch for ch in n if ch.isalpha())] for n in names], dtype=torch.float32)
        assert len(names) == CONFIGS[config]["V"]
        return "1hgv", elements_from_atom_names(names), torch.from_numpy(z["positions"][0].astype(np.float32)), masses, AmberPotentialEnergyTorch(tables)

    z = np.load(os.path.join(ROOT, "tests", "golden", "energy_kat_2olx.npz"))
    names, res, rid = [str(n) for n in z["atom_names"]], [str(r) for r in z["residue_names"]], [int(i) for i in z["residue_ids"]]
    els = [str(e) for e in z["elements"]]
    pos = z["positions"][0].astype(np.float64)
    label = "NNQQ"
    if config == "4aa":
        # NAQQ: the second asparagine cut back to alanine (CG -> HB1 on the CB-CG axis at 0.109 nm; OD1 ND2 HD21 HD22 dropped)
        at = {n: i for i, (n, r) in enumerate(zip(names, rid)) if r == 2}
        cb, cg = at["CB"], at["CG"]
        pos[cg] = pos[cb] + 0.109 * (pos[cg] - pos[cb]) / np.linalg.norm(pos[cg] - pos[cb])
        names[cg], els[cg] = "HB1", "H"
        drop = {at[n] for n in ("OD1", "ND2", "HD21", "HD22")}
        keep = [i for i in range(len(names)) if i not in drop]
        names, rid, els = [names[i] for i in keep], [rid[i] for i in keep], [els[i] for i in keep]
        res = ["ALA" if r == 2 else res[i] for i, r in zip(keep, rid)]
        pos = pos[keep]
        label = "NAQQ"
    tables = amber99sbildn_obc_tables(names, res, rid)
    vocab = {"C": 0, "H": 1, "N": 2, "O": 3, "S": 4}
    types = torch.tensor([vocab[e] for e in els])
    masses = torch.tensor([ELEMENT_MASSES[e] for e in els], dtype=torch.float32)
    assert len(names) == CONFIGS[config]["V"]
    return label, types, torch.from_numpy(pos).float(), masses, AmberPotentialEnergyTorch(tables)


def build_chain(device, seed, proposals, path, config="ad", n_chains=1):
    import timewarp_amd as tw
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain
    from timewarp_amd.utils.multichain import MetropolisHastingsChains

    cfg = CONFIGS[config]
    model = tw.model_constructor(synthetic.transformer_nvp_config() if cfg["model"] == "dense" else synthetic.kernel_transformer_nvp_config())
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), base_seed=0, calibrated=True, **cfg["calibration"]))
    model.execution_path = path
    model = model.to(device).eval()
    name, types, coords, masses, energy = molecule(config)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    batch = single_state_batch(name, types, coords, torch.zeros(cfg["V"], 3))
    if n_chains > 1:
        # `proposals` rows per launch shared by the chains: S = proposals // C proposals per chain and iteration
        chain = LockStepChains(MetropolisHastingsChains([batch] * n_chains, model, device, energy, masses, proposals // n_chains,
                                                        random_velocs=MH_MODE["random_velocs"], resample_velocs=MH_MODE["resample_velocs"]))
        prewarm(model, types, coords, device, (proposals // n_chains) * n_chains)
        return chain, model
    chain = MetropolisHastingsChain(batch, model, device, energy, masses, num_proposal_steps=proposals, **MH_MODE)
    prewarm(model, types, coords, device, proposals)
    return chain, model


class LockStepChains:
    """--chains C (utils/multichain.py): C independent chains of the same molecule evaluated in lock-step - C x S
    rows per flow launch, one accept scan per chain - behind the interface the timed loop drives a single chain through."""

    def __init__(self, chains):
        self.inner = chains

    def step_deferred(self):
        self.inner.step_deferred()

    def flush(self, lag=False):
        # lag: start copying this window's accept results, book the window before it (multichain.py)
        self.inner.flush(lag=lag)

    def _settled(self):
        self.inner.flush()   # everything queued so far is booked before a counter is read
        return self.inner

    accepted = property(lambda self: sum(self._settled().accepted))
    proposals = property(lambda self: self.inner.proposals)
    chain_c = property(lambda self: [t for per_chain in self._settled().chain_c for t in per_chain])

    def trajectory(self):
        self.inner.flush()
        return torch.cat(self.chain_c, dim=0), None


PREWARM_PASSES = 40  # ~0.15 s


def prewarm(model, types, coords, device, proposals):
    """Part of set-up, before the W warm-up steps: a fixed number of flow passes on throw-away inputs.  A fresh process
    starts with the GPU at idle clocks and the packed weights not yet resident in the Infinity Cache: the first ~80
    launches of the dominant kernel run 465-560 us before settling at ~415 us (per-dispatch trace of this bench under
    rocprofv3, profiles/README.md).  The chain's state and random streams are not touched."""
    with torch.no_grad(), model.deferred_range_check():  # no range-flag read-back (a device sync) per pass; the chain looks at its flushes
        V = coords.shape[0]
        at = types[None].to(device)
        xc = coords[None].to(device)
        xv = torch.zeros(1, V, 3, device=device)
        mk = torch.zeros(1, V, dtype=torch.bool, device=device)
        z = torch.zeros(proposals, 1, V, 3, device=device)
        for _ in range(PREWARM_PASSES):
            model.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                               masked_elements=mk, num_samples=proposals, z_coords=z, z_velocs=z)
    torch.cuda.synchronize(device)


def committed_traffic(kernel: str, bench_args: str):
    """HBM / fabric bytes per launch from the newest committed rocprofv3 --pmc summary (profiles/r*_pmc_traffic.json, written by
    tools/pmc_traffic.sh + tools/summarize_profiles.py --traffic) that measured EXACTLY this kernel instantiation under
    exactly this bench configuration - else (None, None).  r04 read whatever the newest file held for the kernel family; a
    rebuilt kernel then carried its predecessor's traffic."""
    import glob

    want = kernel.split(" (")[0].strip()
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(pmc) as f:
            recs = json.load(f)
        for rec in recs.values():
            if not isinstance(rec, dict) or "kernel" not in rec:
                continue
            name = rec["kernel"].replace("void ", "").split("(tw::")[0].strip()
            if name == want and rec.get("bench_args") == bench_args:
                return rec["traffic_bytes_per_launch_corrected"], (
                    f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py "
                    f"{bench_args}`: this instantiation, this configuration)")
    return None, None


def live_kernel(lib) -> str:
    """The net-block instantiation the last flow call of this thread launched (tw_last_netblock_kernel, ABI 7)."""
    return lib.tw_last_netblock_kernel().decode()


def attention_block(model, device, proposals, avg_launch_ms, path_name="h3"):
    """The north-star's second figure: algorithmic FLOP rate of the attention block (values_proj + A.V + out_proj,
    SURVEY 8d) against the bf16/f16 dense MFMA peak.  Measured on one extra, untimed launch of the dominant kernel with
    its s_memtime section stamps switched on (tw_debug_set_flags bit 4; wave 0 of workgroup 0 stamps the shader clock at
    the section boundaries): share of the launch spent in the attention sections x the live average launch time."""
    from timewarp_amd import _lib, synthetic

    types, coords, _ = synthetic.alanine_dipeptide_state()
    g = torch.Generator().manual_seed(5)
    at = types[None].to(device)
    xc = (coords - coords.mean(0, keepdim=True))[None].to(device)
    xv = (torch.randn(1, V_ATOMS, 3, generator=g) * 0.5).to(device)
    zo = (torch.randn(proposals, V_ATOMS, 3, generator=g) * 0.5).to(device)
    mask = torch.zeros(1, V_ATOMS, dtype=torch.bool, device=device)
    lib = _lib.load()
    # bit 4: section stamps; bit 13: from the encoder-stack statement - the PRODUCT build, whose stamps are compiled in (r05:
    # in every statement, five scalar compare-and-branch pairs per layer when off) - not from the per-section build
    lib.tw_debug_set_flags(16 | 8192)
    try:
        for _ in range(2):
            acts, _ = model.debug_netblock(0, 0, at, xc, xv, mask, zo, PATHS[path_name]["path"])
        torch.cuda.synchronize()
        stamped_kernel = live_kernel(lib)
    finally:
        lib.tw_debug_set_flags(0)
    n_layers = model.dims.n_layers
    ts = acts.reshape(-1)[:128].contiguous().view(torch.int64).cpu().tolist()
    # stamps: 0 start, 1 in_mlp, 2 + 4 l + 3 end of layer l, out_mlp; 60: top of the kernel; inside the statement
    # 40 + 4 l + {0, 1} = start / end of layer l's attention block (side-block DMA + all heads: mixing and the folded
    # out_proj . values_proj GEMM)
    total = ts[2 + 4 * n_layers] - (ts[60] or ts[0])
    att = sum(ts[40 + 4 * l + 1] - ts[40 + 4 * l] for l in range(n_layers))
    share = att / total
    d = model.dims
    flop_token_layer = 2 * d.d_model * d.n_heads * d.d_model * 2 + 2 * d.n_heads * V_ATOMS * d.d_model
    flop_launch = flop_token_layer * n_layers * 2 * proposals * V_ATOMS  # both coupling nets of one coupling layer
    achieved = flop_launch / (share * avg_launch_ms * 1e-3) / 1e12
    return {
        "what": "values_proj + A.V + out_proj of all encoder layers (SURVEY 8d), algorithmic FLOPs / time in the attention "
                "sections of the dominant kernel",
        "achieved": achieved, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F16_MFMA_PEAK_TFLOPS,
        "share_of_launch": share, "algorithmic_flop_per_launch": float(flop_launch),
        "attention_cycles_per_launch": att, "stamped_cycles_per_launch": total,
        # shader clock the launch actually ran at: stamped cycles of the launch / live average launch time (the per-section
        # build's cycle count for h3, the product build's own for h1)
        "effective_clock_ghz": total / (avg_launch_ms * 1e6),
        "executed_over_algorithmic": ("0.75 (one fp16 MFMA per product x 1/2 from folding out_proj into values_proj per head + the "
                                      "block-diagonal mixing on K=32+16 MFMAs)" if path_name == "h1" else
                                      "2.26 (3-term fp16 split x 1/2 from folding out_proj into values_proj per head "
                                      "+ the K=48 block-diagonal mixing on K=32+16 MFMAs)"),
        "method": "s_memtime section stamps of one untimed launch of the timed build itself (tw_debug_set_flags 16 | 8192; the "
                  "stamps are compiled into the encoder-stack statement) x live average launch time",
        "stamped_kernel": stamped_kernel,
    }


def cpu_baseline(proposals):
    """The oracle (oracle/flow_oracle.py + oracle/mh_oracle.py + oracle/energy_oracle.c) on this
    box's host cores: a thread-count sweep on a tenth of the workload, then 3 full 1000-proposal MH iterations at the
    fastest setting (>= 3 timed iterations, thread count stated, n = 1 reported)."""
    from oracle import flow_oracle as fo
    from oracle import mh_oracle as mo
    from timewarp_amd import synthetic
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb
    import numpy as np
    import subprocess

    so = os.path.join(ROOT, "oracle", "_build", "libenergy_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    tables = alanine_dipeptide_amber99sb()

    class FF(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("n_atoms", "n_bonds", "n_angles", "n_torsions", "n_exceptions", "has_gbsa")] + \
                   [(n, C.c_double) for n in ("cutoff", "rf_dielectric", "solute_dielectric", "solvent_dielectric", "surface_area_energy")] + \
                   [(n, C.c_void_p) for n in ("bond_idx", "bond_par", "angle_idx", "angle_par", "torsion_idx", "torsion_par", "exc_idx", "exc_par", "atom_par")]

    arrs = [np.ascontiguousarray(a) for a in (
        tables.bond_idx.astype(np.int32), tables.bond_par, tables.angle_idx.astype(np.int32), tables.angle_par,
        tables.torsion_idx.astype(np.int32), tables.torsion_par, tables.exc_idx.astype(np.int32), tables.exc_par, tables.atom_par)]
    ff = FF(tables.n_atoms, len(arrs[0]), len(arrs[2]), len(arrs[4]), len(arrs[6]), int(tables.has_gbsa), tables.cutoff,
            tables.rf_dielectric, tables.solute_dielectric, tables.solvent_dielectric, tables.surface_area_energy,
            *[a.ctypes.data for a in arrs])

    energy_seconds = [0.0]

    class CEnergy:
        kbT = 8.314462618e-3 * 310.0

        def __call__(self, coords):
            t_e = time.perf_counter()
            x = np.ascontiguousarray(coords.reshape(-1, V_ATOMS, 3).numpy(), dtype=np.float32)
            out = np.zeros(x.shape[0])
            lib.oracle_amber_energy(C.byref(ff), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), None,
                                    C.c_int64(x.shape[0]))
            energy_seconds[0] += time.perf_counter() - t_e
            return torch.from_numpy(out).to(torch.float32)[:, None]

    class Noise:
        def __init__(self):
            self.g = torch.Generator().manual_seed(0)

        def randn_like(self, t):
            return torch.randn(t.shape, generator=self.g)

        def latents(self, S, B, V, sc, sv):
            return torch.randn((S, B, V, 3), generator=self.g) * sc, torch.randn((S, B, V, 3), generator=self.g) * sv

        def uniform(self, S):
            return torch.rand(S, generator=self.g)

    spec = fo.FlowSpec(variant="kernel")
    sd = fo.synth_state_dict(fo.make_template(spec), 0, calibrated=True, **CALIBRATION)
    model = mo.OracleModel(sd, spec)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    velocs = torch.zeros(V_ATOMS, 3)
    mask = torch.zeros(1, V_ATOMS, dtype=torch.bool)

    class ChunkedModel(mo.OracleModel):
        """The same two model calls evaluated `chunk` rows at a time.  The reference hands torch all 1000 rows at once
        (flow.py:284-296 tiles the conditioning state over the samples), which on the host leaves the 22 000 x 2048 FFN
        activations to stream through DRAM; row blocks that stay in cache are a fairer statement of what the CPU can do."""

        def __init__(self, sd, spec, chunk):
            super().__init__(sd, spec)
            self.chunk = chunk

        def conditional_sample_with_logp(self, atom_types, x_coords, x_velocs, masked, z_coords, z_velocs):
            outs = [fo.conditional_sample_with_logp(self.sd, self.spec, atom_types, x_coords, x_velocs, masked,
                                                    z_coords[i:i + self.chunk], z_velocs[i:i + self.chunk])
                    for i in range(0, z_coords.shape[0], self.chunk)]
            return tuple(torch.cat(t, dim=0) for t in zip(*outs))

        def log_likelihood(self, atom_types, x_coords, x_velocs, y_coords, y_velocs, masked):
            c = self.chunk
            return torch.cat([fo.log_likelihood(self.sd, self.spec, atom_types[i:i + c], x_coords[i:i + c], x_velocs[i:i + c],
                                                y_coords[i:i + c], y_velocs[i:i + c], masked[i:i + c])
                              for i in range(0, x_coords.shape[0], c)], dim=0)

    def run(n_iter, S, threads, mdl=None):
        """n_iter MH iterations of S proposals on `threads` torch threads; (seconds, accepted, energy seconds)."""
        mdl = mdl or model
        torch.set_num_threads(threads)
        kw = dict(num_proposal_steps=S, **MH_MODE)
        # warm-up at this thread count (thread pool, allocator)
        mo.sample_with_model(types[None], coords[None], velocs[None], mask, mdl, CEnergy(), masses, 1, Noise(),
                             num_proposal_steps=16, **MH_MODE)
        energy_seconds[0] = 0.0
        noise = Noise()  # one stream for the whole run
        accepted, t0 = 0, time.perf_counter()
        x_c, x_v = coords[None], velocs[None]
        for _ in range(n_iter):
            c, v, acc, _ = mo.sample_with_model(types[None], x_c, x_v, mask, mdl, CEnergy(), masses, 1, noise, **kw)
            x_c, x_v = torch.from_numpy(c[-1:]), torch.from_numpy(v[-1:])
            accepted += acc
        return time.perf_counter() - t0, accepted, energy_seconds[0]

    # thread count stated, n = 1 reported, >= 3 timed iterations.  Sweep torch's intra-op threads on a
    # 100-proposal iteration (a tenth of the workload, ~1-4 s per setting), then time full iterations at the best one.
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    candidates = sorted({n for n in (1, 8, 16, 32, 64, ncpu) if n <= ncpu})
    sweep_S = max(1, proposals // 10)
    sweep, skipped = {}, []
    for n in candidates:
        # past the optimum the rate only falls (oversubscribed host threads: 256 threads ran at 0.4 proposals/s, four
        # minutes for this one setting): stop once a setting is more than 1.5x slower than the best seen
        if sweep and sweep[max(sweep)]["proposals_per_s"] * 1.5 < max(v["proposals_per_s"] for v in sweep.values()):
            skipped.append(n)
            continue
        sec, _, esec = run(1, sweep_S, n)
        sweep[n] = {"model_s": sec - esec, "proposals_per_s": sweep_S / sec}
    best = max(sweep, key=lambda n: sweep[n]["proposals_per_s"])
    # (a) what the reference does: every model call on all `proposals` rows at once.  The thread count that wins on 100
    # rows need not win on 1000, so the full-size iteration is also tried at up to 64 threads (one iteration each).
    full = {}
    for n in sorted({best, min(64, ncpu)}):
        sec, acc, esec = run(1, proposals, n)
        full[n] = dict(seconds=sec, accepted=acc, energy_s=esec)
    best_full = min(full, key=lambda n: full[n]["seconds"])
    iters = 2
    elapsed, accepted, esec = run(iters, proposals, best_full)
    # (b) the same iterations with the model calls in blocks of `sweep_S` rows (results identical row for row)
    c_iters = 3
    c_elapsed, c_accepted, c_esec = run(c_iters, proposals, best, ChunkedModel(sd, spec, sweep_S))
    torch.set_num_threads(default_threads)
    n1 = sweep[1]
    whole = {"value": accepted / elapsed, "proposals_per_s": iters * proposals / elapsed, "s_per_iteration": elapsed / iters,
             "threads": best_full, "iterations": iters,
             "one_iteration_s_by_threads": {str(n): round(v["seconds"], 2) for n, v in full.items()}}
    chunked = {"value": c_accepted / c_elapsed, "proposals_per_s": c_iters * proposals / c_elapsed,
               "s_per_iteration": c_elapsed / c_iters, "threads": best, "iterations": c_iters, "rows_per_model_call": sweep_S}
    use_chunked = chunked["value"] > whole["value"]
    top, t_el, t_es, t_it = (chunked, c_elapsed, c_esec, c_iters) if use_chunked else (whole, elapsed, esec, iters)
    return {
        # the better of the two host schedules is the stated baseline; both are reported
        "value": top["value"],
        "unit": "MH-accepted samples/s",
        "cores": top["threads"],
        "kind": "port",
        "sample": f"{t_it} full MH iterations of {proposals} proposals (flow reverse+forward via oracle/flow_oracle.py on torch-CPU "
                  f"fp32 with {top['threads']} threads, model calls on "
                  f"{'blocks of %d rows' % sweep_S if use_chunked else 'all rows at once, as the reference does'}; energies via "
                  f"oracle/energy_oracle.c on 1 core; accept scan) in {t_el:.2f} s; the other schedule is under "
                  f"'{'whole_batch' if use_chunked else 'chunked'}'",
        "proposals_per_s": top["proposals_per_s"],
        "s_per_iteration": top["s_per_iteration"],
        # model part and energy part separately (the energy stand-in for OpenMM is the scalar C oracle)
        "energy_s_per_iteration": t_es / t_it,
        "model_s_per_iteration": (t_el - t_es) / t_it,
        "model_threads": top["threads"], "energy_threads": 1, "host_logical_cpus": ncpu,
        "whole_batch": whole,   # one 1000-row model call per pass: the reference's schedule
        "chunked": chunked,     # 10 x 100-row model calls per pass
        "thread_sweep_proposals_per_s": {str(n): round(v["proposals_per_s"], 2) for n, v in sweep.items()},
        "thread_sweep_skipped": skipped,
        "single_thread": {"cores": 1, "proposals_per_s": n1["proposals_per_s"],
                          "sample": f"one {sweep_S}-proposal MH iteration on 1 torch thread",
                          "s_per_1000_proposal_iteration_extrapolated": proposals / n1["proposals_per_s"]},
    }


def alt_path_record(device, seed, proposals, steps, sync_every, name="f32"):
    """The same workload on another kernel family, timed after the main region on a second chain so the driver's record
    carries them all: "f32" = the exact-f32 fused kernel (TW_EXECUTION_PATH=f32; the C ABI's TW_PATH_AUTO); "h1" = the
    opt-in single-MFMA fast mode (TW_EXECUTION_PATH=h1), with the north star's attention-block figure."""
    from timewarp_amd import _lib

    pinfo = PATHS[name]
    lib = _lib.load()
    chain, model = build_chain(device, seed, proposals, _lib.TW_PATH_AUTO if name == "f32" else pinfo["path"])
    with torch.no_grad():
        chain.step_deferred()
        chain.flush()
    torch.cuda.synchronize()
    acc0 = chain.accepted
    lib.tw_profile_begin()
    t0 = time.perf_counter()
    with torch.no_grad():
        for it in range(steps):
            chain.step_deferred()
            if (it + 1) % sync_every == 0:
                chain.flush(lag=True)
        chain.trajectory()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = C.c_double(0.0), C.c_int64(0)
    lib.tw_profile_end(C.byref(k_ms), C.byref(k_launches))
    avg_ms = k_ms.value / max(int(k_launches.value), 1)
    achieved = FLOP_PER_SAMPLE_PASS * proposals / N_COUPLING / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    kernel = live_kernel(lib) or pinfo["kernel"]
    rec = {
        "execution_path": {"f32": "f32 (exact-f32 fused kernel: TW_EXECUTION_PATH=f32 / the C ABI's TW_PATH_AUTO; model_constructor's default is h3)",
                           "h1": "h1 (opt-in fast mode: TW_EXECUTION_PATH=h1 / TW_PATH_FUSED_H1; one fp16 MFMA per product; NOT a "
                                 "parity path - never the default, never the headline)"}[name],
        "dtype": pinfo["dtype"], "value": (chain.accepted - acc0) / elapsed, "unit": "MH-accepted samples/s",
        "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "range_guard_fired": bool(getattr(model, "demoted", False)),
        "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": pinfo["peak"],
                     "unit": "TFLOP/s", "frac": achieved / pinfo["peak"], "avg_launch_ms": avg_ms,
                     "launches": int(k_launches.value), "mfma_per_fp32_product": pinfo["mfma_per_product"]},
    }
    if proposals == S_PROPOSALS:
        # (the alternative paths run inside the default command, which is what the PMC passes profiled: bench args "")
        rec["roofline"]["traffic"], rec["roofline"]["traffic_source"] = committed_traffic(kernel, "")
    if name == "h1" and proposals == S_PROPOSALS and not rec["range_guard_fired"]:
        rec["roofline"]["attention_block"] = attention_block(model, device, proposals, avg_ms, "h1")
        rec["measured_deviation"] = ("un-calibrated full-size weights vs the reference's vectors: 1.6e-3 (coordinates), 2.0e-4 "
                                     "(log p) relative; forward/reverse round trip of log p: 0.043 max of ~228; accept indicators of "
                                     "whole MH iterations vs the fp32 oracle 384/384 (profiles/r04_h1_accuracy.txt).  This bench's "
                                     "identity-flow calibration makes scale = 1, shift = 0 exactly on every path")
    return rec


def other_config_record(device, seed, config, steps, sync_every, n_chains=1):
    """BASELINE configs[3] / configs[4] (and the lock-step mode of configs[1]) inside the default command, so that the driver's
    record carries a line for each that somebody other than the builder timed (missing 5): a chain of its own on
    the split-fp16 path, 2 warm-up iterations, `steps` timed ones with every 5th net-block launch bracketed by HIP events."""
    from timewarp_amd import _lib

    cfg = CONFIGS[config]
    lib = _lib.load()
    rows = (cfg["S"] // n_chains) * n_chains
    chain, model = build_chain(device, seed, cfg["S"], PATHS["h3"]["path"], config, n_chains)
    with torch.no_grad():
        for _ in range(2):
            chain.step_deferred()
        chain.flush()
    torch.cuda.synchronize()
    acc0, prop0 = chain.accepted, chain.proposals
    stride = os.environ.get("TW_PROFILE_STRIDE")
    os.environ["TW_PROFILE_STRIDE"] = "5"   # short leg: more samples; coprime with the 16 launches of an iteration
    lib.tw_profile_begin()
    t0 = time.perf_counter()
    with torch.no_grad():
        for it in range(steps):
            chain.step_deferred()
            if (it + 1) % sync_every == 0:
                chain.flush(lag=True)
        chain.trajectory()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = C.c_double(0.0), C.c_int64(0)
    lib.tw_profile_end(C.byref(k_ms), C.byref(k_launches))
    if stride is None:
        del os.environ["TW_PROFILE_STRIDE"]
    else:
        os.environ["TW_PROFILE_STRIDE"] = stride
    avg_ms = k_ms.value / max(int(k_launches.value), 1)
    flop_per_launch = cfg["flop_sample_pass"] * rows / N_COUPLING
    achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    kernel = live_kernel(lib)
    bench_args = " ".join(([] if config == "ad" else [f"--config {config}"]))
    traffic, traffic_src = committed_traffic(kernel, bench_args) if n_chains == 1 else (None, None)
    return {
        "workload": cfg["workload"] + ("" if n_chains == 1 else f"; {n_chains} lock-step chains per GPU sharing the launch's {rows} rows (SURVEY 8f-1)"),
        "value": (chain.accepted - acc0) / elapsed, "unit": "MH-accepted samples/s",
        "proposals_per_s": (chain.proposals - prop0) / elapsed,
        "steps": steps, "warmup": 2, "ms_per_step": elapsed / steps * 1e3, "proposals_per_step": rows, "chains_per_gpu": n_chains,
        "dtype": PATHS["h3"]["dtype"], "range_guard_fired": bool(getattr(model, "demoted", False)),
        "roofline": {"bound": "mfma", "kernel": kernel + " (" + cfg.get("kernel_note", "both coupling nets of one coupling layer, all proposals") + ")",
                     "achieved": achieved, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F16_MFMA_PEAK_TFLOPS,
                     "avg_launch_ms": avg_ms, "launches": int(k_launches.value), "algorithmic_flop_per_launch": flop_per_launch,
                     "traffic": traffic, "traffic_source": traffic_src, "mfma_per_fp32_product": 3},
    }


def protein_record(device, seed, steps=8):
    """Whole MH iterations on the reference's 691-atom test protein (CONFIGS["1hgv"]): the flow on the model's default path there
    (TW_PATH_SIMPLE_H3, both passes), the AMBER energy kernel on all 691 atoms, the accept step - one C-ABI call per iteration."""
    from timewarp_amd import _lib

    cfg = CONFIGS["1hgv"]
    chain, model = build_chain(device, seed, cfg["S"], -1, "1hgv")   # -1: PREFER_SPLIT_FP16, the constructor's default
    with torch.no_grad():
        for _ in range(2):
            chain.step_deferred()
        chain.flush()
        torch.cuda.synchronize()
        acc0, prop0 = chain.accepted, chain.proposals
        t0 = time.perf_counter()
        for _ in range(steps):
            chain.step_deferred()
        chain.flush()
        chain.trajectory()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    return {"workload": cfg["workload"], "value": (chain.accepted - acc0) / elapsed, "unit": "MH-accepted samples/s",
            "proposals_per_s": (chain.proposals - prop0) / elapsed, "steps": steps, "warmup": 2, "ms_per_step": elapsed / steps * 1e3,
            "proposals_per_step": cfg["S"], "path": int(model._path_for(cfg["V"])), "dtype": PATHS["h3"]["dtype"],
            "algorithmic_flop_per_step": 2 * cfg["flop_sample_pass"] * cfg["S"],
            "range_guard_fired": bool(getattr(model, "demoted", False))}


def large_molecule_flow_record(device, seed, V, S, passes=4):
    """Molecules above every fused layout (193+ atoms; the reference's own test protein has 691): FLOW PASSES ONLY of the
    kernel-attention model on the path a model takes there by default (TW_PATH_SIMPLE_H3: per-op launches, linears / mixing /
    FFN on split-fp16 MFMAs) - S proposals of a synthetic V-atom molecule drawn from one conditioning state (the reverse pass of an
    MH iteration) and their log-likelihood back, every proposal its own conditioning state (the forward pass).  No energy, no
    accept step: the line exists so that VERDICT r05's 200 x 256 / 256 x 256 figures are timed by the driver as well."""
    import timewarp_amd as tw
    from timewarp_amd import _lib, synthetic

    model = tw.model_constructor(synthetic.kernel_transformer_nvp_config())
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), base_seed=0, calibrated=True, **CALIBRATION))
    model = model.to(device).eval()
    path = model._path_for(V)
    g = torch.Generator().manual_seed(seed)
    at = torch.randint(0, 5, (1, V), generator=g).to(device)
    xc = (torch.randn(1, V, 3, generator=g) * 0.8).to(device)
    xv = torch.randn(1, V, 3, generator=g).to(device)
    mk = torch.zeros(1, V, dtype=torch.bool, device=device)
    rev = lambda: model.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                     masked_elements=mk, num_samples=S)
    with torch.no_grad():
        yc, yv, _ = rev()
        atS, mkS, xcS, xvS = at.repeat(S, 1), mk.repeat(S, 1), xc.repeat(S, 1, 1), xv.repeat(S, 1, 1)
        fwd = lambda: model.log_likelihood(atom_types=atS, x_coords=yc.squeeze(1), x_velocs=yv.squeeze(1), y_coords=xcS, y_velocs=xvS,
                                           adj_list=None, edge_batch_idx=None, masked_elements=mkS)
        fwd()
        ms = {}
        for name, f in (("reverse", rev), ("forward", fwd)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(passes):
                f()
            torch.cuda.synchronize()
            ms[name] = (time.perf_counter() - t0) / passes * 1e3
    flop = 16 * V * F_BLK_KERNEL(V) * S
    rec = {"workload": f"kernel_transformer_nvp.yaml flow passes only, synthetic {V}-atom molecule x {S} proposals (no energy, no accept step)",
           "path": {_lib.TW_PATH_SIMPLE_H3: "TW_PATH_SIMPLE_H3 (per-op launches on split-fp16 MFMAs: the model default above 192 atoms)"}.get(path, str(path)),
           "passes": passes, "algorithmic_flop_per_pass": flop, "dtype": PATHS["h3"]["dtype"],
           "range_guard_fired": bool(getattr(model, "demoted", False))}
    for name in ("reverse", "forward"):
        tf = flop / (ms[name] * 1e-3) / 1e12
        rec[name] = {"ms_per_pass": ms[name], "achieved": tf, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F16_MFMA_PEAK_TFLOPS}
    return rec


def end_timed_region(traj, t0, device, world):
    """The N > 1 leg of the measurement contract: the one collective of the path (all-gather of the trajectories, inside
    the timed region), barrier + device synchronisation on both sides of the clock, MAX of the elapsed time over ranks.
    Device-agnostic so that tests/test_distributed_cpu.py can run it under gloo; returns (gathered list, seconds)."""
    from timewarp_amd import distributed

    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    gathered, _ = distributed.gather_trajectories(traj)  # no-op at N = 1
    sync()
    if world > 1:
        torch.distributed.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    per_rank = [elapsed]
    if world > 1:
        each = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(each, t)  # 8 bytes per rank, after the clock has stopped
        per_rank = [float(e.item()) for e in each]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    end_timed_region.per_rank_seconds = per_rank
    return gathered, float(t.item())


def whole_job_rates(accepted, proposals, states, elapsed, device):
    """Sums of the per-rank counters over the job / the max-over-ranks time: `value` and its two companions."""
    from timewarp_amd import distributed

    accepted, proposals, states = distributed.all_reduce_counters([accepted, proposals, states], device)
    return accepted / elapsed, proposals / elapsed, states / elapsed, accepted


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves - re-run this same
    command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free>` (the form the measurement contract names), pass rank 0's JSON line through and exit with the
    job's status.  Inside a torchrun environment whose WORLD_SIZE is not N the run is refused: a line whose `n_gpus`
    differs from `--gpus` would be read as a measurement of N GPUs."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with --nproc-per-node {args.gpus} "
                             f"(or unset the torchrun variables: `python bench.py --gpus {args.gpus}` starts its own ranks)")
        return
    if args.gpus == 1:
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    import socket
    import subprocess

    with socket.socket() as s:   # a free rendezvous port on the loopback interface (the container hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def plumbing_only(args, rank, world):
    """--plumbing-only: the launch form and the N > 1 leg of the timed region on dummy per-rank trajectories (ragged lengths, as
    at collection).  The line is marked and carries no metric - it exists so that the bare `python bench.py --gpus N` form can
    be exercised without a GPU."""
    traj = torch.full((3 + rank, V_ATOMS, 3), float(rank))
    t0 = time.perf_counter()
    gathered, elapsed = end_timed_region(traj, t0, "cpu", world)
    _, _, states, _ = whole_job_rates(0.0, 0.0, float(traj.shape[0]), 1.0, "cpu")   # the counters' all-reduce: sum of the lengths
    ok = len(gathered) == world and all(g.shape[0] == 3 + r and bool((g == r).all()) for r, g in enumerate(gathered))
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "metric": None, "value": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "per_rank_ms": [t * 1e3 for t in end_timed_region.per_rank_seconds],
                          "gathered_ok": ok, "states_gathered": states,
                          "backend": torch.distributed.get_backend() if world > 1 else None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py --plumbing-only: the gathered trajectories are wrong")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--proposals", type=int, default=None, help="proposals per MH iteration (default: the configuration's)")
    ap.add_argument("--config", choices=sorted(k for k in CONFIGS if k != "1hgv"), default="ad",   # (1hgv: other_configs.protein_1hgv only)
                    help="ad: BASELINE.json configs[1], the headline (default); 4aa: configs[3]; dense: configs[4]")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--sync-every", type=int, default=8,
                    help="MH iterations queued per host read-back of the accept results (sample_with_model's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs on BASELINE configs[3] / configs[4] / lock-step chains that the default line carries")
    ap.add_argument("--chains", type=int, default=1,
                    help="chains per GPU evaluated in lock-step (SURVEY 8f-1): the launch's rows (--proposals) are shared, "
                         "proposals // chains per chain and iteration.  Default 1 = the BASELINE configuration; a line of its own "
                         "otherwise (no alternative paths, no CPU baseline)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch check, NOT a measurement: the ranks rendezvous (gloo unless TW_DIST_BACKEND says otherwise), run "
                         "the timed region's collection leg on dummy trajectories and rank 0 prints a line marked "
                         "\"plumbing_only\": true.  No GPU, no kernels - what tests/test_distributed_cpu.py runs")
    ap.add_argument("--path", choices=sorted(PATHS), default="h3",
                    help="flow execution path: split-fp16 fused kernel (default, the headline), exact-f32 fused kernel, or the "
                         "opt-in single-MFMA fast mode h1 (not a parity path)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.proposals is None:
        args.proposals = cfg["S"]
    if args.config in ("dense", "4aa", "4aa-nnqq") and args.path == "f32":
        ap.error("--config dense / 4aa are measured on the default (split-fp16) path h3 or the opt-in fast mode h1")

    self_launch(args)   # returns only in a process that IS one of the --gpus ranks

    from timewarp_amd import distributed

    # TW_DIST_BACKEND=gloo + one GPU shared by all ranks is a plumbing check of the N>1 path on a 1-GPU box
    # (tools/README.md); the driver's runs use RCCL ("nccl") with one GPU per rank.
    rank, world, local = distributed.init_from_env(os.environ.get("TW_DIST_BACKEND", "gloo" if args.plumbing_only else "nccl"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")   # (self_launch checked the env)
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
    from timewarp_amd import _lib

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
    backend = os.environ.get("TW_DIST_BACKEND", "nccl")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and torch.cuda.device_count() < local_world:
        # one rank per GPU or nothing: two ranks sharing a device would report a "scaling" figure that is really time-slicing
        raise RuntimeError(f"bench.py: {local_world} ranks on this node but only {torch.cuda.device_count()} visible GPU(s); "
                           "the RCCL run needs one GPU per rank (TW_DIST_BACKEND=gloo is the 1-GPU plumbing check)")
    device = torch.device("cuda", local % torch.cuda.device_count() if "TW_DIST_BACKEND" in os.environ else local)
    torch.cuda.set_device(device)
    lib = _lib.load()

    pinfo = PATHS[args.path]
    if args.chains < 1 or args.chains > args.proposals:
        ap.error("--chains: 1 .. --proposals")
    rows = (args.proposals // args.chains) * args.chains   # rows per flow launch
    chain, model = build_chain(device, distributed.chain_seed(args.seed, rank), args.proposals, pinfo["path"], args.config, args.chains)
    with torch.no_grad():
        for _ in range(args.warmup):
            chain.step_deferred()
        chain.flush()
    torch.cuda.synchronize()
    acc0, prop0 = chain.accepted, chain.proposals
    states0 = sum(t.shape[0] for t in chain.chain_c)
    if world > 1:
        # the collection's own collectives once, untimed (ragged lengths, as at collection): whatever RCCL sets up lazily at the
        # first all_gather / all_reduce of a communicator - channels, kernels, staging buffers - is not part of the K timed steps
        distributed.gather_trajectories(torch.zeros((1 + rank % 2, cfg["V"], 3), dtype=torch.float32, device=device))
        distributed.all_reduce_counters([0.0, 0.0, 0.0], device)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # HIP events around every 17th launch of the dominant kernel.  A bracketed launch costs the stream two bubbles of 6-8 us
    # (profiles/r05_iteration_timeline.txt: the gaps in front of and behind launches 1, 6, 11, 16 of an iteration at r04's stride
    # of 5 - 45 us of a 6.5 ms iteration that the product never pays).  17 is coprime with the 16 net-block launches of an
    # iteration, so over the timed region every position of the pass is sampled equally often - the first launch of a pass has
    # no coupling prologue, a stride of 4 would have bracketed it in half of the samples.
    os.environ.setdefault("TW_PROFILE_STRIDE", "17")
    lib.tw_profile_begin()
    t0 = time.perf_counter()
    with torch.no_grad():
        # deferred iterations: tw_mh_accept moves the chain state on the device, the host reads the accept results
        # of `sync_every` iterations with one copy (what sample_with_model does for the bulk of a chain)
        for it in range(args.steps):
            chain.step_deferred()
            if (it + 1) % args.sync_every == 0:
                chain.flush(lag=True)   # as sample_with_model does: book the previous window while this one runs
        traj, _ = chain.trajectory()
        gathered, elapsed = end_timed_region(traj, t0, device, world)
    k_ms, k_launches = C.c_double(0.0), C.c_int64(0)
    lib.tw_profile_end(C.byref(k_ms), C.byref(k_launches))
    states = sum(x.shape[0] for x in chain.chain_c) - states0
    value, proposals_per_s, states_per_s, accepted = whole_job_rates(chain.accepted - acc0, chain.proposals - prop0, states,
                                                                       elapsed, device)

    if rank == 0:
        launches = max(int(k_launches.value), 1)
        avg_ms = k_ms.value / launches
        flop_per_launch = cfg["flop_sample_pass"] * rows / N_COUPLING
        # HBM/fabric bytes per launch come from separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE, WRITE_SIZE; gfx950 correction applied), summarised in profiles/
        # ... of this kernel instantiation under this configuration - null when no committed summary measured exactly that
        kernel_live = live_kernel(lib)
        bench_args = " ".join(([] if args.config == "ad" else [f"--config {args.config}"]) + ([] if args.path == "h3" else [f"--path {args.path}"]))
        traffic, traffic_src = committed_traffic(kernel_live, bench_args) if (args.proposals == cfg["S"] and args.chains == 1) else (None, None)
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out = {
            "metric": "MH-accepted samples/sec (whole node), alanine-dipeptide kernel_transformer_nvp" if args.config == "ad" else
                      "MH-accepted samples/sec (whole node), " + ("4AA tetrapeptide kernel_transformer_nvp" if args.config.startswith("4aa")
                                                                   else "alanine-dipeptide transformer_nvp (dense softmax)"),
            "value": value,
            "unit": "MH-accepted samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": pinfo["dtype"],
            "data": "synthetic",
            "config": {
                "workload": cfg["workload"],
                "bench_config": args.config,
                "proposals_per_step": rows,
                "chains_per_gpu": args.chains,
                "proposals_per_chain_and_step": rows // args.chains,
                "weights": "name-seeded synthetic N(0,1)/sqrt(fan_in); identity flow (last out_mlp layer of every coupling net "
                           "zeroed, SURVEY 8d's idea) with coordinate prior log-scale %g and velocity prior log-scale 0 (NOT "
                           "8d's -5 / -5: tuned so acceptance is non-degenerate against the stiff bonded terms).  `value` depends on this calibration "
                           "(the first accepted proposal ends an iteration: accepted_per_step is 1 per chain); `proposals_per_s` is the "
                           "weight-independent figure" % cfg["calibration"]["coords_log_scale"],
                "mh_mode": "accept=True, random_velocs=True, resample_velocs=True (velocity terms of the exponent cancel)",
                "setup_prewarm": f"{PREWARM_PASSES} untimed flow passes on throw-away inputs before the warm-up steps (GPU clock ramp)",
                "execution_path": args.path,
            },
            "proposals_per_s": proposals_per_s,
            "chain_states_per_s": states_per_s,
            "accepted_per_step": accepted / (args.steps * world),
            "per_rank_ms": [t / args.steps * 1e3 for t in getattr(end_timed_region, "per_rank_seconds", [elapsed])],
            "roofline": {
                "bound": "mfma",
                # the instantiation the timed region's last flow pass launched, as the library reports it (tw_last_netblock_kernel:
                # <NT, ASM, DENSE, WIDE, RFF, ENC, H1, NG6>) - the layout is chosen per launch
                "kernel": kernel_live + " (" + cfg.get("kernel_note", "both coupling nets of one coupling layer, all proposals") + ")",
                "achieved": achieved,
                "peak": pinfo["peak"],
                "unit": "TFLOP/s",
                "frac": achieved / pinfo["peak"],
                "mfma_per_fp32_product": pinfo["mfma_per_product"],
                "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_ms": avg_ms,
                "launches": int(k_launches.value),
                "launches_timed": "every %s-th net-block launch of the timed region bracketed by HIP events on the launch stream "
                                  "(16 launches per iteration: 2 without, 14 with the coupling prologue; a stride coprime with "
                                  "16 samples every position equally); the trailing coupling launch of each pass is not in "
                                  "this figure" % os.environ["TW_PROFILE_STRIDE"],
                "algorithmic_flop_per_launch": flop_per_launch,
            },
        }
        # a run whose range guard fired has been (partly) timed on the f32 kernels - say so instead of
        # reporting an f16 roofline computed from f32 launch times
        out["range_guard_fired"] = bool(getattr(model, "demoted", False))
        if out["range_guard_fired"]:
            raise RuntimeError("bench.py: the fp16 range guard demoted the model to the exact-f32 kernels during the run; "
                               "the line would mislabel the measured path.  Re-run with --path f32")
        if args.config != "ad" or args.chains > 1:
            # one line per extra configuration: roofline from the live HIP events; the headline's companions (attention block,
            # other paths, CPU baseline) belong to --config ad with one chain
            out["roofline"]["algorithmic_tflop_per_iteration"] = 2 * cfg["flop_sample_pass"] * rows / 1e12
            print(json.dumps(out), flush=True)
            if world > 1:
                torch.distributed.destroy_process_group()
            return
        if args.path == "h1" and args.proposals == S_PROPOSALS:
            out["roofline"]["attention_block"] = attention_block(model, device, args.proposals, avg_ms, "h1")
        if args.path == "h3" and args.proposals == S_PROPOSALS:
            out["roofline"]["attention_block"] = attention_block(model, device, args.proposals, avg_ms)
            # what the matrix pipes of THIS box sustain, measured now: a bare 16x16x32 MFMA stream on all 256 CUs (random
            # operands; tw_probe_mfma_clock = tools/probe/mfma_stream_probe.hip as an entry point), two launches of ~10 ms: the first
            # lets the clock settle (a 1.3 ms launch from idle reads 1.74-1.85 GHz where the settled chip runs 2.0-2.1: r03's constant)
            probe_cyc, probe_ms = C.c_int64(0), C.c_double(0.0)
            sustained_ghz, probe = SUSTAINED_MFMA_CLOCK_GHZ, None
            if lib.tw_probe_mfma_clock(256, 32768, C.byref(probe_cyc), C.byref(probe_ms), None) == 0 and probe_ms.value > 0:
                sustained_ghz = probe_cyc.value / (probe_ms.value * 1e6)
                probe = {"workgroups": 256, "mfma_per_wave": 36 * 32768, "cycles": int(probe_cyc.value), "ms": probe_ms.value,
                         "cycles_per_mfma": probe_cyc.value / (36.0 * 32768),
                         "tflops": 256 * 4 * 36 * 32768 * 16384.0 / (probe_ms.value * 1e-3) / 1e12}
            busy_clocks = 16.0 * h3_mfma_per_wave(model.dims)
            floor_ms = busy_clocks / (sustained_ghz * 1e6)
            out["roofline"]["power_bound"] = {
                "what": "launch time if the matrix pipe never idled, at the clock THIS chip sustains with every matrix pipe busy on "
                        "random fp16 operands (mfma_stream_probe below, measured right after the timed region; one CU alone runs "
                        "2.40 GHz); frac = that floor / the live launch time.  Context for `frac` above, not a replacement of it",
                "matrix_pipe_busy_clocks_per_wave": busy_clocks,
                "sustained_clock_ghz": sustained_ghz,
                "sustained_clock_measured_by_this_run": probe is not None,
                "mfma_stream_probe": probe,
                "sustained_f16_mfma_tflops": sustained_ghz * 1e9 * 1024 * 1024 / 1e12,
                "floor_ms": floor_ms,
                "frac": floor_ms / avg_ms,
                "constants": "matrix_pipe_busy_clocks_per_wave = 16 clocks x the MFMA count of the generated statements for this model's "
                             "dimensions (h3_mfma_per_wave: 36 216; SQ_INSTS_MFMA 36.2 k, SQ_VALU_MFMA_BUSY_CYCLES 579.5 k in "
                             "profiles/r05_ad_sq_counters.md) - a property of the instruction stream, not a measurement; sustained_clock_ghz, "
                             "avg_launch_ms and attention_block.effective_clock_ghz are measured by this run (the clock by the "
                             "bare MFMA stream of tw_probe_mfma_clock right after the timed region; 1.74 GHz, the r03 probe "
                             "result, only if that call fails)",
                "source": "tw_probe_mfma_clock (this run); profiles/r04_mfma_shape_probe_warm.txt (the same stream stand-alone: 1.99-2.02 GHz "
                          "settled; a 32x32x16 stream sustains 1.65-1.69), profiles/r05_ad_sq_counters.md",
            }
        if world == 1 and args.path != "f32":
            out["alt_path"] = alt_path_record(device, distributed.chain_seed(args.seed, rank), args.proposals,
                                              max(4, args.steps // 4), args.sync_every)
        if world == 1 and args.path == "h3":
            # the north star's second figure (>= 40 % of the half-precision MFMA roofline on the attention block) belongs to
            # the fast mode (SURVEY section 7 "precision contract"): reported here, beside the headline
            out["alt_path_h1"] = alt_path_record(device, distributed.chain_seed(args.seed, rank), args.proposals,
                                                 max(8, args.steps // 2), args.sync_every, "h1")
        if world == 1 and args.path == "h3" and args.proposals == S_PROPOSALS and not args.no_other_configs:
            seed = distributed.chain_seed(args.seed, rank)
            out["other_configs"] = {
                "4aa": other_config_record(device, seed, "4aa", 12, args.sync_every),       # BASELINE configs[3]
                "dense": other_config_record(device, seed, "dense", 12, args.sync_every),   # BASELINE configs[4]
                # the mode whole-node accepted samples/s rewards - 32 chains x 31 proposals through the headline's launches
                "ad_chains32": other_config_record(device, seed, "ad", 24, args.sync_every, n_chains=32),
                # above every fused layout: flow passes of 200- and 256-atom molecules x 256 proposals
                "flow_200x256": large_molecule_flow_record(device, seed, 200, 256),
                "flow_256x256": large_molecule_flow_record(device, seed, 256, 256),
                # ... and whole MH iterations on the reference's own 691-atom test protein
                "protein_1hgv": protein_record(device, seed),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.proposals)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
