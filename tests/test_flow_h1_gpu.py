"""The opt-in "fast" mode (TW_PATH_FUSED_H1 / TW_EXECUTION_PATH=h1): ONE half-precision MFMA per product on fp16
operands (11 significand bits), fp32 accumulation.  NOT a parity path - SURVEY section 7 "precision contract": the 1e-5
bar belongs to the split-fp16 / f32 kernels (tests/test_flow_gpu.py); here every comparison states the error that was
MEASURED on MI355X and holds the path to a bar a small factor above it, so that a regression of the arithmetic (a
dropped term, a wrong tile) is caught while the documented deviation is not hidden.

What is checked against the reference vectors / the oracle:
  * the full-size kernel_transformer_nvp goldens (un-calibrated weights: every coupling net moves the sample);
  * 48 rows of a 1000-proposal launch (the bench size) against the oracle;
  * the property that keeps Metropolis-Hastings meaningful in this mode: the density the reverse pass reports for a
    proposal and the density the forward pass assigns to the same proposal agree to the arithmetic's own noise;
  * whole MH iterations against the oracle loop - indicator AGREEMENT is reported and bounded from below, not asserted
    bit for bit (a proposal whose acceptance probability sits within the noise of u can flip);
  * ragged / padded batches (one- and two-molecule waves), the path selection by name and the fall-back per call."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from oracle import mh_oracle as mo
from tests import helpers as H

pytestmark = pytest.mark.gpu

H1, H3 = 4, 3

# measured on MI355X (profiles/r04_h1_accuracy.txt), un-calibrated full-size weights, alanine dipeptide:
#   loglik 3.3e-5, proposals 1.6e-3 (coordinates) / 1.2e-3 (velocities), log p(y|x) 2.0e-4, reverse-move density 1.6e-3
BARS = dict(loglik=1.5e-4, s_y_coords=5e-3, s_y_velocs=4e-3, s_logp=8e-4, logp_yx=5e-3)


def _errors(out, d):
    keep = ~d["masked"][0]
    e = {}
    for k in BARS:
        a, b = (out[k][:, :, keep], d[k][:, :, keep]) if k.startswith("s_y") else (out[k], d[k])
        e[k] = H.rel_err(a, b)
    return e


def test_full_kernel_ad_golden_measured_error():
    d, _ = H.load("kernel_full_ad")
    m = H.tw_kernel_model(H.full_kernel_sd(), path=H1)
    out = H.run_model_case(m, d)
    e = _errors(out, d)
    print("h1 vs reference vectors:", {k: f"{v:.2e}" for k, v in e.items()})
    for k, bar in BARS.items():
        assert e[k] < bar, (k, e[k], bar)
    # and it IS the single-MFMA arithmetic, not one of the parity kernels answering under its name
    assert e["s_y_coords"] > 1e-5


def test_calibrated_golden_is_exact():
    """bench.py's calibration (identity flow: last out_mlp layers zeroed): scale = 1, shift = 0 whatever the nets compute,
    so the fast path reproduces the vectors as tightly as the parity kernels do."""
    d, _ = H.load("kernel_full_ad_calibrated")
    m = H.tw_kernel_model(H.full_kernel_sd(calibrated=True), path=H1)
    H.assert_case_close(H.run_model_case(m, d), d, tol=1e-5)


def test_full_size_S1000_rows_and_round_trip():
    """Bench size: 1000 proposals in one launch (250 workgroups).  48 rows spread over the launch against the oracle at
    the measured error; all 1000 rows through the round trip: the forward pass (log_likelihood) of the proposals the
    reverse pass produced gives back the reverse pass's own log p(y|x).  In exact arithmetic the two are equal; here both
    run the same fp16 arithmetic on inputs that differ in the last fp32 bits, which moves a few operand roundings - the
    residual is the noise the acceptance ratio of an MH iteration carries in this mode."""
    sd = H.full_kernel_sd()
    m = H.tw_kernel_model(sd, path=H1)
    d, _ = H.load("kernel_full_ad")
    S = 1000
    g = torch.Generator().manual_seed(5)
    zc, zv = fo.draw_latents(sd, S, (1, 22, 3), g)
    at, xc, xv, mk = d["atom_types"].cuda(), d["x_coords"].cuda(), d["x_velocs"].cuda(), d["masked"].cuda()
    yc, yv, lp = m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None,
                                                edge_batch_idx=None, masked_elements=mk, num_samples=S,
                                                z_coords=zc.cuda(), z_velocs=zv.cuda())
    ll = m.log_likelihood(atom_types=at.repeat(S, 1), x_coords=xc.repeat(S, 1, 1), x_velocs=xv.repeat(S, 1, 1),
                          y_coords=yc.squeeze(1), y_velocs=yv.squeeze(1), adj_list=None, edge_batch_idx=None,
                          masked_elements=mk.repeat(S, 1))
    assert torch.isfinite(lp).all() and torch.isfinite(ll).all()
    H.assert_not_demoted(m)
    rt = (ll.cpu() - lp.squeeze(1).cpu()).abs()
    print(f"h1 round trip |log p forward - log p reverse|: max {float(rt.max()):.3e}, mean {float(rt.mean()):.3e} "
          f"of |log p| ~ {float(lp.abs().mean()):.1f}")
    assert float(rt.max()) < 0.2 and float(rt.mean()) < 0.01      # measured: 0.043 / 9.1e-4 (|log p| ~ 228)
    rows = list(range(0, 8)) + list(range(496, 504)) + list(range(992, 1000))
    rest = [r for r in torch.randperm(S, generator=g).tolist() if r not in rows][:24]
    rows = torch.tensor(sorted(rows + rest))
    ryc, ryv, rlp = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, d["atom_types"], d["x_coords"],
                                                    d["x_velocs"], d["masked"], zc[rows], zv[rows])
    e = dict(coords=H.rel_err(yc.cpu()[rows], ryc), velocs=H.rel_err(yv.cpu()[rows], ryv), logp=H.rel_err(lp.cpu()[rows], rlp))
    print("h1, 48 rows of the 1000-proposal launch vs oracle:", {k: f"{v:.2e}" for k, v in e.items()})
    assert e["coords"] < BARS["s_y_coords"] and e["velocs"] < BARS["s_y_velocs"] and e["logp"] < BARS["s_logp"]


@pytest.mark.parametrize("V,lens", [(22, [22, 20, 22, 17, 22]), (7, [7, 5, 6, 7, 7, 3, 7, 7, 7]), (30, [30, 28, 25]), (48, [48, 40]),
                                    (60, [60, 51, 60, 60, 44]), (100, [100, 87, 100]), (160, [160, 131]), (88, [88, 61, 88]), (176, [176, 150])])
def test_ragged_batches_vs_split_fp16_kernel(V, lens):
    """Padded atoms, several molecules per wave (windowed mixing), one per wave (full mixing), and the wide layout (49 - 160
    atoms: molecules packed over a workgroup's four waves, tw_h1_attns_asm.inc + the per-section in / FFN / out statements): log_likelihood of a
    ragged batch on the fast path against the split-fp16 kernel (itself held to the oracle at 1e-5) at the measured
    error; padded atoms contribute nothing (their inputs are overwritten with garbage and the result must not move)."""
    sd = H.full_kernel_sd()
    g = torch.Generator().manual_seed(V)
    B = len(lens)
    at = torch.randint(0, 5, (B, V), generator=g)
    xc = torch.randn(B, V, 3, generator=g) * 0.3
    xv = torch.randn(B, V, 3, generator=g) * 0.5
    yc = xc + torch.randn(B, V, 3, generator=g) * 0.05
    yv = xv + torch.randn(B, V, 3, generator=g) * 0.05
    mk = torch.zeros(B, V, dtype=torch.bool)
    for i, n in enumerate(lens):
        mk[i, n:] = True
    args = lambda yc_, yv_: dict(atom_types=at.cuda(), x_coords=xc.cuda(), x_velocs=xv.cuda(), y_coords=yc_.cuda(),
                                 y_velocs=yv_.cuda(), adj_list=None, edge_batch_idx=None, masked_elements=mk.cuda())
    fast = H.tw_kernel_model(sd, path=H1)
    ref = H.tw_kernel_model(sd, path=H3)
    a = fast.log_likelihood(**args(yc, yv)).cpu()
    b = ref.log_likelihood(**args(yc, yv)).cpu()
    H.assert_not_demoted(fast)
    e = H.rel_err(a, b)
    print(f"h1 vs h3, V={V}: {e:.2e}")
    assert 0 < e < 3e-3
    yc2, yv2 = yc.clone(), yv.clone()
    yc2[mk] = 7.0
    yv2[mk] = -3.0
    assert torch.equal(fast.log_likelihood(**args(yc2, yv2)).cpu(), a)


def test_path_selection_by_name(monkeypatch):
    """TW_EXECUTION_PATH=h1 asks for the fast kernel wherever it exists (kernel attention, up to 48 atoms) and takes the
    split-fp16 / f32 kernels for the rest, per call; the default never selects it."""
    import ctypes as C
    import timewarp_amd as tw
    from timewarp_amd import _lib, synthetic
    from timewarp_amd.modules import flow

    cfg = synthetic.kernel_transformer_nvp_config()
    monkeypatch.delenv("TW_EXECUTION_PATH", raising=False)
    assert tw.model_constructor(cfg).execution_path == flow.PREFER_SPLIT_FP16
    monkeypatch.setenv("TW_EXECUTION_PATH", "h1")
    m = tw.model_constructor(cfg)
    assert m.execution_path == flow.PREFER_SINGLE_FP16
    assert [m._path_for(v) for v in (1, 22, 48, 49, 60, 80, 90, 160, 192, 193)] == [H1] * 9 + [5]   # (90: since the 96-slot stride; 161 .. 192: six-group windows; above: the per-op path with split-fp16 GEMMs)
    desc = m.dims.to_desc()
    lib = _lib.load()
    assert lib.tw_flow_path_supported(C.byref(desc), 22, H1) == 1 and lib.tw_flow_path_supported(C.byref(desc), 60, H1) == 1
    assert 0 < lib.tw_flow_packed_h1_bytes(C.byref(desc)) < lib.tw_flow_packed_h3_bytes(C.byref(desc)) * 0.6
    # the dense softmax variant has one too (its MLP sections; the attention block stays in split form), since r05 also with
    # position features (the 192-column in-MLP stays in split form); above 48 atoms the preference falls back
    monkeypatch.setenv("TW_EXECUTION_PATH", "h1")
    md = tw.model_constructor(synthetic.transformer_nvp_config())
    assert md._path_for(22) == H1
    mp = H.tw_dense_model(H.full_dense_posenc_sd(), rff_dim=128, path=None)
    assert mp._path_for(22) == H1 and mp._path_for(60) != H1
    # by name on an unsupported shape: an error, not a silent other kernel
    mm = H.tw_kernel_model(H.full_kernel_sd(), path=H1)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1, 200, 3, generator=g) * 0.8).cuda()   # (193+: beyond the 192 slots of a workgroup)
    with pytest.raises(RuntimeError, match="single-MFMA path unsupported"):
        mm.log_likelihood(atom_types=torch.zeros(1, 200, dtype=torch.long).cuda(), x_coords=x, x_velocs=x, y_coords=x, y_velocs=x,
                          adj_list=None, edge_batch_idx=None, masked_elements=torch.zeros(1, 200, dtype=torch.bool).cuda())


@pytest.mark.parametrize("name", ["dense_full_ad", "dense_full_padded", "dense_posenc_full_ad"])
def test_dense_model_fast_mode_measured_error(name):
    """transformer_nvp (BASELINE config 4) on the fast path: in / FFN / out sections with one fp16 MFMA per product, the
    softmax attention block unchanged (split-fp16).  Measured against the reference's vectors: log-likelihood 5.6e-4 /
    6.5e-4 (plain / padded), proposals 1.0-1.4e-3, log-densities 1.4e-4 / 2.0e-3 - the class of the kernel-attention fast
    mode; and clearly not the parity kernel (> 1e-5)."""
    d, _ = H.load(name)
    if "posenc" in name:   # r05: random Fourier position features - the in-MLP in split form, the rest single-MFMA
        m = H.tw_dense_model(H.full_dense_posenc_sd(), rff_dim=128, path=H1)
    else:
        m = H.tw_dense_model(H.full_dense_sd(), path=H1)
    out = H.run_model_case(m, d)
    keep = ~d["masked"][0]
    e = {}
    for k in BARS:
        if k in out:
            a, b = (out[k][:, :, keep], d[k][:, :, keep]) if k.startswith("s_y") else (out[k], d[k])
            e[k] = H.rel_err(a, b)
    print("h1 dense vs the reference vectors:", name, {k: f"{v:.2e}" for k, v in e.items()})
    for k, v in e.items():
        assert v < (2e-3 if k == "loglik" else BARS[k]), (k, v)
    assert e["loglik"] > 1e-5
    assert not m.demoted


def test_dense_model_fast_mode_round_trip_and_determinism():
    """1000 proposals of the dense flow on the fast path: the forward pass returns the latents of the reverse pass to the
    fast mode's own error (both passes use the same arithmetic), and two calls agree bit for bit."""
    d, _ = H.load("dense_full_ad")
    S = 1000
    g = torch.Generator().manual_seed(3)
    m = H.tw_dense_model(H.full_dense_sd(), path=H1)
    a = {k: d[k].cuda() for k in ("atom_types", "x_coords", "x_velocs", "masked")}
    zc, zv = torch.randn(S, 1, 22, 3, generator=g).cuda(), torch.randn(S, 1, 22, 3, generator=g).cuda()
    call = lambda: m.conditional_sample_with_logp(atom_types=a["atom_types"], x_coords=a["x_coords"], x_velocs=a["x_velocs"],
                                                  adj_list=None, edge_batch_idx=None, masked_elements=a["masked"],
                                                  num_samples=S, z_coords=zc, z_velocs=zv)
    yc, yv, lp = call()
    yc2, yv2, lp2 = call()
    assert torch.equal(yc, yc2) and torch.equal(yv, yv2) and torch.equal(lp, lp2)
    assert torch.isfinite(yc).all() and torch.isfinite(lp).all()
    rep = lambda t: t.expand(S, *t.shape[1:]).contiguous()
    ll = m.log_likelihood(atom_types=rep(a["atom_types"]), x_coords=rep(a["x_coords"]), x_velocs=rep(a["x_velocs"]),
                          y_coords=yc[:, 0], y_velocs=yv[:, 0], adj_list=None, edge_batch_idx=None, masked_elements=rep(a["masked"]))
    err = float((ll - lp[:, 0]).abs().max() / lp.abs().max())
    print("dense fast mode: |log p(forward) - log p(reverse)| / max |log p| =", f"{err:.2e}")
    assert err < 5e-3


@pytest.mark.parametrize("flag,layout", [(131072, "wide"), (65536, "64-token")])
def test_wide_layout_v60_golden_measured_error(flag, layout):
    """BASELINE config 3's molecule size on the fast path: the 60-atom vectors from the reference on the wide layout (three
    molecules per workgroup, cross-wave mixing) and on the 64-token build (one molecule per wave; tools/gen_h3_*_asm.py --h1
    --nt=4), each forced by its debug bit, at the measured error - the same class as on alanine dipeptide."""
    from timewarp_amd import _lib

    d, _ = H.load("kernel_full_v60")
    lib = _lib.load()
    try:
        lib.tw_debug_set_flags(flag)
        m = H.tw_kernel_model(H.full_kernel_sd(), path=H1)
        out = H.run_model_case(m, d)
    finally:
        lib.tw_debug_set_flags(0)
    e = _errors(out, d)
    print(f"h1 ({layout}) vs the 60-atom reference vectors:", {k: f"{v:.2e}" for k, v in e.items()})
    for k, bar in BARS.items():
        assert e[k] < (4e-4 if k == "loglik" else bar), (k, e[k], bar)   # measured: loglik 1.4e-4, the rest as on alanine dipeptide
    assert e["s_y_coords"] > 1e-5


class _Rec:
    """Wrappers that keep what one iteration of oracle/mh_oracle.sample_with_model computes for ALL S proposals (its
    ChainStats only hold the rows up to the first accepted one)."""

    def __init__(self, model, energy, noise):
        self.m, self.e, self.n = model, energy, noise
        self.kbT = energy.kbT
        self.energies, self.u = [], None

    # model
    def scales(self):
        return self.m.scales()

    def conditional_sample_with_logp(self, at, xc, xv, mk, zc, zv):
        self.x_v = xv
        out = self.m.conditional_sample_with_logp(at, xc, xv, mk, zc, zv)
        self.y_c, self.y_v, self.p_xy = out
        return out

    def log_likelihood(self, *a):
        self.p_yx = self.m.log_likelihood(*a)
        return self.p_yx

    # energy
    def __call__(self, coords):
        self.energies.append(self.e(coords))
        return self.energies[-1]

    # noise
    def randn_like(self, t):
        return self.n.randn_like(t)

    def latents(self, *a):
        return self.n.latents(*a)

    def uniform(self, S):
        self.u = self.n.uniform(S)
        return self.u


@pytest.mark.parametrize("random_velocs,kind", [(True, "kernel"), (False, "kernel"), (True, "dense")])
def test_mh_iterations_vs_oracle_indicator_agreement(random_velocs, kind):
    """Whole MH iterations (tw_mh_iteration: fast flow kernel + AMBER energy kernel, 64 proposals) against the oracle loop
    on shared host noise, weights whose coupling nets move the proposals.  The oracle runs fp32: a proposal whose
    acceptance probability lies within the fast path's noise of the uniform draw may flip, and after the first flip the
    two chains are different chains.  So every iteration starts FROM THE ORACLE'S STATE (one-iteration chains along the
    reference chain), all 64 proposals of it are compared - exponent of the acceptance ratio and both log-densities at the
    measured error - and the agreement of the accept indicators is REPORTED and bounded from below (98 %), not asserted
    bit for bit."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain

    S, n_iter = 64, 6
    if kind == "dense":   # transformer_nvp on its fast path (MLP sections single-MFMA), same scaling of the output layers
        sd = dict(H.full_dense_sd())
        for k in sd:
            if ".out_mlp._layers.2." in k:
                sd[k] = sd[k] * 1e-4
        sd["coords_prior_log_scale"], sd["velocs_prior_log_scale"] = torch.tensor(-7.0), torch.tensor(0.0)
        spec = H.FULL_DENSE_SPEC
    else:
        sd, spec = H.mh_state_dict("scaled", random_velocs), H.FULL_KERNEL_SPEC
    types, coords, masses = synthetic.alanine_dipeptide_state()
    kw = dict(accept=True, num_proposal_steps=S)
    if random_velocs:
        kw.update(random_velocs=True, resample_velocs=True)
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    model = H.tw_dense_model(sd, path=H1) if kind == "dense" else H.tw_kernel_model(sd, path=H1)
    mask = torch.zeros(1, 22, dtype=torch.bool)
    x_c = coords[None].clone()
    x_v = torch.randn(1, 22, 3, generator=torch.Generator().manual_seed(9)) * 0.05
    agree = total = accepted_ref = 0
    worst = dict(p_xy=0.0, p_yx=0.0, exponent=0.0)
    for it in range(n_iter):
        rec = _Rec(mo.OracleModel(sd, spec), H.OracleAmberEnergy(energy.tables), H.HostNoise(100 + it))
        rc, rv, racc, _ = mo.sample_with_model(types[None], x_c, x_v, mask, rec, rec, masses, 1, rec, **kw)
        kbT = rec.kbT
        e_x, e_y = (t.squeeze(-1) / kbT for t in rec.energies[:2])
        kin = lambda v: mo.compute_kinetic_energy(v, masses, random_velocs, kbT)
        ex_ref = (e_y - e_x) + (kin(rec.y_v.squeeze(1)) - kin(rec.x_v.repeat(S, 1, 1))) + rec.p_xy.reshape(S) - rec.p_yx
        ind_ref = rec.u < torch.clamp(torch.exp(-ex_ref), max=1.0)
        chain = MetropolisHastingsChain(single_state_batch("ad", types, x_c[0], x_v[0]), model, torch.device("cuda"), energy,
                                        masses, noise=H.HostNoise(100 + it, "cuda"), **kw)
        assert chain._fused
        out = chain._compute()
        per = dict(out[6])
        H.assert_not_demoted(model)
        ind = out[5].bool().cpu()
        agree += int((ind == ind_ref).sum())
        total += S
        accepted_ref += int(racc)
        worst["p_xy"] = max(worst["p_xy"], float((per["pxy"].cpu() - rec.p_xy.reshape(S)).abs().max()))
        worst["p_yx"] = max(worst["p_yx"], float((per["pyx"].cpu() - rec.p_yx).abs().max()))
        worst["exponent"] = max(worst["exponent"], float((per["exp"].cpu() - ex_ref).abs().max()))
        x_c, x_v = torch.from_numpy(rc[-1:]), torch.from_numpy(rv[-1:])   # follow the ORACLE's chain
    print(f"h1 MH iterations vs oracle ({kind}, random_velocs={random_velocs}): indicator agreement {agree}/{total} over all proposals "
          f"({accepted_ref} of {n_iter} reference iterations accepted), max abs deviation log p(y|x) {worst['p_xy']:.3e}, "
          f"log p(x|y) {worst['p_yx']:.3e}, exponent {worst['exponent']:.3e}")
    assert accepted_ref >= 1 and agree >= 0.98 * total
    # measured (profiles/r04_h1_accuracy.txt): 9.2e-5, 4.7e-3, 4.6e-3 - the weights of these chains move a proposal by ~1e-4 nm
    assert worst["p_xy"] < 2e-3 and worst["p_yx"] < 0.03 and worst["exponent"] < 0.03


@pytest.mark.parametrize("V,lens", [(70, [70, 44, 70] * 40), (176, [176, 150] * 30)])
def test_fast_mode_wide_statements_repeat_bit_for_bit(V, lens):
    """r04: the attention statements prefetch the score fragments of the "next head" also in the last head; until the statement
    waited for those loads at its exit, one that returned late wrote into registers the code behind had taken over - ~0.5 % of
    the fast mode's launches on the three- and six-group wide statements returned one corrupted workgroup (two 24-MFMA stages
    behind the loads are ~800 cycles; tools/stress_wide.py, profiles/r04_stress_wide*.txt).  Forty repeats with the caches
    flushed in between must return the first run's bits (a smoke alarm, not a proof: the stress tool runs thousands)."""
    sd = H.full_kernel_sd()
    g = torch.Generator().manual_seed(31 + V)
    B = len(lens)
    at = torch.randint(0, 5, (B, V), generator=g)
    xc = torch.randn(B, V, 3, generator=g) * 0.5
    xv = torch.randn(B, V, 3, generator=g) * 0.5
    yc = xc + torch.randn(B, V, 3, generator=g) * 0.02
    yv = torch.randn(B, V, 3, generator=g) * 0.5
    mk = torch.zeros(B, V, dtype=torch.bool)
    for i, n in enumerate(lens):
        mk[i, n:] = True
    args = dict(atom_types=at.cuda(), x_coords=xc.cuda(), x_velocs=xv.cuda(), y_coords=yc.cuda(), y_velocs=yv.cuda(),
                adj_list=None, edge_batch_idx=None, masked_elements=mk.cuda())
    m = H.tw_kernel_model(sd, path=H1)
    first = m.log_likelihood(**args).cpu()
    assert torch.isfinite(first).all()
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB: evicts L2 and the Infinity Cache
    for it in range(40):
        junk.fill_(float(it))
        assert torch.equal(m.log_likelihood(**args).cpu(), first), it
