"""r05: the encoder stack as ONE generated asm statement for the kernel families that still ran compiled glue between their
asm sections (and spilled 200-528 bytes per lane around it): the 64-token build (49-64 atoms, BASELINE configs[3]), the wide
layout, the dense softmax kernel.  Each statement against the oracle, against the per-section build of the same kernel
(tw_debug_set_flags bit 12) and against itself (a repeated run must be bit-identical)."""
import pytest
import torch

from oracle import flow_oracle as fo
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-5
H3, H1 = 3, 4
NT4, PER_SECTION = 65536, 4096


def _ragged(V, lens, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.5
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b, n in enumerate(lens):
        mask[b, n:] = True
    return g, at, x_c, x_v, y_c, y_v, mask


def _loglik(m, at, x_c, x_v, y_c, y_v, mask):
    return m.log_likelihood(atom_types=at.cuda(), x_coords=x_c.cuda(), x_velocs=x_v.cuda(), y_coords=y_c.cuda(), y_velocs=y_v.cuda(),
                            adj_list=None, edge_batch_idx=None, masked_elements=mask.cuda()).cpu()


def _sample(m, at, x_c, x_v, mask, zc, zv):
    yc, yv, lp = m.conditional_sample_with_logp(atom_types=at.cuda(), x_coords=x_c.cuda(), x_velocs=x_v.cuda(), adj_list=None,
                                                edge_batch_idx=None, masked_elements=mask.cuda(), num_samples=zc.shape[0],
                                                z_coords=zc.cuda(), z_velocs=zv.cuda())
    return yc.cpu(), yv.cpu(), lp.cpu()


@pytest.mark.parametrize("V,lens", [(49, [49, 49, 40, 49, 49, 49, 49, 31, 49]), (60, [60, 60, 60, 60, 51, 60, 60, 60, 60, 60, 42]),
                                    (64, [64, 64, 50, 64, 64, 64])])
def test_64_token_encoder_stack_statement(V, lens):
    """tools/gen_h3_enc_asm.py --nt=4: forward pass on a ragged batch of more than one workgroup (padding tokens in the last
    one, two or three token tiles) and the reverse pass of one conditioning state, against the oracle and the per-section build."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 1500 + V)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    S = 9
    zc, zv = fo.draw_latents(sd, S, (1, V, 3), g)
    rs = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)
    m = H.tw_kernel_model(sd, path=H3)

    def run():
        return (_loglik(m, at, x_c, x_v, y_c, y_v, mask),) + _sample(m, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)

    try:
        lib.tw_debug_set_flags(NT4)
        stack, again = run(), run()
        lib.tw_debug_set_flags(NT4 | PER_SECTION)
        sections = run()
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    for a, b in zip(stack, again):
        assert torch.equal(a, b)
    assert not torch.equal(stack[0], sections[0])   # two different kernels did run
    keep = ~mask[0]
    for name, out in (("stack", stack), ("sections", sections)):
        errs = (H.rel_err(out[0], ref), H.rel_err(out[1][:, :, keep], rs[0][:, :, keep]), H.rel_err(out[2][:, :, keep], rs[1][:, :, keep]),
                H.rel_err(out[3], rs[2]))
        print(f"64-token {name} vs oracle, V = {V}:", errs)
        assert max(errs) < TOL, (name, errs)


@pytest.mark.parametrize("n_layers", [1, 2, 4])
def test_64_token_encoder_stack_layer_counts(n_layers):
    """The layer loop (scales, side blocks, score fragments advancing per layer) lives inside the statement."""
    from timewarp_amd import _lib

    lib = _lib.load()
    spec = fo.FlowSpec(variant="kernel", num_transformer_layers=n_layers, num_coupling_layers=2)
    sd = fo.synth_state_dict(fo.make_template(spec), 0)
    V, lens = 52, [52, 52, 45, 52, 52]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 1600 + n_layers)
    ref = fo.log_likelihood(sd, spec, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H3, n_coupling=2, n_layers=n_layers)
    try:
        lib.tw_debug_set_flags(NT4)
        out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < TOL, H.rel_err(out, ref)


def test_64_token_encoder_stack_chebyshev_fragments_per_layer():
    """chebyshev_kernel: one score-fragment set per (net, layer) - the statement advances its fragment pointer (an SGPR pair
    in this build) by the variant stride per layer."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_cheb_sd()
    V, lens = 52, [52, 47, 52, 52, 52, 39]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 1700)
    ref = fo.log_likelihood(sd, H.FULL_CHEB_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H3, attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=True)
    try:
        lib.tw_debug_set_flags(NT4)
        out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        lib.tw_debug_set_flags(NT4 | PER_SECTION)
        sec = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < 2e-5 and H.rel_err(sec, ref) < 2e-5, (H.rel_err(out, ref), H.rel_err(sec, ref))
    assert H.rel_err(out, sec) < 5e-6, H.rel_err(out, sec)


def test_64_token_encoder_stack_fast_mode():
    """The single-MFMA form of the same statement (tw_h1n4_enc_asm.inc): NOT a parity path - held to the per-section fast
    build at the fast mode's own noise and to the oracle at the measured deviation of that mode (tests/test_flow_h1_gpu.py)."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    V, lens = 60, [60, 60, 60, 60, 51, 60, 60, 60, 60]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 1800)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H1)
    try:
        lib.tw_debug_set_flags(NT4)
        stack, again = _loglik(m, at, x_c, x_v, y_c, y_v, mask), _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        lib.tw_debug_set_flags(NT4 | PER_SECTION)
        sections = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    assert torch.equal(stack, again)
    e_ref, e_sec = H.rel_err(stack, ref), H.rel_err(stack, sections)
    print("fast mode, 64-token statement: vs oracle", e_ref, "vs per-section build", e_sec, "per-section vs oracle", H.rel_err(sections, ref))
    assert e_ref < 1.5e-3 and e_sec < 1.5e-3, (e_ref, e_sec)
    assert e_ref > 1e-6   # it is the single-MFMA arithmetic


# ---- the wide layout (25-192 atoms: molecules packed over the workgroup's 192 token slots) ----
ALWAYS_WIDE, FIVE_GROUPS_ONLY = 32768, 262144


@pytest.mark.parametrize("V,lens,flags", [
    (30, [30, 28, 30, 30, 25, 30, 30, 30], ALWAYS_WIDE),           # six molecules per workgroup, five-group windows
    (48, [48, 40, 48, 48, 48], ALWAYS_WIDE),                        # four per workgroup: every molecule is one wave's tokens
    (65, [65, 65, 50, 65, 65], 0),                                  # 96-slot stride, three-group windows (tw_h3w3_enc_asm.inc)
    (88, [88, 88, 61, 88, 88], 0),
    (70, [70, 44, 70], FIVE_GROUPS_ONLY),                           # the same sizes on the five-group statement
    (100, [100, 87, 100], 0),                                       # one molecule per workgroup, padding waves
    (150, [150, 131], 0),
    (176, [176, 176, 133], 0),                                      # six-group windows (tw_h3w6_enc_asm.inc)
    (192, [192, 161, 192], 0)])
def test_wide_encoder_stack_statement(V, lens, flags):
    """tools/gen_h3_enc_asm.py --wide [--ng=3|6]: forward pass on a ragged batch of more than one workgroup and the reverse pass of
    one conditioning state, against the oracle, against the per-section build (bit 12) and against itself."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 2500 + V)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    S = 7
    zc, zv = fo.draw_latents(sd, S, (1, V, 3), g)
    rs = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)
    m = H.tw_kernel_model(sd, path=H3)

    def run():
        return (_loglik(m, at, x_c, x_v, y_c, y_v, mask),) + _sample(m, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)

    try:
        lib.tw_debug_set_flags(flags)
        stack, again = run(), run()
        lib.tw_debug_set_flags(flags | PER_SECTION)
        sections = run()
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    for a, b in zip(stack, again):
        assert torch.equal(a, b)
    assert not torch.equal(stack[0], sections[0])   # two different kernels did run
    keep = ~mask[0]
    for name, out in (("stack", stack), ("sections", sections)):
        errs = (H.rel_err(out[0], ref), H.rel_err(out[1][:, :, keep], rs[0][:, :, keep]), H.rel_err(out[2][:, :, keep], rs[1][:, :, keep]),
                H.rel_err(out[3], rs[2]))
        print(f"wide {name} vs oracle, V = {V}:", errs)
        assert max(errs) < TOL, (name, errs)


@pytest.mark.parametrize("n_layers,V,lens", [(1, 72, [72, 72, 60]), (2, 110, [110, 95]), (4, 36, [36, 36, 36, 30, 36, 36, 36])])
def test_wide_encoder_stack_layer_counts(n_layers, V, lens):
    from timewarp_amd import _lib

    lib = _lib.load()
    spec = fo.FlowSpec(variant="kernel", num_transformer_layers=n_layers, num_coupling_layers=2)
    sd = fo.synth_state_dict(fo.make_template(spec), 0)
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 2600 + n_layers)
    ref = fo.log_likelihood(sd, spec, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H3, n_coupling=2, n_layers=n_layers)
    try:
        lib.tw_debug_set_flags(ALWAYS_WIDE)
        out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < TOL, H.rel_err(out, ref)


def test_wide_encoder_stack_chebyshev_fragments_per_layer():
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_cheb_sd()
    V, lens = 90, [90, 77, 90]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 2700)
    ref = fo.log_likelihood(sd, H.FULL_CHEB_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H3, attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=True)
    try:
        out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        lib.tw_debug_set_flags(PER_SECTION)
        sec = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < 2e-5 and H.rel_err(sec, ref) < 2e-5, (H.rel_err(out, ref), H.rel_err(sec, ref))
    assert H.rel_err(out, sec) < 5e-6, H.rel_err(out, sec)


@pytest.mark.parametrize("V,lens,flags", [(40, [40, 40, 33, 40, 40, 40], ALWAYS_WIDE), (65, [65, 65, 50, 65, 65], 0), (120, [120, 99, 120], 0),
                                          (180, [180, 150], 0)])
def test_wide_encoder_stack_fast_mode(V, lens, flags):
    """tw_h1w{,3,6}_enc_asm.inc: NOT a parity path - held to the per-section fast build and to the oracle at that mode's deviation."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 2800 + V)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H1)
    try:
        lib.tw_debug_set_flags(flags)
        stack, again = _loglik(m, at, x_c, x_v, y_c, y_v, mask), _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        lib.tw_debug_set_flags(flags | PER_SECTION)
        sections = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    assert torch.equal(stack, again)
    e_ref, e_sec = H.rel_err(stack, ref), H.rel_err(stack, sections)
    print(f"fast mode, wide statement, V = {V}: vs oracle", e_ref, "vs per-section build", e_sec, "per-section vs oracle", H.rel_err(sections, ref))
    assert e_ref < 1.5e-3 and e_sec < 1.5e-3, (e_ref, e_sec)
    assert e_ref > 1e-6   # it is the single-MFMA arithmetic


# ---- the dense softmax model (transformer_nvp; BASELINE configs[4]) ----
def _dense_case(sd, spec, V, lens, seed, S=7):
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, seed)
    ref = fo.log_likelihood(sd, spec, at, x_c, x_v, y_c, y_v, mask)
    zc, zv = fo.draw_latents(sd, S, (1, V, 3), g)
    rs = fo.conditional_sample_with_logp(sd, spec, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)
    return (at, x_c, x_v, y_c, y_v, mask, zc, zv), ref, rs


@pytest.mark.parametrize("V,lens", [(22, [22, 20, 22, 17, 22, 22, 22, 22, 22]), (7, [7, 5, 6, 7, 7, 3, 7] * 5), (16, [16, 13, 16, 16, 16, 10, 16] * 2),
                                    (30, [30, 28, 25, 30, 30]), (48, [48, 40, 33, 48, 48]),
                                    # r06: 64-token waves (tools/gen_h3_enc_asm.py --dense --nt=4: the softmax block in two query halves)
                                    (49, [49, 49, 41, 49, 49, 33]), (60, [60, 52, 60, 60, 57, 60]), (64, [64, 64, 50, 64, 64])])
def test_dense_encoder_stack_statement(V, lens):
    """tools/gen_h3_enc_asm.py --dense [--nt=4]: ragged forward pass over more than one workgroup and the reverse pass of one
    conditioning state, against the oracle, the per-section build (bit 12) and itself."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_dense_sd()
    (at, x_c, x_v, y_c, y_v, mask, zc, zv), ref, rs = _dense_case(sd, H.FULL_DENSE_SPEC, V, lens, 3500 + V)
    m = H.tw_dense_model(sd, path=H3)

    def run():
        return (_loglik(m, at, x_c, x_v, y_c, y_v, mask),) + _sample(m, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)

    try:
        stack, again = run(), run()
        lib.tw_debug_set_flags(PER_SECTION)
        sections = run()
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    for a, b in zip(stack, again):
        assert torch.equal(a, b)
    assert not torch.equal(stack[0], sections[0])   # two different kernels did run
    keep = ~mask[0]
    for name, out in (("stack", stack), ("sections", sections)):
        errs = (H.rel_err(out[0], ref), H.rel_err(out[1][:, :, keep], rs[0][:, :, keep]), H.rel_err(out[2][:, :, keep], rs[1][:, :, keep]),
                H.rel_err(out[3], rs[2]))
        print(f"dense {name} vs oracle, V = {V}:", errs)
        assert max(errs) < TOL, (name, errs)


def test_dense_encoder_stack_position_features():
    """transformer_nvp_posenc.yaml: the in-MLP of the 128 random Fourier features stays compiled C++ in front of the statement.
    Against the reference's own vectors (tests/golden/dense_posenc_full_ad.npz), the statement and the per-section build."""
    from timewarp_amd import _lib

    lib = _lib.load()
    d, _ = H.load("dense_posenc_full_ad")
    m = H.tw_dense_model(H.full_dense_posenc_sd(), rff_dim=128, path=H3)
    try:
        out = H.run_model_case(m, d)
        lib.tw_debug_set_flags(PER_SECTION)
        sec = H.run_model_case(m, d)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    H.assert_case_close(out, d, tol=TOL)
    H.assert_case_close(sec, d, tol=TOL)
    assert any(not torch.equal(torch.as_tensor(out[k]), torch.as_tensor(sec[k])) for k in out)   # two different kernels did run


@pytest.mark.parametrize("n_layers", [1, 2, 4])
def test_dense_encoder_stack_layer_counts(n_layers):
    """The double-buffered side blocks and out_proj's scale a layer ahead: one, an even and more layers than buffers."""
    spec = fo.FlowSpec(variant="dense", num_transformer_layers=n_layers, num_coupling_layers=2)
    sd = fo.synth_state_dict(fo.make_template(spec), 0)
    V, lens = 20, [20, 18, 20, 20, 20, 11, 20, 20, 20]
    (at, x_c, x_v, y_c, y_v, mask, zc, zv), ref, rs = _dense_case(sd, spec, V, lens, 3700 + n_layers)
    m = H.tw_dense_model(sd, path=H3, n_coupling=2, n_layers=n_layers)
    out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < TOL, H.rel_err(out, ref)


def test_dense_encoder_stack_fast_mode():
    """tw_h1d_enc_asm.inc (MLP sections single-MFMA, the softmax attention block in split form): NOT a parity path."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_dense_sd()
    V, lens = 22, [22, 20, 22, 17, 22, 22, 22, 22, 22]
    (at, x_c, x_v, y_c, y_v, mask, zc, zv), ref, rs = _dense_case(sd, H.FULL_DENSE_SPEC, V, lens, 3800)
    m = H.tw_dense_model(sd, path=H1)
    try:
        stack, again = _loglik(m, at, x_c, x_v, y_c, y_v, mask), _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        lib.tw_debug_set_flags(PER_SECTION)
        sections = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    assert torch.equal(stack, again)
    e_ref, e_sec = H.rel_err(stack, ref), H.rel_err(stack, sections)
    print("fast mode, dense statement: vs oracle", e_ref, "vs per-section build", e_sec, "per-section vs oracle", H.rel_err(sections, ref))
    assert e_ref < 1.5e-3 and e_sec < 1.5e-3, (e_ref, e_sec)
    assert e_ref > 1e-6


# ---- the paired 64-token layout (97-128 atoms: one molecule per pair of waves, two per workgroup) ----
NO_PAIR = 1048576


@pytest.mark.parametrize("V,lens", [(97, [97, 97, 80, 97, 97]), (100, [100, 87, 100]), (112, [112, 112, 112, 99]), (128, [128, 128, 101, 128, 128])])
def test_paired_64_token_layout_vs_oracle_and_wide_layout(V, lens):
    """tools/gen_h3_enc_asm.py --nt=4 --pair: ragged forward pass (odd and even row counts: the last workgroup holds one molecule
    or two) and the reverse pass of one conditioning state, against the oracle, against the 48-token wide layout (bit 20: one
    molecule per workgroup, its own statement and fragment order) and against itself."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 4500 + V)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    S = 7
    zc, zv = fo.draw_latents(sd, S, (1, V, 3), g)
    rs = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)
    m = H.tw_kernel_model(sd, path=H3)

    def run():
        out = (_loglik(m, at, x_c, x_v, y_c, y_v, mask),) + _sample(m, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)
        return out, lib.tw_last_netblock_kernel().decode()

    try:
        (paired, k_pair), (again, _) = run(), run()
        lib.tw_debug_set_flags(NO_PAIR)
        wide, k_wide = run()
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert k_pair == "tw::netblock_h3_kernel<4, true, false, true, false, true, false, false>", k_pair
    assert k_wide == "tw::netblock_h3_kernel<3, true, false, true, false, true, false, false>", k_wide
    for a, b in zip(paired, again):
        assert torch.equal(a, b)
    keep = ~mask[0]
    for name, out in (("paired", paired), ("wide", wide)):
        errs = (H.rel_err(out[0], ref), H.rel_err(out[1][:, :, keep], rs[0][:, :, keep]), H.rel_err(out[2][:, :, keep], rs[1][:, :, keep]),
                H.rel_err(out[3], rs[2]))
        print(f"{name} vs oracle, V = {V}:", errs)
        assert max(errs) < TOL, (name, errs)


def test_paired_64_token_layout_chebyshev_fragments_per_layer():
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_cheb_sd()
    V, lens = 104, [104, 91, 104]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 4700)
    ref = fo.log_likelihood(sd, H.FULL_CHEB_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H3, attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=True)
    try:
        out = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        assert lib.tw_last_netblock_kernel().decode().startswith("tw::netblock_h3_kernel<4, true, false, true")
        lib.tw_debug_set_flags(NO_PAIR)
        wide = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    assert H.rel_err(out, ref) < 2e-5 and H.rel_err(wide, ref) < 2e-5, (H.rel_err(out, ref), H.rel_err(wide, ref))
    assert H.rel_err(out, wide) < 5e-6, H.rel_err(out, wide)


def test_paired_64_token_layout_fast_mode():
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_kernel_sd()
    V, lens = 120, [120, 99, 120]
    g, at, x_c, x_v, y_c, y_v, mask = _ragged(V, lens, 4800)
    ref = fo.log_likelihood(sd, H.FULL_KERNEL_SPEC, at, x_c, x_v, y_c, y_v, mask)
    m = H.tw_kernel_model(sd, path=H1)
    try:
        paired, again = _loglik(m, at, x_c, x_v, y_c, y_v, mask), _loglik(m, at, x_c, x_v, y_c, y_v, mask)
        assert lib.tw_last_netblock_kernel().decode() == "tw::netblock_h3_kernel<4, true, false, true, false, true, true, false>"
        lib.tw_debug_set_flags(NO_PAIR)
        wide = _loglik(m, at, x_c, x_v, y_c, y_v, mask)
    finally:
        lib.tw_debug_set_flags(0)
    assert torch.equal(paired, again)
    e_ref, e_wide = H.rel_err(paired, ref), H.rel_err(paired, wide)
    print("fast mode, paired layout: vs oracle", e_ref, "vs wide layout", e_wide)
    assert e_ref < 1.5e-3 and e_wide < 1.5e-3 and e_ref > 1e-6, (e_ref, e_wide)


# ---- the dense softmax model on 64-token waves (49-64 atoms): r06 the encoder-stack statement (softmax block in two query halves);
# ---- the per-section build (asm MLP sections, compiled-C++ attention block) behind tw_debug_set_flags bit 12 ----
@pytest.mark.parametrize("V,lens", [(52, [52, 52, 41, 52, 52]), (61, [61, 61, 61, 48, 61, 61, 61]), (64, [64, 50])])
def test_dense_model_on_64_token_waves_vs_oracle(V, lens):
    """r05: the dense model above 48 atoms ran the exact-f32 fused kernel (3x slower) on the default path until now.  Forward pass
    on a ragged batch of more than one workgroup and the reverse pass of one conditioning state against the oracle; the instantiation
    that ran; a repeated run bit-identical; the fast mode (which has no such build) falls back per call."""
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_dense_sd()
    (at, x_c, x_v, y_c, y_v, mask, zc, zv), ref, rs = _dense_case(sd, H.FULL_DENSE_SPEC, V, lens, 5500 + V, S=6)
    m = H.tw_dense_model(sd, path=None)           # the constructor's default must land on the split-fp16 kernel
    assert m._path_for(V) == H3

    def run():
        return (_loglik(m, at, x_c, x_v, y_c, y_v, mask),) + _sample(m, at[:1], x_c[:1], x_v[:1], mask[:1], zc, zv)

    out, again = run(), run()
    assert lib.tw_last_netblock_kernel().decode() == "tw::netblock_h3_kernel<4, true, true, false, false, true, false, false>"   # ENC since r06
    try:
        lib.tw_debug_set_flags(PER_SECTION)
        sections = run()
        assert lib.tw_last_netblock_kernel().decode() == "tw::netblock_h3_kernel<4, true, true, false, false, false, false, false>"
    finally:
        lib.tw_debug_set_flags(0)
    H.assert_not_demoted(m)
    for a, b in zip(out, again):
        assert torch.equal(a, b)
    assert H.rel_err(sections[0], ref) < TOL and not torch.equal(sections[0], out[0])   # the r05 build: still at the bar, a different kernel
    keep = ~mask[0]
    errs = (H.rel_err(out[0], ref), H.rel_err(out[1][:, :, keep], rs[0][:, :, keep]), H.rel_err(out[2][:, :, keep], rs[1][:, :, keep]),
            H.rel_err(out[3], rs[2]))
    print(f"dense, 64-token waves, V = {V}:", errs)
    assert max(errs) < TOL, errs
    fast = H.tw_dense_model(sd, path=H1)
    with pytest.raises(RuntimeError, match="unsupported"):
        _loglik(fast, at, x_c, x_v, y_c, y_v, mask)
