"""Bit-for-bit repeatability under disturbance, per kernel family x {split-fp16, fast mode} x {forward, reverse pass}: the same
inputs N times with the caches flushed and another kernel family's bytes left in the LDS in between.  r04 found - with the
manual tools/stress_all.py, not with a test - a score-fragment prefetch that could land after an asm statement's exit in EVERY
attention statement (commit 14f66a9: 6 of 1200 launches returned a corrupted workgroup); this is that tool cut down to the
suite's time budget.  One pass = 16 net-block launches (8 coupling layers x forward or reverse), so N = 40 passes are 640
launches of the family's kernel."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

N = 40
ALWAYS_WIDE = 32768

# (label, dense, atoms, rows of the forward batch / proposals of the reverse pass, debug flags)
FAMILIES = [
    ("48-token waves, several molecules (alanine dipeptide)", False, 22, 600, 0),
    ("48-token waves, one molecule each", False, 44, 300, 16384),
    ("64-token waves", False, 60, 256, 65536),
    ("wide layout, five-group windows", False, 30, 384, ALWAYS_WIDE),
    ("wide layout, three-group windows", False, 65, 192, 0),
    ("paired 64-token waves (one molecule per pair of waves)", False, 110, 128, 0),
    ("wide layout, one molecule per workgroup", False, 140, 96, 0),
    ("wide layout, six-group windows", False, 176, 96, 0),
    ("dense softmax model", True, 22, 600, 0),
    ("dense softmax model, 64-token waves", True, 60, 256, 0),
]


def _batch(V, B, seed):
    g = torch.Generator().manual_seed(seed)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * (0.3 if V <= 30 else 0.5)
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b in range(0, B, 3):
        mask[b, V - 1 - (b % 5):] = True
    return [t.cuda() for t in (at, x_c, x_v, y_c, y_v, mask)]


@pytest.fixture(scope="module")
def disturb():
    junk = torch.empty(1 << 26, dtype=torch.float32, device="cuda")       # 256 MiB: L2 and Infinity Cache turn over
    other = H.tw_kernel_model(H.full_kernel_sd(), path=1)                 # the exact-f32 fused kernel: another LDS layout
    at, x_c, x_v, y_c, y_v, mask = _batch(22, 64, 1)

    def run(it):
        junk.fill_(float(it))
        other.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, adj_list=None,
                             edge_batch_idx=None, masked_elements=mask)
    return run


@pytest.mark.parametrize("path", [3, 4], ids=["split-fp16", "fast-mode"])
@pytest.mark.parametrize("label,dense,V,B,flags", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_repeated_launches_are_bit_identical(disturb, label, dense, V, B, flags, path):
    from timewarp_amd import _lib

    lib = _lib.load()
    sd = H.full_dense_sd() if dense else H.full_kernel_sd()
    at, x_c, x_v, y_c, y_v, mask = _batch(V, B, 11 + V)
    g = torch.Generator().manual_seed(5)
    zc, zv = torch.randn(B, 1, V, 3, generator=g).cuda() * 0.1, torch.randn(B, 1, V, 3, generator=g).cuda()
    m = H.tw_dense_model(sd, path=path) if dense else H.tw_kernel_model(sd, path=path)
    import ctypes as C

    desc = m.dims.to_desc()
    if lib.tw_flow_path_supported(C.byref(desc), V, path) != 1:
        pytest.skip("this family has no build on this path (the dense model's 64-token waves: split-fp16 only)")

    def fwd():
        return m.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, adj_list=None,
                                edge_batch_idx=None, masked_elements=mask)

    def rev():
        return torch.cat([t.reshape(-1) for t in m.conditional_sample_with_logp(
            atom_types=at[:1], x_coords=x_c[:1], x_velocs=x_v[:1], adj_list=None, edge_batch_idx=None,
            masked_elements=mask[:1] & False, num_samples=B, z_coords=zc, z_velocs=zv)])

    try:
        lib.tw_debug_set_flags(flags)
        for what, fn in (("forward", fwd), ("reverse", rev)):
            first = fn().clone()
            assert bool(torch.isfinite(first).all()), (label, what)
            bad = 0
            for it in range(N):
                disturb(it)
                bad += int(not torch.equal(fn(), first))
            assert bad == 0, f"{label}, {what} pass: {bad} of {N} runs differ from the first"
    finally:
        lib.tw_debug_set_flags(0)
    if path == 3:
        H.assert_not_demoted(m)
