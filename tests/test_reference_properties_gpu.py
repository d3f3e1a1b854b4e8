"""The properties the REFERENCE's own tests pin on this path, held by the HIP path through the C ABI
(tests/test_reference_properties.py holds the oracle to the same ones on the CPU):
  * tests/test_kernel_attention.py:163-173  known-answer: rational Chebyshev expansion at 0.7 (five values from a Julia implementation)
  * tests/test_kernel_attention.py:19-46    normalised RBF scores sum to 1 over the keys (atol 1e-3)
  * tests/test_batching.py:132-177          batched log_likelihood == per-item (rtol = atol = 1e-4): padding-mask handling
  * tests/test_distributional_equivariance.py:24-67   p(Ty given Tx) = p(y given x) under a translation T
  * tests/test_losses.py:143-224            same seed => same stochastic result, another seed => another (here: the MH chain)
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_reference_properties import CHEB_AT_0_7, _ragged

pytestmark = pytest.mark.gpu

SIMPLE, FUSED, H3 = 2, 1, 3


def test_chebyshev_expansion_known_answer_through_the_scores_kernel():
    """Two atoms sqrt(0.7) lengthscales apart, unit coefficient vectors, no normalisation: the off-diagonal score of head h is
    R_h(0.7) - the reference's Julia values (its attention squares the scaled distance before expanding, kernel_attention.py:30)."""
    from timewarp_amd import _lib

    lib = _lib.load()
    H_, order = 5, 5
    x = torch.zeros(1, 2, 3)
    x[0, 1, 0] = float(np.sqrt(0.7))
    ls = torch.ones(H_)
    coeffs = torch.eye(H_, order)
    out = torch.empty(1, H_, 2, 2, device="cuda")
    mask = torch.zeros(1, 2, dtype=torch.uint8)
    xd, md, ld, cd = x.cuda(), mask.cuda(), ls.cuda(), coeffs.cuda()
    _lib.check(lib.tw_kernel_scores_cheb(xd.data_ptr(), md.data_ptr(), ld.data_ptr(), cd.data_ptr(), order, 0, H_, 1, 2, 0, 0,
                                         out.data_ptr(), None), "tw_kernel_scores_cheb")
    got = out.cpu()[0, :, 0, 1]
    assert torch.allclose(got, torch.tensor(CHEB_AT_0_7), rtol=1e-5, atol=1e-6), got
    # the diagonal: distance 0 -> x = 0 -> R_n(0) = T_n(-1) = (-1)^n
    assert torch.allclose(out.cpu()[0, :, 0, 0], torch.tensor([1.0, -1.0, 1.0, -1.0, 1.0]))


@pytest.mark.parametrize("V", [9, 22, 60])
def test_kernel_scores_are_normalised(V):
    from timewarp_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(V)
    B = 4
    x = 0.1 * torch.randn(B, V, 3, generator=g)
    mask = torch.zeros(B, V, dtype=torch.bool)
    mask[2, V - 3:] = True
    ls = torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])
    out = torch.empty(B, 6, V, V, device="cuda")
    xd, md, ld = x.cuda(), mask.to(torch.uint8).cuda(), ls.cuda()
    _lib.check(lib.tw_kernel_scores(xd.data_ptr(), md.data_ptr(), ld.data_ptr(), 6, B, V, 1, int(V > 25), out.data_ptr(), None),
               "tw_kernel_scores")
    a = out.cpu()
    assert a.shape == (B, 6, V, V)
    assert torch.allclose(a.sum(-1), torch.ones(B, 6, V), atol=1e-3)
    assert float(a[2, :, :, V - 3:].abs().max()) == 0.0


def _models():
    yield "kernel, exact-f32 kernel", H.tw_kernel_model(H.full_kernel_sd(), path=FUSED)
    yield "kernel, split-fp16 kernel", H.tw_kernel_model(H.full_kernel_sd(), path=H3)
    yield "kernel, per-op path", H.tw_kernel_model(H.full_kernel_sd(), path=SIMPLE)
    yield "dense, split-fp16 kernel", H.tw_dense_model(H.full_dense_sd(), path=H3)
    yield "chebyshev kernel, split-fp16 kernel", H.tw_kernel_model(H.full_cheb_sd(), path=H3, attention_type="chebyshev_kernel",
                                                                    cheb_order=6, force_asymptotic_zero=True)


def _ll(m, at, x_c, x_v, y_c, y_v, mask):
    return m.log_likelihood(atom_types=at.cuda(), x_coords=x_c.cuda(), x_velocs=x_v.cuda(), y_coords=y_c.cuda(), y_velocs=y_v.cuda(),
                            adj_list=None, edge_batch_idx=None, masked_elements=mask.cuda()).cpu()


def test_batched_log_likelihood_equals_per_item():
    """Full-size models on a ragged batch: every row of the batched call equals the call on that molecule alone, cut to its own
    length (the reference: rtol = atol = 1e-4 on tiny models; here 1e-5 relative)."""
    at, x_c, x_v, y_c, y_v, mask = _ragged(22, [22, 15, 22, 9, 18, 22], 7)
    for name, m in _models():
        batched = _ll(m, at, x_c, x_v, y_c, y_v, mask)
        for b in range(at.shape[0]):
            n = int((~mask[b]).sum())
            one = _ll(m, at[b:b + 1, :n], x_c[b:b + 1, :n], x_v[b:b + 1, :n], y_c[b:b + 1, :n], y_v[b:b + 1, :n], mask[b:b + 1, :n])
            assert abs(float(batched[b] - one[0])) < 1e-5 * abs(float(one[0])) + 1e-4, (name, b, batched[b], one[0])
        H.assert_not_demoted(m)


def test_translation_equivariance_of_the_conditional_density():
    at, x_c, x_v, y_c, y_v, mask = _ragged(22, [22, 17, 22, 22], 8)
    t = torch.tensor([1.5, -2.0, 0.7])
    for name, m in _models():
        a = _ll(m, at, x_c, x_v, y_c, y_v, mask)
        b = _ll(m, at, x_c + t, x_v, y_c + t, y_v, mask)
        # centring x + t in fp32 costs ~1e-7 relative of |t| = 2.6 on coordinates of O(0.4): held to the reference's 1e-4
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4), (name, (a - b).abs().max())


def test_same_seed_same_chain_other_seed_other_chain():
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, sample_with_model

    sd = H.mh_state_dict("scaled", True)
    model = H.tw_kernel_model(sd, path=H3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")

    def run(seed):
        return sample_with_model(single_state_batch("ad", types, coords), model, dev, energy, masses, 40, accept=True,
                                 num_proposal_steps=16, random_velocs=True, resample_velocs=True, noise=DeviceNoise(dev, seed=seed))

    a, b, c = run(42), run(42), run(43)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert all(np.array_equal(getattr(a[3], f), getattr(b[3], f)) for f in ("acceptance", "exponent", "p_xy", "p_yx", "energies_pot"))
    assert not np.array_equal(a[0], c[0])
    H.assert_not_demoted(model)
