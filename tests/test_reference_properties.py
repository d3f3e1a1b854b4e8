"""The properties the REFERENCE's own tests pin on this path, restated for the oracle on the CPU - the
product is held to the same ones through the C ABI in tests/test_reference_properties_gpu.py:
  * tests/test_kernel_attention.py:163-173  known-answer: rational Chebyshev expansion at 0.7 (five values from a Julia implementation)
  * tests/test_kernel_attention.py:19-46    normalised RBF scores sum to 1 over the keys (atol 1e-3), shape [B, H, Q, M]
  * tests/test_batching.py:132-177          batched log_likelihood == per-item (rtol = atol = 1e-4): padding-mask handling
  * tests/test_distributional_equivariance.py:24-67   p(Ty given Tx) = p(y given x) under a translation T
"""
import pytest
import torch

from oracle import flow_oracle as fo
from tests import helpers as H

# the reference's own vector (tests/test_kernel_attention.py:169-171): data, "Reference values from Julia implementation"
CHEB_AT_0_7 = [1.0, -0.17647058823529416, -0.9377162629757785, 0.507429269285569, 0.7586235796985188]


def test_chebyshev_expansion_known_answer():
    # The reference's test expands x = 0.7 directly (`chebyshev_expansion`); its attention squares the scaled distance first
    # (`chebyshev_basis_function`, kernel_attention.py:30), which is what the oracle restates: scaled = sqrt(0.7).
    scaled = torch.full((1, 5, 1, 1), 0.7).sqrt()               # [B, H, Q, M]; head h picks term h through unit coefficients
    out = fo.chebyshev_basis(scaled, torch.eye(5), False).flatten()
    assert torch.allclose(out, torch.tensor(CHEB_AT_0_7))       # torch.allclose defaults, as the reference's test


def test_kernel_scores_are_normalised():
    g = torch.Generator().manual_seed(0)
    x = 0.1 * torch.randn(7, 9, 3, generator=g)                 # (the reference's test: 0.1 * randn positions)
    mask = torch.zeros(7, 9, dtype=torch.bool)
    mask[2, 6:] = True
    a = fo.kernel_scores(x, mask, torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2]))
    assert a.shape == (7, 6, 9, 9)
    assert torch.allclose(a.sum(-1), torch.ones(7, 6, 9), atol=1e-3)
    assert float(a[2, :, :, 6:].abs().max()) == 0.0             # masked keys carry no weight


def _ragged(V, lens, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.4
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.05
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b, n in enumerate(lens):
        mask[b, n:] = True
    return at, x_c, x_v, y_c, y_v, mask


TINY = {"kernel": (H.TINY_KERNEL_SPEC, {}), "learnable_kernel": (H.TINY_LEARNABLE_SPEC, {}), "dense": (H.TINY_DENSE_SPEC, {})}


def _tiny_sd(name):
    spec, kw = TINY[name]
    sd = dict(fo.synth_state_dict(fo.make_template(spec, atom_embedding_dim=4, d_model=8, dim_feedforward=16, mlp_hidden=(8,),
                                                   lengthscales=(0.1, 0.5, 1.0), **kw), 3))
    if spec.attention_type == "learnable_kernel":   # kernel_attention.py:217-252: a parameter log_lengthscales instead of the buffer
        for k in [k for k in sd if k.endswith(".attention.lengthscales")]:
            g = torch.Generator().manual_seed(len(k))
            sd[k.replace("lengthscales", "log_lengthscales")] = torch.log(sd.pop(k)) + 0.1 * torch.randn(3, generator=g)
    return spec, sd


@pytest.mark.parametrize("name", sorted(TINY))
def test_batched_log_likelihood_equals_per_item(name):
    spec, sd = _tiny_sd(name)
    at, x_c, x_v, y_c, y_v, mask = _ragged(11, [11, 7, 11, 4, 9], 5)
    batched = fo.log_likelihood(sd, spec, at, x_c, x_v, y_c, y_v, mask)
    for b in range(5):
        n = int((~mask[b]).sum())
        one = fo.log_likelihood(sd, spec, at[b:b + 1, :n], x_c[b:b + 1, :n], x_v[b:b + 1, :n], y_c[b:b + 1, :n], y_v[b:b + 1, :n],
                                mask[b:b + 1, :n])
        assert torch.allclose(batched[b], one[0], rtol=1e-4, atol=1e-4), (b, batched[b], one[0])


@pytest.mark.parametrize("name", sorted(TINY))
def test_translation_equivariance_of_the_conditional_density(name):
    spec, sd = _tiny_sd(name)
    at, x_c, x_v, y_c, y_v, mask = _ragged(9, [9, 6, 9], 6)
    t = torch.tensor([1.5, -2.0, 0.7])
    a = fo.log_likelihood(sd, spec, at, x_c, x_v, y_c, y_v, mask)
    b = fo.log_likelihood(sd, spec, at, x_c + t, x_v, y_c + t, y_v, mask)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
