"""Flow-pass timings of the BASELINE configs that are parity cases rather than the bench line:
cfg 3 (kernel flow, ~60 atoms, 512 proposals: exact-f32 fused kernel with 64-token waves) and cfg 4 (dense softmax
transformer_nvp on alanine dipeptide: fused dense kernel and per-op path)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


g = torch.Generator().manual_seed(0)
# cfg 3
V, S = 60, 512
at = torch.randint(0, 5, (1, V), generator=g).cuda()
xc = (torch.randn(1, V, 3, generator=g) * 0.45).cuda()
xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
mk = torch.zeros(1, V, dtype=torch.bool).cuda()
flop = 16 * V * (4478976 + 4608 * V) * S
res3 = {}
for path, name in ((1, "fused-f32 (64-token waves)"), (3, "split-fp16, wide layout (3 molecules per workgroup)"),
                   (4, "FAST MODE (single fp16 MFMA, not a parity path), wide layout")):
    m = H.tw_kernel_model(H.full_kernel_sd(), path=path)
    res3[path] = timed(lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                              masked_elements=mk, num_samples=S))
    print(f"cfg3 kernel flow V={V} S={S} {name}: {res3[path]:.2f} ms per reverse pass, {flop / res3[path] / 1e9:.1f} TFLOP/s algorithmic")
print(f"  cfg3: split-fp16 / f32 speed-up {res3[1] / res3[3]:.2f}x")
# the same molecule at a proposal count that fills whole rounds of the chip (768 = 256 workgroups x 3 molecules per net... x 2 nets = 2 rounds)
for S2 in (384, 768):
    flop2 = 16 * V * (4478976 + 4608 * V) * S2
    m = H.tw_kernel_model(H.full_kernel_sd(), path=3)
    ms = timed(lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                      masked_elements=mk, num_samples=S2))
    print(f"  V={V} S={S2} split-fp16 wide: {ms:.2f} ms per reverse pass, {flop2 / ms / 1e9:.1f} TFLOP/s algorithmic")
# molecule sizes between the bench case and cfg 3: the split-fp16 kernel takes every molecule that fits a 48-token wave
for V, S in ((17, 1000), (30, 1000), (40, 1000), (48, 1000), (64, 1000), (100, 500)):
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.35).cuda()
    xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    flop = 16 * V * (4478976 + 4608 * V) * S
    res = {}
    for path, name in ((1 if V <= 64 else 2, "fused-f32" if V <= 64 else "per-op path"), (3, "split-fp16")):
        mm = H.tw_kernel_model(H.full_kernel_sd(), path=path)
        res[path] = timed(lambda: mm.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None,
                                                                  edge_batch_idx=None, masked_elements=mk, num_samples=S))
        print(f"kernel flow V={V} S={S} {name}: {res[path]:.2f} ms per reverse pass, {flop / res[path] / 1e9:.1f} TFLOP/s algorithmic")
    print(f"  V={V}: split-fp16 speed-up {res[1 if V <= 64 else 2] / res[3]:.2f}x")
# cfg 4
V, S = 22, 1000
at = torch.randint(0, 5, (1, V), generator=g).cuda()
xc = (torch.randn(1, V, 3, generator=g) * 0.3).cuda()
xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
mk = torch.zeros(1, V, dtype=torch.bool).cuda()
flop = 16 * V * 3726336 * S
for path, name, iters in ((3, "split-fp16 dense net-block kernel", 5), (1, "fused f32-MFMA dense net-block kernel", 5), (2, "per-op path", 2)):
    md = H.tw_dense_model(H.full_dense_sd(), path=path)
    ms = timed(lambda: md.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                       masked_elements=mk, num_samples=S), iters=iters)
    print(f"cfg4 dense flow V={V} S={S} {name}: {ms:.2f} ms per reverse pass, {flop / ms / 1e9:.1f} TFLOP/s algorithmic")
