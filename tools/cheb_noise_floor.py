"""Where the 1e-5 bar sits for the full-size chebyshev_kernel model (tests/test_flow_gpu.py::
test_full_chebyshev_attention_all_paths): per golden row, the deviation of the reverse-move log-density log p(x~|y~)
  (a) of the REFERENCE'S OWN fp32 arithmetic from the same computation in fp64 (oracle/flow_oracle.py on torch-CPU: its fp32
      run reproduces the reference's vectors bit for bit), and
  (b) of every HIP execution path from the reference's vectors,
both relative to max |log p| over the rows (the tests' `rel_err`).  If (b) is of the size of (a), the kernels are as close
to the reference as the reference is to exact arithmetic, and a tighter bar would test rounding order, not correctness."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H

torch.set_num_threads(min(16, os.cpu_count() or 1))
for name, sd, spec, kw in (("kernel_cheb_full_ad", H.full_cheb_sd(), H.FULL_CHEB_SPEC,
                            dict(attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=True)),
                           ("kernel_full_ad", H.full_kernel_sd(), H.FULL_KERNEL_SPEC, {})):
    d, _ = H.load(name)
    S = d["s_y_coords"].shape[0]
    gy, gv = d["s_y_coords"].squeeze(1), d["s_y_velocs"].squeeze(1)

    def oracle(dt):
        c = lambda t: t.to(dt) if t.is_floating_point() else t
        return fo.log_likelihood({k: c(v) for k, v in sd.items()}, spec, d["atom_types"].repeat(S, 1), c(gy), c(-gv),
                                 c(d["x_coords"]).repeat(S, 1, 1), c(-d["x_velocs"]).repeat(S, 1, 1), d["masked"].repeat(S, 1))

    ref = d["logp_yx"].double()
    scale = float(ref.abs().max())
    a32, a64 = oracle(torch.float32).double(), oracle(torch.float64)
    own = (ref - a64).abs() / scale
    print(f"{name}: oracle fp32 == reference vectors: {bool(torch.equal(a32.float(), d['logp_yx']))}; max |log p| {scale:.1f}")
    print(f"  (a) reference fp32 vs fp64 arithmetic, per row: max {float(own.max()):.2e}, 90th percentile {float(own.quantile(0.9)):.2e}, "
          f"median {float(own.median()):.2e}; worst rows {own.topk(4).indices.tolist()}")
    if torch.cuda.is_available():
        for path, label in ((2, "per-op"), (1, "fused f32"), (3, "split-fp16")):
            m = H.tw_kernel_model(sd, path=path, **kw)
            out = H.run_model_case(m, d)["logp_yx"].double()
            e = (out - ref).abs() / scale
            print(f"  (b) {label:10s} vs reference vectors, per row: max {float(e.max()):.2e}, 90th percentile {float(e.quantile(0.9)):.2e}, "
                  f"median {float(e.median()):.2e}; vs fp64 arithmetic: max {float(((out - a64).abs() / scale).max()):.2e}")
