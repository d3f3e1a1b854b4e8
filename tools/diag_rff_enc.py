"""Encoder-stack build (default) against the per-section build (tw_debug_set_flags 4096) of the dense softmax model over\n(position features, encoder layers, coupling layers, rows): which combinations demote / differ.  Used in r05 to isolate the\n%[cur] / %[ring] register sharing of the position-feature build (DESIGN_LOG.md, round 5)."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import helpers as H
from oracle import flow_oracle as fo
from timewarp_amd import _lib
lib = _lib.load()

def case(rff, n_layers, n_coupling, B, V=22):
    spec = fo.FlowSpec(variant="dense", n_head=8, num_transformer_layers=n_layers, num_coupling_layers=n_coupling)
    t = fo.make_template(spec, rff_dim=rff) if rff else fo.make_template(spec)
    sd = fo.synth_state_dict(t, 0)
    g = torch.Generator().manual_seed(1)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.3
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    outs = {}
    for flags in (4096, 0):
        m = H.tw_dense_model(sd, rff_dim=rff, n_coupling=n_coupling, n_layers=n_layers, path=3)
        lib.tw_debug_set_flags(flags)
        ll = m.log_likelihood(atom_types=at.cuda(), x_coords=x_c.cuda(), x_velocs=x_v.cuda(), y_coords=y_c.cuda(), y_velocs=y_v.cuda(),
                              adj_list=None, edge_batch_idx=None, masked_elements=mask.cuda()).cpu()
        lib.tw_debug_set_flags(0)
        outs[flags] = (ll, getattr(m, "demoted", False))
    print(f"rff={rff} layers={n_layers} coupling={n_coupling} B={B}: enc demoted={outs[0][1]} sections demoted={outs[4096][1]} "
          f"diff={H.rel_err(outs[0][0], outs[4096][0]):.2e}", flush=True)

for rff in (0, 128):
    for n_layers in (1, 3):
        for n_coupling in (2, 8):
            for B in (1, 9):
                case(rff, n_layers, n_coupling, B)
