import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
sd = H.full_kernel_sd()
def case(B, V, lens, seed, types_max=5):
    g = torch.Generator().manual_seed(seed)
    at = torch.randint(0, types_max, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.3
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b, n in enumerate(lens): mask[b, n:] = True
    return at, x_c, x_v, y_c, y_v, mask
models = {p: H.tw_kernel_model(sd, path=p) for p in (1, 3)}
for label, args in (("B5 nomask", (5, 22, [22]*5, 1)), ("B4 nomask", (4, 22, [22]*4, 2)), ("B8 nomask", (8, 22, [22]*8, 3)),
                    ("B4 mask", (4, 22, [22, 20, 22, 17], 4)), ("B8 types<4", (8, 22, [22]*8, 5, 4)), ("B1", (1,22,[22],6))):
    at, x_c, x_v, y_c, y_v, mask = case(*args)
    outs = {}
    for p, m in models.items():
        outs[p] = m.log_likelihood(atom_types=at.cuda(), x_coords=x_c.cuda(), x_velocs=x_v.cuda(), y_coords=y_c.cuda(),
                                   y_velocs=y_v.cuda(), adj_list=None, edge_batch_idx=None, masked_elements=mask.cuda()).cpu()
    print(label, "f32", outs[1].tolist(), "\n   h3", outs[3].tolist(), flush=True)
# stage-level: B=4 no mask, compare debug netblock of chain 0 scale net
at, x_c, x_v, y_c, y_v, mask = case(4, 22, [22]*4, 2)
xc = x_c - fo.centre_of_mass(x_c, mask)
z_other = y_v  # chain 0 transforms positions -> input uses z_velocs
for net in (0,):
    a1, o1 = models[1].debug_netblock(0, net, at.cuda(), xc.cuda(), x_v.cuda(), mask.cuda(), z_other.cuda(), 1)
    a3, o3 = models[3].debug_netblock(0, net, at.cuda(), xc.cuda(), x_v.cuda(), mask.cuda(), z_other.cuda(), 3)
    for i in range(a1.shape[0]):
        for r in range(4):
            print("stage", i, "row", r, "rel", H.rel_err(a3[i, r].cpu(), a1[i, r].cpu()))
    print("out", H.rel_err(o3.cpu(), o1.cpu()))
