"""Debug aid: dump the split-fp16 net-block activations for a fixed input under given debug flags."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
from timewarp_amd import _lib, build as _b
if os.environ.get("TW_DEBUG_LIB"):
    _b.LIB_PATH = os.environ["TW_DEBUG_LIB"]
tag, flags = sys.argv[1], int(sys.argv[2])
sd = H.full_kernel_sd()
g = torch.Generator().manual_seed(2)
B, V = 4, 22
at = torch.randint(0, 5, (B, V), generator=g)
x_c = torch.randn(B, V, 3, generator=g) * 0.3
x_v = torch.randn(B, V, 3, generator=g) * 0.5
y_v = torch.randn(B, V, 3, generator=g) * 0.5
mask = torch.zeros(B, V, dtype=torch.bool)
xc = x_c - fo.centre_of_mass(x_c, mask)
m = H.tw_kernel_model(sd, path=3)
_lib.load().tw_debug_set_flags(flags)
outs = []
for rep in range(3):
    a3, o3 = m.debug_netblock(0, 0, at.cuda(), xc.cuda(), x_v.cuda(), mask.cuda(), y_v.cuda(), 3)
    outs.append((a3.cpu(), o3.cpu()))
for rep in (1, 2):
    print("rep", rep, "identical to rep 0:", torch.equal(outs[rep][0], outs[0][0]), torch.equal(outs[rep][1], outs[0][1]))
os.makedirs("gpurun_out", exist_ok=True)
torch.save(outs[0], f"gpurun_out/dump_{tag}_{flags}.pt")
print("saved", tag, flags, float(outs[0][0].abs().mean()))
