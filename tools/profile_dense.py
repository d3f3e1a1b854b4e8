import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import helpers as H
from torch.profiler import profile, ProfilerActivity
g = torch.Generator().manual_seed(0)
V, S = 22, 1000
md = H.tw_dense_model(H.full_dense_sd(), path=2)
at = torch.randint(0, 5, (1, V), generator=g).cuda()
xc = (torch.randn(1, V, 3, generator=g) * 0.3).cuda(); xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
mk = torch.zeros(1, V, dtype=torch.bool).cuda()
f = lambda: md.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None, masked_elements=mk, num_samples=S)
f(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    f(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=70))
