import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
chain, model = bench.build_chain(dev, 1234, 1000, bench.PATHS["h3"]["path"])
with torch.no_grad():
    for _ in range(3): chain.step_deferred()
    chain.flush(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(4): chain.step_deferred()
        chain.flush(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
