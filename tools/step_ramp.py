import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
chain, model = bench.build_chain(dev, 1234, 1000, bench.PATHS["h3"]["path"])
ts = []
with torch.no_grad():
    for i in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        chain.step_deferred(); chain.flush()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
