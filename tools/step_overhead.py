"""Per-step time of the collective path with ONE rank (RCCL world_size 1) at the per-rank shapes of an N-GPU job."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from ffvd_amd import synthetic
from ffvd_amd.distributed import ShardedElbo
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for S in (4, 8, 32):
    params, Y, c, meta = synthetic.make_named("c2", S=S)
    sh = ShardedElbo(params, Y, c, meta, rank=0, world=1, mode="chains", device=0, always_reduce=True, route="gram")
    for _ in range(5): sh.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): sh.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    ms = sh.engine.time_elbo(20) / 20
    print("S=%d step %.3f ms (kernels alone %.3f ms, overhead %.0f us)" % (S, dt * 1e3, ms, (dt * 1e3 - ms) * 1e3))
    del sh
dist.destroy_process_group()
