"""Dev tool: RCCL process group with ONE rank next to the library's own HIP stream (the 8-GPU bench path, as far as one GPU
allows): init, barrier, all_reduce, a filter run in between, destroy."""
import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import numpy as np, bayesssm_amd as b
from bench import simulate_lg
m = b.models.linear_gaussian()
ctx = b.Context(0, 1 << 16, 1)
dist.barrier(); torch.cuda.synchronize()
r = b.bootstrap_filter(simulate_lg(50), 1 << 16, m.init_fn, m.transition_fn, m.log_likelihood_fn, return_particles=False, seed=1, ctx=ctx,
                       phi=0.8, sigma_x=1.0, sigma_y=1.0)
t = torch.tensor([r["loglike"]], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
from bayesssm_amd.pmmh import gather_chains
g = gather_chains({0: np.arange(6.0).reshape(2, 3)}, 1, 2, 3, dist)
dist.barrier(); dist.destroy_process_group()
print("nccl rehearsal ok", float(t.item()), g.shape)
