"""Rehearsal of the data-parallel training step (BASELINE cfg 5 control flow) with N processes on ONE GPU and gloo
standing in for RCCL:  LDM_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29533 tools/ddp_rehearsal.py
Checks: identical decisions and parameters on all ranks after two steps, and equality with a single-process step on
the concatenated batch (gradient averaging over equal shards == the full-batch mean)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import dist as ldist, synth  # noqa: E402
from ldm_image_generator_amd.ddpm import DDPM  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

rank, world, _ = ldist.init_from_env()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = dict(input_channels=8, stages=[1, 2, 2], channels=[128, 128, 256])


def make():
    net = UNet(**cfg)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    d = DDPM(model=net.to(dev).train())
    return d, torch.optim.SGD(d.parameters(), lr=1e-2)


full = torch.randn(4 * world, 8, 16, 16, generator=torch.Generator().manual_seed(0))
d, opt = make()
for step in range(2):
    torch.manual_seed(100 + step + 1000 * rank)             # per-rank noise / timesteps, shared Python-random decisions
    ldist.train_step(d, opt, full[4 * rank:4 * rank + 4].to(dev), step_seed=step, world=world)
flat = torch.cat([p.detach().reshape(-1) for p in d.parameters()])
ref = flat.clone()
dist.broadcast(ref, src=0)
same = bool(torch.equal(flat, ref))
ok = torch.tensor([1.0 if same else 0.0], device=dev)
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    print("ranks %d: parameters identical on all ranks after 2 steps: %s; finite: %s" % (world, bool(ok.item() == 1.0), bool(torch.isfinite(flat).all())))
dist.barrier()
dist.destroy_process_group()
