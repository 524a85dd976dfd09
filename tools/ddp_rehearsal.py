"""Rehearsal of the data-parallel training step (BASELINE cfg 5 control flow) with N processes on ONE GPU and gloo standing in
for RCCL (the collective is the only thing that differs from the 8-GPU run):
    LDM_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ddp_rehearsal.py
Asserts (exit code 1 on failure):
  * after two steps the parameters are bit-identical on all ranks (same Python-random decisions, one averaged gradient);
  * ranks drew DIFFERENT timesteps / noise (train_step seeds torch with step_seed * world + rank);
  * the first data-parallel step equals, on rank 0, a single process that runs the shards one after the other with the ranks'
    seeds and averages their gradients -- BITWISE (the backward has no float atomics: every gradient is bit-reproducible);
  * the bucketed, overlapped all-reduce (dist.GradSync: one bucket per UNet level, launched while the backward continues) gives
    bit for bit the parameters of the flat, blocking all-reduce after the backward;
  * bf16 on the wire (fp32 accumulation on receipt) stays within 1e-3 of the fp32 exchange on the updated parameters."""
import os
import random
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


def flat_params(d):
    return torch.cat([p.detach().reshape(-1) for p in d.parameters()])


full = torch.randn(4 * world, 8, 16, 16, generator=torch.Generator().manual_seed(0))
d, opt = make()
failures = []
stats = {}
loss0 = ldist.train_step(d, opt, full[4 * rank:4 * rank + 4].to(dev), step_seed=0, world=world, stats=stats)
after1 = flat_params(d).clone()
if world > 1 and "allreduce_ms_exposed" not in stats:
    failures.append("the bucketed all-reduce did not run")
# the same first step with the flat, blocking all-reduce, and with bf16 on the wire
ldist.BUCKETED = False
d_flat, opt_flat = make()
ldist.train_step(d_flat, opt_flat, full[4 * rank:4 * rank + 4].to(dev), step_seed=0, world=world)
ldist.BUCKETED = True
if not torch.equal(flat_params(d_flat), after1):
    failures.append("bucketed all-reduce != flat all-reduce (bitwise)")
ldist.WIRE_DTYPE = torch.bfloat16
d_w16, opt_w16 = make()
ldist.train_step(d_w16, opt_w16, full[4 * rank:4 * rank + 4].to(dev), step_seed=0, world=world)
ldist.WIRE_DTYPE = None
w16_err = float((flat_params(d_w16).double() - after1.double()).norm() / after1.double().norm())
if not w16_err < 1e-3:
    failures.append("bf16 wire format deviates %.3e from the fp32 exchange" % w16_err)
del d_flat, opt_flat, d_w16, opt_w16
t_probe = torch.randint(0, 1 << 30, (1,))                    # torch's CPU stream after the rank-specific seed: must differ between ranks
ldist.train_step(d, opt, full[4 * rank:4 * rank + 4].to(dev), step_seed=1, world=world)
flat = flat_params(d)
ref = flat.clone()
dist.broadcast(ref, src=0)
same = torch.tensor([1.0 if torch.equal(flat, ref) else 0.0], device=dev)
dist.all_reduce(same, op=dist.ReduceOp.MIN)
if same.item() != 1.0:
    failures.append("parameters differ between ranks after two steps")
probes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(probes, t_probe)
if len({int(p) for p in probes}) != world:
    failures.append("ranks share one torch random stream (same timesteps / noise on every shard)")
if not bool(torch.isfinite(flat).all()):
    failures.append("non-finite parameters")
if rank == 0:
    # single-process restatement of step 0: shard by shard with the ranks' seeds, gradients averaged
    d1, opt1 = make()
    acc = None
    for r in range(world):
        random.seed(0)
        torch.manual_seed(0 * world + r)
        opt1.zero_grad()
        d1.calculate_loss(full[4 * r:4 * r + 4].to(dev)).backward()
        grads = [None if p.grad is None else p.grad.clone() for p in d1.parameters()]
        acc = grads if acc is None else [None if a is None else a + g for a, g in zip(acc, grads)]
    for p, g in zip(d1.parameters(), acc):
        p.grad = None if g is None else g / world
    opt1.step()
    want = flat_params(d1)
    err = float((after1.double() - want.double()).norm() / want.double().norm())
    if not (torch.equal(after1, want) if world == 2 else err < 1e-6):       # two ranks: a + b == b + a, so the sums agree bit for bit
        failures.append("data-parallel step != sequential shards with averaged gradients (rel %.3e)" % err)
    print("ddp rehearsal, %d ranks (gloo): loss %.5f, step-0 deviation from the sequential restatement %.3e, bf16-wire deviation %.3e, "
          "all-reduce exposed %.2f ms of %d bytes" % (world, float(loss0), err, w16_err, stats.get("allreduce_ms_exposed", 0.0), stats.get("allreduce_bytes", 0)))
fail = torch.tensor([float(len(failures))], device=dev)
dist.all_reduce(fail, op=dist.ReduceOp.SUM)
for f in failures:
    print("rank %d FAILED: %s" % (rank, f), flush=True)
if rank == 0 and fail.item() == 0:
    print("DDP_REHEARSAL_OK", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if fail.item() else 0)
