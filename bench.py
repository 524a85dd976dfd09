"""Headline benchmark: 256x256, 50-step latent-diffusion sampling (UNet denoise loop +
VAE decode), synthetic formula weights, fp32 on the exact-fp32 MFMA path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full pass of the hot path over one batch: 50 denoise steps (UNet forward
+ DDIM update) on this rank's 256 latents [256, 8, 32, 32], VAE decode to [256, 3, 256, 256]
and (N > 1) the single all-gather of the images (BASELINE.json configs[2]/[3]).  x_T is
already resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md "Peak FP32 (matrix)"
UNET_GFLOP_PER_SAMPLE_STEP = 13.74  # SURVEY.md 8(d), algorithmic minimum @ latent 32x32
DECODE_GFLOP_PER_IMAGE = 80.586


def cpu_baseline(threads):
    """The CPU oracle (port of the reference's algorithm, pinned by goldens) on a bounded sample
    of the same workload: full-size UNet, 2 eval-mode denoise steps on 4 latents, 2 decodes."""
    from oracle import ldm_oracle as O
    torch.set_num_threads(threads)
    usd = O.formula_state(O.unet_state_shapes())
    dsd = O.formula_state(O.decoder_state_shapes())
    x = torch.randn(4, 8, 32, 32, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        random.seed(0)
        O.unet_forward(usd, x[:1], torch.full((1,), 999), training=False)          # warm-up
        t0 = time.perf_counter()
        for t in (999, 978):
            O.unet_forward(usd, x, torch.full((4,), t), training=False)
        t_step = (time.perf_counter() - t0) / 8.0                                   # s per sample-step
        O.vae_decode(dsd, x[:1])
        t0 = time.perf_counter()
        O.vae_decode(dsd, x[:2])
        t_dec = (time.perf_counter() - t0) / 2.0                                    # s per image
    return dict(value=1.0 / (50 * t_step + t_dec), unit="images/s", cores=threads, kind="port",
                sample="oracle/ldm_oracle.py: full-size UNet eval-mode, 2 denoise steps x 4 latents + 2 decodes; "
                       "images/s = 1/(50*t_sample_step + t_decode)",
                sample_steps_per_sec=1.0 / t_step, decode_images_per_sec=1.0 / t_dec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--num-steps", type=int, default=50, help="DDIM steps per image")
    ap.add_argument("--mode", default="eval", choices=["eval", "train"],
                    help="eval: all 36 blocks run (headline, FLOPs deterministic); train: the reference's stochastic depth")
    ap.add_argument("--gather", default="f32", choices=["f32", "u8"],
                    help="what the single all-gather moves: the fp32 images SURVEY 8(d) defines the metric on (default), or the "
                         "device-side uint8 HWC post-process of sample_ldm.py:75-77 (a quarter of the bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-mode-leg", action="store_true",
                    help="skip the secondary measurement in the reference-faithful mode (no .eval(): stochastic depth live)")
    ap.add_argument("--no-split-leg", action="store_true",
                    help="skip the secondary measurement under GEMM schedule 2 (bf16x3 split consumer)")
    args = ap.parse_args()

    from ldm_image_generator_amd import dist as ldist
    from ldm_image_generator_amd import ops, synth
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    from ldm_image_generator_amd.vae import Decoder, to_uint8_images
    import torch.distributed as dist

    rank, world, local = ldist.init_from_env()
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if os.environ.get("LDM_BENCH_ONE_DEVICE"):      # rehearsal of the multi-rank path on a one-GPU box (with LDM_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if world > 1:                      # bring RCCL up outside the timed region even when --warmup 0
        dist.all_reduce(torch.zeros(1, device=dev))
    net = UNet()
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    dec = Decoder()
    dec.load_state_dict(synth.fill_state_dict(dec.state_dict()))
    net, dec = net.to(dev), dec.to(dev)
    net.train(args.mode == "train")
    ddpm = DDPM(model=net)

    B, T = args.batch, args.num_steps
    gb = B * world
    lo, hi = ldist.shard_bounds(gb, rank, world)
    x_t = ldist.global_noise(gb, (8, 32, 32), seed=0)[lo:hi].to(dev)

    def one_pass(seed):
        z = ddpm.sample((B, 8, 32, 32), seed=seed, num_steps=T, x_init=x_t, progress=False)
        img = dec(z)
        if args.gather == "u8":
            img = to_uint8_images(img)
        return ldist.gather_images(img, gb, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(warmup, steps):
        """W untimed passes, then exactly K timed passes bracketed by barrier + synchronize; max over ranks."""
        out = None
        for i in range(warmup):
            one_pass(i)
        fence()
        ops.prof_enable(rank == 0)
        t0 = time.perf_counter()
        for i in range(steps):
            out = one_pass(100 + i)
        fence()
        dt = time.perf_counter() - t0
        prof = ops.prof_read() if rank == 0 else (0, 0.0, 0.0)
        ops.prof_enable(False)
        t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item()), prof, out

    dt, (launches, gemm_ms, gemm_flops), out = measure(args.warmup, args.steps)
    finite = bool(torch.isfinite(out.float()).all().item())

    # secondary leg, never the headline: the same passes under GEMM schedule 2 (fp32 operands cut exactly into three
    # bf16 pieces, six bf16 MFMAs per product, fp32 accumulate -- DESIGN.md 3.1), with its deviation from the
    # exact-fp32 images of the same seed
    # secondary leg: what the reference's scripts actually do -- they never call .eval(), so SwinBlocks are skipped with
    # p = 0.25 during sampling too (unet.py:39); fewer FLOPs per image, hence reported beside, not as, the headline
    train_leg = None
    if not args.no_train_mode_leg and args.mode == "eval":
        net.train(True)
        dt3, (l3, ms3, fl3), out3 = measure(1, args.steps)
        net.train(False)
        train_leg = {"value": gb * args.steps / dt3, "unit": "images/s", "ms_per_step": dt3 / args.steps * 1e3,
                     "gemm_tflops": fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else None,
                     "executed_gflop_per_sample_step": fl3 / 1e9 / (B * args.steps) / T if rank == 0 else None,
                     "outputs_finite": bool(torch.isfinite(out3.float()).all().item()),
                     "note": "reference-faithful train mode (stochastic depth live while sampling), same seeds"}
        del out3
    split = None
    if not args.no_split_leg:
        keep = out[: min(16, out.shape[0])].clone()
        del out
        old = ops.gemm_variant(2)
        dt2, (l2, ms2, fl2), out2 = measure(1, args.steps)
        ops.gemm_variant(old)
        d = (out2[: keep.shape[0]].double() - keep.double())
        split = {"value": gb * args.steps / dt2, "unit": "images/s", "ms_per_step": dt2 / args.steps * 1e3,
                 "gemm_tflops_fp32_equivalent": fl2 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else None,
                 "rel_l2_vs_exact_images": float(d.norm() / keep.double().norm()),
                 "note": "GEMM schedule 2: v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 splits of the fp32 operands, "
                         "fp32 accumulate; opt-in, not the headline"}
        del out2

    if rank == 0:
        images = gb * args.steps
        ms_per_step = dt / args.steps * 1e3
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        algo_flops = images * (T * UNET_GFLOP_PER_SAMPLE_STEP + DECODE_GFLOP_PER_IMAGE) * 1e9
        line = {
            "metric": "images_per_sec_256x256_50step_ldm", "value": images / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sample_ldm 256x256, %d DDIM steps, batch %d per GPU: UNet(385.7M) + VAE Decoder, "
                                   "formula weights, %s-mode" % (T, B, args.mode),
                       "global_batch": gb, "latent": [8, 32, 32], "parallelism": "dp%d" % world, "gathered": args.gather},
            "denoise_steps_per_sec": images * T / dt / B, "sample_steps_per_sec": images * T / dt,
            "algorithmic_tflops": algo_flops / dt / 1e12 if args.mode == "eval" else None,
            "outputs_finite": finite,
            "roofline": {"bound": "mfma", "kernel": "ldm_gemm_f32 family (gemm_stream_kernel, gconv3x3_kernel; v_mfma_f32_32x32x2_f32), all launches of the timed region",
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "launches": launches, "kernel_ms": gemm_ms, "gflop_per_sample_step_measured":
                             None if args.mode != "eval" else gemm_flops / 1e9 / (B * args.steps) / T},
        }
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):      # PMC passes are separate runs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): tools/traffic_summary.py
            tj = json.load(open(tpath))
            line["roofline"]["traffic"] = tj["gemm_hbm_bytes_per_launch"]
            line["roofline"]["traffic_unit"] = "bytes per GEMM launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r01_traffic.md)"
            line["roofline"]["algorithmic_bytes_per_launch"] = None
        if train_leg is not None:
            line["train_mode"] = train_leg
        if split is not None:
            line["split_schedule"] = split
        if not args.no_cpu_baseline and world == 1:            # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 64))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
