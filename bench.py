"""Headline benchmark: 256x256, 50-step latent-diffusion sampling (UNet denoise loop +
VAE decode), synthetic formula weights, fp32 on the exact-fp32 MFMA path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full pass of the hot path over one batch: 50 denoise steps (UNet forward
+ DDIM update) on this rank's 256 latents [256, 8, 32, 32], VAE decode to [256, 3, 256, 256]
and (N > 1) the single all-gather of the images (BASELINE.json configs[2]/[3]).  x_T is
already resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Secondary objects on the same line (never `value`): `train_mode` (the reference-faithful sampling mode),
`split_schedule` (GEMM schedule 2), `autocast_bf16` (the opt-in bf16 sampling + decode mode: ddpm.py:52,75), `cfg2` (BASELINE.json
configs[1]: pixel-space 64x64, batch 64, UNet only), `train_step` (BASELINE.json configs[4]: one optimisation step of train_ldm.py on
latents [128, 8, 64, 64] per GPU, AdamW included, fp32 and bf16 operands; for N > 1 with the bucketed, overlapped gradient all-reduce),
`vae_train_step` (one iteration of train_vae.py's loop), `cpu_baseline` (with `gpu_vs_cpu_rel_l2`: the HIP path checked
against the oracle outputs of the baseline's own timed sample), and at N = 1 `slices_check` (4 rows of the timed batch re-run alone).
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
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md "HBM3E peak BW" (spec; 6290 measured with a float4 copy)
UNET_GFLOP_PER_SAMPLE_STEP = 13.74  # SURVEY.md 8(d), algorithmic minimum @ latent 32x32
DECODE_GFLOP_PER_IMAGE = 80.586
PROF_CLASSES = {0: "ldm_gemm_f32", 1: "ldm_gemm_tn_f32", 2: "ldm_gconv3x3_wgrad_f32", 3: "ldm_gemm_bf16", 4: "ldm_gemm_tn_bf16",
                5: "ldm_gconv3x3_bf16"}


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
        random.seed(1)                                                             # expert choices of the two timed forwards (gpu_probe draws the same)
        t0 = time.perf_counter()
        eps = [O.unet_forward(usd, x, torch.full((4,), t), training=False) for t in CPU_SAMPLE_STEPS]
        t_step = (time.perf_counter() - t0) / 8.0                                   # s per sample-step
        O.vae_decode(dsd, x[:1])
        t0 = time.perf_counter()
        img = O.vae_decode(dsd, x[:2])
        t_dec = (time.perf_counter() - t0) / 2.0                                    # s per image
    line = dict(value=1.0 / (50 * t_step + t_dec), unit="images/s", cores=threads, kind="port",
                sample="oracle/ldm_oracle.py: full-size UNet eval-mode, 2 denoise steps x 4 latents + 2 decodes; "
                       "images/s = 1/(50*t_sample_step + t_decode)",
                sample_steps_per_sec=1.0 / t_step, decode_images_per_sec=1.0 / t_dec)
    return line, dict(eps=eps, img=img)


CPU_SAMPLE_STEPS = (999, 978)


def gpu_probe(net, dec, dev):
    """The HIP path on exactly the inputs cpu_baseline() times the oracle on (same formula weights, same x, same timesteps): two
    UNet forwards on 4 latents and the decode of 2 of them.  Taken while the weights are still the formula weights (the
    training-step leg moves them)."""
    was_training = net.training
    net.eval()
    with torch.no_grad():
        x = torch.randn(4, 8, 32, 32, generator=torch.Generator().manual_seed(0)).to(dev)
        eps = []
        random.seed(1)                                          # the expert choices cpu_baseline() makes for its two timed forwards
        for t in CPU_SAMPLE_STEPS:
            eps.append(net(x=x, time=torch.full((4,), t, device=dev), condition=None).cpu())
        img = dec(x[:2]).cpu()
    net.train(was_training)
    return dict(eps=eps, img=img)


def gpu_vs_cpu(gpu, cpu):
    """GPU output compared to the CPU output of this run.  Stated fp32 tolerance: rel-L2 <= 2e-5 each."""
    errs = {}
    for t, got, ref in zip(CPU_SAMPLE_STEPS, gpu["eps"], cpu["eps"]):
        errs["unet_t%d" % t] = float((got.double() - ref.double()).norm() / ref.double().norm())
    errs["decode"] = float((gpu["img"].double() - cpu["img"].double()).norm() / cpu["img"].double().norm())
    worst = max(errs.values())
    return {"rel_l2": errs, "worst": worst, "tolerance": 2e-5, "ok": bool(worst <= 2e-5),
            "note": "HIP path vs the oracle outputs of cpu_baseline's own timed sample (4 latents x 2 UNet forwards, 2 decodes)"}


def traffic_table():
    """PMC-derived HBM traffic of the GEMM family (separate rocprofv3 --pmc passes: tools/traffic_summary.py); newest round first."""
    for name in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            return name, json.load(open(path))
    return None, None


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
    ap.add_argument("--no-slices-check", action="store_true", help="skip the N = 1 self-check (4 rows of the timed batch re-run alone)")
    ap.add_argument("--no-cfg2-leg", action="store_true", help="skip the BASELINE configs[1] leg (pixel-space 64x64, batch 64, UNet only)")
    ap.add_argument("--no-train-mode-leg", action="store_true",
                    help="skip the secondary measurement in the reference-faithful mode (no .eval(): stochastic depth live)")
    ap.add_argument("--no-autocast-leg", action="store_true", help="skip the secondary measurement in the opt-in bf16 (autocast) mode")
    ap.add_argument("--no-split-leg", action="store_true",
                    help="skip the secondary measurement under GEMM schedule 2 (bf16x3 split consumer)")
    ap.add_argument("--no-vae-train-leg", action="store_true", help="skip the VAE training iteration (SURVEY 8 f4: train_vae.py's loop body)")
    ap.add_argument("--vae-batch", type=int, default=8)
    ap.add_argument("--vae-size", type=int, default=256)
    ap.add_argument("--vae-steps", type=int, default=3)
    ap.add_argument("--no-train-step-leg", action="store_true", help="skip the training-step measurement (BASELINE cfg 5 per-GPU shape)")
    ap.add_argument("--train-batch", type=int, default=128, help="training-step leg: samples per GPU (cfg 5: 1024 / 8)")
    ap.add_argument("--train-latent", type=int, default=64, help="training-step leg: latent edge (512 px / 8)")
    ap.add_argument("--train-steps", type=int, default=3, help="training-step leg: timed optimisation steps")
    ap.add_argument("--train-warmup", type=int, default=2, help="training-step leg: untimed steps before the timed ones")
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

    @torch.no_grad()                   # VAE.decode (vae.py:50-52) is a no_grad method; with gradients enabled Decoder.forward keeps a training tape
    def decode(z):
        img = dec(z)
        return to_uint8_images(img) if args.gather == "u8" else img

    def one_pass(seed):
        z = ddpm.sample((B, 8, 32, 32), seed=seed, num_steps=T, x_init=x_t, progress=False)
        return ldist.gather_images(decode(z), gb, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item())

    def measure(warmup, steps, prof=True):
        """W untimed passes, then exactly K timed passes bracketed by barrier + synchronize; max over ranks."""
        out = None
        for i in range(warmup):
            one_pass(i)
        fence()
        ops.prof_enable(rank == 0 and prof)
        t0 = time.perf_counter()
        for i in range(steps):
            out = one_pass(100 + args.steps - steps + i)      # every leg ends on seed 100 + K - 1: their last outputs are comparable
        fence()
        dt = time.perf_counter() - t0
        abytes = ops.prof_read_bytes(-1) if rank == 0 else 0.0
        prof = ops.prof_read() if rank == 0 else (0, 0.0, 0.0)
        ops.prof_enable(False)
        return max_over_ranks(dt), prof + (abytes,), out

    # what produced `value`: the process-wide GEMM knobs as the library reports them (LDM_GEMM_VARIANT in the environment would show here)
    knobs = {"gemm_variant": ops.gemm_variant(-1), "wide_epilogue": ops.gemm_wide_epilogue(-1), "gemm_ring": ops.gemm_ring(-1)}
    dt, (launches, gemm_ms, gemm_flops, gemm_bytes), out = measure(args.warmup, args.steps)
    finite = bool(torch.isfinite(out.float()).all().item())

    # N > 1, outside the timed region: rank 0 re-runs 4 samples of ANOTHER rank's shard alone (same seed -> same expert
    # decisions) and compares them with the rows the all-gather delivered -- sharded == unsharded, sample for sample
    shard_check = None
    if world > 1:
        lo1, _ = ldist.shard_bounds(gb, 1, world)
        if rank == 0:
            xs = ldist.global_noise(gb, (8, 32, 32), seed=0)[lo1:lo1 + 4].to(dev)
            z = ddpm.sample((4, 8, 32, 32), seed=100 + args.steps - 1, num_steps=T, x_init=xs, progress=False)
            alone = decode(z).double()
            got = out[lo1:lo1 + 4].double()
            err = float((alone - got).norm() / alone.norm().clamp_min(1e-30))
            tol = 5e-6 if args.gather == "f32" else 2e-2
            shard_check = {"rel_l2": err, "tolerance": tol, "ok": bool(err <= tol), "rows": [lo1, lo1 + 4],
                           "note": "rank 0 alone vs rows gathered from rank 1 (batch 4 vs %d: other GEMM tile paths, fp32 re-association only)" % B}
        fence()
    slices_check = None
    if world == 1 and not args.no_slices_check and B >= 8:
        # N = 1: the same property inside one GPU -- 4 rows of the timed batch re-run alone (same seed -> same expert decisions;
        # batch 4 takes other GEMM tile / split-K paths than batch B: fp32 re-association only)
        lo1 = (B // 2) & ~3
        z = ddpm.sample((4, 8, 32, 32), seed=100 + args.steps - 1, num_steps=T, x_init=x_t[lo1:lo1 + 4], progress=False)
        alone = decode(z).double()
        got = out[lo1:lo1 + 4].double()
        err = float((alone - got).norm() / alone.norm().clamp_min(1e-30))
        tol = 5e-6 if args.gather == "f32" else 2e-2
        slices_check = {"rel_l2": err, "tolerance": tol, "ok": bool(err <= tol), "rows": [lo1, lo1 + 4],
                        "note": "rows of the last timed pass re-run alone as a batch of 4 (what makes batch sharding exact)"}
        del alone, got, z

    probe = gpu_probe(net, dec, dev) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    sec_steps = max(2, args.steps // 4)            # passes of the secondary sampling legs (they are never `value`)
    # secondary leg: what the reference's scripts actually do -- they never call .eval(), so SwinBlocks are skipped with
    # p = 0.25 during sampling too (unet.py:39); fewer FLOPs per image, hence reported beside, not as, the headline
    train_leg = None
    if not args.no_train_mode_leg and args.mode == "eval":
        net.train(True)
        dt3, (l3, ms3, fl3, _), out3 = measure(1, sec_steps)
        net.train(False)
        train_leg = {"value": gb * sec_steps / dt3, "unit": "images/s", "ms_per_step": dt3 / sec_steps * 1e3, "steps": sec_steps,
                     "gemm_tflops": fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else None,
                     "executed_gflop_per_sample_step": fl3 / 1e9 / (B * sec_steps) / T if rank == 0 else None,
                     "outputs_finite": bool(torch.isfinite(out3.float()).all().item()),
                     "note": "reference-faithful train mode (stochastic depth live while sampling), same seeds"}
        del out3
    # secondary leg, never the headline: the same passes under GEMM schedule 2 (fp32 operands cut exactly into three
    # bf16 pieces, six bf16 MFMAs per product, fp32 accumulate -- DESIGN.md 3.1), with its deviation from the
    # exact-fp32 images of the same seed
    split = None
    keep = out[: min(16, out.shape[0])].clone()      # exact-fp32 images of the last timed pass: what the reduced-precision legs are compared with
    del out
    if not args.no_split_leg:
        old = ops.gemm_variant(2)
        dt2, (l2, ms2, fl2, _), out2 = measure(1, sec_steps)
        ops.gemm_variant(old)
        d = (out2[: keep.shape[0]].double() - keep.double())
        eq = fl2 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else None
        split = {"value": gb * sec_steps / dt2, "unit": "images/s", "ms_per_step": dt2 / sec_steps * 1e3, "steps": sec_steps,
                 "gemm_tflops_fp32_equivalent": eq,
                 "roofline": None if eq is None else {"bound": "mfma", "achieved": eq, "peak": BF16_MFMA_PEAK_TFLOPS / 6.0, "unit": "TFLOP/s",
                                                      "frac": eq / (BF16_MFMA_PEAK_TFLOPS / 6.0),
                                                      "note": "peak = dense bf16 MFMA / 6 (six bf16 products per fp32 product)"},
                 "rel_l2_vs_exact_images": float(d.norm() / keep.double().norm()),
                 "note": "GEMM schedule 2: v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 splits of the fp32 operands, "
                         "fp32 accumulate; opt-in, not the headline"}
        del out2

    # secondary leg, never the headline: the opt-in bf16 ("autocast") mode -- ddpm.py:52,75: on a GPU the reference samples under
    # 16-bit autocast.  bf16 GEMM operands with fp32 accumulation in the UNet (fp32 residual stream, attention core, DDIM update) and
    # bf16 activations in the VAE decoder; same seeds, deviation from the exact-fp32 images of the same pass reported beside it.
    amp = None
    if not args.no_autocast_leg and args.gather == "f32":
        from ldm_image_generator_amd import autocast
        autocast.set_autocast_dtype(net, torch.bfloat16)
        autocast.set_compute_dtype(dec, torch.bfloat16)
        try:
            dta, _, outa = measure(1, sec_steps * 2, prof=False)
            # a second, profiled pass set for the per-class kernel times (hipEvents around every MFMA launch slow short kernels down a little)
            ops.prof_enable(rank == 0)
            one_pass(100 + args.steps - 1)                       # every rank: the pass ends in the all-gather
            fence()
            if rank == 0:
                per, tot_ms, tot_fl, tot_by = {}, 0.0, 0.0, 0.0
                for cls, name in PROF_CLASSES.items():
                    n, ms, fl = ops.prof_read_class(cls)
                    if n:
                        by = ops.prof_read_bytes(cls)
                        per[name] = {"launches": n, "ms": ms, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                     "algorithmic_gb_per_s": by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None}
                        tot_ms, tot_fl, tot_by = tot_ms + ms, tot_fl + fl, tot_by + by
                ops.prof_read()
            ops.prof_enable(False)
        finally:
            autocast.set_autocast_dtype(net, None)
            autocast.set_compute_dtype(dec, None)
        nsteps_a = sec_steps * 2
        da = (outa[: keep.shape[0]].double() - keep.double())
        amp = {"value": gb * nsteps_a / dta, "unit": "images/s", "ms_per_step": dta / nsteps_a * 1e3, "steps": nsteps_a, "dtype": "bf16",
               "denoise_steps_per_sec": gb * nsteps_a * T / dta / B,
               "over_exact_fp32": (gb * nsteps_a / dta) / (gb * args.steps / dt),
               "rel_l2_vs_exact_images": float(da.norm() / keep.double().norm()),
               "outputs_finite": bool(torch.isfinite(outa).all().item()),
               "algorithmic_tflops": gb * nsteps_a * (T * UNET_GFLOP_PER_SAMPLE_STEP + DECODE_GFLOP_PER_IMAGE) * 1e9 / dta / 1e12 if args.mode == "eval" else None,
               "note": "opt-in (autocast.set_autocast_dtype / set_compute_dtype): bf16 GEMM operands, fp32 accumulate; never `value`"}
        if rank == 0 and tot_ms > 0:
            ach, gbs = tot_fl / (tot_ms * 1e-3) / 1e12, tot_by / (tot_ms * 1e-3) / 1e9
            amp["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "kernel": "all MFMA kernels of one pass (bf16 NT GEMMs, bf16 grouped conv, fp32 FiLM / ch_conv GEMMs), hipEvents per launch",
                               "note": "algorithmic operand bytes (every operand once) / kernel time",
                               "other_view": {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK_TFLOPS},
                               "mfma_kernel_ms_per_pass": tot_ms, "per_kernel": per}
        del outa

    # BASELINE.json configs[1]: sample_ddpm.py, pixel space 64x64, 50 steps, batch 64, UNet only (no VAE).  The reference's script
    # crashes on 3-channel input with the default 8-channel UNet, so the harness uses
    # UNet(input_channels=3); its 36 SwinBlocks and ch_convs are the SAME modules as the headline net (only stem / head differ).
    cfg2 = None
    if not args.no_cfg2_leg and args.mode == "eval":
        try:
            net2 = UNet(input_channels=3, stages=[0, 0, 0, 0])                # cheap shell: stem, head and empty stages ...
            for i in range(len(net.encoder_stages)):                          # ... filled with the headline net's stages
                net2.encoder_stages[i] = net.encoder_stages[i]
                net2.decoder_stages[i] = net.decoder_stages[i]
            ends = {k: v for k, v in net2.state_dict().items() if k.startswith(("encoder_first", "decoder_last"))}
            net2.load_state_dict(synth.fill_state_dict(ends), strict=False)
            net2 = net2.to(dev).eval()
            d2 = DDPM(model=net2)
            b2 = 64
            xp = torch.randn(b2, 3, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
            d2.sample((b2, 3, 64, 64), seed=0, num_steps=T, x_init=xp, progress=False)
            fence()
            ops.prof_enable(rank == 0)
            t0 = time.perf_counter()
            for i in range(2):
                o2 = d2.sample((b2, 3, 64, 64), seed=100 + i, num_steps=T, x_init=xp, progress=False)
            fence()
            dtc = max_over_ranks(time.perf_counter() - t0)
            l2c, msc, flc = ops.prof_read() if rank == 0 else (0, 0.0, 0.0)
            ops.prof_enable(False)
            famc = flc / (msc * 1e-3) / 1e12 if msc > 0 else None
            cfg2 = {"value": b2 * world * 2 / dtc, "unit": "images/s", "denoise_steps_per_sec": 2 * T / dtc, "steps": 2, "ms_per_step": dtc / 2 * 1e3,
                    "config": {"workload": "sample_ddpm 64x64 pixel space, %d DDIM steps, batch %d per GPU, UNet(input_channels=3) only, eval-mode" % (T, b2)},
                    "gemm_tflops": famc, "outputs_finite": bool(torch.isfinite(o2).all().item()),
                    "roofline": None if famc is None else {"bound": "mfma", "achieved": famc, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                           "frac": famc / FP32_MFMA_PEAK_TFLOPS, "kernel": "ldm_gemm_f32 family, hipEvents per launch"},
                    "algorithmic_tflops": b2 * 2 * T * 54.98e9 / dtc / 1e12}
            del net2, d2, o2, xp
            torch.cuda.empty_cache()
        except Exception as exc:           # a secondary leg never takes the headline line down with it
            import traceback
            traceback.print_exc(file=sys.stderr)
            cfg2 = {"error": "%s: %s" % (type(exc).__name__, exc)}
            ops.prof_enable(False)

    # BASELINE.json configs[4]: one optimisation step of train_ldm.py:76-86 at the per-GPU shape of "global batch 1024 over
    # 8 GPUs, 512x512 -> latents [128, 8, 64, 64]": q-sample, UNet forward with the tape, L1 loss, hand-written backward,
    # (N > 1) the gradient all-reduce, AdamW.  Random-init formula weights, train mode (stochastic depth live), synthetic latents.
    train_step = None
    if not args.no_train_step_leg:
        try:
            from ldm_image_generator_amd import train as ltrain
            torch.cuda.empty_cache()
            net.train(True)
            # train_ldm.py:67 constructs torch.optim.AdamW; its fused=True flavour (one kernel per parameter chunk, same update rule)
            # keeps the host out of the way: the foreach default costs 10-28 ms of host-bound time per step on 1376 tensors
            opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, fused=True)
            # AdamW creates a parameter's state (step, exp_avg, exp_avg_sq: three zero-fills) the first time that parameter has a gradient;
            # with 2-of-4 experts drawn per block and step, "first times" keep happening for dozens of steps (833 tiny fills per step in
            # round 2's profile).  A run of any length has all of it allocated after its first epoch: allocate it before the timed steps.
            for group in opt.param_groups:
                for p_ in group["params"]:
                    st_ = opt.state[p_]
                    if len(st_) == 0:
                        st_["step"] = torch.zeros((), dtype=torch.float32, device=p_.device)
                        st_["exp_avg"] = torch.zeros_like(p_, memory_format=torch.preserve_format)
                        st_["exp_avg_sq"] = torch.zeros_like(p_, memory_format=torch.preserve_format)
            xb = torch.randn(args.train_batch, 8, args.train_latent, args.train_latent,
                             generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
            train_step = {"unit": "samples/s", "config": {"workload": "train_ldm step, latents [%d, 8, %d, %d] per GPU, UNet(385.7M) train mode, "
                                                                      "L1 loss, torch.optim.AdamW(fused=True)" % (args.train_batch, args.train_latent, args.train_latent),
                                                          "global_batch": args.train_batch * world, "steps": args.train_steps, "warmup": args.train_warmup}}
            for prec in getattr(ltrain, "PRECISIONS", ("f32",)):
                ltrain.set_precision(net, prec) if hasattr(ltrain, "set_precision") else None
                torch.cuda.reset_peak_memory_stats()
                for wi in range(args.train_warmup):                                   # warm-up: allocator, RCCL buffers, weight caches, and the
                    ldist.train_step(ddpm, opt, xb, 10000 + wi, world)                # optimizer state of the experts a step happens to pick
                fence()
                tstats = {}
                t0 = time.perf_counter()
                for i in range(args.train_steps):
                    loss = ldist.train_step(ddpm, opt, xb, 1 + i, world, stats=tstats)
                fence()
                dts = max_over_ranks(time.perf_counter() - t0)
                leg = {"ms_per_step": dts / args.train_steps * 1e3, "value": args.train_batch * world * args.train_steps / dts,
                       "loss": float(loss.detach()), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
                # per-kernel times come from the SAME steps run once more with a hipEvent pair around every MFMA launch (~1 000 events per
                # step: they cost the bf16 step 1-3 ms, so they stay out of the timed steps, as in the autocast leg); every rank runs them
                ops.prof_enable(rank == 0)
                t0 = time.perf_counter()
                for i in range(args.train_steps):
                    ldist.train_step(ddpm, opt, xb, 1 + i, world)
                fence()
                leg["ms_per_step_with_events"] = max_over_ranks(time.perf_counter() - t0) / args.train_steps * 1e3
                if world > 1:                                  # bucketed all-reduce overlapped with the backward (dist.GradSync): what was NOT hidden
                    leg["allreduce_ms_exposed"] = tstats.get("allreduce_ms_exposed")
                    leg["allreduce_bytes_per_step"] = tstats.get("allreduce_bytes")
                if prec == "bf16":
                    leg["gemm_ring"] = ops.gemm_ring(-1)      # 1 = 256x256 ring kernel for the large plain NT GEMMs (bit-identical to 0)
                if rank == 0:
                    per, tot_ms, tot_fl = {}, 0.0, 0.0
                    tot_by = 0.0
                    for cls, name in PROF_CLASSES.items():
                        n, ms, fl = ops.prof_read_class(cls)
                        if n:
                            by = ops.prof_read_bytes(cls)
                            per[name] = {"launches_per_step": n // args.train_steps, "ms_per_step": ms / args.train_steps,
                                         "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                         "algorithmic_gb_per_s": by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None}
                            tot_ms += ms
                            tot_fl += fl
                            tot_by += by
                    ops.prof_read()
                    peak = BF16_MFMA_PEAK_TFLOPS if prec == "bf16" else FP32_MFMA_PEAK_TFLOPS
                    ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
                    gbs = tot_by / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
                    leg["executed_gflop_per_step"] = tot_fl / 1e9 / args.train_steps
                    leg["mfma_kernel_ms_per_step"] = tot_ms / args.train_steps
                    mfma_view = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
                    hbm_view = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "note": "algorithmic operand bytes of the MFMA kernels (every operand once) / their kernel time"}
                    # fp32 operands: the exact-fp32 MFMA paces the step; bf16 operands: the same GEMMs are 16x cheaper and the step
                    # is paced by operand traffic (ridge 2500 / 6.3 = 400 FLOP/B vs 96-384 FLOP/B of these layers)
                    leg["roofline"] = dict(hbm_view if prec == "bf16" else mfma_view)
                    leg["roofline"].update({"kernel": "all MFMA kernels of the step (NT, TN weight-gradient, grouped conv), hipEvents per launch",
                                            "other_view": mfma_view if prec == "bf16" else hbm_view, "per_kernel": per})
                ops.prof_enable(False)
                train_step[prec] = leg
            if hasattr(ltrain, "set_precision"):
                ltrain.set_precision(net, "f32")
            if "bf16" in train_step and "f32" in train_step:
                train_step["bf16_over_f32"] = train_step["bf16"]["value"] / train_step["f32"]["value"]
            del opt, xb
            net.train(args.mode == "train")
        except Exception as exc:           # a secondary leg never takes the headline line down with it
            import traceback
            traceback.print_exc(file=sys.stderr)
            train_step = {"error": "%s: %s" % (type(exc).__name__, exc)}
            ops.prof_enable(False)
            net.train(args.mode == "train")

    # one iteration of train_vae.py's loop (train_vae.py:104-127) -- VAE objective (L1 reconstruction x 10 + VQ loss +
    # 0.1 x generator hinge through the Discriminator) backward + optimizer, then the Discriminator's own hinge step -- on synthetic
    # 256x256 images, batch 8 (the script's defaults are batch 1, 192x192 crops), exact fp32.  torch.optim.AdamW(fused=True) stands in
    # for transformers' Adafactor (not on the hot path; `transformers` optimizers are out of scope).
    vae_step = None
    if not args.no_vae_train_leg:
        try:
            from ldm_image_generator_amd.vae import VAE, Discriminator, Encoder, VectorQuantizer
            torch.cuda.empty_cache()
            enc_v, dec_v, disc_v = Encoder(), Decoder(), Discriminator()
            for m_ in (enc_v, dec_v, disc_v):
                m_.load_state_dict(synth.fill_state_dict(m_.state_dict()))
            torch.manual_seed(1234)
            vq_v = VectorQuantizer()
            vae_v = VAE(enc_v, dec_v, vq_v).to(dev)
            disc_v = disc_v.to(dev)
            opt_v = torch.optim.AdamW(vae_v.parameters(), lr=1e-4, fused=True)
            opt_dv = torch.optim.AdamW(disc_v.parameters(), lr=1e-4, fused=True)
            vb, vs = args.vae_batch, args.vae_size
            img_v = (torch.rand(vb, 3, vs, vs, generator=torch.Generator().manual_seed(7 + rank)) * 2 - 1).to(dev)

            def vae_iter():
                opt_v.zero_grad()
                recon, reg, y = vae_v.calclate_loss(img_v)
                adv = torch.relu(-disc_v.calclate_logit(y)).mean()
                (recon * 10.0 + reg * 1.0 + adv * 0.1).backward()
                opt_v.step()
                opt_dv.zero_grad()
                y = y.detach()
                d_loss = torch.relu(1 + disc_v.calclate_logit(y)).mean() + torch.relu(1 - disc_v.calclate_logit(img_v)).mean()
                d_loss.backward()
                opt_dv.step()
                return recon, reg, adv, d_loss

            for _ in range(2):
                vae_iter()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.vae_steps):
                losses = vae_iter()
            fence()
            dtv = max_over_ranks(time.perf_counter() - t0)
            ops.prof_enable(rank == 0)                       # per-kernel times: the same iterations once more, with hipEvents (see train_step)
            for _ in range(args.vae_steps):
                vae_iter()
            fence()
            vae_step = {"ms_per_step": dtv / args.vae_steps * 1e3, "value": vb * world * args.vae_steps / dtv, "unit": "images/s", "dtype": "f32",
                        "config": {"workload": "train_vae.py iteration: VAE loss + generator hinge backward + AdamW, then Discriminator hinge step + AdamW; "
                                               "images [%d, 3, %d, %d] per GPU, Encoder / Decoder / VectorQuantizer(8192 x 8) / Discriminator at default widths" % (vb, vs, vs),
                                   "steps": args.vae_steps, "warmup": 2},
                        "losses": {"recon": float(losses[0]), "reg": float(losses[1]), "adv": float(losses[2]), "disc": float(losses[3])},
                        "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
            if rank == 0:
                per, tot_ms, tot_fl, tot_by = {}, 0.0, 0.0, 0.0
                for cls, name in PROF_CLASSES.items():
                    n, ms, fl = ops.prof_read_class(cls)
                    if n:
                        by = ops.prof_read_bytes(cls)
                        per[name] = {"launches_per_step": n // args.vae_steps, "ms_per_step": ms / args.vae_steps, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else None}
                        tot_ms, tot_fl, tot_by = tot_ms + ms, tot_fl + fl, tot_by + by
                ops.prof_read()
                if tot_ms > 0:
                    ach = tot_fl / (tot_ms * 1e-3) / 1e12
                    vae_step["executed_gflop_per_step"] = tot_fl / 1e9 / args.vae_steps
                    vae_step["mfma_kernel_ms_per_step"] = tot_ms / args.vae_steps
                    vae_step["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
                                            "kernel": "all MFMA kernels of the iteration (implicit 3x3 NT GEMMs, TN / NT weight-gradient GEMMs), hipEvents per launch",
                                            "per_kernel": per}
            ops.prof_enable(False)
            del vae_v, disc_v, opt_v, opt_dv, img_v
            torch.cuda.empty_cache()
        except Exception as exc:           # a secondary leg never takes the headline line down with it
            import traceback
            traceback.print_exc(file=sys.stderr)
            vae_step = {"error": "%s: %s" % (type(exc).__name__, exc)}
            ops.prof_enable(False)

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
            "gemm_variant": knobs["gemm_variant"], "wide_epilogue": knobs["wide_epilogue"], "gemm_ring": knobs["gemm_ring"],
            "gemm_variant_note": "1 = exact fp32 (v_mfma_f32_32x32x2_f32), the schedule `value` was measured under; 2 appears only in split_schedule",
            "film_tables": ("per_loop: every sample() call computes the FiLM tables of all its timesteps in its first denoise step (they depend on t, "
                            "not on x); bit-identical to per-step" if getattr(net, "hoist_films", False) else "per_step"),
            "roofline": {"bound": "mfma", "kernel": "ldm_gemm_f32 family (gemm_ring_kernel, gemm_stream_kernel, gconv3x3_pipe_kernel; v_mfma_f32_32x32x2_f32), all launches of the timed region",
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "launches": launches, "kernel_ms": gemm_ms,
                         "algorithmic_bytes_per_launch": gemm_bytes / launches if launches else None,
                         "gflop_per_sample_step_measured": None if args.mode != "eval" else gemm_flops / 1e9 / (B * args.steps) / T},
        }
        tname, tj = traffic_table()
        if tj is not None:      # PMC passes are separate runs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
            line["roofline"]["traffic"] = tj["gemm_hbm_bytes_per_launch"]
            line["roofline"]["traffic_unit"] = "bytes per GEMM launch (2*FETCH_SIZE + WRITE_SIZE, profiles/%s)" % tname.replace(".json", ".md")
            if launches:
                line["roofline"]["traffic_over_algorithmic"] = tj["gemm_hbm_bytes_per_launch"] / (gemm_bytes / launches)
            for k in ("fetch_over_algorithmic", "write_over_algorithmic", "per_instance"):
                if k in tj:
                    line["roofline"][k] = tj[k]
        if world > 1:
            line["dist_world_size"] = dist.get_world_size()
            line["dist_backend"] = dist.get_backend()
            line["sharded_equals_unsharded"] = shard_check
        if slices_check is not None:
            line["slices_check"] = slices_check
        if train_leg is not None:
            line["train_mode"] = train_leg
        if split is not None:
            line["split_schedule"] = split
        if amp is not None:
            line["autocast_bf16"] = amp
        if train_step is not None:
            line["train_step"] = train_step
        if cfg2 is not None:
            line["cfg2"] = cfg2
        if vae_step is not None:
            line["vae_train_step"] = vae_step
        if not args.no_cpu_baseline and world == 1:            # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"], cpu_out = cpu_baseline(min(os.cpu_count() or 1, 64))
            line["cpu_baseline"]["gpu_vs_cpu_rel_l2"] = gpu_vs_cpu(probe, cpu_out)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
