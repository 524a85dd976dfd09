"""Summarise the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<tag>_traffic.{md,json}.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half the bytes of a wide
coalesced 16-B/lane stream (global_load and LDS-DMA alike), hence the x2 (guide, section HBM)."""
import collections
import csv
import glob
import json
import sys


def short_name(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("gemm_stream_kernel", "gemm_ring_kernel", "gemm_f32_kernel"):
        if k in name:
            return name[name.find(k):].split("(")[0]
    return name.split("(")[0].split("<")[0][-48:]


def in_gemm_family(key):
    """Kernels launched by ldm_gemm_f32 (the roofline's "dominant kernel" family)."""
    return key.startswith("gemm_") or key.startswith("gconv3x3") or key.startswith("splitk")


def per_kernel(pattern, counter):
    f = glob.glob(pattern, recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = short_name(name)
        a = agg[key]
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


if __name__ == "__main__":
    fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    F = per_kernel(fetch_dir + "/**/*counter_collection*.csv", "FETCH_SIZE")
    W = per_kernel(write_dir + "/**/*counter_collection*.csv", "WRITE_SIZE")
    rows, tot = [], [0, 0.0, 0.0]
    for k in sorted(F, key=lambda k: -F[k][2]):
        n, fb, ns = F[k]
        wb = W.get(k, [1, 0.0, 0.0])[1]
        rows.append(dict(kernel=k, launches=n, fetch_bytes_x2_per_launch=2 * fb / n, write_bytes_per_launch=wb / n,
                         avg_us=ns / n / 1e3))
        if in_gemm_family(k):
            tot[0] += n
            tot[1] += 2 * fb + wb
            tot[2] += ns
    summary = dict(gemm_launches=tot[0], gemm_hbm_bytes_per_launch=tot[1] / max(tot[0], 1),
                   gemm_avg_us=tot[2] / max(tot[0], 1) / 1e3, kernels=rows[:14])
    # ALGORITHMIC bytes per launch of the two dominant instances at the bench workload (B = 256, latent 32x32: M = 262144 >> level,
    # C = 128 << level, 6 / 6 / 18 / 6 SwinBlocks per level), every operand once:
    #   gated MoE GEMM: reads x [M, C] + 2 x 3 weight matrices [C, C], writes the hidden [M, 3C]
    #   VAE dense 3x3 conv (C >= 128 instance): reads the input once + weights, writes the output (+ reads the residual where fused)
    blocks = [6, 6, 18, 6]
    gate_r = sum(n * ((262144 >> (2 * i)) * (128 << i) * 4 + 6 * (128 << i) ** 2 * 4) for i, n in enumerate(blocks)) / 36.0
    gate_w = sum(n * ((262144 >> (2 * i)) * 3 * (128 << i) * 4) for i, n in enumerate(blocks)) / 36.0
    inst = {}
    # the gated GEMM runs on two kernels since round 2: the ring kernel's gated instance (stages 0-2) and the stream kernel's (stage 3)
    gate_rows = [r for r in rows if r["kernel"].startswith("gemm_stream_kernel<2, 2, 2, 1, true, 0, 0, true") or
                 r["kernel"].startswith("gemm_ring_kernel<0, 2, true")]
    if gate_rows:
        n = sum(r["launches"] for r in gate_rows)
        fb = sum(r["fetch_bytes_x2_per_launch"] * r["launches"] for r in gate_rows) / n
        wb = sum(r["write_bytes_per_launch"] * r["launches"] for r in gate_rows) / n
        inst["gated MoE GEMM"] = dict(kernel=" + ".join(r["kernel"] for r in gate_rows), launches=n, algorithmic_fetch=gate_r, algorithmic_write=gate_w,
                                      fetch_over_algorithmic=fb / gate_r, write_over_algorithmic=wb / gate_w,
                                      total_over_algorithmic=(fb + wb) / (gate_r + gate_w))
    summary["per_instance"] = inst
    json.dump(summary, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    with open("profiles/%s_traffic.md" % tag, "w") as out:
        out.write("# HBM-side traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n")
        out.write("command: `bench.py --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline` (B = 256, 2 denoise steps + decode)\n\n")
        out.write("FETCH_SIZE x2 (gfx950 wide-stream correction), WRITE_SIZE as read.  GEMM family: %d launches, "
                  "%.1f MB per launch on average, %.1f us per launch (profiled).\n\n" % (tot[0], summary["gemm_hbm_bytes_per_launch"] / 1e6, summary["gemm_avg_us"]))
        out.write("| kernel | launches | fetch x2 MB/launch | write MB/launch | avg us |\n|---|---|---|---|---|\n")
        for r in rows[:14]:
            out.write("| `%s` | %d | %.1f | %.1f | %.1f |\n" % (r["kernel"][:70], r["launches"], r["fetch_bytes_x2_per_launch"] / 1e6,
                                                           r["write_bytes_per_launch"] / 1e6, r["avg_us"]))
    print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}))
