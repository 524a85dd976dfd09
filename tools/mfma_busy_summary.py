"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE` pass into
profiles/<tag>_mfma_busy.md.

Units (guide /opt/skills/guides/MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over all
SIMDs (64 per v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel's cycle count is
GRBM_GUI_ACTIVE / 8 and the chip offers 256 CUs x 4 SIMDs x that many MFMA-issue cycles.

    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import sys

SIMDS = 256 * 4


def short_name(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("gemm_stream_kernel", "gemm_ring_kernel", "gemm_f32_kernel"):
        if k in name:
            return name[name.find(k):].split("(")[0]
    return name.split("(")[0].split("<")[0][-48:]


def in_gemm_family(key):
    """Kernels launched by ldm_gemm_f32 (the roofline's "dominant kernel" family)."""
    return key.startswith("gemm_") or key.startswith("gconv3x3") or key.startswith("splitk")


def main(pmc_dir, tag):
    f = glob.glob(pmc_dir + "/**/*counter_collection*.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = short_name(name)
        d = per[(key, r["Dispatch_Id"])]
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for (key, _), d in per.items():
        a = agg[key]
        a["n"] += 1
        for k, v in d.items():
            a[k] += v
    rows = sorted(agg.items(), key=lambda kv: -kv[1]["ns"])
    tot = collections.defaultdict(float)
    with open("profiles/%s_mfma_busy.md" % tag, "w") as out:
        out.write("# MFMA-busy per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE)\n\n")
        out.write("command: `bench.py --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline` (B = 256, 2 denoise steps + decode)\n\n")
        out.write("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / wall time.\n\n")
        out.write("| kernel | launches | avg us | MFMA busy | clock GHz | share of GPU time |\n|---|---|---|---|---|---|\n")
        all_ns = sum(a["ns"] for _, a in rows)
        for key, a in rows[:14]:
            cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * cyc) if cyc else 0.0
            out.write("| `%s` | %d | %.1f | %.3f | %.2f | %.1f %% |\n" % (key[:64], a["n"], a["ns"] / a["n"] / 1e3, busy,
                                                                      cyc / max(a["ns"], 1), 100.0 * a["ns"] / all_ns))
            if in_gemm_family(key):
                for k in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "ns"):
                    tot[k] += a.get(k, 0.0)
        cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
        if cyc:
            out.write("\nGEMM family, time-weighted: MFMA busy %.3f at %.2f GHz effective clock (profiled pass).\n"
                      % (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), cyc / tot["ns"]))
    print(open("profiles/%s_mfma_busy.md" % tag).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
