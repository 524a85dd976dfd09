"""Turn rocprofv3's <tag>_kernel_stats.csv (from --kernel-trace --stats) into profiles/<tag>_kernel_stats.{csv,md}."""
import collections
import csv
import glob
import shutil
import statistics
import sys


def main(stats_dir, tag, title):
    found = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)
    if found:
        shutil.copy(found[0], "profiles/%s_kernel_stats.csv" % tag)
    else:                               # rocpd output (no stats csv): aggregate the kernel trace ourselves, same columns
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(glob.glob(stats_dir + "/**/*kernel_trace.csv", recursive=True)[0])):
            agg.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = float(sum(sum(v) for v in agg.values()))
        with open("profiles/%s_kernel_stats.csv" % tag, "w", newline="") as out:
            w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100 * sum(v) / tot, 2), min(v), max(v),
                            statistics.pstdev(v) if len(v) > 1 else 0.0])
    rows = list(csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag)))
    with open("profiles/%s_kernel_stats.md" % tag, "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats, %s\n\n" % title)
        out.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:16]:
            out.write("| `%s` | %s | %.1f | %.1f | %.1f |\n" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        gemm = [r for r in rows if "gemm" in r["Name"]]
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        out.write("\nGEMM family: %d launches, %.1f %% of GPU time, %.1f us per launch on average.\n"
                  % (sum(int(r["Calls"]) for r in gemm), 100 * sum(float(r["TotalDurationNs"]) for r in gemm) / tot,
                     sum(float(r["TotalDurationNs"]) for r in gemm) / max(sum(int(r["Calls"]) for r in gemm), 1) / 1e3))
    print(open("profiles/%s_kernel_stats.md" % tag).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else
         "`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline` (2 passes of 50 steps + decode, B=256)")
