#!/usr/bin/env python3
"""Prints the figures DESIGN.md quotes from the files under profiles/r02/ (so that prose and evidence cannot drift):
   python tools/design_numbers.py [profiles/r02]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02")
PEAK = 78.6


def rows(name):
    path = os.path.join(P, name)
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def main():
    b = json.loads(open(os.path.join(P, "bench_n1.json")).read().strip().split("\n")[-1])
    r = b["roofline"]
    print("bench: value %.1f M codon-frames/s, kernel %.2f ms, %.1f TFLOP/s, frac %.3f; strong leg %.1f M"
          % (b["value"] / 1e6, r["kernel_ms"], r["achieved"], r["frac"], b["strong"]["value"] / 1e6))
    e = b["end_to_end"]
    c = b["cpu_baseline"]
    print("end to end 1 Mb: %.3f s (runs %s) = %.1f M cf/s; process wall %.2f s; CPU driver %.1f s = %.3f M cf/s -> %.0fx; one thread %.1f k cf/s; differing wig lines %s"
          % (e["seconds"], e["seconds_all_runs"], e["codon_frames_per_s"] / 1e6, e["process_wall_seconds"], c["end_to_end_seconds"],
             c["end_to_end_value"] / 1e6, e["vs_cpu_driver"], c["single_thread_value"] / 1e3, e.get("wig_lines_differing_from_cpu_driver")))
    s = e.get("steady_state", {})
    if "seconds" in s:
        print("steady state (x10): %.3f s = %.1f M cf/s, stage busy %s" % (s["seconds"], s["codon_frames_per_s"] / 1e6, s["stage_busy"]))
    print("cpu_baseline kernel level: %.3f M cf/s on %d threads (%s)" % (c["value"] / 1e6, c["cores"], c["cpu_model"]))
    for row in rows("kernel_stats_bench.csv"):
        if "prune_fixed" in row["Name"]:
            ms = float(row["AverageNs"]) / 1e6
            tf = 2_000_000 * r["flop_per_codon_frame"] / (ms * 1e-3) / 1e12
            print("rocprofv3 bench: prune_fixed avg %.3f ms over %s calls -> %.1f TFLOP/s = %.3f of peak" % (ms, row["Calls"], tf, tf / PEAK))
    pm = json.load(open(os.path.join(P, "pmc_summary.json")))
    g = lambda k: pm[k]["mean_per_dispatch"]
    print("PMC per launch: SQ_INSTS_MFMA %.3g, SQ_INSTS_VALU %.3g (non-MFMA %.3g = %.1f per 64-MFMA product), MFMA busy cycles / 64 = %.3g, "
          "FETCH_SIZE*2+WRITE_SIZE = %.2f GB, L2 hit rate %.3f, clock %.2f GHz"
          % (g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA"), (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / (g("SQ_INSTS_MFMA") / 64),
             g("SQ_VALU_MFMA_BUSY_CYCLES") / 64, (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9, g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
             g("GRBM_GUI_ACTIVE") / 8 / (r["kernel_ms"] * 1e-3) / 1e9))
    print("block strategies (516 blocks):")
    for name in ("blocks_bench_516.txt", "blocks_bench_eigen_forms.txt", "blocks_bench_516_explicit_only.txt", "blocks_bench_2064.txt"):
        path = os.path.join(P, name)
        if os.path.exists(path):
            for line in open(path):
                if line.startswith("GPU MLE") or line.startswith("GPU OMEGA") or line.startswith("run_omega: 516") or line.startswith("run_omega: 2064"):
                    print("   %-36s %s" % (name, line.strip()))
    tot = sum(float(x["TotalDurationNs"]) for x in rows("blocks_kernel_stats.csv") if "pcsf::" in x["Name"])
    for x in rows("blocks_kernel_stats.csv"):
        if "pcsf::" in x["Name"]:
            print("   kernel %-40s %5.1f %%  %4s launches, avg %.3f ms" % (x["Name"].split("(")[0][-40:], 100 * float(x["TotalDurationNs"]) / tot, x["Calls"], float(x["AverageNs"]) / 1e6))
    path = os.path.join(P, "blocks_pmc_summary.json")
    if os.path.exists(path):
        d = json.load(open(path))
        dur = {x["Name"].split("(")[0]: float(x["TotalDurationNs"]) * 1e-9 for x in rows("blocks_kernel_stats.csv")}
        for k, v in d.items():
            if "pcsf" not in k:
                continue
            val = lambda c: v.get(c, {}).get("sum", 0.0)
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs: utilisation while the kernel runs =
            # busy / (1024 SIMDs x seconds the kernel ran x clock); durations from the kernel-trace run of the same command
            secs = dur.get(k, 0.0)
            util = val("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * secs * 2.39e9) if secs else 0.0
            print("   PMC %-40s VALU %.3g (MFMA %.3g) LDS %.3g bank-conflict cycles %.3g waves %.3g; per wave: %.0f VALU + %.0f LDS instructions; MFMA pipe busy while running %.3f"
                  % (k[-40:], val("SQ_INSTS_VALU"), val("SQ_INSTS_MFMA"), val("SQ_INSTS_LDS"), val("SQ_LDS_BANK_CONFLICT"), val("SQ_WAVES"),
                     val("SQ_INSTS_VALU") / max(val("SQ_WAVES"), 1), val("SQ_INSTS_LDS") / max(val("SQ_WAVES"), 1), util))
    for name in ("config4/text_run1.txt", "config4/text_run2.txt", "config4/packed_run1.txt", "config4/packed_run2.txt", "config4/pack.txt", "config4/outputs.txt"):
        path = os.path.join(P, name)
        if os.path.exists(path):
            t = open(path).read()
            m = re.search(r"(\d+) codon-frames.*\n\s+seconds: total ([\d.]+)", t)
            if m:
                print("config 4, %-24s %s codon-frames in %s s = %.1f M cf/s" % (name, m.group(1), m.group(2), int(m.group(1)) / float(m.group(2)) / 1e6))
            else:
                print("config 4, %-24s %s" % (name, t.strip().split("\n")[0][:160]))


if __name__ == "__main__":
    main()
