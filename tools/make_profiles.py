#!/usr/bin/env python
"""Condense the rocprofv3 outputs under gpurun_out/ into the committed summaries in profiles/.

Expects the directories written by the measurement command recorded in profiles/README.md:
prof_stats (--kernel-trace --stats), prof_fetch / prof_write (--pmc FETCH_SIZE / WRITE_SIZE on
bench.py), prof_calib (--pmc FETCH_SIZE on tools/microbench/fetch_calib.bin), prof_sq / prof_grbm
(SQ and GRBM counters on a 32-image launch) and bench_default.json.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
G = os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):
    files = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


def short_name(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def counters(path, match):
    per = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = short_name(r["Kernel_Name"])
    return per, names


def main():
    os.makedirs(OUT, exist_ok=True)
    lines = []
    # ---- kernel stats
    stats = newest("prof_stats/*/*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(OUT, "%s_bench_kernel_stats.csv" % TAG), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs",
                    "MaxNs", "StdDev"])
        for r in rows[:10]:
            name = r["Name"] if len(r["Name"]) <= 140 else r["Name"][:137] + "..."
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    jbf = [r for r in rows if "jbf" in r["Name"]][0]
    avg_ms = float(jbf["AverageNs"]) / 1e6
    # ---- bench line
    bench = json.load(open(os.path.join(G, "bench_default.json")))
    with open(os.path.join(OUT, "%s_bench.json" % TAG), "w") as fh:
        json.dump(bench, fh, indent=1)
    n, h, w_ = (bench["config"]["batch_per_gpu"], bench["config"]["height"],
                bench["config"]["width"])
    # ---- calibration of FETCH_SIZE
    cal, names = counters(newest("prof_calib/*/*counter_collection.csv"), "read1")
    calib = {}
    for d, v in cal.items():
        calib["read12" if "read12" in names[d] else "read16"] = v["FETCH_SIZE"] * 1024.0
    true_bytes = float(3 << 30)
    f12 = true_bytes / calib["read12"]
    f16 = true_bytes / calib["read16"]
    # ---- traffic of the bench launch
    fe, _ = counters(newest("prof_fetch/*/*counter_collection.csv"), "jbf")
    wr, _ = counters(newest("prof_write/*/*counter_collection.csv"), "jbf")
    fetch_raw = sorted(v["FETCH_SIZE"] for v in fe.values())[len(fe) // 2] * 1024.0
    write = sorted(v["WRITE_SIZE"] for v in wr.values())[len(wr) // 2] * 1024.0
    fetch = fetch_raw * f12
    traffic = {"batch": n, "height": h, "width": w_,
               "fetch_size_raw_bytes": fetch_raw, "fetch_calibration_factor": f12,
               "fetch_bytes": fetch, "write_bytes": write,
               "hbm_bytes_per_launch": fetch + write,
               "algorithmic_bytes_per_launch": 9.0 * n * h * w_,
               "note": "FETCH_SIZE/WRITE_SIZE are in KiB; separate --pmc passes; FETCH_SIZE "
                       "scaled by the factor measured with tools/microbench/fetch_calib.hip for "
                       "12-byte-per-lane loads (16-byte-per-lane factor %.3f, cf. "
                       "MI355X_MICROARCH.md HBM section)" % f16}
    with open(os.path.join(OUT, "jbf_pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    # ---- SQ counters: grey-src launch (the metric's input) and, when profiled, colour-src launch
    def sq_column(sq_dir, grbm_dir):
        sq_path = newest("%s/*/*counter_collection.csv" % sq_dir)
        gr_path = newest("%s/*/*counter_collection.csv" % grbm_dir)
        if not sq_path or not gr_path:
            return None
        sq, sqn = counters(sq_path, "jbf")
        gr, _ = counters(gr_path, "jbf")
        first = max(sq, key=lambda d: sq[d]["SQ_INSTS_VALU"])      # the big launch, not a warm-up probe
        c = sq[first]
        gui = max(v["GRBM_GUI_ACTIVE"] for v in gr.values()) / 8.0
        # kernel duration of that dispatch from the trace of the GRBM pass
        return {"name": sqn[first], "c": c, "gui": gui}

    simds, cus = 1024.0, 256.0
    cols = [("grey src (the metric's launch shape, 32 images)", sq_column("prof_sq", "prof_grbm"))]
    if cols[0][1]:
        # bench.py's `lds.busy_measured` reads this (the LDS pipeline's busy share of the metric kernel)
        c0 = cols[0][1]
        traffic["lds_busy"] = c0["c"]["SQ_LDS_IDX_ACTIVE"] / cus / c0["gui"]
        traffic["lds_busy_source"] = ("profiles/%s_jbf_pmc.md: SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8), "
                                      "rocprofv3 --pmc passes of this kernel at batch 32 (tools/prof_bench.sh)" % TAG)
        with open(os.path.join(OUT, "jbf_pmc_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    colour = sq_column("prof_sq_colour", "prof_grbm_colour")
    if colour:
        cols.append(("3-channel colour src (32 images)", colour))
    lines += [
        "# %s — rocprofv3 summaries for bench.py (one MI355X)" % TAG, "",
        "Kernel: `%s`, launch = %d x %dx%d images." % (short_name(jbf["Name"]), n, w_, h), "",
        "| quantity | value |", "|---|---|",
        "| bench.py value | %.0f MP/s (%.1f ms per 256-image step) |" % (bench["value"], bench["ms_per_step"]),
        "| kernel average duration, `--kernel-trace --stats` (%s calls) | %.2f ms |" % (jbf["Calls"], avg_ms),
        "| kernel duration from HIP events inside bench.py | %.2f ms |" % bench["roofline"]["kernel_ms"],
        "| shader clock under the launch (bench.py probe, s_memtime / s_memrealtime) | %.0f MHz |" % bench.get("valu", {}).get("clock_mhz", float("nan")),
        "| algorithmic bytes per launch (9 B/px) | %.3f GB |" % (traffic["algorithmic_bytes_per_launch"] / 1e9),
        "| achieved on algorithmic bytes | %.1f GB/s = %.3f %% of 8 TB/s |" % (bench["roofline"]["achieved"], 100 * bench["roofline"]["frac"]),
        "| WRITE_SIZE per launch | %.3f GB (algorithmic 3 B/px = %.3f GB) |" % (write / 1e9, 3.0 * n * h * w_ / 1e9),
        "| FETCH_SIZE per launch, raw | %.3f GB (x%.1f calibration = %.3f GB; algorithmic 6 B/px = %.3f GB) |" % (fetch_raw / 1e9, f12, fetch / 1e9, 6.0 * n * h * w_ / 1e9),
        "| FETCH_SIZE calibration (3 GiB read once): 12 B/lane loads | counter reads 1/%.3f of the bytes |" % f12,
        "| FETCH_SIZE calibration: 16 B/lane loads | counter reads 1/%.3f of the bytes |" % f16,
        "| HBM-side traffic per launch (calibrated fetch + write) | %.3f GB = %.2f x algorithmic |" % ((fetch + write) / 1e9, (fetch + write) / traffic["algorithmic_bytes_per_launch"]),
        "| colour-src launch (secondary) | %.0f MP/s |" % bench.get("colour_src", {}).get("value", float("nan")),
        "| CPU baseline (oracle, %d threads) | %.2f MP/s |" % (bench["cpu_baseline"]["cores"], bench["cpu_baseline"]["value"]),
        "",
        "SQ counters of the same kernel (`%s`), one column per src kind: the grey tap loop issues 26 "
        "VALU instructions per 4-output column step, the colour loop 44 (three accumulators per "
        "output)." % cols[0][1]["name"], "",
        "| counter | " + " | ".join(t for t, _ in cols) + " | reading |",
        "|---|" + "---|" * len(cols) + "---|",
    ]

    def row(label, fn, reading):
        return "| %s | %s | %s |" % (label, " | ".join(fn(v["c"], v["gui"]) for _, v in cols), reading)

    lines += [
        row("GRBM_GUI_ACTIVE / 8", lambda c, g: "%.3e" % g, "kernel length in shader cycles"),
        row("SQ_INSTS_VALU", lambda c, g: "%.3e" % c["SQ_INSTS_VALU"], "VALU wave-instructions"),
        row("cycles per VALU wave-instruction per SIMD", lambda c, g: "%.2f" % (g * simds / c["SQ_INSTS_VALU"]),
            "floor 2.0 (one per 2 cycles per SIMD)"),
        row("VALU per LDS instruction", lambda c, g: "%.2f" % (c["SQ_INSTS_VALU"] / c["SQ_INSTS_LDS"]), ""),
        row("LDS busy (SQ_LDS_IDX_ACTIVE)", lambda c, g: "%.0f %%" % (100 * c["SQ_LDS_IDX_ACTIVE"] / cus / g),
            "share of the kernel"),
        row("bank conflicts", lambda c, g: "%.2f %%" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]),
            "of LDS-active cycles"),
        row("issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)", lambda c, g: "%.0f %%" % (100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
            "share of wave time"),
        row("ready, not issued (SQ_WAIT_INST_ANY)", lambda c, g: "%.0f %%" % (100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
            "pipe busy"),
        row("parked (SQ_WAIT_ANY)", lambda c, g: "%.0f %%" % (100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]),
            "s_waitcnt / barrier"),
    ]
    with open(os.path.join(OUT, "%s_jbf_pmc.md" % TAG), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
