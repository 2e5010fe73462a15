#!/usr/bin/env python
"""Condense tools/prof_c5_traffic.sh's passes into profiles/TAG_c5_traffic.{md,json}.

    python tools/make_profiles_c5.py TAG [STEPS_PROFILED]

The profiled command is `bench.py --config c5 --steps 2 --warmup 1` = 3 steps of 128 x 4K images x
3 guided passes: per guided-filter kernel the launches, their average and summed duration, and
FETCH_SIZE / WRITE_SIZE (KiB in the counter files) summed over the launches of ONE step, as bytes
per pixel of a pass (128 x 3840 x 2160 pixels x 3 passes per step).  FETCH_SIZE is corrected with the
factor tools/microbench/fetch_calib.hip measured in the same run for the load width each kernel uses
(MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-byte requests as 64 bytes).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, H, W, PASSES = 128, 2160, 3840, 3


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rf::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name).replace("unsigned char", "u8")


def one(pattern):
    files = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


def counter_sum(run, name, want=lambda k: "rf::" in k):
    path = one("%s_%s/*/*counter_collection.csv" % (TAG, run))
    per = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if want(r["Kernel_Name"]) and r["Counter_Name"] == name:
            per[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
            cnt[short(r["Kernel_Name"])] += 1
    return per, cnt


def calibration():
    """bytes really read / FETCH_SIZE bytes, per kernel of fetch_calib.bin"""
    per, _ = counter_sum("calib", "FETCH_SIZE", want=lambda k: "read" in k)
    truth = {"read12_kernel": 3 << 30, "read16_kernel": 3 << 30, "readT_kernel<u8>": 1 << 30,
             "readT_kernel<unsigned int>": 3 << 30, "readT_kernel<HIP_vector_type<unsigned int, 2u> >": 3 << 30}
    out = {}
    for k, v in per.items():
        for t, b in truth.items():
            if k.startswith(t) or k == t:
                out[k] = {"bytes_read": b, "fetch_size_bytes": v, "factor": b / v if v else None}
    return out


def main():
    px_pass = N * H * W
    stats = collections.OrderedDict()
    for r in csv.DictReader(open(one("%s_c5_stats/*/*kernel_stats.csv" % TAG))):
        if "rf::" in r["Name"]:
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                      "total_ms": float(r["TotalDurationNs"]) / 1e6}
    fetch, fcnt = counter_sum("c5_fetch", "FETCH_SIZE")
    write, _ = counter_sum("c5_write", "WRITE_SIZE")
    calib = calibration()
    doc = {"tag": TAG, "steps_profiled": STEPS, "calibration": calib, "kernels": {}}
    lines = ["# %s - guided filter at the C5 shard: per-kernel time and memory-side traffic" % TAG, "",
             "`tools/prof_c5_traffic.sh %s`: `rocprofv3 --kernel-trace --stats`, `--pmc FETCH_SIZE` and "
             "`--pmc WRITE_SIZE` (three separate runs) of `python3 bench.py --config c5 --steps 2 --warmup 1 "
             "--traffic off --cpu-seconds 0 --no-extras` (%d steps of 128 x 3840x2160 x 3 passes; two halves of "
             "a chunk run on two streams, so kernel durations add up to more than the wall time; counter "
             "passes serialise the kernels)." % (TAG, STEPS), "",
             "FETCH_SIZE calibration in the same run (`tools/microbench/fetch_calib.bin`, every kernel reads "
             "its buffer exactly once):", "",
             "| load width | bytes read | FETCH_SIZE reported | factor |", "|---|---|---|---|"]
    for k, c in calib.items():
        lines.append("| `%s` | %.3f GB | %.3f GB | %.2f |" % (k, c["bytes_read"] / 1e9,
                                                              c["fetch_size_bytes"] / 1e9, c["factor"]))
    lines += ["", "| kernel | launches per step | avg ms | ms per step (sum) | FETCH B/px/pass (x2) | WRITE B/px/pass | B/px/pass |",
              "|---|---|---|---|---|---|---|"]
    tot = tot_ms = 0.0
    for k, s in stats.items():
        f = 2.0 * fetch.get(k, 0.0) / STEPS / PASSES / px_pass
        wr = write.get(k, 0.0) / STEPS / PASSES / px_pass
        ms_step = s["total_ms"] / STEPS
        doc["kernels"][k] = {"launches_per_step": s["calls"] / STEPS, "avg_ms": s["avg_ms"],
                             "ms_per_step": ms_step, "fetch_b_px_pass": f, "write_b_px_pass": wr}
        if ms_step < 0.05:
            continue
        tot += f + wr
        tot_ms += ms_step
        lines.append("| `%s` | %.0f | %.3f | %.2f | %.1f | %.1f | %.1f |"
                     % (k, s["calls"] / STEPS, s["avg_ms"], ms_step, f, wr, f + wr))
    doc["bytes_per_px_pass"] = tot
    doc["kernel_ms_per_step"] = tot_ms
    log = open(os.path.join(G, "%s_c5_stats.log" % TAG)).read()
    for ln in log.splitlines():
        if ln.startswith("{"):
            b = json.loads(ln)
            doc["bench_value_mp_s"] = b["value"]
            doc["bench_ms_per_step"] = b["ms_per_step"]
    lines += ["", "Sum: **%.1f B/px per pass** through the memory side (algorithmic 9; 21 for the chain of three "
              "with shared guide) ; kernels %.1f ms per step added up; the step under the profiler: %.2f ms = "
              "%.0f MP/s." % (tot, tot_ms, doc.get("bench_ms_per_step", 0), doc.get("bench_value_mp_s", 0)), ""]
    with open(os.path.join(ROOT, "profiles", "%s_c5_traffic.md" % TAG), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(ROOT, "profiles", "%s_c5_traffic.json" % TAG), "w") as fh:
        json.dump(doc, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
