#!/usr/bin/env python
"""Condense the rocprofv3 passes of tools/prof_gf_cnn.sh into profiles/TAG_gf_cnn.{md,json}.

    python tools/make_profiles_gf.py TAG [GF_BATCH]

Per kernel of the guided filter (grey and colour src, 4K) and of the CNN (256 IIW images):
average duration from `--kernel-trace --stats`, FETCH_SIZE and WRITE_SIZE from their own
`--pmc` passes (KiB -> bytes; FETCH_SIZE doubled: on gfx950 the counter tallies 128-B requests
at 64 B, MI355X_MICROARCH.md section HBM, checked by tools/microbench/fetch_calib.hip), and the
resulting bytes per pixel next to the algorithmic figure of SURVEY.md 8d.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rf::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name).replace("unsigned char", "u8")


def one(pattern):
    files = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


def stats(run):
    path = one("%s_%s_stats/*/*kernel_stats.csv" % (TAG, run))
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "rf::" in r["Name"]:
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
    return out


def counter(run, name):
    path = one("%s_%s_%s/*/*counter_collection.csv" % (TAG, run, name))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "rf::" in r["Kernel_Name"] and r["Counter_Name"] == name:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sorted(v)[len(v) // 2] for k, v in per.items()}


def main():
    runs = (("gf", "guided filter r=45 eps=3, %d x 3840x2160, flat guide, grey src (1-channel "
             "planes)" % NB, NB * 2160 * 3840, 9.0),
            ("gfc", "guided filter r=45 eps=3, %d x 3840x2160, flat guide, colour src" % NB,
             NB * 2160 * 3840, 9.0),
            ("cnn", "1x1 CNN, 256 x 500x333", 256 * 333 * 500, 8.0))
    doc = {"tag": TAG, "runs": {}}
    lines = ["# %s - rocprofv3 summaries, guided filter and CNN (one MI355X)" % TAG, "",
             "Commands: `tools/prof_gf_cnn.sh %s %d` (per run one `--kernel-trace --stats` pass and one "
             "`--pmc` pass each for FETCH_SIZE and WRITE_SIZE); FETCH_SIZE doubled (gfx950: 128-B "
             "requests counted as 64 B)." % (TAG, NB), ""]
    for run, title, px, alg in runs:
        if not one("%s_%s_stats/*/*kernel_stats.csv" % (TAG, run)):
            continue
        st, fe, wr = stats(run), counter(run, "FETCH_SIZE"), counter(run, "WRITE_SIZE")
        lines += ["## " + title, "",
                  "| kernel | calls | avg ms per launch | FETCH x2 per launch (GB) | WRITE per launch (GB) | B/px moved per pass |",
                  "|---|---|---|---|---|---|"]
        # the profiled program makes 3 calls; a kernel launched once per half of the batch shows
        # 6 calls: columns are per LAUNCH, the totals per pass (= per call) count every launch
        tot_ms = tot_b = 0.0
        rec = {}
        for k, s in st.items():
            f, w = 2.0 * fe.get(k, 0.0), wr.get(k, 0.0)
            if s["avg_ms"] < 0.02:        # flag zeroing / probes / workgroups of the other variant
                continue
            per_pass = 1 if run == "cnn" else max(1, s["calls"] // 3)
            tot_ms += s["avg_ms"] * per_pass
            tot_b += (f + w) * per_pass
            rec[k] = {"avg_ms": s["avg_ms"], "launches_per_pass": per_pass, "fetch_bytes": f,
                      "write_bytes": w, "bytes_per_px": (f + w) * per_pass / px}
            lines.append("| `%s` | %d | %.3f | %.3f | %.3f | %.1f |"
                         % (k, s["calls"], s["avg_ms"], f / 1e9, w / 1e9, (f + w) * per_pass / px))
        wall_ms = tot_ms
        wall_note = "kernel durations added up"
        wj = one("gf_profile_wall_%s.json" % {"gf": "grey", "gfc": "colour"}.get(run, "none"))
        ev = one("%s_cnn_events.json" % TAG) if run == "cnn" else None
        if ev:
            evd = json.load(open(ev))
            wall_ms = evd["event_ms_median"]
            wall_note = ("median of HIP events around the same %d launches, in the same process, "
                         "under the profiler" % evd["launches"])
        if wj:
            wall_ms = json.load(open(wj))["wall_ms"]
            wall_note = ("wall time of a call, HIP events, under the profiler; the two halves of the "
                         "batch overlap on two streams, so it is less than the kernels added up")
        gbs = alg * px / (wall_ms * 1e-3) / 1e9
        lines += ["", "Per pass: %.3f ms of kernels added up, %.3f ms (%s); %.1f B/px through the memory "
                  "side against %.0f B/px algorithmic (%.1fx); %.0f MP/s; algorithmic bytes at %.1f GB/s "
                  "= %.2f %% of 8 TB/s; moved bytes at %.2f TB/s."
                  % (tot_ms, wall_ms, wall_note, tot_b / px, alg, tot_b / px / alg,
                     px / 1e6 / (wall_ms * 1e-3), gbs, 100 * gbs / 8000.0,
                     tot_b / (wall_ms * 1e-3) / 1e12), ""]
        doc["runs"][run] = {"title": title, "pixels": px, "kernels": rec, "kernel_ms_per_pass": tot_ms,
                            "wall_ms_per_pass": wall_ms,
                            "bytes_per_px_moved": tot_b / px, "algorithmic_bytes_per_px": alg,
                            "mp_per_s": px / 1e6 / (wall_ms * 1e-3),
                            "roofline_frac_algorithmic": gbs / 8000.0}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "%s_gf_cnn.md" % TAG), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(ROOT, "profiles", "%s_gf_cnn.json" % TAG), "w") as fh:
        json.dump(doc, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
