#!/usr/bin/env python
"""Who runs beside whom in a guided-filter step: overlap table from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d gpurun_out/ovl -- python3 bench.py --config c5 --steps 2 --warmup 1 \
        --traffic off --cpu-seconds 0 --no-extras
    python tools/gf_overlap.py gpurun_out/ovl [--label aligned] [--out profiles/r05_c5_overlap.md]

Reads every *_kernel_trace.csv below the directory, keeps the guided-filter kernels, splits them
into steps at idle gaps (> --gap-ms between the end of everything so far and the next start), and
for the LAST step reports
  * per kernel class (stage 1 / row walk / column walk / probe): launches, summed duration, and the
    time during which at least one launch of the class was running (union);
  * the step's span and the time covered by stage 1 only, by walks only, by both, by neither;
  * per stage-1 launch: its duration and the fraction of it during which a walk kernel of ANOTHER
    queue was running (the co-run the two-stream schedule is meant to produce).
Timestamps are the dispatch begin / end stamps of the trace (ns); a kernel "runs" between them,
which says nothing about how many of its workgroups are resident - occupancy is argued separately
(DESIGN.md 3.2).
"""
import argparse
import csv
import glob
import os
import sys


def classify(name):
    if "gf_stage1_kernel" in name:
        return "stage1"
    if "gf_rowstate_kernel" in name or "gf_rowsum_kernel" in name or "gf_rowhead_kernel" in name:
        return "row walk"
    if "gf_colwalk_kernel" in name or "gf_colsum_apply_kernel" in name:
        return "column walk"
    if "gf_grey_probe_kernel" in name:
        return "probe"
    return None


def union(intervals):
    out = []
    for a, b in sorted(intervals):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(u1, u2):
    i = j = 0
    out = []
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if a < b:
            out.append([a, b])
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return out


def load(path):
    files = [path] if os.path.isfile(path) else sorted(
        glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                cls = classify(r["Kernel_Name"])
                if cls is None:
                    continue
                rows.append({"cls": cls, "q": (r.get("Queue_Id"), r.get("Stream_Id")),
                             "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]),
                             "name": r["Kernel_Name"].split("(")[0].replace("void rf::(anonymous namespace)::", "")
                             .replace("void rf::", "")})
    rows.sort(key=lambda r: r["t0"])
    return rows


def split_steps(rows, gap_ns):
    steps, cur, end = [], [], None
    for r in rows:
        if end is not None and r["t0"] - end > gap_ns:
            steps.append(cur)
            cur = []
        cur.append(r)
        end = r["t1"] if end is None else max(end, r["t1"])
    if cur:
        steps.append(cur)
    return steps


def report(step, label, brief=False):
    ms = 1e-6
    t0 = min(r["t0"] for r in step)
    t1 = max(r["t1"] for r in step)
    lines = []
    lines.append("## %s: one step, %.2f ms from the first guided-filter dispatch to the last end, "
                 "%d launches on %d queues" % (label, (t1 - t0) * ms, len(step),
                                               len({r["q"] for r in step})))
    lines.append("")
    lines.append("| kernel class | launches | summed duration ms | running (union) ms |")
    lines.append("|---|---|---|---|")
    un = {}
    for cls in ("stage1", "row walk", "column walk", "probe"):
        sel = [r for r in step if r["cls"] == cls]
        if not sel:
            continue
        un[cls] = union([(r["t0"], r["t1"]) for r in sel])
        lines.append("| %s | %d | %.2f | %.2f |" % (cls, len(sel), sum(r["t1"] - r["t0"] for r in sel) * ms,
                                                    length(un[cls]) * ms))
    s1 = un.get("stage1", [])
    walks = union([(r["t0"], r["t1"]) for r in step if r["cls"] in ("row walk", "column walk")])
    both = intersect(s1, walks)
    anyk = union([(r["t0"], r["t1"]) for r in step])
    lines.append("")
    lines.append("| of the step's %.2f ms | ms | share |" % ((t1 - t0) * ms))
    lines.append("|---|---|---|")
    span = float(t1 - t0)
    for what, v in (("stage 1 and a walk kernel both running", length(both)),
                    ("stage 1 only", length(s1) - length(both)),
                    ("walk kernels only", length(walks) - length(both)),
                    ("neither (probe / idle)", span - length(union([tuple(x) for x in s1] + [tuple(x) for x in walks])))):
        lines.append("| %s | %.2f | %.0f %% |" % (what, v * ms, 100.0 * v / span))
    lines.append("| any guided-filter kernel running | %.2f | %.0f %% |" % (length(anyk) * ms, 100.0 * length(anyk) / span))
    lines.append("")
    if not brief:
        lines.append("Per stage-1 launch (in start order; launches of the instantiation that does not apply "
                     "to the images exit at once and are left out): duration, and the share of it during which a "
                     "row- or column-walk kernel of ANOTHER queue was running:")
        lines.append("")
        lines.append("| # | queue | start ms | duration ms | walk of another queue beside it |")
        lines.append("|---|---|---|---|---|")
    fr = []
    for k, r in enumerate([r for r in step if r["cls"] == "stage1" and r["t1"] - r["t0"] > 300000]):
        other = union([(x["t0"], x["t1"]) for x in step
                       if x["cls"] in ("row walk", "column walk") and x["q"] != r["q"]])
        cov = length(intersect([[r["t0"], r["t1"]]], other))
        f = cov / float(max(1, r["t1"] - r["t0"]))
        fr.append((f, r["t1"] - r["t0"]))
        if not brief:
            lines.append("| %d | %s | %.2f | %.2f | %.0f %% |" % (k, "/".join(str(x) for x in r["q"]),
                                                                 (r["t0"] - t0) * ms, (r["t1"] - r["t0"]) * ms, 100 * f))
    if fr:
        tot = sum(d for _, d in fr)
        lines.append("")
        lines.append("Duration-weighted mean over the stage-1 launches: **%.0f %%**."
                     % (100.0 * sum(f * d for f, d in fr) / tot))
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace", help="a *_kernel_trace.csv or a directory that holds some")
    ap.add_argument("--label", default="schedule")
    ap.add_argument("--gap-ms", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=0,
                    help="the trace holds this many back-to-back steps (no idle gap between them): cut "
                         "the launches, in start order, into that many equal runs instead of at gaps")
    ap.add_argument("--brief", action="store_true", help="leave the per-launch table out")
    ap.add_argument("--out", default=None, help="append the table to this file")
    args = ap.parse_args()
    rows = load(args.trace)
    if not rows:
        sys.exit("gf_overlap: no guided-filter kernels in %s" % args.trace)
    if args.steps > 0:
        per = len(rows) // args.steps
        steps = [rows[k * per:(k + 1) * per] for k in range(args.steps)]
    else:
        steps = split_steps(rows, args.gap_ms * 1e6)
    steps = [s for s in steps if any(r["cls"] == "stage1" for r in s)]
    txt = report(steps[-1], "%s (%d steps in the trace, the last one shown)" % (args.label, len(steps)),
                 brief=args.brief)
    sys.stdout.write(txt)
    if args.out:
        with open(args.out, "a") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
