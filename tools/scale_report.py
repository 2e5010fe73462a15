#!/usr/bin/env python
"""1 -> N GPU scaling curve of bench.py in one command (image batches shard with no
collective, weak scaling - every GPU keeps the config's per-GPU batch).

    python tools/scale_report.py [--gpus 1,2,4,8] [--config north_star|c4|c5|c3|c2]
                                 [--steps 5] [--warmup 1] [--out profiles/scale_<config>.json]

Runs `bench.py --gpus N --config C` for every N the node has devices for (bench.py starts its
own ranks, one process per GPU), prints absolute MP/s and the ratio to N = 1, and writes the
parsed bench lines to --out.  Counts above the visible device count are skipped, not faked.
The north-star target is >= 7.5x at 8 GPUs.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def visible_devices():
    import torch                       # device_count() does not initialise the GPU
    return torch.cuda.device_count()


def run_bench(n, config, steps, warmup, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", config,
           "--steps", str(steps), "--warmup", str(warmup), "--cpu-seconds", "0", "--no-extras"]
    cmd += extra
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, cwd=ROOT)
    if proc.returncode != 0:
        raise SystemExit("bench.py --gpus %d failed with exit code %d" % (n, proc.returncode))
    lines = [l for l in proc.stdout.decode().splitlines() if l.startswith("{")]
    if len(lines) != 1:
        raise SystemExit("bench.py --gpus %d printed %d JSON lines" % (n, len(lines)))
    return json.loads(lines[0])


def report(rows):
    base = rows[0]["value"] / rows[0]["n_gpus"]
    out = []
    for r in rows:
        out.append({"n_gpus": r["n_gpus"], "value": r["value"], "unit": r["unit"],
                    "ms_per_step": r["ms_per_step"], "ratio_to_one_gpu": r["value"] / base,
                    "per_gpu": r["value"] / r["n_gpus"]})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--config", default="north_star")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (smoke runs)")
    args = ap.parse_args(argv)
    counts = sorted({int(x) for x in args.gpus.split(",") if x})
    have = visible_devices() if os.environ.get("RF_BENCH_STUB") != "1" else max(counts)
    extra = ["--batch", str(args.batch)] if args.batch else []
    rows, skipped = [], []
    for n in counts:
        if n > have:
            skipped.append(n)
            continue
        rows.append(run_bench(n, args.config, args.steps, args.warmup, extra))
    if not rows:
        raise SystemExit("no GPU count in %s fits the %d visible device(s)" % (counts, have))
    table = report(rows)
    print("%-6s %14s %12s %10s" % ("GPUs", rows[0]["unit"], "ms/step", "x 1 GPU"))
    for t in table:
        print("%-6d %14.1f %12.2f %10.2f" % (t["n_gpus"], t["value"], t["ms_per_step"],
                                             t["ratio_to_one_gpu"]))
    if skipped:
        print("skipped (only %d device(s) visible): %s" % (have, ", ".join(map(str, skipped))))
    doc = {"config": args.config, "metric": rows[0]["metric"], "workload": rows[0]["config"],
           "scaling": rows[0]["scaling"], "visible_devices": have, "skipped": skipped,
           "curve": table, "bench_lines": rows}
    out = args.out or os.path.join(ROOT, "gpurun_out", "scale_%s.json" % args.config)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", out)
    return doc


if __name__ == "__main__":
    main()
