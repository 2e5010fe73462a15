#!/usr/bin/env python
"""Condense the SQ / GRBM counter passes of tools/prof_gf_sq.sh into a per-kernel table.

    python tools/gf_sq_summary.py TAG KIND [--md out.md]

Reads gpurun_out/TAG_gfsq_KIND, TAG_gfsq2_KIND, TAG_gfgrbm_KIND (rocprofv3 counter_collection and
kernel_trace CSVs; counters are summed over the chip by rocprofv3) and prints, per kernel (median
dispatch): duration, waves, VGPRs, SQ_WAVE_CYCLES and SQ_BUSY_CYCLES-derived waves per SIMD, VALU
instructions per wave, share of wave time issuing / stalled on issue / parked in s_waitcnt,
cycles per VALU instruction per SIMD, LDS busy share, bank-conflict share, effective clock.
SQ_* "cycles" counters tick once per 4 shader cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rf::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)


def load(dirname):
    """{kernel: {counter: median value, '_ms': median duration, '_vgpr':..}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(G, dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[k]["_ms"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
            out[k]["_vgpr"].append(float(r.get("VGPR_Count", 0) or 0) + float(r.get("Accum_VGPR_Count", 0) or 0))
            out[k]["_lds"].append(float(r.get("LDS_Block_Size", 0) or 0))
            out[k]["_grid"].append(float(r.get("Grid_Size", 0) or 0))
            out[k]["_wg"].append(float(r.get("Workgroup_Size", 0) or 0))
    med = {}
    for k, d in out.items():
        med[k] = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    return med


def main():
    tag, kind = sys.argv[1], sys.argv[2]
    a = load("%s_gfsq_%s" % (tag, kind))
    b = load("%s_gfsq2_%s" % (tag, kind))
    g = load("%s_gfgrbm_%s" % (tag, kind))
    rows = []
    for k in sorted(a, key=lambda k: -a[k]["_ms"]):
        if not k.startswith("gf_") or a[k]["_ms"] < 0.02:
            continue
        x, y, z = a[k], b.get(k, {}), g.get(k, {})
        wave_cyc = x["SQ_WAVE_CYCLES"] * 4          # shader cycles summed over waves
        busy = y.get("SQ_BUSY_CYCLES", 0) * 4       # per-SE busy; not used for occupancy
        waves = y.get("SQ_WAVES", 0)
        ms = x["_ms"]
        clk = z.get("GRBM_GUI_ACTIVE", 0) / 8.0 / (z.get("_ms", ms) * 1e-3) / 1e6 if z else 0
        cyc_kernel = ms * 1e-3 * (clk or 2400) * 1e6
        waves_per_simd = wave_cyc / (cyc_kernel * 1024) if cyc_kernel else 0
        valu = x["SQ_INSTS_VALU"]
        rows.append({
            "kernel": k, "ms": ms, "waves": waves, "vgpr": x["_vgpr"], "lds_B": x["_lds"],
            "wg": x["_wg"], "waves_per_simd": waves_per_simd,
            "valu_per_wave": valu / waves if waves else 0,
            "lds_inst_per_wave": x["SQ_INSTS_LDS"] / waves if waves else 0,
            "vmem_rd_per_wave": y.get("SQ_INSTS_VMEM_RD", 0) / waves if waves else 0,
            "salu_per_wave": y.get("SQ_INSTS_SALU", 0) / waves if waves else 0,
            "issuing": x["SQ_ACTIVE_INST_ANY"] / x["SQ_WAVE_CYCLES"],
            "issue_stall": x["SQ_WAIT_INST_ANY"] / x["SQ_WAVE_CYCLES"],
            "waitcnt": x["SQ_WAIT_ANY"] / x["SQ_WAVE_CYCLES"],
            "cyc_per_valu_per_simd": cyc_kernel * 1024 / valu if valu else 0,
            "lds_busy": x["SQ_LDS_IDX_ACTIVE"] / (cyc_kernel * 256) if cyc_kernel else 0,
            "lds_conflict": x["SQ_LDS_BANK_CONFLICT"] / max(1.0, x["SQ_LDS_IDX_ACTIVE"]),
            "clock_mhz": clk,
        })
    hdr = ("kernel", "ms", "waves", "vgpr", "lds_B", "waves_per_simd", "valu_per_wave",
           "lds_inst_per_wave", "vmem_rd_per_wave", "salu_per_wave", "issuing", "issue_stall",
           "waitcnt", "cyc_per_valu_per_simd", "lds_busy", "lds_conflict", "clock_mhz")
    lines = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
    for r in rows:
        cells = []
        for h in hdr:
            v = r[h]
            cells.append(v if isinstance(v, str) else ("%.3f" % v if abs(v) < 10 else "%.0f" % v))
        lines.append("| " + " | ".join(cells) + " |")
    text = "\n".join(lines)
    print(text)
    if "--md" in sys.argv:
        with open(sys.argv[sys.argv.index("--md") + 1], "a") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
