#!/bin/bash
# SQ counters of the float joint-bilateral kernels at 8 x 1080p (through gpurun from the repo root):
#   tools/f32_sq.sh
export TMPDIR=/tmp
rm -rf gpurun_out/f32sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/f32sq -- python3 tools/f32_time.py 8 > gpurun_out/f32sq.log 2>&1
python3 - <<"PY"
import csv, glob, collections
f = glob.glob("gpurun_out/f32sq/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "jbf_f32_quad" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("jbf_f32_quad_kernel")[1].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        acc[k]["_vgpr"].append(float(r["VGPR_Count"]))
        acc[k]["_lds"].append(float(r["LDS_Block_Size"]))
for k, v in acc.items():
    m = {c: sorted(x)[len(x) // 2] for c, x in v.items()}
    wc = m["SQ_WAVE_CYCLES"]
    cyc = m["_dur"] * 1e-3 * 2.3e9
    print(k, "ms %.2f vgpr %d lds %d" % (m["_dur"], m["_vgpr"], m["_lds"]),
          "issuing %.2f issue-stall %.2f waitcnt %.2f" % (m["SQ_ACTIVE_INST_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_WAIT_ANY"] / wc),
          "waves/simd %.2f" % (wc * 4 / (cyc * 1024)), "cycles/valu/simd %.2f" % (cyc * 1024 / m["SQ_INSTS_VALU"]),
          "lds busy %.2f conflicts %.2f" % (m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), m["SQ_LDS_BANK_CONFLICT"] / max(1, m["SQ_LDS_IDX_ACTIVE"])))
PY
