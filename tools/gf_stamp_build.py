#!/usr/bin/env python
"""Diagnostic build of the guided filter's column walk with s_memtime stamps around its phases.

Compiles reflectance_filtering_amd/csrc/rf_gf_fused.hpp with -DRF_GF_STAMP for ONE radius into
reflectance_filtering_amd/librf_hip.so.stamp (the other objects are the product's own; the
product library is untouched).  The stamps are summed over waves in a device buffer of their own
(rf_debug_gf_stamps reads and clears it); no output depends on them.

    make -C reflectance_filtering_amd/csrc && python tools/gf_stamp_build.py [radius] \
        && gpurun -- python3 tools/gf_stamp_run.py [radius]

RF_STAMP_DEFS adds compiler flags (the experiment switches of rf_gf_fused.hpp: -DRF_GF_EXP_ALIAS,
-DRF_GF_EXP_HOT - operand fetches that hit in cache, results wrong, timing only), RF_STAMP_SUFFIX
names the library (librf_hip.so.stamp<suffix>; gf_stamp_run.py reads the same variable).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reflectance_filtering_amd", "csrc")
radius = int(sys.argv[1]) if len(sys.argv) > 1 else 45
src = r'''
#define RF_GF_STAMP 1
#include "rf_gf_fused.hpp"
namespace rf {
#define RF_PART(K) GfFusedLaunch gf_fused_part_##K(int radius) { return radius == %d ? &gf_fused_launch<%d> : nullptr; }
RF_PART(0) RF_PART(1) RF_PART(2) RF_PART(3) RF_PART(4) RF_PART(5) RF_PART(6) RF_PART(7)
#define RF_LARGE(K) GfFusedLaunch gf_fused_large_##K(int) { return nullptr; }
RF_LARGE(0) RF_LARGE(1) RF_LARGE(2) RF_LARGE(3)
}
extern "C" int rf_debug_gf_stamps(unsigned long long *out16)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rf::g_cw_stamps), sizeof(zero)) != hipSuccess)
        return -4;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rf::g_cw_stamps), zero, sizeof(zero)) != hipSuccess)
        return -4;
    return 0;
}
''' % (radius, radius)
path = os.path.join(CSRC, "_gf_stamp_tmp.hip")
with open(path, "w") as fh:
    fh.write(src)
obj = os.path.join(CSRC, "_gf_stamp_tmp.o")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-gpu-rdc",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
flags += os.environ.get("RF_STAMP_DEFS", "").split()   # experiment switches of rf_gf_fused.hpp
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", path, "-o", obj])
objs = [os.path.join(CSRC, o) for o in ("rf_api.o", "rf_jbf.o", "rf_gf.o", "rf_cnn.o",
                                        "rf_colorize.o", "rf_whdr.o")]
out = os.path.join(ROOT, "reflectance_filtering_amd",
                   "librf_hip.so.stamp" + os.environ.get("RF_STAMP_SUFFIX", ""))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out,
                       obj] + objs)
os.remove(path)
os.remove(obj)
print("built", out)
