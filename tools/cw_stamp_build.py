#!/usr/bin/env python
"""Diagnostic build of the guided filter's column walk with s_memtime stamps around its phases
(staging, fetch issue, row chains, column phase, flush).  Patches a COPY of rf_gf.hip, compiles it
and links reflectance_filtering_amd/librf_hip.so.st; tools/cw_exp.py runs it and prints cycles per
sub-tile.  The stamps go to a buffer of their own and no output depends on them; the product
library is untouched.

    make -C reflectance_filtering_amd/csrc && python tools/cw_stamp_build.py && gpurun -- python3 tools/cw_exp.py
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reflectance_filtering_amd", "csrc")
s = open(os.path.join(CSRC, "rf_gf.hip")).read()


def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)


rep("constexpr int kSB = 16;      // columns per state block and per column-walk wave",
    "__device__ unsigned long long g_cw_stamps[8];\nconstexpr int kSB = 16;")
rep('''        const int uu_ = (u_);                                                                \\
        RF_ROWTAB(uu_ + 1);''', '''        const int uu_ = (u_);                                                                \\
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();                         \\
        RF_ROWTAB(uu_ + 1);''')
rep('''        double s_ = pst;                                                                     \\
        __syncthreads();                                                                     \\
        if (!(FILL))''', '''        double s_ = pst;                                                                     \\
        __syncthreads();                                                                     \\
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                         \\
        if (!(FILL))''')
rep('''        if (chain) {                                                                         \\
            /* all operand differences first''', '''        const unsigned long long t2_ = __builtin_amdgcn_s_memtime();                         \\
        if (chain) {                                                                         \\
            /* all operand differences first''')
rep('''        __syncthreads();                                                                     \\
        {                                                                                    \\
            /* the sub-tile's row sums first''', '''        __syncthreads();                                                                     \\
        const unsigned long long t3_ = __builtin_amdgcn_s_memtime();                         \\
        {                                                                                    \\
            /* the sub-tile's row sums first''')
rep('''        if (!(FILL)) {                                                                       \\
            _Pragma("unroll") for (int k = 0; k < NG; k++)''', '''        const unsigned long long t4_ = __builtin_amdgcn_s_memtime();                         \\
        acc_st[0] += t1_ - t0_; acc_st[1] += t2_ - t1_; acc_st[2] += t3_ - t2_; acc_st[3] += t4_ - t3_; \\
        if (!(FILL)) {                                                                       \\
            _Pragma("unroll") for (int k = 0; k < NG; k++)''')
rep('''            __syncthreads();                                                                 \\
        }                                                                                    \\
    } while (0)

    RF_ROWTAB(0);''', '''            __syncthreads();                                                                 \\
            acc_st[4] += __builtin_amdgcn_s_memtime() - t4_;                                 \\
        }                                                                                    \\
    } while (0)

    unsigned long long acc_st[5] = {0, 0, 0, 0, 0};
    RF_ROWTAB(0);''')
rep('''#undef RF_SUB
#undef RF_GUIDE_FETCH''', '''    if (lane == 0 && item == 100) {
        for (int q = 0; q < 5; q++)
            g_cw_stamps[q] = acc_st[q];
        g_cw_stamps[5] = (unsigned long long)nsub;
    }
#undef RF_SUB
#undef RF_GUIDE_FETCH''')
rep('''extern "C" size_t rf_gf_workspace_bytes(''', '''extern "C" int rf_cw_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rf::g_cw_stamps), 64);
}

extern "C" size_t rf_gf_workspace_bytes(''')
src = "/tmp/rf_gf_stamped.hip"
open(src, "w").write(s)
obj = "/tmp/rf_gf_stamped.o"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                       "-ffp-contract=off", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                       "-I" + CSRC, "-c", src, "-o", obj])
objs = [os.path.join(CSRC, o) for o in ("rf_api.o", "rf_jbf.o", "rf_cnn.o", "rf_colorize.o", "rf_whdr.o")]
out = os.path.join(ROOT, "reflectance_filtering_amd", "librf_hip.so.st")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
print("built", out)
