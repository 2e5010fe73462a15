// rf_gf_fused_inst.hip -- instantiates the fused guided-filter stage 2 (rf_gf_fused.hpp) for the
// radii r = RF_GF_PART + 1, RF_GF_PART + 1 + RF_GF_PARTS, ... <= kGfFusedMaxRadius.  The Makefile
// compiles this file RF_GF_PARTS times (rf_gf_fused_<part>.o) so that the per-radius kernels build
// in parallel; rf_gf.hip asks the parts for a radius's launcher (gf_fused_launcher).
#include "rf_gf_fused.hpp"

#ifndef RF_GF_PART
#error "compile with -DRF_GF_PART=<0..RF_GF_PARTS-1> -DRF_GF_PARTS=<n>"
#endif

namespace rf {
namespace {
template <int R>
GfFusedLaunch find(int radius)
{
    if constexpr (R > kGfFusedMaxRadius) {
        return nullptr;
    } else {
#ifdef RF_GF_DEV_ONLY  // development builds: only this radius is instantiated (make GF_DEV_ONLY=45)
        if constexpr (R == RF_GF_DEV_ONLY)
#endif
        if (radius == R)
            return &gf_fused_launch<R>;
        return find<R + RF_GF_PARTS>(radius);
    }
}
}  // namespace

#define RF_CAT2(a, b) a##b
#define RF_CAT(a, b) RF_CAT2(a, b)
GfFusedLaunch RF_CAT(gf_fused_part_, RF_GF_PART)(int radius) { return find<RF_GF_PART + 1>(radius); }

}  // namespace rf
