// rf_gf_fused_inst.hip -- instantiates the fused guided-filter stage 2 (rf_gf_fused.hpp) for the
// radii r = RF_GF_PART + 1, RF_GF_PART + 1 + RF_GF_PARTS, ... <= kGfFusedSmallMax (rf_gf_fused_<part>.o),
// or - with -DRF_GF_LARGE - for r = kGfFusedSmallMax + 1 + RF_GF_PART, + RF_GF_PARTS, ... <=
// kGfFusedMaxRadius (rf_gf_fused_L<part>.o: the large radii, whose unrolled bodies take a third of
// the compile time at -O1 and lose nothing - their registers are spoken for either way).  The
// Makefile compiles this file once per part so that the per-radius kernels build in parallel; rf_gf.hip
// asks the parts for a radius's launcher (gf_fused_launcher).
#include "rf_gf_fused.hpp"

#ifndef RF_GF_PART
#error "compile with -DRF_GF_PART=<0..RF_GF_PARTS-1> -DRF_GF_PARTS=<n>"
#endif

#ifdef RF_GF_LARGE
#define RF_GF_FIRST (rf::kGfFusedSmallMax + 1 + RF_GF_PART)
#define RF_GF_LAST rf::kGfFusedMaxRadius
#define RF_GF_FN RF_CAT(gf_fused_large_, RF_GF_PART)
#else
#define RF_GF_FIRST (RF_GF_PART + 1)
#define RF_GF_LAST rf::kGfFusedSmallMax
#define RF_GF_FN RF_CAT(gf_fused_part_, RF_GF_PART)
#endif
#define RF_CAT2(a, b) a##b
#define RF_CAT(a, b) RF_CAT2(a, b)

namespace rf {
namespace {
template <int R>
GfFusedLaunch find(int radius)
{
    if constexpr (R > RF_GF_LAST) {
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

GfFusedLaunch RF_GF_FN(int radius) { return find<RF_GF_FIRST>(radius); }

}  // namespace rf
