// pair_kernels_inst.hip — one translation unit per compiled SH order.
// Built once per order with -DSHP_L=<L> (L = -1: run-time order, loop kernel);
// exports shp_launch_L<L>() to the dispatch table in shpair_api.hip.
#include "pair_kernel.hpp"

#ifndef SHP_L
#error "compile with -DSHP_L=<order>"
#endif

#define SHP_CAT2(a, b) a##b
#define SHP_CAT(a, b) SHP_CAT2(a, b)
#if SHP_L < 0
#define SHP_FN shp_launch_Lrt
#define SHP_AFN shp_attr_Lrt
#else
#define SHP_FN SHP_CAT(shp_launch_L, SHP_L)
#define SHP_AFN SHP_CAT(shp_attr_L, SHP_L)
#endif

namespace shp {
void SHP_FN(const PairParams& P, bool needv, hipStream_t st, hipEvent_t w) { launch_pair_contact<SHP_L>(P, needv, st, w); }
hipError_t SHP_AFN(bool needv, bool weighted, hipFuncAttributes* a, bool jpoly, bool split, bool spec) { return pair_contact_attributes<SHP_L>(needv, weighted, a, jpoly, split, spec); }
}  // namespace shp
