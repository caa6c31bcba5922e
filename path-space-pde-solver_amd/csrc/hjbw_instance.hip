// One (d, H) instantiation of the wide rollout kernels; compiled once per line of wide_instances.def.
#include "hjbw_kernels.h"
#include "hjbc_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
// six or more 32-feature steps (d > 160): the cooperative split-product forward (hjbc_kernels.h) beside the tile-per-wave one
template <int D_, int H_>
static void psp_add_coop(psp::HjbInstance& r) {
    if constexpr (psp::GeoW<D_, H_>::KS8 >= 6) {
        r.coop_lds_bytes = &psp::HjbcLaunch<D_, H_>::lds_bytes;
        r.launch_fwd_coop = &psp::HjbcLaunch<D_, H_>::fwd;
    }
}
#undef PSP_DEFINE_WIDE_INSTANCE
#define PSP_DEFINE_WIDE_INSTANCE(D_, H_)                                       \
    extern "C" psp::HjbInstance psp_wide_instance_##D_##_##H_() {              \
        psp::HjbInstance r = psp::HjbwLaunch<D_, H_>::instance();              \
        psp_add_coop<D_, H_>(r);                                               \
        return r;                                                              \
    }
#define PSP_DEFINE_W_(D_, H_) PSP_DEFINE_WIDE_INSTANCE(D_, H_)
PSP_DEFINE_W_(PSP_D, PSP_H)
