// One (d, H) instantiation of the wide rollout kernels; compiled once per line of wide_instances.def.
#include "hjbw_kernels.h"
#include <cstdlib>

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
// d <= 256: the split-product role-specialised backward lives in its own translation unit (hjbwx_instance.hip: other compiler flags)
#if PSP_D <= 256
#define PSP_WBX_CAT_(a, b, c) a##b##_##c
#define PSP_WBX_NAME(pre, D_, H_) PSP_WBX_CAT_(pre, D_, H_)
extern "C" int PSP_WBX_NAME(psp_wide_bwd2x_ok_, PSP_D, PSP_H)();
extern "C" hipError_t PSP_WBX_NAME(psp_wide_bwd2x_, PSP_D, PSP_H)(const psp::HjbArgs* a, int grid, hipStream_t s);
static hipError_t psp_bwd2x_thunk(const psp::HjbArgs& a, int grid, hipStream_t s) {
    return PSP_WBX_NAME(psp_wide_bwd2x_, PSP_D, PSP_H)(&a, grid, s);
}
static hipError_t (*psp_bwd2_fp32)(const psp::HjbArgs&, int, hipStream_t) = nullptr;
// PSP_WIDE_BWD_X3=0 keeps the fp32 kernel (A/B; read per call)
static hipError_t psp_bwd2x_or_fp32(const psp::HjbArgs& a, int grid, hipStream_t s) {
    const char* e = getenv("PSP_WIDE_BWD_X3");
    if (e && e[0] == '0') return psp_bwd2_fp32(a, grid, s);
    return psp_bwd2x_thunk(a, grid, s);
}
static void psp_add_bwd2x(psp::HjbInstance& r) {
    if (r.bwd2_one_per_cu && PSP_WBX_NAME(psp_wide_bwd2x_ok_, PSP_D, PSP_H)()) {      // (the role-specialised fp32 kernel is what it replaces)
        psp_bwd2_fp32 = r.launch_bwd2;
        r.launch_bwd2_x3 = &psp_bwd2x_or_fp32;
    }
}
#else
static void psp_add_bwd2x(psp::HjbInstance&) {}
#endif
#undef PSP_DEFINE_WIDE_INSTANCE
#define PSP_DEFINE_WIDE_INSTANCE(D_, H_)                                       \
    extern "C" psp::HjbInstance psp_wide_instance_##D_##_##H_() {              \
        psp::HjbInstance r = psp::HjbwLaunch<D_, H_>::instance();              \
        psp_add_coop<D_, H_>(r);                                               \
        psp_add_bwd2x(r);                                                      \
        return r;                                                              \
    }
#define PSP_DEFINE_W_(D_, H_) PSP_DEFINE_WIDE_INSTANCE(D_, H_)
PSP_DEFINE_W_(PSP_D, PSP_H)
