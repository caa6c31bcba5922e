// One (d, H) instantiation of the rollout kernels; compiled once per line of instances.def.
#include "hjb_kernels.h"
#include "hjbs_kernels.h"
#include "hjba_kernels.h"
#include "hjbq_kernels.h"
#include "hjbx_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_CAT_(a, b) a##b
#define PSP_DEFINE_(D_, H_) PSP_DEFINE_INSTANCE(D_, H_)
#undef PSP_DEFINE_INSTANCE
#define PSP_DEFINE_INSTANCE(D_, H_)                                                       \
    extern "C" psp::HjbInstance psp_instance_##D_##_##H_() {                                \
        psp::HjbInstance r = psp::HjbLaunch<D_, H_>::instance();                           \
        r.split_lds_bytes = &psp::HjbsLaunch<D_, H_>::lds_bytes;                           \
        r.launch_fwd_split = &psp::HjbsLaunch<D_, H_>::fwd;                                \
        r.launch_adj = &psp::HjbaLaunch<D_, H_>::adj;                                      \
        r.launch_fwd_bf16 = &psp::HjbLaunch<D_, H_>::fwd_bf16;                             \
        r.fwd_x3_lds_bytes = &psp::HjbLaunch<D_, H_>::fwd_x3_lds;                          \
        r.launch_fwd_x3 = &psp::HjbLaunch<D_, H_>::fwd_x3;                                 \
        r.bwd2_x3_lds_bytes = &psp::HjbxLaunch<D_, H_>::lds_bytes;                         \
        r.launch_bwd2_x3 = &psp::HjbxLaunch<D_, H_>::bwd;                                  \
        r.launch_adj_x3 = &psp::HjbaLaunch<D_, H_>::adj_x3;                                \
        r.quad_lds_bytes = &psp::HjbqLaunch<D_, H_>::lds_bytes;                            \
        r.launch_fwd_quad = &psp::HjbqLaunch<D_, H_>::fwd;                                 \
        r.launch_adj_quad = &psp::HjbqLaunch<D_, H_>::adj;                                 \
        return r;                                                                           \
    }
PSP_DEFINE_(PSP_D, PSP_H)
