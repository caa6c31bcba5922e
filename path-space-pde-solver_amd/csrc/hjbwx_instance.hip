// One (d, H) instantiation of the split-product role-specialised backward of the wide family (hjbwx_kernels.h, d <= 256); compiled
// once per line of wide_instances.def with d <= 256, WITHOUT the SLP vectoriser and with the two-instruction operand split.
#include "hjbwx_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_WBX_CAT_(a, b, c) a##b##_##c
#define PSP_WBX_DEFINE_(D_, H_)                                                                                      \
    extern "C" int PSP_WBX_CAT_(psp_wide_bwd2x_ok_, D_, H_)() { return psp::HjbwxLaunch<D_, H_>::kOk ? 1 : 0; }       \
    extern "C" hipError_t PSP_WBX_CAT_(psp_wide_bwd2x_, D_, H_)(const psp::HjbArgs* a, int grid, hipStream_t s) {     \
        return psp::HjbwxLaunch<D_, H_>::bwd(*a, grid, s);                                                           \
    }
#define PSP_WBX_DEFINE(D_, H_) PSP_WBX_DEFINE_(D_, H_)
PSP_WBX_DEFINE(PSP_D, PSP_H)
