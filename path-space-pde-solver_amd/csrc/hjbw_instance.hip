// One (d, H) instantiation of the wide rollout kernels; compiled once per line of wide_instances.def.
#include "hjbw_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_DEFINE_W_(D_, H_) PSP_DEFINE_WIDE_INSTANCE(D_, H_)
PSP_DEFINE_W_(PSP_D, PSP_H)
