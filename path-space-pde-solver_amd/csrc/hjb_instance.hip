// One (d, H) instantiation of the rollout kernels; compiled once per line of instances.def.
#include "hjb_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_CAT_(a, b) a##b
#define PSP_DEFINE_(D_, H_) PSP_DEFINE_INSTANCE(D_, H_)
PSP_DEFINE_(PSP_D, PSP_H)
