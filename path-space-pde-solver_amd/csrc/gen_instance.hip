// One (d, H) instantiation of the GeneralSolver kernels; compiled once per line of gen_instances.def.
#include "gen_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_DEFINE_GEN_(D_, H_) PSP_DEFINE_GEN_INSTANCE(D_, H_)
PSP_DEFINE_GEN_(PSP_D, PSP_H)
