// One (d, H) instantiation of the DenseNet-control forward kernels; compiled once per line of dense_instances.def.
#include "hjbd_kernels.h"
#ifndef PSP_D
#error "compile with -DPSP_D=<d> -DPSP_H=<H>"
#endif
#define PSP_DEFINE_D_(D_, H_) PSP_DEFINE_DNET_INSTANCE(D_, H_)
PSP_DEFINE_D_(PSP_D, PSP_H)
