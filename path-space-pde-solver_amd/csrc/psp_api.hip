// psp_api.hip -- C ABI of libpsp_hip.so (see include/psp.h) + the small streaming kernels.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>   // types only: the entry points are bound with dlsym at first use (psp_comm_*)

#include "../../include/psp.h"
#include "hjb_kernels.h"
#include "gen_kernels.h"
#include "hjbw_kernels.h"
#include "hjbd_kernels.h"
#include "genl_kernels.h"

#define X(D_, H_) PSP_DECLARE_DNET_INSTANCE(D_, H_)
#include "dense_instances.def"
#undef X
#define X(D_, H_) PSP_DECLARE_INSTANCE(D_, H_)
#include "instances.def"
#undef X
#define X(D_, H_) PSP_DECLARE_WIDE_INSTANCE(D_, H_)
#include "wide_instances.def"
#undef X
#define X(D_, H_) PSP_DECLARE_GEN_INSTANCE(D_, H_)
#include "gen_instances.def"
#undef X

namespace {

thread_local char g_err[512] = "";
unsigned long long* g_dbg = nullptr;   // psp_debug_set_stamp_buffer
long long g_dbg_n = 0;

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
int fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return -10;
}

typedef psp::HjbInstance (*InstanceFn)();
struct Entry { int d, H; InstanceFn fn; };
const Entry kTable[] = {
#define X(D_, H_) {D_, H_, &psp_instance_##D_##_##H_},
#include "instances.def"
#undef X
};

const Entry kWideTable[] = {
#define X(D_, H_) {D_, H_, &psp_wide_instance_##D_##_##H_},
#include "wide_instances.def"
#undef X
};

// narrow family first (state panel in 256 registers, tables in LDS); the wide family covers larger d.
// PSP_FORCE_WIDE=1 prefers the wide instance where both exist (parity tests of one family against the other).
bool find_instance(int d, int H, psp::HjbInstance* out) {
    static const char* fw = getenv("PSP_FORCE_WIDE");
    const bool force_wide = fw && fw[0] == '1';
    if (!force_wide)
        for (const Entry& e : kTable)
            if (e.d == d && e.H == H) { *out = e.fn(); return true; }
    for (const Entry& e : kWideTable)
        if (e.d == d && e.H == H) { *out = e.fn(); return true; }
    if (force_wide)
        for (const Entry& e : kTable)
            if (e.d == d && e.H == H) { *out = e.fn(); return true; }
    return false;
}

typedef psp::GenInstance (*GenInstanceFn)();
struct GenEntry { int d, H; GenInstanceFn fn; };
const GenEntry kGenTable[] = {
#define X(D_, H_) {D_, H_, &psp_gen_instance_##D_##_##H_},
#include "gen_instances.def"
#undef X
};
bool find_gen_instance(int d, int H, psp::GenInstance* out) {
    for (const GenEntry& e : kGenTable)
        if (e.d == d && e.H == H) { *out = e.fn(); return true; }
    return false;
}

typedef psp::DnetInstance (*DnetInstanceFn)();
struct DnetEntry { int d, H; DnetInstanceFn fn; };
const DnetEntry kDnetTable[] = {
#define X(D_, H_) {D_, H_, &psp_dnet_instance_##D_##_##H_},
#include "dense_instances.def"
#undef X
};
bool find_dnet_instance(int d, int H, psp::DnetInstance* out) {
    for (const DnetEntry& e : kDnetTable)
        if (e.d == d && e.H == H) { *out = e.fn(); return true; }
    return false;
}

constexpr int kMaxLds = 160 * 1024;

struct Plan {
    psp::HjbInstance inst;
    int ntile16, fwd_waves, fwd_grid, bwd_waves, bwd_grid;
    bool fwd_coop = false;   // wide family, split products, d > 256: hjbc_fwd_kernel (fwd_waves = 2 or 4 TILES per 512-thread workgroup)
    bool bwd_specialised;       // hjb_bwd2_kernel (producer / consumer waves) instead of hjb_bwd_kernel
    bool fwd_split;             // hjbs_fwd_kernel (four waves per tile) instead of hjb_fwd_kernel
    bool fwd_quad;              // hjbq_fwd_kernel (four trajectories per workgroup) for the smallest K
};

int n_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;     // MI355X; also used when no device is visible (size queries on CPU)
    }
    return cus;
}

int make_plan(const psp_hjb_config* c, Plan* p) {
    if (!c) return fail(-1, "null config");
    if (c->d <= 0 || c->H <= 0 || c->K_local <= 0 || c->N <= 0) return fail(-1, "non-positive d/H/K/N");
    if (!find_instance(c->d, c->H, &p->inst)) {
        snprintf(g_err, sizeof(g_err), "no compiled HJB kernel instance for d=%d H=%d", c->d, c->H);
        return -2;
    }
    if (c->drift_kind < 0 || c->drift_kind > 3 || c->sigma_kind < 0 || c->sigma_kind > 2 ||
        c->runcost_kind < 0 || c->runcost_kind > 1 || c->term_kind < 0 || c->term_kind > 2 ||
        c->loss_kind < 0 || c->loss_kind > 3 || c->noise_mode < 0 || c->noise_mode > 1 || c->store_path < 0 ||
        c->store_path > 4)
        return fail(-1, "config enum out of range");
    if (c->store_path == 4) {                                  // the backward regenerates xi: only where the image IS xi
        if (p->inst.wide) return fail(-2, "store_path 4 (xi regenerated by the backward) exists in the narrow kernel family only");
        if (c->noise_mode != PSP_NOISE_PHILOX || !c->adaptive)
            return fail(-1, "store_path 4 needs on-device noise and the adaptive forward process (the stored image is then xi itself)");
    }
    if (c->mlp_dtype == PSP_MLP_F16X3) {
        if (!p->inst.launch_fwd_x3) return fail(-3, "the split-product mode (PSP_MLP_F16X3) is not built for this kernel family");
        if (p->inst.fwd_x3_lds_bytes(c->drift_kind, c->sigma_kind) > kMaxLds)
            return fail(-3, "split-product forward tables do not fit the 160 KiB LDS for this (d,H,drift,sigma)");
        if (c->range_flag && p->inst.fwd_lds_bytes(c->drift_kind, c->sigma_kind) > kMaxLds)
            return fail(-3, "range guard: the fp32-MFMA forward tables do not fit the 160 KiB LDS for this (d,H,drift,sigma)");
    } else if (p->inst.fwd_lds_bytes(c->drift_kind, c->sigma_kind) > kMaxLds)
        return fail(-3, "forward kernel weights do not fit the 160 KiB LDS for this (d,H,drift,sigma)");
    p->ntile16 = (c->K_local + 15) / 16;
    if ((long long)c->N * p->ntile16 >= (1LL << 31)) return fail(-1, "N * ceil(K/16) must stay below 2^31");
    const int cus = n_cus();
    // forward: one 16-trajectory tile per wave; 1..8 waves per workgroup so that small K still spreads over CUs
    // (wide family: 1..4 waves, one per SIMD)
    int fw = (p->ntile16 + cus - 1) / cus;
    if (fw < 1) fw = 1;
    if (fw > 8) fw = 8;
    if (p->inst.wide && fw > 4) fw = 4;   // wide family: one wave per SIMD
    if (p->inst.wide && c->mlp_dtype == PSP_MLP_F16X3) fw = 4;   // its split-product forward shares the table stream between FOUR waves
    // ... unless the cooperative forward serves the launch (hjbc_kernels.h: on-device noise, no running cost, no u_L2 log): TWO tiles
    // per workgroup; the range guard's fp32 twin then runs hjbw_fwd_kernel on the same grid with two waves per workgroup
    // (PSP_FWD_COOP=0 keeps the tile-per-wave kernel: A/B and parity tests of one against the other)
    {
        const char* fc = getenv("PSP_FWD_COOP");
        // four tiles per workgroup while that still gives every CU one (a single round at K = 16 384), else two
        const int nt = (fc && (fc[0] == '2' || fc[0] == '4')) ? fc[0] - '0' : (p->ntile16 >= 4 * cus ? 4 : 2);
        // measured (same box, K = 32 768): d = 200 tile-per-wave 3.53, two tiles 3.81, four tiles 3.27 ms; d = 256: 5.47 / 4.59 / 3.98; d = 500
        // (K = 16 384): 16.9 / 12.5 / 11.9.  The KERNEL must not depend on the launch size -- a K-chunked run reproduces the resident
        // run's D bit for bit (tests/test_gpu_full_size.py), and two and four tiles sum in the same order while the tile-per-wave
        // kernel does not -- so the cooperative kernel serves every size where it exists (8 % slower at d = 200 below K = 16 384)
        const bool coop_default = true;
        p->fwd_coop = p->inst.wide && c->mlp_dtype == PSP_MLP_F16X3 && p->inst.launch_fwd_coop && p->inst.coop_lds_bytes(nt) <= kMaxLds &&
                      c->noise_mode == PSP_NOISE_PHILOX && c->runcost_kind == 0 && c->u_ref == nullptr && !(fc && fc[0] == '0') &&
                      (coop_default || (fc && fc[0] != '0'));
        if (p->fwd_coop) fw = nt;
    }
    p->fwd_waves = fw;
    p->fwd_grid = (p->ntile16 + fw - 1) / fw;
    // few tiles: the feature-split forward (four waves per tile, weights in registers) cuts the per-step latency ~3x.
    // It wins while one-wave-per-tile would leave SIMDs idle: up to 2 tiles per CU (PSP_FWD_VARIANT=1 / 2 / 3 force the
    // tile-per-wave, the feature-split or the quad kernel)
    const char* fv = getenv("PSP_FWD_VARIANT");       // (read per call: tests switch variants inside one process)
    p->fwd_split = !p->inst.wide && p->inst.launch_fwd_split && p->inst.split_lds_bytes() <= kMaxLds &&
                   c->mlp_dtype != PSP_MLP_BF16_FWD && c->mlp_dtype != PSP_MLP_F16X3 &&   // (these modes exist in hjb_fwd_kernel only)
                   ((fv && fv[0] == '2') || (!(fv && fv[0] == '1') && p->ntile16 <= 2 * cus));
    if (p->fwd_split) { p->fwd_waves = 8; p->fwd_grid = p->ntile16; }   // (the kernel itself fixes 4 or 8 waves per tile)
    // fewer tiles than a quarter of the CUs: four trajectories per workgroup (hjbq_kernels.h), so that K = 1024 still covers the
    // chip (PSP_FWD_VARIANT=3 forces it, 1 / 2 exclude it)
    p->fwd_quad = !p->inst.wide && p->inst.launch_fwd_quad && p->inst.quad_lds_bytes() <= kMaxLds &&
                  c->mlp_dtype != PSP_MLP_BF16_FWD && c->mlp_dtype != PSP_MLP_F16X3 &&
                  ((fv && fv[0] == '3') || (!fv && 4 * p->ntile16 <= cus));
    if (p->fwd_quad) { p->fwd_split = false; p->fwd_waves = 8; p->fwd_grid = 4 * p->ntile16; }
    // backward: persistent over rounds of 4 sample blocks; 4-wave workgroups, two per CU
    // (<= 256 VGPRs and <= 80 KiB LDS each), fewer when there is little work
    const long long nblk = (long long)c->N * p->ntile16;
    const long long nround = (nblk + 3) / 4;
    // adaptive runs use the role-specialised kernel (one 8-wave workgroup per CU) when its double-buffered
    // exchange area fits the LDS; PSP_BWD_VARIANT=1 forces the two-workgroups-per-CU kernel (A/B timing)
    static const char* force = getenv("PSP_BWD_VARIANT");
    p->bwd_specialised = p->inst.bwd2_lds_bytes() <= kMaxLds && !(force && force[0] == '1' && p->inst.launch_bwd);
    if (p->inst.wide) p->bwd_specialised = true;          // launch_bwd2 = hjbw_bwd_kernel (4 waves)
    if (!p->bwd_specialised && !p->inst.launch_bwd)
        return fail(-3, "the role-specialised backward does not fit the LDS for this (d,H) and the library was built without "
                        "the legacy backward kernels (-DPSP_LEGACY_BWD)");
    if (!p->bwd_specialised && p->inst.bwd_lds_bytes(c->adaptive) > kMaxLds)
        return fail(-3, "backward kernel staging does not fit the 160 KiB LDS for this (d,H)");
    if (c->store_path == 4 && !p->bwd_specialised)
        return fail(-2, "store_path 4 needs the role-specialised backward kernels (hjb_bwd2_kernel / hjb_bwd3_kernel)");
    p->bwd_waves = (p->bwd_specialised && !p->inst.wide) ? 8 : 4;
    long long g = nround;
    const long long gmax = (p->bwd_specialised && !(p->inst.wide && c->d <= 256 && !p->inst.bwd2_one_per_cu)) ? cus : 2LL * cus;   // hjbw_bwd_kernel, d <= 256: two per CU
    if (g > gmax) g = gmax;
    if (g < 1) g = 1;
    p->bwd_grid = (int)g;
    return 0;
}

void fill_args(const psp_hjb_config* c, const Plan& p, psp::HjbArgs* a) {
    memset(a, 0, sizeof(*a));
    a->drift = c->drift; a->sigma = c->sigma; a->runcost = c->runcost; a->term = c->term;
    a->k_offset = c->k_offset; a->K_global = c->K_global; a->K_local = c->K_local; a->N = c->N;
    a->ntile16 = p.ntile16; a->dt = c->dt; a->sqdt = c->sqrt_dt; a->sigma_scale = c->sigma_scale;
    a->drift_kind = c->drift_kind; a->sigma_kind = c->sigma_kind; a->runcost_kind = c->runcost_kind;
    a->term_kind = c->term_kind; a->adaptive = c->adaptive; a->loss_kind = c->loss_kind;
    a->noise_mode = c->noise_mode; a->store_path = c->store_path;
    a->uref = c->u_ref; a->ul2 = c->u_l2_out;
    a->iter_dev = c->iter_dev;
    // diagnostic stamp buffer: [forward: fwd_grid x 8 waves x 8][backward: bwd_grid x 4 waves x 8]
    a->dbg = (g_dbg && g_dbg_n >= ((long long)p.fwd_grid * 8 + (long long)p.bwd_grid * 8) * 8) ? g_dbg : nullptr;
}

int check_ptrs(const psp_hjb_config* c) {
    if ((c->drift_kind != PSP_DRIFT_ZERO) && !c->drift) return fail(-1, "drift parameters missing");
    if (c->sigma_kind == PSP_SIGMA_DENSE && !c->sigma) return fail(-1, "sigma matrix missing");
    if (c->runcost_kind == PSP_RUNCOST_DIAG_QUAD && !c->runcost) return fail(-1, "running-cost vector missing");
    if (!c->term) return fail(-1, "terminal-cost vector missing");
    if (c->u_ref && !c->u_l2_out) return fail(-1, "u_ref set but u_l2_out is null");
    return 0;
}

struct GenPlan {
    psp::GenInstance inst;
    int ntile16, fwd_waves, fwd_grid, bwd_grid;
    bool bwd_specialised;       // gen_bwd2_kernel (producer / consumer waves) instead of gen_bwd_kernel
};

int n_cus();

int make_gen_plan(const psp_gen_config* c, GenPlan* p) {
    if (!c) return fail(-1, "null config");
    if (c->d <= 0 || c->H <= 0 || c->K_local <= 0 || c->N <= 0) return fail(-1, "non-positive d/H/K/N");
    if (!find_gen_instance(c->d, c->H, &p->inst)) {
        snprintf(g_err, sizeof(g_err), "no compiled GeneralSolver kernel instance for d=%d H=%d", c->d, c->H);
        return -2;
    }
    if ((c->drift_kind != PSP_DRIFT_ZERO && c->drift_kind != PSP_DRIFT_DOUBLE_WELL && c->drift_kind != PSP_DRIFT_DIAG) || c->h_kind < 0 ||
        c->h_kind > PSP_GH_EXPBALL_SIN || c->noise_mode < 0 || c->noise_mode > 1 || c->domain_kind < 0 ||
        c->domain_kind > PSP_DOM_ANNULUS)
        return fail(-1, "config enum out of range");
    if (c->domain_kind == PSP_DOM_SPHERE && !(c->dom_a > 0.f)) return fail(-1, "sphere radius must be positive");
    if (c->domain_kind == PSP_DOM_BOX && !(c->dom_a < c->dom_b)) return fail(-1, "box bounds must satisfy X_l < X_r");
    if (c->domain_kind == PSP_DOM_ANNULUS && !(c->dom_a >= 0.f && c->dom_a < c->dom_b)) return fail(-1, "annulus radii must satisfy 0 <= r_1 < r_2");
    if (c->drift_kind != PSP_DRIFT_ZERO && !c->drift) return fail(-1, "drift vector missing (double-well kappa / diagonal of A)");
    if ((c->v_steps_out == nullptr) != (c->y_steps_out == nullptr)) return fail(-1, "v_steps_out and y_steps_out go together");
    if (p->inst.fwd_lds_bytes() > kMaxLds)
        return fail(-3, "GeneralSolver kernel tables do not fit the 160 KiB LDS for this (d,H)");
    if (c->mlp_dtype == PSP_MLP_F16X3 && (!p->inst.launch_fwd_x3 || p->inst.fwd_x3_lds_bytes() > kMaxLds))
        return fail(-3, "split-product forward tables do not fit the 160 KiB LDS for this (d,H)");
    p->ntile16 = (c->K_local + 15) / 16;
    const int cus = n_cus();
    int fw = (p->ntile16 + cus - 1) / cus;
    if (fw < 1) fw = 1;
    if (fw > 8) fw = 8;
    p->fwd_waves = fw;
    p->fwd_grid = (p->ntile16 + fw - 1) / fw;
    const long long nround = ((long long)(c->N + 1) * p->ntile16 + 3) / 4;
    static const char* force = getenv("PSP_BWD_VARIANT");
    p->bwd_specialised = p->inst.bwd2_lds_bytes() <= kMaxLds && !(force && force[0] == '1' && p->inst.launch_bwd);
    if (!p->bwd_specialised && !p->inst.launch_bwd)
        return fail(-3, "the role-specialised backward does not fit the LDS for this (d,H) and the library was built without "
                        "the legacy backward kernels (-DPSP_LEGACY_BWD)");
    long long g = nround;
    const long long gmax = p->bwd_specialised ? cus : 2LL * cus;
    if (g > gmax) g = gmax;
    if (g < 1) g = 1;
    p->bwd_grid = (int)g;
    return 0;
}

void fill_gen_args(const psp_gen_config* c, const GenPlan& p, psp::GenArgs* a) {
    memset(a, 0, sizeof(*a));
    a->drift = c->drift; a->k_offset = c->k_offset; a->K_local = c->K_local; a->N = c->N; a->ntile16 = p.ntile16;
    a->dt = c->dt; a->sqdt = c->sqrt_dt; a->T = c->T; a->sigma_scale = c->sigma_scale;
    a->drift_kind = c->drift_kind; a->h_kind = c->h_kind; a->adaptive = c->adaptive;
    a->noise_mode = c->noise_mode; a->store_path = c->store_path;
    a->domain_kind = c->domain_kind; a->dom_a = c->dom_a; a->dom_b = c->dom_b;
    a->d_real = (c->d_real > 0 && c->d_real < c->d) ? c->d_real : c->d;
    for (int i = 0; i < 4; ++i) a->h_par[i] = c->h_par[i];
    a->Vsteps = c->v_steps_out; a->Ysteps = c->y_steps_out; a->per_sample = c->per_sample_weights ? 1 : 0;
    a->path16 = (c->mlp_dtype == PSP_MLP_BF16) ? 1 : 0;       // both kernels on bf16 MFMA: bf16-pair path block
}

// ---- small kernels -------------------------------------------------------------------
__global__ void reduce_partials_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    // single workgroup, fixed summation order: thread t sums entries t, t+256, ... then a tree
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = s0[0]; out[1] = s1[0]; }
}

// partial sums -> (sum D, sum D^2) and the loss value (single rank: local sums are global)
__global__ void reduce_partials_loss_kernel(const double* __restrict__ part, int n, double* __restrict__ out, int loss_kind,
                                            double invK, float* __restrict__ loss_log, const uint32_t* __restrict__ index_dev) {
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s0[0]; out[1] = s1[0];
        if (loss_log) {
            const double m = s0[0] * invK;
            double loss = s1[0] * invK - m * m;                       // log-variance: mean(D^2) - mean(D)^2 (solver.py:167-168)
            if (loss_kind == PSP_LOSS_MOMENT) loss = s1[0] * invK;    // :165-166
            if (loss_kind == PSP_LOSS_REL_ENTROPY) loss = -m;         // D = -(Zsum + g) (:179-180)
            loss_log[index_dev ? *index_dev : 0u] = (float)loss;
        }
    }
}

// Range guard of the split-product mode (include/psp.h: range_flag).  flag[0] = 1 iff any of the n doubles is non-finite
// (a per-workgroup partial of (sum D, sum D^2): an f16x3 operand beyond 65504 makes D_k NaN), flag[1] counts the 1s.
__global__ void range_flag_partials_kernel(const double* __restrict__ part, int n, int* __restrict__ flag) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = part[i];
        mine |= !(fabs(v) <= 1.7976931348623157e308);            // NaN and +-inf
    }
    if (mine) bad = 1;                                           // benign race: every writer stores 1
    __syncthreads();
    if (threadIdx.x == 0) { flag[0] = bad; flag[1] += bad; }
}
__global__ void snapshot_u64_kernel(const unsigned long long* src, unsigned long long* dst) { *dst = *src; }
// the same from two fp32 arrays of n entries (GeneralSolver: V(X_N) and Y_N per trajectory)
__global__ void range_flag_arrays_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, int* __restrict__ flag,
                                         unsigned long long* counter, const unsigned long long* counter_before) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        mine |= !(fabsf(a[i]) <= 3.402823466e38f);
        mine |= !(fabsf(b[i]) <= 3.402823466e38f);
    }
    if (mine) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        flag[0] = bad; flag[1] += bad;
        if (bad && counter) *counter = *counter_before;          // the predicated fp32 kernel counts the active steps afresh
    }
}

__global__ void iter_advance_kernel(psp_iter_state* st, double b1, double b2) {
    st->iter += 1u; st->step += 1u; st->beta1_pow *= b1; st->beta2_pow *= b2;
}

// Adam with the bias corrections of the step held in a device psp_iter_state (same arithmetic as adam_kernel: the host
// path forms step_size = lr / (1 - b1^step) and sqrt(1 - b2^step) in double and rounds to fp32, so does every thread here)
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long long n, const psp_iter_state* __restrict__ st, float lr, float b1,
                                float b2, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float step_size = (float)((double)lr / (1.0 - st->beta1_pow));
    const float bc2_sqrt = (float)sqrt(1.0 - st->beta2_pow);
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// Launch-bound sizes (graph replay): partial gradients -> gradient -> Adam -> iteration state, ONE launch instead of three.
// Per parameter: fixed-order sum of the workgroup partials (reduce_partials_8way, as reduce_grad_kernel), the Adam update of
// adam_dev_kernel with the bias corrections of the CURRENT state, then the last workgroup to finish advances the state
// (every other workgroup has read it before its ticket).  grad_out still receives the gradient (diagnostics, tests).
// Fixed-order sum over the workgroup partials of parameter i, spread over EIGHT threads (slice u sums the workgroups
// w = u (mod 8) in increasing order, the slices meet in LDS as ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7))): the same grouping -- and
// the same bits -- as one thread with eight accumulators, but nwg / 8 dependent loads deep instead of nwg (256 partials of a
// 17 k-parameter net took 12.7 us, a fifteenth of the K = 1024 iteration).  blockDim = (32 parameters) x (8 slices).
__device__ __forceinline__ float reduce_partials_8way(const float* __restrict__ part, int nwg, int P, int i, float* sh) {
    const int tx = threadIdx.x & 31, u = threadIdx.x >> 5;
    float s = 0.f;
    if (i < P) {
        int w = u;
        for (; w + 24 < nwg; w += 32) {                        // four loads in flight
            const float a0 = part[(size_t)w * P + i], a1 = part[(size_t)(w + 8) * P + i];
            const float a2 = part[(size_t)(w + 16) * P + i], a3 = part[(size_t)(w + 24) * P + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; w < nwg; w += 8) s += part[(size_t)w * P + i];
    }
    sh[u * 32 + tx] = s;
    __syncthreads();
    return ((sh[tx] + sh[32 + tx]) + (sh[64 + tx] + sh[96 + tx])) + ((sh[128 + tx] + sh[160 + tx]) + (sh[192 + tx] + sh[224 + tx]));
}

__global__ void reduce_grad_adam_advance_kernel(const float* __restrict__ part, int nwg, int P, float* __restrict__ grad_out,
                                                float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                psp_iter_state* st, unsigned int* ticket, float lr, float b1, float b2, float eps) {
    __shared__ float sh[256];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31);
    const double b1p = st->beta1_pow, b2p = st->beta2_pow;
    const float gi = reduce_partials_8way(part, nwg, P, i, sh);
    if (i < P && threadIdx.x < 32) {
        grad_out[i] = gi;
        const float step_size = (float)((double)lr / (1.0 - b1p));
        const float bc2_sqrt = (float)sqrt(1.0 - b2p);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
    __syncthreads();                                           // every thread of this workgroup has read the state
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int t = atomicAdd(ticket, 1u);
        if (t == gridDim.x - 1) {                              // last workgroup: all others read the state before their ticket
            st->iter += 1u; st->step += 1u; st->beta1_pow = b1p * (double)b1; st->beta2_pow = b2p * (double)b2;
            *ticket = 0u;
        }
    }
}

__global__ void reduce_grad_kernel(const float* __restrict__ part, int nwg, int P, float* __restrict__ out) {
    __shared__ float sh[256];
    const int p = blockIdx.x * 32 + (threadIdx.x & 31);
    const float g = reduce_partials_8way(part, nwg, P, p, sh);
    if (p < P && threadIdx.x < 32) out[p] = g;
}

// torch.optim.Adam single-tensor semantics (torch/optim/adam.py, _single_tensor_adam)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            float step_size, float bc2_sqrt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);              // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;             // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);                         // param.addcdiv_(exp_avg, denom, -step_size)
}

__global__ void philox_fill_kernel(float* __restrict__ out, int N, int K, int d, long long k_offset,
                                   uint32_t seed_lo, uint32_t seed_hi, uint32_t iter) {
    // one thread per Philox call: (step n, trajectory k, call idx = 4b+q) -> features 16b+4r+q
    const int ncall = ((d + 15) / 16) * 4;
    const long long total = (long long)N * K * ncall;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int idx = (int)(t % ncall);
    const long long r0 = t / ncall;
    const int k = (int)(r0 % K), n = (int)(r0 / K);
    const psp::f32x4 z = psp::philox_block((uint32_t)(k_offset + k), (uint32_t)n, (uint32_t)idx, iter, seed_lo, seed_hi);
    const int b = idx >> 2, q = idx & 3;
    for (int r = 0; r < 4; ++r) {
        const int f = 16 * b + 4 * r + q;
        if (f < d) out[((size_t)(n + 1) * K + k) * d + f] = z[r];
    }
}

__global__ void zero_kernel(float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.f;
}

// u = -Z(t, x): plain per-output kernel (evaluation only, not on the training hot path)
__global__ void control_eval_kernel(int d, int H, const float* __restrict__ P, const float* __restrict__ X,
                                    int K, float t, float* __restrict__ out) {
    extern __shared__ float sm[];                 // h1[H], h2[H] for one trajectory per workgroup
    float* h1 = sm; float* h2 = sm + H;
    const int k = blockIdx.x;
    const int oW1 = 0, ob1 = H * (d + 1), oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + d * H;
    for (int o = threadIdx.x; o < H; o += blockDim.x) {
        float s = P[ob1 + o];
        s = fmaf(P[oW1 + o * (d + 1)], t, s);
        for (int i = 0; i < d; ++i) s = fmaf(P[oW1 + o * (d + 1) + 1 + i], X[(size_t)k * d + i], s);
        h1[o] = tanhf(s);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < H; o += blockDim.x) {
        float s = P[ob2 + o];
        for (int i = 0; i < H; ++i) s = fmaf(P[oW2 + o * H + i], h1[i], s);
        h2[o] = tanhf(s);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d; o += blockDim.x) {
        float s = P[ob3 + o];
        for (int i = 0; i < H; ++i) s = fmaf(P[oW3 + o * H + i], h2[i], s);
        out[(size_t)k * d + o] = -s;
    }
}

}  // namespace

extern "C" {

int psp_version(void) { return PSP_VERSION; }
int psp_abi_struct_sizes(int32_t out[6]) {
    static_assert(sizeof(psp_iter_state) == 24, "psp_iter_state layout");
    if (!out) return fail(-1, "null output");
    out[0] = (int32_t)sizeof(psp_hjb_config); out[1] = (int32_t)sizeof(psp_hjb_sizes);
    out[2] = (int32_t)sizeof(psp_gen_config); out[3] = (int32_t)sizeof(psp_gen_sizes);
    out[4] = (int32_t)sizeof(psp_dnet_config); out[5] = (int32_t)sizeof(psp_dnet_sizes);
    return 0;
}
int psp_abi_struct_sizes2(int32_t out[2]) {
    if (!out) return fail(-1, "null output");
    out[0] = (int32_t)sizeof(psp_genl_config); out[1] = (int32_t)sizeof(psp_genl_sizes);
    return 0;
}
const char* psp_last_error(void) { return g_err; }

// ---- DenseNet control (hjbd_kernels.h): time_approx='outer' and DenseNet(d+1 -> d) controls ---------------------
namespace {
struct DnetPlan { psp::DnetInstance inst; int ntile16, grid; long long table_floats, n_params; int slices, bwd_grid, bwd_ok; };
int make_dnet_plan(const psp_dnet_config* c, DnetPlan* p) {
    if (!c) return fail(-1, "null config");
    const psp_hjb_config& b = c->base;
    if (b.d <= 0 || b.H <= 0 || b.K_local <= 0 || b.N <= 0) return fail(-1, "non-positive d/H/K/N");
    if (!find_dnet_instance(b.d, b.H, &p->inst)) {
        snprintf(g_err, sizeof(g_err), "no compiled DenseNet-control kernel instance for d=%d H=%d", b.d, b.H);
        return -2;
    }
    if (c->d_real <= 0 || c->d_real > b.d || c->H_real <= 0 || c->H_real > b.H)
        return fail(-1, "d_real / H_real must lie in [1, d] / [1, H] of the instance");
    if (b.drift_kind < 0 || b.drift_kind > 3 || b.sigma_kind < 0 || b.sigma_kind > 2 || b.runcost_kind < 0 ||
        b.runcost_kind > 1 || b.term_kind < 0 || b.term_kind > 2 || b.noise_mode < 0 || b.noise_mode > 1 ||
        b.store_path < 0 || b.store_path > 3 || b.loss_kind < 0 || b.loss_kind > 3)
        return fail(-1, "config enum out of range");
    if (p->inst.lds_bytes > kMaxLds) return fail(-3, "DenseNet-control kernel images do not fit the 160 KiB LDS");
    if (b.mlp_dtype == PSP_MLP_F16X3 && (!p->inst.launch_fwd_x3 || p->inst.lds_bytes_x3 > kMaxLds))
        return fail(-3, "split-product forward images do not fit the 160 KiB LDS for this (d,H)");
    p->ntile16 = (b.K_local + 15) / 16;
    p->grid = (p->ntile16 + 3) / 4;
    p->table_floats = (long long)p->inst.shared_floats + (long long)(c->per_step ? b.N : 1) * p->inst.set_floats +
                      (long long)b.N * p->inst.vec_floats;
    const long long di = c->d_real + (c->time_input ? 1 : 0), h = c->H_real, d = c->d_real;
    p->n_params = di * h + h + (di + h) * h + h + (di + 2 * h) * d + d;
    // backward work items = (step, slice): about two waves of workgroups over the chip, at least one round per item
    const int cus = n_cus();
    int S = (2 * cus + b.N / 2) / b.N;
    const int smax = (p->ntile16 + 3) / 4;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    p->slices = S;
    const long long items = (long long)b.N * S;
    p->bwd_grid = (int)(items < cus ? items : cus);
    p->bwd_ok = (p->inst.bwd_lds_bytes <= kMaxLds && p->inst.bwd_passes > 0) ? 1 : 0;   // else: too many accumulator tiles per wave
    return 0;
}
}  // namespace

extern "C" int psp_dnet_instance_count(void) { return (int)(sizeof(kDnetTable) / sizeof(kDnetTable[0])); }
extern "C" int psp_dnet_instance_get(int32_t i, int32_t* d, int32_t* H) {
    if (i < 0 || i >= psp_dnet_instance_count() || !d || !H) return fail(-1, "instance index out of range");
    *d = kDnetTable[i].d; *H = kDnetTable[i].H;
    return 0;
}

extern "C" int psp_dnet_query(const psp_dnet_config* cfg, psp_dnet_sizes* out) {
    DnetPlan p;
    int rc = make_dnet_plan(cfg, &p);
    if (rc) return rc;
    if (!out) return fail(-1, "null output");
    out->table_bytes = p.table_floats * 4;
    out->fwd_partial_bytes = (int64_t)p.grid * 2 * 8;
    out->n_params_per_set = p.n_params;
    out->fwd_workgroups = p.grid;
    out->reserved = 0;
    out->image_bytes = (int64_t)cfg->base.N * p.ntile16 * p.inst.image_block_floats * 4;
    out->partial_bytes = (int64_t)cfg->base.N * p.slices * p.inst.partial_floats * 4;
    out->bwd_supported = p.bwd_ok; out->slices = p.slices; out->padded_params = p.inst.partial_floats;
    out->bwd_workgroups = p.bwd_grid;
    return 0;
}

extern "C" int psp_dnet_terminal_reduce(const psp_dnet_config* cfg, const double* fwd_partial, double* sums_out, void* stream) {
    DnetPlan p;
    int rc = make_dnet_plan(cfg, &p);
    if (rc) return rc;
    if (!fwd_partial || !sums_out) return fail(-1, "null buffer passed to psp_dnet_terminal_reduce");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fwd_partial, p.grid, sums_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_partials_kernel launch");
    return 0;
}

extern "C" int psp_dnet_rollout_bwd(const psp_dnet_config* cfg, const float* params, const float* images, const float* w,
                                    float* partial, void* stream) {
    DnetPlan p;
    int rc = make_dnet_plan(cfg, &p);
    if (rc) return rc;
    if (!p.bwd_ok) return fail(-3, "the hand-written DenseNet-control backward does not cover this instance (accumulator tiles)");
    if (!params || !images || !w || !partial) return fail(-1, "null buffer passed to psp_dnet_rollout_bwd");
    const psp_hjb_config* b = &cfg->base;
    psp::DnetArgs a;
    memset(&a, 0, sizeof(a));
    a.h.params = params; a.h.K_local = b->K_local; a.h.N = b->N; a.h.ntile16 = p.ntile16; a.h.sqdt = b->sqrt_dt; a.h.dt = b->dt;
    a.pimg = const_cast<float*>(images); a.wts = w; a.partial = partial; a.slices = p.slices;
    a.d_real = cfg->d_real; a.h_real = cfg->H_real; a.time_input = cfg->time_input ? 1 : 0; a.per_step = cfg->per_step ? 1 : 0;
    // split-product outer products where the stored image is the Brownian increment itself (detached adaptive run: the power-of-two
    // scale of the weight-carrying tiles then needs nothing but the weights); guarded by the fp32 kernel like the other split kernels
    const bool x3 = b->mlp_dtype == PSP_MLP_F16X3 && b->adaptive && b->store_path == 1 && p.inst.launch_bwd_x3;
    const bool guard = x3 && b->range_flag != nullptr;
    if (guard) { a.h.cond = b->range_flag; a.h.cond_want = 0; }
    hipError_t e = x3 ? p.inst.launch_bwd_x3(a, p.bwd_grid, (hipStream_t)stream) : p.inst.launch_bwd(a, p.bwd_grid, (hipStream_t)stream);
    if (e == hipSuccess && guard) {
        a.h.cond_want = 1;
        e = p.inst.launch_bwd(a, p.bwd_grid, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "hjbd_bwd_kernel launch");
    return 0;
}

extern "C" int psp_dnet_adjoint_sweep(const psp_dnet_config* cfg, const float* params, float* images, const float* XN,
                                      const float* mu, const float* nu, const float* wT, float* tables, void* stream) {
    DnetPlan p;
    int rc = make_dnet_plan(cfg, &p);
    if (rc) return rc;
    const psp_hjb_config* b = &cfg->base;
    if ((rc = check_ptrs(b))) return rc;
    if (!params || !images || !XN || !mu || !tables) return fail(-1, "null buffer passed to psp_dnet_adjoint_sweep");
    if (b->store_path != 2 && b->store_path != 3) return fail(-1, "psp_dnet_adjoint_sweep needs store_path 2 or 3 (the forward's image kind)");
    if (b->store_path == 3 && !nu) return fail(-1, "store_path 3 (relative entropy) needs nu");
    psp::DnetArgs a;
    memset(&a, 0, sizeof(a));
    psp::HjbArgs& h = a.h;
    h.drift = b->drift; h.sigma = b->sigma; h.runcost = b->runcost; h.term = b->term;
    h.K_local = b->K_local; h.N = b->N; h.ntile16 = p.ntile16; h.dt = b->dt; h.sqdt = b->sqrt_dt; h.sigma_scale = b->sigma_scale;
    h.drift_kind = b->drift_kind; h.sigma_kind = b->sigma_kind; h.runcost_kind = b->runcost_kind; h.term_kind = b->term_kind;
    h.adaptive = b->adaptive; h.store_path = b->store_path;
    h.params = params; h.XN = const_cast<float*>(XN); h.adj_mu = mu; h.adj_nu = nu; h.adj_wT = wT;
    a.tbl = tables; a.pimg = images;
    a.d_real = cfg->d_real; a.h_real = cfg->H_real; a.time_input = cfg->time_input ? 1 : 0; a.per_step = cfg->per_step ? 1 : 0;
    const bool x3 = b->mlp_dtype == PSP_MLP_F16X3 && p.inst.launch_adj_x3 && p.inst.lds_bytes_x3 <= kMaxLds;
    const bool guard = x3 && b->range_flag != nullptr;       // range guard (psp_hjb_config.range_flag): both sweeps, predicated
    if (guard) { h.cond = b->range_flag; h.cond_want = 0; }
    hipError_t e = x3 ? p.inst.launch_adj_x3(a, p.grid, (hipStream_t)stream) : p.inst.launch_adj(a, p.grid, (hipStream_t)stream);
    if (e == hipSuccess && guard) {
        h.cond_want = 1;
        e = p.inst.launch_adj(a, p.grid, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "hjbd_adj_kernel launch");
    return 0;
}

extern "C" int psp_dnet_rollout_fwd(const psp_dnet_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                                    const float* y0, const float* xi, uint64_t seed, uint32_t iter, const float* tfeat,
                                    float* px, float* pxi, float* D_out, float* Fint_out, float* XN_out, float* Y_out,
                                    double* fwd_partial, float* tables, void* stream) {
    DnetPlan p;
    int rc = make_dnet_plan(cfg, &p);
    if (rc) return rc;
    const psp_hjb_config* b = &cfg->base;
    if ((rc = check_ptrs(b))) return rc;
    if (!params || !x0 || !D_out || !fwd_partial || !tables) return fail(-1, "null buffer passed to psp_dnet_rollout_fwd");
    if (x0_stride != 0 && x0_stride != b->d) return fail(-1, "x0_stride must be 0 or d");
    if (b->noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    if (b->store_path && !cfg->images_out && (!px || !pxi)) return fail(-1, "store_path set but the X / xi stores are null");
    if (b->store_path >= 2 && !cfg->images_out)
        return fail(-1, "store_path 2 / 3 (adjoint sweep) need the register images (psp_dnet_config.images_out)");
    psp::DnetArgs a;
    memset(&a, 0, sizeof(a));
    psp::HjbArgs& h = a.h;
    h.drift = b->drift; h.sigma = b->sigma; h.runcost = b->runcost; h.term = b->term;
    h.k_offset = b->k_offset; h.K_global = b->K_global; h.K_local = b->K_local; h.N = b->N;
    h.ntile16 = p.ntile16; h.dt = b->dt; h.sqdt = b->sqrt_dt; h.sigma_scale = b->sigma_scale;
    h.drift_kind = b->drift_kind; h.sigma_kind = b->sigma_kind; h.runcost_kind = b->runcost_kind;
    h.term_kind = b->term_kind; h.adaptive = b->adaptive; h.loss_kind = b->loss_kind;
    h.noise_mode = b->noise_mode; h.store_path = b->store_path;
    h.params = params; h.x0 = x0; h.x0_stride = x0_stride; h.y0 = y0; h.xi = xi; h.tfeat = tfeat;
    h.D = D_out; h.Fint = Fint_out; h.XN = XN_out; h.Yout = Y_out; h.fwd_partial = fwd_partial;
    h.seed_lo = (uint32_t)seed; h.seed_hi = (uint32_t)(seed >> 32); h.iter = iter;
    a.tbl = tables; a.px = px; a.pxi = pxi;
    if ((cfg->r1_out == nullptr) != (cfg->r2_out == nullptr)) return fail(-1, "r1_out and r2_out go together");
    a.pr1 = cfg->r1_out; a.pr2 = cfg->r2_out; a.pimg = cfg->images_out;
    a.d_real = cfg->d_real; a.h_real = cfg->H_real; a.time_input = cfg->time_input ? 1 : 0; a.per_step = cfg->per_step ? 1 : 0;
    const bool x3 = b->mlp_dtype == PSP_MLP_F16X3 && p.inst.launch_fwd_x3 && p.inst.lds_bytes_x3 <= kMaxLds;
    if (b->mlp_dtype == PSP_MLP_F16X3 && !x3) return fail(-3, "split-product forward images do not fit the 160 KiB LDS for this (d,H)");
    hipError_t e = x3 ? p.inst.launch_fwd_x3(a, p.grid, (hipStream_t)stream) : p.inst.launch_fwd(a, p.grid, (hipStream_t)stream);
    if (e == hipSuccess && x3 && b->range_flag) {
        // range guard (psp_hjb_config.range_flag): non-finite partials -> flag -> the fp32-MFMA rollout, predicated
        hipLaunchKernelGGL(range_flag_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fwd_partial, 2 * p.grid,
                           b->range_flag);
        if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "range_flag_partials_kernel launch");
        h.cond = b->range_flag; h.cond_want = 1;
        e = p.inst.launch_fwd(a, p.grid, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "hjbd_fwd_kernel launch");
    return 0;
}

int psp_debug_set_stamp_buffer(unsigned long long* buf, int64_t n_entries) {
    g_dbg = buf;
    g_dbg_n = buf ? n_entries : 0;
#ifdef PSP_STAMPS
    return 1;
#else
    return 0;
#endif
}

static int64_t grad_rows_bytes(const Plan& p) {       // partial-gradient rows, padded to 256 B
    const int64_t b = (int64_t)p.bwd_grid * p.inst.n_params * 4;
    return (b + 255) / 256 * 256;
}

int psp_hjb_supported(int32_t d, int32_t H) {
    psp::HjbInstance inst;
    return find_instance(d, H, &inst) ? 1 : 0;
}

int psp_hjb_family(int32_t d, int32_t H) {
    psp::HjbInstance inst;
    if (!find_instance(d, H, &inst)) return 0;
    return inst.wide ? 2 : 1;
}

int psp_hjb_instance_count(void) {
    return (int)(sizeof(kTable) / sizeof(kTable[0]) + sizeof(kWideTable) / sizeof(kWideTable[0]));
}

int psp_hjb_instance_get(int32_t i, int32_t* d, int32_t* H, int32_t* family) {
    const int nn = (int)(sizeof(kTable) / sizeof(kTable[0])), nw = (int)(sizeof(kWideTable) / sizeof(kWideTable[0]));
    if (i < 0 || i >= nn + nw || !d || !H || !family) return fail(-1, "instance index out of range");
    const Entry& e = i < nn ? kTable[i] : kWideTable[i - nn];
    *d = e.d; *H = e.H; *family = i < nn ? 1 : 2;
    return 0;
}

int psp_hjb_query(const psp_hjb_config* cfg, psp_hjb_sizes* out) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if (!out) return fail(-1, "null output");
    memset(out, 0, sizeof(*out));
    out->n_params = p.inst.n_params;
    out->fwd_workgroups = p.fwd_grid;
    out->bwd_workgroups = p.bwd_grid;
    out->fwd_coop_tiles = p.fwd_coop ? p.fwd_waves : 0;
    out->path_bytes = cfg->store_path
        ? (int64_t)cfg->N * p.ntile16 * (int64_t)p.inst.path_floats_per_tile_step * 4 : 0;
    // the wide family keeps its A-operand tables behind the partial sums in the same caller-owned scratch
    out->fwd_partial_bytes = (int64_t)p.fwd_grid * 2 * 8 + (int64_t)p.inst.fwd_table_floats * 4;
    out->grad_partial_bytes = grad_rows_bytes(p) + (int64_t)p.inst.bwd_table_floats * 4;
    return 0;
}

int psp_hjb_rollout_fwd(const psp_hjb_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                        const float* y0, const float* xi, uint64_t seed, uint32_t iter, float* path,
                        float* D_out, float* XN_out, float* Y_out, double* fwd_partial, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if ((rc = check_ptrs(cfg))) return rc;
    if (!params || !x0 || !D_out || !fwd_partial) return fail(-1, "null buffer passed to psp_hjb_rollout_fwd");
    if (x0_stride != 0 && x0_stride != cfg->d) return fail(-1, "x0_stride must be 0 or d");
    if (cfg->noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    if (cfg->store_path && !path) return fail(-1, "store_path set but path buffer is null");
    psp::HjbArgs a;
    fill_args(cfg, p, &a);
    a.params = params; a.x0 = x0; a.x0_stride = x0_stride; a.y0 = y0; a.xi = xi; a.path = path;
    a.D = D_out; a.XN = XN_out; a.Yout = Y_out; a.fwd_partial = fwd_partial;
    a.tables = reinterpret_cast<float*>(fwd_partial + 2 * (size_t)p.fwd_grid);
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.iter = iter;
    hipError_t e;
    if (cfg->mlp_dtype == PSP_MLP_BF16_FWD) {
        if (!p.inst.launch_fwd_bf16) return fail(-3, "the bf16 control-net mode exists for the narrow kernel family only");
        e = p.inst.launch_fwd_bf16(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);      // make_plan kept the tile-per-wave forward
    } else if (cfg->mlp_dtype == PSP_MLP_F16X3) {
        e = p.fwd_coop ? p.inst.launch_fwd_coop(a, p.fwd_grid, p.fwd_waves, (hipStream_t)stream)
                       : p.inst.launch_fwd_x3(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
        if (e == hipSuccess && cfg->range_flag) {
            // range guard: non-finite partials -> flag -> the fp32-MFMA forward of the same launch, predicated on the flag
            // (same grid, same partials / D / path-store layout; it overwrites what the split kernel left)
            hipLaunchKernelGGL(range_flag_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fwd_partial,
                               2 * p.fwd_grid, cfg->range_flag);
            if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "range_flag_partials_kernel launch");
            a.cond = cfg->range_flag; a.cond_want = 1;
            e = p.inst.launch_fwd(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
        }
    } else if (cfg->mlp_dtype != PSP_MLP_FP32) {
        return fail(-1, "mlp_dtype out of range for the HJB rollout (fp32, bf16_fwd or f16x3)");
    } else {
        e = p.fwd_quad ? p.inst.launch_fwd_quad(a, p.fwd_grid, (hipStream_t)stream)
            : p.fwd_split ? p.inst.launch_fwd_split(a, p.fwd_grid, (hipStream_t)stream)
                          : p.inst.launch_fwd(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "hjb_fwd_kernel launch");
    return 0;
}

int psp_hjb_rollout_eval(const psp_hjb_config* cfg, const float* params, const float* x0, int32_t x0_stride,
                         const float* xi, uint64_t seed, uint32_t iter, const float* tfeat, float* D_out,
                         float* Fint_out, float* XN_out, double* fwd_partial, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if ((rc = check_ptrs(cfg))) return rc;
    if (!params || !x0 || !D_out || !fwd_partial) return fail(-1, "null buffer passed to psp_hjb_rollout_eval");
    if (x0_stride != 0 && x0_stride != cfg->d) return fail(-1, "x0_stride must be 0 or d");
    if (cfg->noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    psp::HjbArgs a;
    fill_args(cfg, p, &a);
    a.store_path = 0;
    a.uref = nullptr; a.ul2 = nullptr;      // the evaluation grid is not the training grid (utilities.py:296-299)
    a.params = params; a.x0 = x0; a.x0_stride = x0_stride; a.xi = xi; a.tfeat = tfeat;
    a.D = D_out; a.Fint = Fint_out; a.XN = XN_out; a.fwd_partial = fwd_partial;
    a.tables = reinterpret_cast<float*>(fwd_partial + 2 * (size_t)p.fwd_grid);
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.iter = iter;
    hipError_t e = p.fwd_quad ? p.inst.launch_fwd_quad(a, p.fwd_grid, (hipStream_t)stream)
                   : p.fwd_split ? p.inst.launch_fwd_split(a, p.fwd_grid, (hipStream_t)stream)
                                 : p.inst.launch_fwd(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "hjb_fwd_kernel (eval) launch");
    return 0;
}

int psp_hjb_terminal_reduce(const psp_hjb_config* cfg, const double* fwd_partial, double* sums_out, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if (!fwd_partial || !sums_out) return fail(-1, "null buffer passed to psp_hjb_terminal_reduce");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fwd_partial,
                       p.fwd_grid, sums_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_partials_kernel launch");
    return 0;
}

int psp_hjb_terminal_reduce_loss(const psp_hjb_config* cfg, const double* fwd_partial, double* sums_out, float* loss_log,
                                 const uint32_t* index_dev, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if (!fwd_partial || !sums_out) return fail(-1, "null buffer passed to psp_hjb_terminal_reduce_loss");
    if (loss_log && cfg->loss_kind == PSP_LOSS_WEIGHTS)
        return fail(-1, "psp_hjb_terminal_reduce_loss: the caller forms the loss of a PSP_LOSS_WEIGHTS run");
    if (cfg->K_global <= 0) return fail(-1, "K_global must be positive");
    hipLaunchKernelGGL(reduce_partials_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fwd_partial, p.fwd_grid,
                       sums_out, cfg->loss_kind, 1.0 / (double)cfg->K_global, loss_log, index_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_partials_loss_kernel launch");
    return 0;
}

int psp_iter_state_init(psp_iter_state* host_out, uint32_t iter, int32_t step, float beta1, float beta2) {
    if (!host_out || step <= 0) return fail(-1, "psp_iter_state_init needs an output struct and a 1-based step");
    host_out->iter = iter; host_out->step = (uint32_t)step;
    host_out->beta1_pow = pow((double)beta1, (double)step);
    host_out->beta2_pow = pow((double)beta2, (double)step);
    return 0;
}

int psp_iter_state_advance(psp_iter_state* dev_state, float beta1, float beta2, void* stream) {
    if (!dev_state) return fail(-1, "null state passed to psp_iter_state_advance");
    hipLaunchKernelGGL(iter_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev_state, (double)beta1, (double)beta2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "iter_advance_kernel launch");
    return 0;
}

int psp_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const psp_iter_state* dev_state, float lr, float beta1, float beta2, float eps, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !dev_state) return fail(-1, "null buffer passed to psp_adam_step_dev");
    if (n <= 0) return fail(-1, "psp_adam_step_dev needs n > 0");
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad,
                       exp_avg, exp_avg_sq, (long long)n, dev_state, lr, beta1, beta2, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "adam_dev_kernel launch");
    return 0;
}

namespace {
// the backward kernel(s) of a config: the split-product kernel where it exists, guarded by its fp32-MFMA twin when
// cfg->range_flag is set (both enqueued, predicated on range_flag[0] == 0 / == 1; same grid and partial-gradient layout)
hipError_t launch_hjb_bwd(const psp_hjb_config* cfg, const Plan& p, psp::HjbArgs& a, hipStream_t st) {
    const bool bwd_x3 = cfg->mlp_dtype == PSP_MLP_F16X3 && p.bwd_specialised && p.inst.launch_bwd2_x3 &&
                        (p.inst.wide || p.inst.bwd2_x3_lds_bytes() <= kMaxLds);   // (else the fp32 backward: same results, same store)
    auto fp32 = [&]() {
        return p.bwd_specialised ? p.inst.launch_bwd2(a, p.bwd_grid, st) : p.inst.launch_bwd(a, p.bwd_grid, p.bwd_waves * 64, st);
    };
    if (!bwd_x3) return fp32();
    const bool guard = cfg->range_flag != nullptr;
    if (guard) { a.cond = cfg->range_flag; a.cond_want = 0; }
    hipError_t e = p.inst.launch_bwd2_x3(a, p.bwd_grid, st);
    if (e != hipSuccess || !guard) return e;
    a.cond_want = 1;
    e = fp32();
    a.cond = nullptr; a.cond_want = 0;
    return e;
}
}  // namespace

int psp_hjb_rollout_bwd(const psp_hjb_config* cfg, const float* params, const float* xi, uint64_t seed,
                        uint32_t iter, const float* path, const float* D, const double* sums,
                        float* grad_partial, float* grad_out, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !path || !D || !sums || !grad_partial || !grad_out)
        return fail(-1, "null buffer passed to psp_hjb_rollout_bwd");
    if (cfg->noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    psp::HjbArgs a;
    fill_args(cfg, p, &a);
    a.params = params; a.xi = xi; a.path = const_cast<float*>(path); a.D = const_cast<float*>(D);
    a.sums = sums; a.grad_partial = grad_partial;
    a.tables = reinterpret_cast<float*>(reinterpret_cast<char*>(grad_partial) + grad_rows_bytes(p));
    if (a.dbg) a.dbg += (size_t)p.fwd_grid * 8 * 8;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.iter = iter;
    hipError_t e = launch_hjb_bwd(cfg, p, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "hjb_bwd_kernel launch");
    const int P = p.inst.n_params;
    hipLaunchKernelGGL(reduce_grad_kernel, dim3((P + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       grad_partial, p.bwd_grid, P, grad_out);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_grad_kernel launch");
    return 0;
}

int psp_hjb_rollout_bwd_step(const psp_hjb_config* cfg, float* params, const float* path, const float* D, const double* sums,
                             float* grad_partial, float* grad_out, float* exp_avg, float* exp_avg_sq, psp_iter_state* dev_state,
                             uint32_t* ticket, float lr, float beta1, float beta2, float eps, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !path || !D || !sums || !grad_partial || !grad_out || !exp_avg || !exp_avg_sq || !dev_state || !ticket)
        return fail(-1, "null buffer passed to psp_hjb_rollout_bwd_step");
    if (cfg->store_path == 4) return fail(-1, "psp_hjb_rollout_bwd_step takes no Philox seed: store_path 4 goes through psp_hjb_rollout_bwd");
    psp::HjbArgs a;
    fill_args(cfg, p, &a);
    a.params = params; a.path = const_cast<float*>(path); a.D = const_cast<float*>(D);
    a.sums = sums; a.grad_partial = grad_partial;
    a.tables = reinterpret_cast<float*>(reinterpret_cast<char*>(grad_partial) + grad_rows_bytes(p));
    if (a.dbg) a.dbg += (size_t)p.fwd_grid * 8 * 8;
    hipError_t e = launch_hjb_bwd(cfg, p, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "hjb_bwd_kernel launch");
    const int P = p.inst.n_params;
    hipLaunchKernelGGL(reduce_grad_adam_advance_kernel, dim3((P + 31) / 32), dim3(256), 0, (hipStream_t)stream, grad_partial,
                       p.bwd_grid, P, grad_out, params, exp_avg, exp_avg_sq, dev_state, ticket, lr, beta1, beta2, eps);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_grad_adam_advance_kernel launch");
    return 0;
}

int psp_hjb_adjoint_sweep(const psp_hjb_config* cfg, const float* params, float* path, const float* XN,
                          const float* mu, const float* nu, const float* wT, double* fwd_partial, void* stream) {
    Plan p;
    int rc = make_plan(cfg, &p);
    if (rc) return rc;
    if ((rc = check_ptrs(cfg))) return rc;
    if (!p.inst.launch_adj) return fail(-2, "the adjoint sweep is not built for this kernel instance");
    if (!params || !path || !XN || !mu || !fwd_partial) return fail(-1, "null buffer passed to psp_hjb_adjoint_sweep");
    if (cfg->store_path != 2 && cfg->store_path != 3)
        return fail(-1, "psp_hjb_adjoint_sweep needs the path written with store_path = 2 or 3");
    if (!cfg->adaptive) return fail(-1, "without the adaptive forward process the state path carries no gradient");
    psp::HjbArgs a;
    fill_args(cfg, p, &a);
    a.params = params; a.path = path; a.XN = const_cast<float*>(XN); a.adj_mu = mu; a.adj_nu = nu; a.adj_wT = wT;
    // the wide family keeps its (transposed) operand tables where the forward kernel kept its own: behind the partial
    // sums of the forward scratch, whatever forward variant wrote them
    a.tables = reinterpret_cast<float*>(fwd_partial + 2 * (size_t)p.fwd_grid);
    if (p.fwd_quad && p.inst.launch_adj_quad) {           // smallest K: four trajectories per workgroup (hjbq_adj_kernel)
        hipError_t eq = p.inst.launch_adj_quad(a, 4 * p.ntile16, (hipStream_t)stream);
        if (eq != hipSuccess) return fail_hip(eq, "hjbq_adj_kernel launch");
        return 0;
    }
    // one wave per 16-trajectory tile, like the forward kernels (the recursion is sequential in time)
    int fw = (p.ntile16 + n_cus() - 1) / n_cus();
    if (fw < 1) fw = 1;
    if (fw > (p.inst.wide ? 4 : 8)) fw = p.inst.wide ? 4 : 8;
    const int grid = (p.ntile16 + fw - 1) / fw;
    const bool adj_x3 = cfg->mlp_dtype == PSP_MLP_F16X3 && p.inst.launch_adj_x3;    // (make_plan checked the LDS fit of the forward carve)
    const bool guard = adj_x3 && cfg->range_flag != nullptr;                         // range guard: both sweeps, predicated
    if (guard) { a.cond = cfg->range_flag; a.cond_want = 0; }
    hipError_t e = adj_x3 ? p.inst.launch_adj_x3(a, grid, fw * 64, (hipStream_t)stream)
                          : p.inst.launch_adj(a, grid, fw * 64, (hipStream_t)stream);
    if (e == hipSuccess && guard) {
        a.cond_want = 1;
        e = p.inst.launch_adj(a, grid, fw * 64, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "hjb_adj_kernel launch");
    return 0;
}

int psp_gen_supported(int32_t d, int32_t H) {
    psp::GenInstance inst;
    return find_gen_instance(d, H, &inst) ? 1 : 0;
}

int psp_gen_instance_count(void) { return (int)(sizeof(kGenTable) / sizeof(kGenTable[0])); }

int psp_gen_instance_get(int32_t i, int32_t* d, int32_t* H) {
    const int n = (int)(sizeof(kGenTable) / sizeof(kGenTable[0]));
    if (i < 0 || i >= n || !d || !H) return fail(-1, "instance index out of range");
    *d = kGenTable[i].d; *H = kGenTable[i].H;
    return 0;
}

int psp_gen_query(const psp_gen_config* cfg, psp_gen_sizes* out) {
    GenPlan p;
    int rc = make_gen_plan(cfg, &p);
    if (rc) return rc;
    if (!out) return fail(-1, "null output");
    memset(out, 0, sizeof(*out));
    out->n_params = p.inst.n_params;
    out->fwd_workgroups = p.fwd_grid;
    out->bwd_workgroups = p.bwd_grid;
    const int64_t blk = cfg->mlp_dtype == PSP_MLP_BF16 ? p.inst.path_dwords_per_block16 : p.inst.path_floats_per_block;
    out->path_bytes = cfg->store_path ? (int64_t)(cfg->N + 1) * p.ntile16 * blk * 4 : 0;
    out->ahat_bytes = (int64_t)(cfg->N + 1) * p.ntile16 * 16 * 4;
    out->grad_partial_bytes = (int64_t)p.bwd_grid * p.inst.n_params * 4;
    return 0;
}

int psp_gen_rollout_fwd(const psp_gen_config* cfg, const float* params, const float* x0, const float* t0,
                        const float* xi, uint64_t seed, uint32_t iter, float* path, float* ahat, float* VN,
                        float* YN, float* XN, float* tN, unsigned long long* kcount, void* stream) {
    GenPlan p;
    int rc = make_gen_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !x0 || !t0 || !VN || !YN || !XN || !tN || !kcount)
        return fail(-1, "null buffer passed to psp_gen_rollout_fwd");
    if (cfg->noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    if (cfg->store_path && (!path || !ahat)) return fail(-1, "store_path set but path / ahat buffer is null");
    psp::GenArgs a;
    fill_gen_args(cfg, p, &a);
    a.params = params; a.x0 = x0; a.t0 = t0; a.xi = xi; a.path = path; a.ahat = ahat;
    a.VN = VN; a.YN = YN; a.XN = XN; a.tN = tN; a.kcount = kcount;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.iter = iter;
    if (cfg->mlp_dtype < PSP_MLP_FP32 || cfg->mlp_dtype > PSP_MLP_F16X3) return fail(-1, "mlp_dtype out of range");
    if (cfg->mlp_dtype == PSP_MLP_F16X3 && (!p.inst.launch_fwd_x3 || p.inst.fwd_x3_lds_bytes() > kMaxLds))
        return fail(-3, "split-product forward tables do not fit the 160 KiB LDS for this (d,H)");
    if (cfg->mlp_dtype == PSP_MLP_F16X3 && cfg->range_flag) {
        hipLaunchKernelGGL(snapshot_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, kcount,
                           reinterpret_cast<unsigned long long*>(cfg->range_flag + 2));
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail_hip(es, "snapshot_u64_kernel launch");
    }
    hipError_t e = cfg->mlp_dtype == PSP_MLP_F16X3 ? p.inst.launch_fwd_x3(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream)
                   : cfg->mlp_dtype != PSP_MLP_FP32 ? p.inst.launch_fwd_bf16(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream)
                                                    : p.inst.launch_fwd(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
    if (e == hipSuccess && cfg->mlp_dtype == PSP_MLP_F16X3 && cfg->range_flag) {
        // range guard (psp_gen_config.range_flag): a non-finite V(X_N) / Y_N -> flag -> the fp32-MFMA forward, predicated.
        // kcount accumulates (atomicAdd): the flag kernel takes back what the split kernel added when it raises the flag
        hipLaunchKernelGGL(range_flag_arrays_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, VN, YN, cfg->K_local,
                           cfg->range_flag, kcount, reinterpret_cast<const unsigned long long*>(cfg->range_flag + 2));
        if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "range_flag_arrays_kernel launch");
        a.cond = cfg->range_flag; a.cond_want = 1;
        e = p.inst.launch_fwd(a, p.fwd_grid, p.fwd_waves * 64, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "gen_fwd_kernel launch");
    return 0;
}

int psp_gen_rollout_bwd(const psp_gen_config* cfg, const float* params, const float* path, const float* ahat,
                        const float* wY, const float* wV, float* grad_partial, float* grad_out, void* stream) {
    GenPlan p;
    int rc = make_gen_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !path || !ahat || !wY || (!wV && !cfg->per_sample_weights) || !grad_partial || !grad_out)
        return fail(-1, "null buffer passed to psp_gen_rollout_bwd");
    psp::GenArgs a;
    fill_gen_args(cfg, p, &a);
    a.params = params; a.path = const_cast<float*>(path); a.ahat = const_cast<float*>(ahat);
    a.wY = wY; a.wV = wV; a.grad_partial = grad_partial;
    a.dbg = (g_dbg && g_dbg_n >= (long long)p.bwd_grid * 8 * 8) ? g_dbg : nullptr;      // (-DPSP_STAMPS builds; the kernels ignore it otherwise)
    if (cfg->mlp_dtype < PSP_MLP_FP32 || cfg->mlp_dtype > PSP_MLP_F16X3) return fail(-1, "mlp_dtype out of range");
    if (cfg->mlp_dtype == PSP_MLP_BF16 && !p.bwd_specialised)
        return fail(-3, "the bf16 backward exists for the role-specialised kernel only (LDS budget / PSP_BWD_VARIANT)");
    // (PSP_MLP_F16X3 with shared trajectory weights: the split-product consumers; per-sample weights keep the fp32 kernel)
    const bool bwd_x3 = cfg->mlp_dtype == PSP_MLP_F16X3 && p.bwd_specialised && !cfg->per_sample_weights && p.inst.launch_bwd2_x3;
    const bool guard = bwd_x3 && cfg->range_flag != nullptr;     // range guard: split and fp32-MFMA backward, predicated
    if (guard) { a.cond = cfg->range_flag; a.cond_want = 0; }
    hipError_t e = cfg->mlp_dtype == PSP_MLP_BF16 ? p.inst.launch_bwd2_bf16(a, p.bwd_grid, (hipStream_t)stream)
                   : bwd_x3 ? p.inst.launch_bwd2_x3(a, p.bwd_grid, (hipStream_t)stream)
                   : p.bwd_specialised ? p.inst.launch_bwd2(a, p.bwd_grid, (hipStream_t)stream)
                                       : p.inst.launch_bwd(a, p.bwd_grid, 256, (hipStream_t)stream);
    if (e == hipSuccess && guard) {
        a.cond_want = 1;
        e = p.inst.launch_bwd2(a, p.bwd_grid, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail_hip(e, "gen_bwd_kernel launch");
    const int P = p.inst.n_params;
    hipLaunchKernelGGL(reduce_grad_kernel, dim3((P + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       grad_partial, p.bwd_grid, P, grad_out);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_grad_kernel launch");
    return 0;
}


// ---- value nets of any depth (genl_kernels.h) ----------------------------------------------------------------------------
namespace {
struct GenlPlan { psp::GenlArgs a; int ntile16; long long table_floats; long long n_params; int fwd_lds, bwd_lds, nw_fwd, nw_bwd,
                  bwd_grid, bwd_groups, tlds; };
int make_genl_plan(const psp_genl_config* c, GenlPlan* p) {
    if (!c) return fail(-1, "null config");
    const psp_gen_config& b = c->base;
    if (b.d <= 0 || b.K_local <= 0 || b.N <= 0) return fail(-1, "non-positive d/K/N");
    const int L = c->n_hidden;
    if (L < 1 || L > psp::GENL_MAXL) return fail(-2, "value net: 1 to 4 hidden layers");
    const int D0 = b.d + (c->has_time ? 1 : 0);
    if (D0 > 16 * psp::GENL_MAXDB) return fail(-2, "value net: input width above 112");
    if ((b.drift_kind != PSP_DRIFT_ZERO && b.drift_kind != PSP_DRIFT_DOUBLE_WELL && b.drift_kind != PSP_DRIFT_DIAG) || b.h_kind < 0 ||
        b.h_kind > PSP_GH_EXPBALL_SIN || b.noise_mode < 0 || b.noise_mode > 1 || b.domain_kind < 0 || b.domain_kind > PSP_DOM_ANNULUS)
        return fail(-1, "config enum out of range");
    if (c->activation < 0 || c->activation > PSP_ACT_TANH) return fail(-1, "value net: unknown activation");
    if (b.domain_kind == PSP_DOM_SPHERE && !(b.dom_a > 0.f)) return fail(-1, "sphere radius must be positive");
    if (b.domain_kind == PSP_DOM_BOX && !(b.dom_a < b.dom_b)) return fail(-1, "box bounds must satisfy X_l < X_r");
    if (b.domain_kind == PSP_DOM_ANNULUS && !(b.dom_a >= 0.f && b.dom_a < b.dom_b)) return fail(-1, "annulus radii must satisfy 0 <= r_1 < r_2");
    if (b.drift_kind != PSP_DRIFT_ZERO && !b.drift) return fail(-1, "drift vector missing (double-well kappa / diagonal of A)");
    psp::GenlArgs& a = p->a;
    memset(&a, 0, sizeof(a));
    a.d = b.d; a.D0 = D0; a.has_time = c->has_time ? 1 : 0; a.L = L;
    a.act = c->activation; a.linear_layout = c->linear_layout ? 1 : 0;
    a.time_first = (c->has_time && c->time_first) ? 1 : 0;
    a.time_scale = (c->has_time && c->time_scale != 0.f) ? c->time_scale : 1.0f;
    a.DB0 = (D0 + 15) / 16;
    a.off[0] = 0; a.roff[0] = 0;
    int blocks = a.DB0, real = D0, hbsum = 0, tiles = 0;
    long long pofs = 0, tofs = 0;
    for (int i = 0; i < L; ++i) {
        const int Hi = c->widths[i];
        if (Hi < 1 || Hi > 16 * psp::GENL_MAXHB) return fail(-2, "value net: hidden widths between 1 and 128");
        a.H[i] = Hi; a.HB[i] = (Hi + 15) / 16;
        a.off[i + 1] = blocks; a.roff[i + 1] = real; a.inw[i] = real;
        a.oW[i] = (int)pofs; pofs += (long long)real * Hi;
        a.ob[i] = (int)pofs; pofs += Hi;
        a.tF[i] = tofs; tofs += (long long)a.HB[i] * 4 * blocks * 64;           // [HB_i][4 * input blocks][64]
        a.tR[i] = tofs; tofs += (long long)blocks * 4 * a.HB[i] * 64;           // [input blocks][4 HB_i][64]
        a.vB[i] = tofs; tofs += (long long)a.HB[i] * 16;
        a.tcum[i] = tiles; tiles += blocks * a.HB[i];
        blocks += a.HB[i]; real += Hi; hbsum += a.HB[i];
    }
    a.tcum[L] = tiles;
    a.inw[L] = real;                                                            // (input width of the output layer)
    a.oW[L] = (int)pofs; pofs += real;
    a.ob[L] = (int)pofs; pofs += 1;
    a.TB = blocks; a.HBsum = hbsum; a.P = pofs;
    tiles += blocks;                                                            // the output layer as a layer of one unit
    a.n_tiles = tiles;
    a.vW = tofs; tofs += (long long)blocks * 16;
    p->table_floats = tofs; p->n_params = pofs;
    p->ntile16 = (b.K_local + 15) / 16;
    // waves per tile: one (no barriers, many tiles per CU) for small nets -- always for the smallest, for the others once the
    // batch fills the chip several times over --, else eight waves cutting every product by output block
    const bool small = hbsum <= 8 && blocks <= 16;
    const int cus = n_cus();
    const char* force = getenv("PSP_GENL_NW");
    int nw = (small && (hbsum <= 5 || p->ntile16 >= 4 * cus)) ? 1 : 8;
    if (force && force[0] == '1' && small) nw = 1;
    if (force && force[0] == '8') nw = 8;
    p->nw_fwd = nw; p->nw_bwd = nw;
    if (nw == 8 && p->ntile16 >= 2 * cus) p->nw_fwd = 4;                        // two tiles per CU in flight (genl_kernels.h)
    if (force && force[0] == '4' && nw == 8) p->nw_fwd = 4;
    if (force && force[0] == '8') p->nw_fwd = 8;
    p->fwd_lds = psp::genl_fwd_lds_bytes(a.TB); p->bwd_lds = psp::genl_bwd_lds_bytes(a.TB, a.DB0, p->nw_bwd);
    // LDS-resident tables for the one-wave instances of small nets (genl_fwd_kernel TLDS) are OPT-IN (PSP_GENL_TLDS=1): measured on
    // the committor notebook's net, they shorten the step chain of a K = 200 batch by 6 % (12.1 -> 11.3 ms: the chain is bound by
    // its instruction count, not by the L2 latency of the 29 KB of tables) and cost a batch that fills the chip 75 % (72.7 -> 127 ms
    // at K = 65536: three workgroups per CU instead of twelve, and every workgroup copies the tables)
    a.table_floats = tofs;
    p->tlds = 0;
    {
        const char* et = getenv("PSP_GENL_TLDS");
        const int tb = (int)(4 * ((tofs + 3) / 4 * 4));
        if (nw == 1 && tb <= 36 * 1024 && et && et[0] == '1') {
            p->tlds = 1; p->fwd_lds += tb; p->bwd_lds += tb;
        }
    }
    if (p->fwd_lds > kMaxLds || p->bwd_lds > kMaxLds)
        return fail(-3, "value net: the activation images exceed the 160 KiB LDS (sum of the padded widths too large)");
    const long long nblk = (long long)(b.N + 1) * p->ntile16;
    if (nblk >= (1LL << 31)) return fail(-1, "(N + 1) * ceil(K/16) must stay below 2^31");
    const int per_group = p->nw_bwd * psp::GenlGeo<1>::MAXT;
    p->bwd_groups = (tiles + per_group - 1) / per_group;
    // workgroups of the backward kernel: what the CUs hold at once (LDS-limited), never more than there are sample blocks
    long long per_cu = nw == 1 ? 8 : (p->bwd_lds > kMaxLds / 2 ? 1 : 2);
    if (nw == 1 && (long long)p->bwd_lds * per_cu > kMaxLds) per_cu = kMaxLds / p->bwd_lds;
    long long grid = per_cu * cus;
    if (grid > nblk) grid = nblk;
    p->bwd_grid = (int)grid;
    // the GenArgs part: as fill_gen_args
    psp::GenArgs& g = a.g;
    g.drift = b.drift; g.k_offset = b.k_offset; g.K_local = b.K_local; g.N = b.N; g.ntile16 = p->ntile16;
    g.dt = b.dt; g.sqdt = b.sqrt_dt; g.T = b.T; g.sigma_scale = b.sigma_scale;
    g.drift_kind = b.drift_kind; g.h_kind = b.h_kind; g.adaptive = b.adaptive;
    g.noise_mode = b.noise_mode; g.store_path = b.store_path;
    g.domain_kind = b.domain_kind; g.dom_a = b.dom_a; g.dom_b = b.dom_b; g.d_real = b.d;
    for (int i = 0; i < 4; ++i) g.h_par[i] = b.h_par[i];
    return 0;
}
// the step counts of the tiles live behind the (N + 1) x 16 ceil(K/16) coefficients of `ahat`
int* genl_nexec(const psp_genl_config* cfg, const GenlPlan& p, const float* ahat) {
    return reinterpret_cast<int*>(const_cast<float*>(ahat)) + (size_t)(cfg->base.N + 1) * p.ntile16 * 16;
}
}  // namespace

int psp_genl_query(const psp_genl_config* cfg, psp_genl_sizes* out) {
    GenlPlan p;
    int rc = make_genl_plan(cfg, &p);
    if (rc) return rc;
    if (!out) return fail(-1, "null output");
    memset(out, 0, sizeof(*out));
    const int64_t nblk = (int64_t)(cfg->base.N + 1) * p.ntile16;
    out->table_bytes = p.table_floats * 4;
    out->path_bytes = cfg->base.store_path ? nblk * 2 * p.a.DB0 * 256 * 4 : 0;
    out->ahat_bytes = nblk * 16 * 4 + (int64_t)p.ntile16 * 4;
    out->n_params = p.n_params;
    out->grad_partial_bytes = (int64_t)p.bwd_grid * p.n_params * 4;
    out->n_blocks = (int32_t)nblk;
    out->fwd_workgroups = p.ntile16;
    out->bwd_workgroups = p.bwd_grid * p.bwd_groups;
    out->waves_per_tile = p.nw_fwd;
    for (int i = 0; i <= p.a.L; ++i) out->seg_block_offset[i] = p.a.off[i];
    return 0;
}

int psp_genl_rollout_fwd(const psp_genl_config* cfg, const float* params, const float* x0, const float* t0,
                                    const float* xi, uint64_t seed, uint32_t iter, float* tables, float* path, float* ahat,
                                    float* VN, float* YN, float* XN, float* tN, unsigned long long* kcount, void* stream) {
    GenlPlan p;
    int rc = make_genl_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !x0 || !tables || !VN || !YN || !XN || !tN || !kcount || !ahat) return fail(-1, "null buffer passed to psp_genl_rollout_fwd");
    if (cfg->has_time && !t0) return fail(-1, "t0 missing");
    if (cfg->base.noise_mode == PSP_NOISE_SUPPLIED && !xi) return fail(-1, "supplied-noise mode needs xi");
    if (cfg->base.store_path && !path) return fail(-1, "store_path set but the path buffer is null");
    psp::GenlArgs& a = p.a;
    a.tables = tables; a.tables_w = tables;
    a.nexec = genl_nexec(cfg, p, ahat);
    psp::GenArgs& g = a.g;
    g.params = params; g.x0 = x0; g.t0 = t0; g.xi = xi; g.path = path; g.ahat = ahat;
    g.VN = VN; g.YN = YN; g.XN = XN; g.tN = tN; g.kcount = kcount;
    g.Vsteps = cfg->base.v_steps_out; g.Ysteps = cfg->base.y_steps_out;
    g.seed_lo = (uint32_t)seed; g.seed_hi = (uint32_t)(seed >> 32); g.iter = iter;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(psp::genl_tables_kernel, dim3(128), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "genl_tables_kernel launch");
    e = p.nw_fwd == 1 ? (p.tlds ? psp::genl_launch_fwd<1, true>(p.a, p.ntile16, p.fwd_lds, st) : psp::genl_launch_fwd<1>(p.a, p.ntile16, p.fwd_lds, st))
        : p.nw_fwd == 4 ? psp::genl_launch_fwd<4>(p.a, p.ntile16, p.fwd_lds, st) : psp::genl_launch_fwd<8>(p.a, p.ntile16, p.fwd_lds, st);
    if (e != hipSuccess) return fail_hip(e, "genl_fwd_kernel launch");
    return 0;
}

int psp_genl_rollout_bwd(const psp_genl_config* cfg, const float* params, const float* tables, const float* path,
                         const float* ahat, const float* wY, const float* wV, float* grad_partial, float* grad_out, void* stream) {
    GenlPlan p;
    int rc = make_genl_plan(cfg, &p);
    if (rc) return rc;
    if (!params || !tables || !path || !ahat || !wY || (!wV && !cfg->base.per_sample_weights) || !grad_partial || !grad_out)
        return fail(-1, "null buffer passed to psp_genl_rollout_bwd");
    psp::GenlArgs& a = p.a;
    a.tables = tables;
    a.nexec = genl_nexec(cfg, p, ahat);
    a.gpart = grad_partial;
    psp::GenArgs& g = a.g;
    g.params = params; g.path = const_cast<float*>(path); g.ahat = const_cast<float*>(ahat); g.wY = wY; g.wV = wV;
    g.per_sample = cfg->base.per_sample_weights ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = p.nw_bwd == 1 ? (p.tlds ? psp::genl_launch_bwd<1, true>(p.a, p.bwd_grid, p.bwd_groups, p.bwd_lds, st) : psp::genl_launch_bwd<1>(p.a, p.bwd_grid, p.bwd_groups, p.bwd_lds, st)) : (p.a.HBsum <= 24 ? psp::genl_launch_bwd<8, false, 3>(p.a, p.bwd_grid, p.bwd_groups, p.bwd_lds, st)
                           : psp::genl_launch_bwd<8>(p.a, p.bwd_grid, p.bwd_groups, p.bwd_lds, st));     // (three slots per wave suffice: genl_kernels.h)
    if (e != hipSuccess) return fail_hip(e, "genl_bwd_kernel launch");
    const int P = (int)p.n_params;
    hipLaunchKernelGGL(reduce_grad_kernel, dim3((P + 31) / 32), dim3(256), 0, st, grad_partial, p.bwd_grid, P, grad_out);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "reduce_grad_kernel launch");
    return 0;
}

// ---- collectives: RCCL on the caller's stream, bound lazily so that the library loads without librccl ---------------
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int bind_rccl() {
    if (g_rccl.handle) return 0;
    // a process that already loaded RCCL (torch ships its own librccl.so.1) gets that copy back: same soname
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(-20, "cannot load librccl.so.1: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
        return fail(-20, "librccl.so.1 lacks an expected entry point");
    g_rccl = r;
    return 0;
}
int fail_rccl(ncclResult_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
    return -21;
}
}  // namespace

int psp_comm_unique_id(unsigned char id_out[PSP_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == PSP_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return fail(-1, "null id buffer");
    int rc = bind_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t e = g_rccl.GetUniqueId(&id);
    if (e != ncclSuccess) return fail_rccl(e, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int psp_comm_init(void** comm_out, int32_t nranks, int32_t rank, const unsigned char id[PSP_COMM_ID_BYTES]) {
    if (!comm_out || !id) return fail(-1, "null argument to psp_comm_init");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(-1, "psp_comm_init needs 0 <= rank < nranks");
    int rc = bind_rccl();
    if (rc) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    ncclResult_t e = g_rccl.CommInitRank(&comm, nranks, uid, rank);
    if (e != ncclSuccess) return fail_rccl(e, "ncclCommInitRank");
    *comm_out = comm;
    return 0;
}

int psp_comm_destroy(void* comm) {
    if (!comm) return 0;
    int rc = bind_rccl();
    if (rc) return rc;
    ncclResult_t e = g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
    if (e != ncclSuccess) return fail_rccl(e, "ncclCommDestroy");
    return 0;
}

int psp_allreduce(void* buf, int64_t n, int32_t dtype, void* comm, void* stream) {
    if (!buf || !comm) return fail(-1, "null buffer / communicator passed to psp_allreduce");
    if (n <= 0) return fail(-1, "psp_allreduce needs n > 0");
    if (dtype != PSP_DT_F32 && dtype != PSP_DT_F64) return fail(-1, "psp_allreduce: dtype must be PSP_DT_F32 or PSP_DT_F64");
    int rc = bind_rccl();
    if (rc) return rc;
    ncclResult_t e = g_rccl.AllReduce(buf, buf, (size_t)n, dtype == PSP_DT_F32 ? ncclFloat32 : ncclFloat64, ncclSum,
                                      static_cast<ncclComm_t>(comm), (hipStream_t)stream);
    if (e != ncclSuccess) return fail_rccl(e, "ncclAllReduce");
    return 0;
}

int psp_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  int32_t step, float lr, float beta1, float beta2, float eps, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq) return fail(-1, "null buffer passed to psp_adam_step");
    if (n <= 0 || step <= 0) return fail(-1, "psp_adam_step needs n > 0 and a 1-based step");
    // bias corrections in double on the host, as torch does for python-float steps
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params,
                       grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, step_size, bc2_sqrt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "adam_kernel launch");
    return 0;
}

int psp_philox_normal_fill(float* out, int32_t N, int32_t K_local, int32_t d, int64_t k_offset,
                           uint64_t seed, uint32_t iter, void* stream) {
    if (!out || N <= 0 || K_local <= 0 || d <= 0) return fail(-1, "bad arguments to psp_philox_normal_fill");
    const long long n0 = (long long)K_local * d;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n0);
    const long long total = (long long)N * K_local * (((d + 15) / 16) * 4);
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       out, N, K_local, d, (long long)k_offset, (uint32_t)seed, (uint32_t)(seed >> 32), iter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "philox_fill_kernel launch");
    return 0;
}

int psp_hjb_control_eval(int32_t d, int32_t H, const float* params, const float* X, int32_t K, float t,
                         float* minus_Z_out, void* stream) {
    if (!params || !X || !minus_Z_out || d <= 0 || H <= 0 || K <= 0)
        return fail(-1, "bad arguments to psp_hjb_control_eval");
    hipLaunchKernelGGL(control_eval_kernel, dim3(K), dim3(64), 2 * H * sizeof(float), (hipStream_t)stream, d, H,
                       params, X, K, t, minus_Z_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "control_eval_kernel launch");
    return 0;
}

}  // extern "C"
