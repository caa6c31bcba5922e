// gen_kernels.h -- CDNA4 kernels for GeneralSolver.train (diffusion / BSDE loss, unbounded domains).
// Reference: solver.py:1001-1206 (step :1091-1160); V network: function_space.py:116-140 (DenseNet,
// two hidden layers, relu(.)^2, weights stored (in, out), input = [x, t] with time LAST).
//
// Same register-chained "T layout" as hjb_kernels.h (16 trajectories per wave, weights as
// pre-permuted MFMA A-operand tables in LDS).  Per time step the forward kernel evaluates
//   z1 = W1^T x0 + b1, h1 = relu(z1)^2 ; z2 = W2^T [x0,h1] + b2, h2 = relu(z2)^2 ; V = W3^T [x0,h1,h2] + b3
//   grad_x0 V by the reverse sweep (g_z2 = w3h2 * 2relu(z2), g_h1 = w3h1 + W2h g_z2, ...), Z = s grad_x V,
//   the masked Euler / Y update, and -- for the backward pass -- the loss-weight-independent part of
//   the tangent pass in direction u^ = act ((-h_z + c) dt + xi sqrt(dt)):  z1^ = W1^T (s u^), z2^ = ...
// The backward kernel then needs, per sample, only  grad_theta [ a V + w (s u^) . grad_x V ]
// (reverse-over-forward), with a = w_Y a^ (a^ stored) or w_V at the final point.
#pragma once
#include "hjb_kernels.h"

namespace psp {

enum { GH_ZERO = 0, GH_QUAD = 1, GH_ALLEN_CAHN = 2, GH_EXPBALL_LIN = 3, GH_EXPBALL_SQ = 4, GH_EXPBALL_SIN = 5 };
// exit test of a bounded domain (solver.py:1119-1129, :758-767): sphere |X_n| < a (the state BEFORE the move), boxes on the proposal,
// annulus a < |X_n| < b ('two_spheres', :1122-1123 / :752-753; the state before the move).  'square-corner' (:759-760) tests
// any(X_proposal <= X_r): that is DOM_BOX_UPPER_ANY
enum { DOM_NONE = 0, DOM_SPHERE = 1, DOM_BOX = 2, DOM_BOX_UPPER_ALL = 3, DOM_BOX_UPPER_ANY = 4, DOM_ANNULUS = 5 };

struct GenArgs {
    const float* params;
    const float* x0;      // (K_local, d) initial points
    const float* t0;      // (K_local) initial times
    const float* xi;      // supplied noise (N, K_local, d) or null
    float* path;          // N+1 slots x ntile16 blocks of GGeo::PB floats
    float* ahat;          // (N+1, 16*ntile16) per-sample value-gradient coefficient (without the loss weight)
    float* VN;            // (K_local) V(X_N, t_N)
    float* YN;            // (K_local)
    float* XN;            // (K_local, d)
    float* tN;            // (K_local)
    unsigned long long* kcount;   // active-step counter (K_log, solver.py:1152)
    const float* drift;   // kappa (d) for the double well
    const float* wY;      // (K_local) dLoss/dY_N      (backward)
    const float* wV;      // (K_local) dLoss/dV(X_N)   (backward)
    float* Vsteps;        // optional (N, 16*ntile16): V(X_n, t_n) of every step (value_function ansatz of Solver, solver.py:438-440)
    float* Ysteps;        // optional (N, 16*ntile16): the running Y BEFORE the increment of step n
    int path16;           // bf16-pair path block (GGeo::q*) instead of the fp32 register images: set with mlp_dtype == PSP_MLP_BF16
    int per_sample;       // backward: wY is (N+1, 16*ntile16) tangent weights per sample and ahat holds the value-gradient coefficient itself
    float* grad_partial;
    long long k_offset;
    int K_local, N, ntile16;
    float dt, sqdt, T, sigma_scale;
    int drift_kind, h_kind, adaptive, noise_mode, store_path;
    int domain_kind;      // DOM_*
    int d_real;           // state components the exit test / |x|^2 see (a zero-padded instance carries noise in the padding)
    float dom_a, dom_b;   // sphere radius / box bounds X_l, X_r
    float h_par[4];       // GH_EXPBALL_*: alpha, d (real dimension), coefficient of the extra -y, 1 if the exponent carries 2 t
    uint32_t seed_lo, seed_hi, iter;
    const int* cond;      // launch predicate of the guarded split-product mode (hjb_kernels.h PSP_COND_EXIT)
    int cond_want;
    unsigned long long* dbg;   // diagnostic builds (-DPSP_STAMPS): per-wave cycle sums of gen_bwd2_kernel (tools/r4/gen_stamps.py)
};

template <int D, int H>
struct GGeo {
    static constexpr int DI = D + 1;                       // network input: [x (D), t]
    static constexpr int DBI = cdiv(DI, 16), KSI = cdiv(DI, 4), HB = cdiv(H, 16), KSH = cdiv(H, 4);
    // DenseNet flat parameter offsets (registration order W1,b1,W2,b2,W3,b3; weights are (in, out))
    static constexpr int oW1 = 0, ob1 = DI * H, oW2 = ob1 + H, ob2 = oW2 + (DI + H) * H, oW3 = ob2 + H,
                         ob3 = oW3 + DI + 2 * H, P = ob3 + 1;
    // time feature position in the T layout (feature index D)
    static constexpr int TB = D / 16, TR = (D % 16) / 4, TQ = D % 4;
    // path block (16 samples): register images padded to whole blocks
    static constexpr int pX = 0, pU = pX + 4 * DBI * 64, pD1 = pU + 4 * DBI * 64, pD2 = pD1 + 4 * HB * 64,
                         pZ1 = pD2 + 4 * HB * 64, pZ2 = pZ1 + 4 * HB * 64, PB = pZ2 + 4 * HB * 64;
    // bf16 path block (psp_gen_config.mlp_dtype == PSP_MLP_BF16: both rollout kernels on bf16 MFMA, whose operands are rounded
    // to bf16 anyway): every image as bf16 PAIRS -- dword (2b + r', lane) packs k-steps 4b + r' (low half) and 4b + r' + 2
    // (high half) of lane (j, q).  One dword per lane is two T-layout registers; the 16 bytes a lane reads in F layout hold
    // four samples of feature 16b + 4r' + q (low halves) and of feature + 8 (high halves).  960 instead of 1 920 bytes per
    // sample: both kernels of that mode are bound by the path store, not by the matrix pipe.
    static constexpr int qX = 0, qU = qX + 2 * DBI * 64, qD1 = qU + 2 * DBI * 64, qD2 = qD1 + 2 * HB * 64,
                         qZ1 = qD2 + 2 * HB * 64, qZ2 = qZ1 + 2 * HB * 64, PB16 = qZ2 + 2 * HB * 64;
    // forward LDS carve (floats): six A-operand tables + per-feature vectors
    static constexpr int fW1f = 0, fW2xf = fW1f + HB * KSI * 64, fW2hf = fW2xf + HB * KSI * 64,
                         fW2hr = fW2hf + HB * KSH * 64, fW2xr = fW2hr + HB * KSH * 64,
                         fW1r = fW2xr + DBI * KSH * 64, fVec = fW1r + DBI * KSH * 64;
    static constexpr int vb1 = fVec, vb2 = vb1 + HB * 16, vw3h1 = vb2 + HB * 16, vw3h2 = vw3h1 + HB * 16,
                         vw3x = vw3h2 + HB * 16, vdr = vw3x + DBI * 16, fRed = vdr + DBI * 16, fEnd = fRed + 64;
    static int fwd_lds_floats() { return fEnd; }
    // split-product forward (gen_fwd_kernel<.., X3>): the same six tables as hi / lo f16 images (SplitGeo, hjb_kernels.h); the
    // vectors and the reduction slot follow in the same order
    static constexpr int xW1f = 0, xW2xf = xW1f + SplitGeo<KSI, DBI>::floats(HB), xW2hf = xW2xf + SplitGeo<KSI, DBI>::floats(HB),
                         xW2hr = xW2hf + SplitGeo<KSH, HB>::floats(HB), xW2xr = xW2hr + SplitGeo<KSH, HB>::floats(HB),
                         xW1r = xW2xr + SplitGeo<KSH, HB>::floats(DBI), xVec = xW1r + SplitGeo<KSH, HB>::floats(DBI);
    static int fwd_x3_lds_floats() { return xVec + (fEnd - fVec); }
    // backward: 4 waves arranged WH (column blocks of H) x WD (row blocks).  Rows are split first: the
    // A operands (X, U, d1, z1^ images) then go to exactly one wave each, while the B operands (adjoint
    // panels) are shared through the LDS exchange tiles.
    static constexpr int WD = (DBI >= 4) ? 4 : (DBI >= 2 ? 2 : 1), WH = 4 / WD;
    static constexpr int NIB = cdiv(HB, WH);               // column blocks per wave
    static constexpr int NRX = cdiv(DBI, WD), NRH = cdiv(HB, WD);   // x-row / h-row blocks per wave
    static constexpr int gW2hr = 0, gEx = gW2hr + HB * KSH * 64;
    // exchange tiles (1 KiB) per wave: [gz2 | gz2' | gz1 | gz1'].  d=100, H=64: 16 KiB table + 64 KiB
    // exchange = exactly 80 KiB -> two workgroups per CU (the w3 vectors are read from global instead)
    static constexpr int EXT = 4 * HB;
    static constexpr int gEnd = gEx + 4 * EXT * 256;
    static int bwd_lds_floats() { return gEnd; }
};

// dispatch of one product: fp32 16x16x4 chain (gemm_T) or the bf16 16x16x32 one
// (MODE 0: fp32, 1: bf16, 2: fp32-grade split products on the f16 pipe, gemm_Tx)
template <int MODE, int MB, int KS, int INB>
__device__ __forceinline__ void gen_gemm(f32x4 (&acc)[MB], const float* wlds, const f32x4 (&in)[INB], int lane) {
    if constexpr (MODE == 2) gemm_Tx<MB, KS, INB, 1>(acc, wlds, in, lane);   // (one unit per chunk: the kernel sits on its 256 registers)
    else if constexpr (MODE == 1) gemm_Tb<MB, INB>(acc, wlds, in, lane);
    else gemm_T<MB, KS, INB>(acc, wlds, in, lane);
}
template <int MODE, int KS, int INB, class F>
__device__ __forceinline__ void gen_stage(float* dst, int MB, int tid, int nthr, F src) {
    if constexpr (MODE == 2) stage_aop_x3<KS, INB>(dst, MB, tid, nthr, src);
    else if constexpr (MODE == 1) stage_aop_bf16(dst, MB, (INB + 1) / 2, tid, nthr, src);
    else stage_aop(dst, MB, KS, tid, nthr, src);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    f32x4 o;
    o[0] = fmaxf(v[0], 0.f); o[1] = fmaxf(v[1], 0.f); o[2] = fmaxf(v[2], 0.f); o[3] = fmaxf(v[3], 0.f);
    return o;
}
template <int MODE>
__device__ __forceinline__ f32x4 relu4m(f32x4 v) {
    if constexpr (MODE == 2) return relu4n(v);
    else return relu4(v);
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc);
    acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    return acc;
}
// 2 * [v > 0] per component  (phi''(z) for phi = relu^2, evaluated on d = 2 relu(z))
__device__ __forceinline__ f32x4 step2(f32x4 d) {
    f32x4 o;
    o[0] = d[0] > 0.f ? 2.f : 0.f; o[1] = d[1] > 0.f ? 2.f : 0.f;
    o[2] = d[2] > 0.f ? 2.f : 0.f; o[3] = d[3] > 0.f ? 2.f : 0.f;
    return o;
}

// =======================================================================================
// Forward kernel
// =======================================================================================
// ---- bf16-pair images (GGeo::q*): pack / unpack
__device__ __forceinline__ float pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float bf_lo(float packed) { return __uint_as_float(__float_as_uint(packed) << 16); }
__device__ __forceinline__ float bf_hi(float packed) { return __uint_as_float(__float_as_uint(packed) & 0xffff0000u); }
// T-layout read of hidden block m of an image: `base` = block pointer + lane; fp32 image at float offset o32, pair image at o16
template <bool P16>
__device__ __forceinline__ f32x4 image_get_T(const float* base, int o32, int o16, int m) {
    f32x4 v;
#if defined(PSP_GEN_ABLATE) && (PSP_GEN_ABLATE & 2)
    v[0] = v[1] = v[2] = v[3] = 1.f;                        // diagnostic build: producers skip their path-store loads
    return v;
#endif
    if constexpr (P16) {
        const float u0 = base[o16 + (2 * m) * 64], u1 = base[o16 + (2 * m + 1) * 64];
        v[0] = bf_lo(u0); v[2] = bf_hi(u0); v[1] = bf_lo(u1); v[3] = bf_hi(u1);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = base[o32 + (4 * m + r) * 64];
    }
    return v;
}

// PHILOX: on-device noise decided at launch -- no supplied-noise loads (and no join behind them) in the time loop
// X3 (psp_gen_config.mlp_dtype = PSP_MLP_F16X3): the nine value-net products per step as split f16 products, fp32-grade
// SPECK >= 0 (round 4): the problem switches as compile-time constants -- SPECK = drift_kind | h_kind << 4 on an UNBOUNDED domain,
// non-adaptive forward process, path store on, no per-step value output: the training launch of the parabolic notebooks' problems
// (DoubleWell_multidim_for_general_solver 'HJB': DRIFT_DWELL | GH_QUAD << 4, BASELINE configs[2]; AllenCahn: DRIFT_ZERO |
// GH_ALLEN_CAHN << 4).  The general instance tests every one of them per step and per state block: scalar branches that cut the
// time loop into basic blocks (hjb_kernels.h, hjb_fwd_kernel FAST_ = 2).
template <int D, int H, bool BF16 = false, bool PHILOX = false, bool X3 = false, int SPECK = -1>
__global__ __launch_bounds__(512) void gen_fwd_kernel(const GenArgs a_) {
    PSP_COND_EXIT(a_);
    constexpr bool SPEC = SPECK >= 0;
    // a by-value copy of the arguments whose switches are constants in a specialised instance
    GenArgs a = a_;
    if constexpr (SPEC) {
        a.drift_kind = SPECK & 15; a.h_kind = (SPECK >> 4) & 15; a.domain_kind = DOM_NONE; a.adaptive = 0; a.store_path = 1;
        a.Vsteps = nullptr; a.Ysteps = nullptr;
    }
    using G = GGeo<D, H>;
    constexpr int DI = G::DI, DBI = G::DBI, KSI = G::KSI, HB = G::HB, KSH = G::KSH;
    constexpr int MODE = X3 ? 2 : (BF16 ? 1 : 0);
    constexpr int oW1f = X3 ? G::xW1f : G::fW1f, oW2xf = X3 ? G::xW2xf : G::fW2xf, oW2hf = X3 ? G::xW2hf : G::fW2hf,
                  oW2hr = X3 ? G::xW2hr : G::fW2hr, oW2xr = X3 ? G::xW2xr : G::fW2xr, oW1r = X3 ? G::xW1r : G::fW1r;
    constexpr int VSH = X3 ? G::xVec - G::fVec : 0;    // the vectors and the reduction slot follow the (larger) split tables
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;

    // forward tables: out^T = W^T in^T  ->  A[row = out][k = in] = W[in][out]
    gen_stage<MODE, KSI, DBI>(lds + oW1f, HB, tid, nthr, [&](int row, int col) {
        return (row < H && col < DI) ? P[G::oW1 + col * H + row] : 0.f; });
    gen_stage<MODE, KSI, DBI>(lds + oW2xf, HB, tid, nthr, [&](int row, int col) {
        return (row < H && col < DI) ? P[G::oW2 + col * H + row] : 0.f; });
    gen_stage<MODE, KSH, HB>(lds + oW2hf, HB, tid, nthr, [&](int row, int col) {
        return (row < H && col < H) ? P[G::oW2 + (DI + col) * H + row] : 0.f; });
    // reverse tables: g_in = W g_out  ->  A[row = in][k = out] = W[in][out]
    gen_stage<MODE, KSH, HB>(lds + oW2hr, HB, tid, nthr, [&](int row, int col) {
        return (row < H && col < H) ? P[G::oW2 + (DI + row) * H + col] : 0.f; });
    gen_stage<MODE, KSH, HB>(lds + oW2xr, DBI, tid, nthr, [&](int row, int col) {
        return (row < DI && col < H) ? P[G::oW2 + row * H + col] : 0.f; });
    gen_stage<MODE, KSH, HB>(lds + oW1r, DBI, tid, nthr, [&](int row, int col) {
        return (row < DI && col < H) ? P[G::oW1 + row * H + col] : 0.f; });
    stage_vec(lds + VSH + G::vb1, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob1 + f] : 0.f; });
    stage_vec(lds + VSH + G::vb2, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob2 + f] : 0.f; });
    stage_vec(lds + VSH + G::vw3h1, HB, tid, nthr, [&](int f) { return f < H ? P[G::oW3 + DI + f] : 0.f; });
    stage_vec(lds + VSH + G::vw3h2, HB, tid, nthr, [&](int f) { return f < H ? P[G::oW3 + DI + H + f] : 0.f; });
    stage_vec(lds + VSH + G::vw3x, DBI, tid, nthr, [&](int f) { return f < DI ? P[G::oW3 + f] : 0.f; });
    stage_vec(lds + VSH + G::vdr, DBI, tid, nthr, [&](int f) {
        return (f < D && (a.drift_kind == DRIFT_DWELL || a.drift_kind == DRIFT_DIAG)) ? a.drift[f] : 0.f; });
    __syncthreads();
    const float b3 = P[G::ob3];

    const int t16 = blockIdx.x * nwave + wave;
    const bool wave_valid = t16 < a.ntile16;
    const int k = t16 * 16 + j;
    const bool kvalid = wave_valid && k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    const float dt = a.dt, sqdt = a.sqdt, sig = a.sigma_scale, Tend = a.T;
    unsigned long long nact = 0;

    if (wave_valid) {
        const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds + VSH + G::fVec) + q;
        // feature masks of this lane: f < D (state features) ; f == D is the time input
        f32x4 X[DBI];
#pragma unroll
        for (int b = 0; b < DBI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * b + 4 * r + q;
                const float v = a.x0[(size_t)(kvalid ? k : 0) * D + (f < D ? f : D - 1)];
                X[b][r] = (f < D && kvalid) ? v : 0.f;
            }
        float t = kvalid ? a.t0[k] : 0.f;
        bool stopped = !kvalid;
        float Y = 0.f;
        int msteps = 0;                                             // active steps of this trajectory (exact)

        // one network evaluation at the current (X, t): fills r1, r2 (relu(z)), returns V
        auto net_value = [&](const f32x4* vecs, f32x4 (&r1)[HB], f32x4 (&r2)[HB]) {
            const f32x4* vb1 = vecs + (G::vb1 - G::fVec) / 4;
            const f32x4* vb2 = vecs + (G::vb2 - G::fVec) / 4;
            const f32x4* vw3h1 = vecs + (G::vw3h1 - G::fVec) / 4;
            const f32x4* vw3h2 = vecs + (G::vw3h2 - G::fVec) / 4;
            const f32x4* vw3x = vecs + (G::vw3x - G::fVec) / 4;
#pragma unroll
            for (int m = 0; m < HB; ++m) r1[m] = vb1[m * 4];
            gen_gemm<MODE, HB, KSI, DBI>(r1, lds + oW1f, X, lane);
#pragma unroll
            for (int m = 0; m < HB; ++m) r1[m] = relu4m<MODE>(r1[m]);
            f32x4 h1[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) h1[m] = r1[m] * r1[m];
#pragma unroll
            for (int m = 0; m < HB; ++m) r2[m] = vb2[m * 4];
            gen_gemm<MODE, HB, KSI, DBI>(r2, lds + oW2xf, X, lane);
            gen_gemm<MODE, HB, KSH, HB>(r2, lds + oW2hf, h1, lane);
#pragma unroll
            for (int m = 0; m < HB; ++m) r2[m] = relu4m<MODE>(r2[m]);
            float v = 0.f;
#pragma unroll
            for (int b = 0; b < DBI; ++b) v = dot4(vw3x[b * 4], X[b], v);
#pragma unroll
            for (int m = 0; m < HB; ++m) { v = dot4(vw3h1[m * 4], h1[m], v); v = dot4(vw3h2[m * 4], r2[m] * r2[m], v); }
            return qsum(v) + b3;
        };
        auto put_time = [&](float tv) {
            if (q == G::TQ) X[G::TB][G::TR] = tv;
        };
        put_time(t);

        for (int n = 0; n < a.N; ++n) {
            const f32x4* vecs = opaque(vecs0);
            const f32x4* vw3h1 = vecs + (G::vw3h1 - G::fVec) / 4;
            const f32x4* vw3h2 = vecs + (G::vw3h2 - G::fVec) / 4;
            const f32x4* vw3x = vecs + (G::vw3x - G::fVec) / 4;
            const f32x4* vdr = vecs + (G::vdr - G::fVec) / 4;
            // path block of (n, tile): wave-uniform base forced into SGPRs, stores are "base + lane * 4 + immediate < 4 KiB"
            // (distinct large offsets otherwise become hoisted / spilled 64-bit address registers, and a spill reload
            // waits with vmcnt(0) behind the store in front of it)
            typedef __attribute__((address_space(1))) float* gwptr_t;
            auto pbase = [&](int slot, int ofs) __attribute__((always_inline)) {
                return (gwptr_t)sgpr_block_addr(a.path, (unsigned long long)slot * a.ntile16 + t16, (unsigned)G::PB, (unsigned)ofs);
            };
            auto pbase16 = [&](int slot, int ofs) __attribute__((always_inline)) {      // bf16-pair path block (GGeo::q*)
                return (gwptr_t)sgpr_block_addr(a.path, (unsigned long long)slot * a.ntile16 + t16, (unsigned)G::PB16, (unsigned)ofs);
            };
            const bool p16 = BF16 && a.path16;                    // wave-uniform
            const unsigned ul = (unsigned)lane;
            // ---- V(X,t) and the activations (solver.py:1100)
            f32x4 r1[HB], r2[HB];
            const float Vnow = net_value(vecs, r1, r2);
            if (n == 0) Y = Vnow;                                   // solver.py:1081
            if (a.Vsteps && q == 0) {                               // Solver's value_function ansatz: sum_n (Y_n(X_n) - Y)^2 (solver.py:438-440)
                a.Vsteps[(size_t)n * (a.ntile16 * 16) + k] = Vnow;
                a.Ysteps[(size_t)n * (a.ntile16 * 16) + k] = Y;
            }
            // ---- grad_x V by the reverse sweep (replaces autograd.grad of solver.py:1103)
            f32x4 gz2[HB], gz1[HB], gx[DBI];
#pragma unroll
            for (int m = 0; m < HB; ++m) gz2[m] = vw3h2[m * 4] * (2.0f * r2[m]);
            // the d2 = 2 relu(z2) image leaves for the path store HERE: nothing below reads r2 again, and holding its 4 HB registers
            // across the reverse sweep, the Euler step and the tangent products is what pushed this kernel over its 256 registers
            // (26 spilled dwords inside the time loop of the split-product instance, reloaded behind the path stores)
            if (a.store_path) {
                if (p16) {
                    gwptr_t p2 = pbase16(n, G::qD2);
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp)
                            p2[(unsigned)(2 * m + rp) * 64 + ul] = pk_bf16(2.0f * r2[m][rp], 2.0f * r2[m][rp + 2]);
                } else {
                    gwptr_t p2 = pbase(n, G::pD2);
#pragma unroll
                    for (int ks = 0; ks < 4 * HB; ++ks) p2[ks * 64 + ul] = 2.0f * r2[ks >> 2][ks & 3];
                }
            }
#pragma unroll
            for (int m = 0; m < HB; ++m) gz1[m] = vw3h1[m * 4];
            gen_gemm<MODE, HB, KSH, HB>(gz1, lds + oW2hr, gz2, lane);     // g_h1 = w3h1 + W2h g_z2
#pragma unroll
            for (int m = 0; m < HB; ++m) gz1[m] = gz1[m] * (2.0f * r1[m]);
#pragma unroll
            for (int b = 0; b < DBI; ++b) gx[b] = vw3x[b * 4];
            gen_gemm<MODE, DBI, KSH, HB>(gx, lds + oW2xr, gz2, lane);
            gen_gemm<MODE, DBI, KSH, HB>(gx, lds + oW1r, gz1, lane);
            // Z = sigma^T grad_x V (sigma = s I), state features only (solver.py:1104)
            const float alivef = stopped ? 0.f : 1.f;
            auto noise_block = [&](int b) __attribute__((always_inline)) {
                f32x4 xi;
                if (PHILOX || a.noise_mode == NOISE_PHILOX) {
                    // (trajectory id and block index made opaque per call: the first Philox round multiplies them by constants,
                    //  which is invariant over the time loop -- hoisted, that was three registers per state block, spilled
                    //  under the 256-register cap and reloaded per block behind an s_waitcnt vmcnt(0) that also waits for
                    //  every path store in flight)
                    xi = philox_block((uint32_t)opaque_i((int)kglob), (uint32_t)n, (uint32_t)(4 * b + opaque_i(q)), a.iter,
                                      a.seed_lo, a.seed_hi);
                } else {
                    const float* xrow = a.xi + ((size_t)n * a.K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * b + 4 * r + q;
                        xi[r] = xrow[f < D ? f : D - 1];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) xi[r] = ((16 * b + 4 * r + q) < D && kvalid) ? xi[r] : 0.f;
                return xi;
            };
            auto z_block = [&](int b) __attribute__((always_inline)) {
                f32x4 Z;
#pragma unroll
                for (int r = 0; r < 4; ++r) Z[r] = ((16 * b + 4 * r + q) < D) ? sig * gx[b][r] : 0.f;
                return Z;
            };
            // (X_proposal - X) of solver.py:1116-1117: ((b(X) + sigma c) dt + sigma xi sqrt(dt)) * alive,  c = -Z (adaptive) or 0
            auto move_block = [&](int b, const f32x4& Z, const f32x4& xi) __attribute__((always_inline)) {
                const f32x4 cdt = a.adaptive ? (-dt) * Z : 0.f * Z;
                f32x4 drift = 0.f * Z;
                if (a.drift_kind == DRIFT_DWELL) drift = -(4.0f * vdr[b * 4] * (X[b] * (X[b] * X[b] - 1.0f)));
                else if (a.drift_kind == DRIFT_DIAG) drift = vdr[b * 4] * X[b];          // b(x) = diag(a) x (problems.py:36-37 with a diagonal A)
                const f32x4 step = (drift * dt + sig * cdt + (sig * sqdt) * xi) * alivef;
                return step;
            };
            // ---- exit test of a bounded domain (solver.py:1119-1129; EllipticSolver :758-767) and |x|^2 for the x-dependent h
            float rr = 0.f;
            if (a.domain_kind == DOM_SPHERE || a.domain_kind == DOM_ANNULUS || a.h_kind >= GH_EXPBALL_LIN) {
#pragma unroll
                for (int b = 0; b < DBI; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((16 * b + 4 * r + q) < a.d_real) rr = fmaf(X[b][r], X[b][r], rr);
                rr = qsum(rr);
            }
            bool inside = true;
            if (a.domain_kind == DOM_SPHERE) {
                inside = sqrtf(rr) < a.dom_a;                        // the state BEFORE the move (:1121)
            } else if (a.domain_kind == DOM_ANNULUS) {
                const float rad = sqrtf(rr);                         // :1122-1123 / :752-753, the state before the move
                inside = rad > a.dom_a && rad < a.dom_b;
            } else if (a.domain_kind >= DOM_BOX) {
                // the boxes test the PROPOSAL (:1126-1129): one extra pass over the blocks (noise regenerated: the
                // move itself needs the verdict, and keeping both images would cost the unbounded case registers)
                float n_out = 0.f, n_le = 0.f;
#pragma unroll
                for (int b = 0; b < DBI; ++b) {
                    const f32x4 xi = noise_block(b);
                    const f32x4 Xp = X[b] + move_block(b, z_block(b), xi);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if ((16 * b + 4 * r + q) < a.d_real) {
                            const bool lo_ok = a.domain_kind != DOM_BOX || Xp[r] >= a.dom_a, hi_ok = Xp[r] <= a.dom_b;
                            n_out += (lo_ok && hi_ok) ? 0.f : 1.f;
                            n_le += hi_ok ? 1.f : 0.f;
                        }
                    }
                }
                n_out = qsum(n_out); n_le = qsum(n_le);
                inside = a.domain_kind == DOM_BOX_UPPER_ANY ? n_le > 0.f : n_out == 0.f;
            }
            const bool in_time = inside && (t + dt) <= Tend;         // new_selection of solver.py:1119-1131 (fp32)
            const bool act = in_time && !stopped;
            const float actf = act ? 1.f : 0.f;
            float S = 0.f, Pz = 0.f;
            f32x4 U[DBI];                                            // s * u^ : tangent direction in x-space
#pragma unroll
            for (int b = 0; b < DBI; ++b) {
                const f32x4 xi = noise_block(b);
                const f32x4 Z = z_block(b);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S = fmaf(Z[r], Z[r], S);
                    Pz = fmaf(Z[r], xi[r], Pz);
                }
                // c = -Z (adaptive) or 0, detached (solver.py:1110-1114)
                const f32x4 cdt = a.adaptive ? (-dt) * Z : 0.f * Z;
                // tangent direction u^ = act ((-h_z + c) dt + xi sqrt(dt)),  -h_z = Z for h = -|z|^2/2
                f32x4 u = sqdt * xi + cdt;
                if (a.h_kind == GH_QUAD) u += dt * Z;
                U[b] = (actf * sig) * u;
                const f32x4 step = move_block(b, Z, xi);
                f32x4 Xn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool fx = (16 * b + 4 * r + q) < D;
                    Xn[r] = (fx && act) ? X[b][r] + step[r] : X[b][r];
                }
                // keep the OLD X in the path store (the sample point), move afterwards
                if (a.store_path) {
                    if (p16) {
                        gwptr_t px = pbase16(n, G::qX + b * 128), pu = pbase16(n, G::qU + b * 128);
                        px[ul] = pk_bf16(X[b][0], X[b][2]); px[64 + ul] = pk_bf16(X[b][1], X[b][3]);
                        pu[ul] = pk_bf16(U[b][0], U[b][2]); pu[64 + ul] = pk_bf16(U[b][1], U[b][3]);
                    } else {
                        gwptr_t px = pbase(n, G::pX + b * 256), pu = pbase(n, G::pU + b * 256);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            px[r * 64 + ul] = X[b][r];
                            pu[r * 64 + ul] = U[b][r];
                        }
                    }
                }
                X[b] = Xn;
            }
            S = qsum(S); Pz = qsum(Pz);
            // ---- Y update (solver.py:1141-1142): h sees V(X,t) (not the running Y)
            float minus_h = 0.f, hy = 0.f;
            if (a.h_kind == GH_QUAD) minus_h = 0.5f * S;
            else if (a.h_kind == GH_ALLEN_CAHN) { minus_h = -(Vnow - Vnow * Vnow * Vnow); hy = 1.0f - 3.0f * Vnow * Vnow; }
            else if (a.h_kind >= GH_EXPBALL_LIN) {
                // h = -2 al y (2 al |x|^2 + d) - e y + {0, E - y^2, sin(E - y^2)},  E = exp(2 al |x|^2 + 2 t), t = n dt
                // (problems.py:985, :1022, :1058, :1130, :1166; h sees y = V(X,t), solver.py:1141)
                const float al = a.h_par[0];
                const float lin = 2.0f * al * (2.0f * al * rr + a.h_par[1]) + a.h_par[2];
                float nl = 0.f, nly = 0.f;
                if (a.h_kind != GH_EXPBALL_LIN) {
                    const float arg = expf(2.0f * al * rr + 2.0f * a.h_par[3] * ((float)n * dt)) - Vnow * Vnow;
                    if (a.h_kind == GH_EXPBALL_SQ) { nl = arg; nly = -2.0f * Vnow; }
                    else { nl = sinf(arg); nly = -2.0f * Vnow * cosf(arg); }
                }
                minus_h = Vnow * lin - nl;
                hy = nly - lin;
            }
            const float zc = a.adaptive ? -S : 0.f;
            Y = Y + ((minus_h + zc) * dt + Pz * sqdt) * actf;
            if (a.store_path) {
                // loss-weight-independent part of the tangent pass: z1^ = W1^T U, z2^ = W2x^T U + W2h^T (d1 z1^)
                f32x4 z1h[HB], z2h[HB];
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < HB; ++m) { z1h[m] = zero4; z2h[m] = zero4; }
                gen_gemm<MODE, HB, KSI, DBI>(z1h, lds + oW1f, U, lane);
                f32x4 h1d[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) h1d[m] = (2.0f * r1[m]) * z1h[m];
                // d1 = 2 relu(z1) and z1^ leave as soon as h1' is formed: r1 and z1^ are dead during the two z2^ products
                if (p16) {
                    gwptr_t p1 = pbase16(n, G::qD1), p3 = pbase16(n, G::qZ1);
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp) {
                            const unsigned o = (unsigned)(2 * m + rp) * 64 + ul;
                            p1[o] = pk_bf16(2.0f * r1[m][rp], 2.0f * r1[m][rp + 2]);
                            p3[o] = pk_bf16(z1h[m][rp], z1h[m][rp + 2]);
                        }
                } else {
                    gwptr_t p1 = pbase(n, G::pD1), p3 = pbase(n, G::pZ1);
#pragma unroll
                    for (int ks = 0; ks < 4 * HB; ++ks) {
                        p1[ks * 64 + ul] = 2.0f * r1[ks >> 2][ks & 3];
                        p3[ks * 64 + ul] = z1h[ks >> 2][ks & 3];
                    }
                }
                gen_gemm<MODE, HB, KSI, DBI>(z2h, lds + oW2xf, U, lane);
                gen_gemm<MODE, HB, KSH, HB>(z2h, lds + oW2hf, h1d, lane);
                if (p16) {
                    gwptr_t p4 = pbase16(n, G::qZ2);
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp) p4[(unsigned)(2 * m + rp) * 64 + ul] = pk_bf16(z2h[m][rp], z2h[m][rp + 2]);
                } else {
                    gwptr_t p4 = pbase(n, G::pZ2);
#pragma unroll
                    for (int ks = 0; ks < 4 * HB; ++ks) p4[ks * 64 + ul] = z2h[ks >> 2][ks & 3];
                }
                // coefficient of grad_theta V at this sample: -h_y dt act, plus 1 for V(X_0,t_0) at n = 0
                if (q == 0) a.ahat[(size_t)n * (a.ntile16 * 16) + k] = (n == 0 ? 1.f : 0.f) - hy * dt * actf;
            }
            // ---- time / stop bookkeeping (solver.py:1148-1155)
            t = t + dt * actf;
            put_time(t);
            if (act && q == 0) ++nact;
            msteps += act ? 1 : 0;
            stopped = stopped || !in_time;
        }
        // ---- final point: V(X_N, t_N) (solver.py:1163) as an extra value-only sample
        {
            const f32x4* vecs = opaque(vecs0);
            f32x4 r1[HB], r2[HB];
            const float VN = net_value(vecs, r1, r2);
            if (a.store_path && BF16 && a.path16) {
                float* pblk = a.path + ((size_t)a.N * a.ntile16 + t16) * (size_t)G::PB16 + lane;
#pragma unroll
                for (int b = 0; b < DBI; ++b)
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        pblk[G::qX + (2 * b + rp) * 64] = pk_bf16(X[b][rp], X[b][rp + 2]);
                        pblk[G::qU + (2 * b + rp) * 64] = 0.f;
                    }
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        pblk[G::qD1 + (2 * m + rp) * 64] = pk_bf16(2.0f * r1[m][rp], 2.0f * r1[m][rp + 2]);
                        pblk[G::qD2 + (2 * m + rp) * 64] = pk_bf16(2.0f * r2[m][rp], 2.0f * r2[m][rp + 2]);
                        pblk[G::qZ1 + (2 * m + rp) * 64] = 0.f;
                        pblk[G::qZ2 + (2 * m + rp) * 64] = 0.f;
                    }
                if (q == 0) a.ahat[(size_t)a.N * (a.ntile16 * 16) + k] = 1.f;
            } else if (a.store_path) {
                float* pblk = a.path + ((size_t)a.N * a.ntile16 + t16) * (size_t)G::PB + lane;
#pragma unroll
                for (int ks = 0; ks < 4 * DBI; ++ks) { pblk[G::pX + ks * 64] = X[ks >> 2][ks & 3]; pblk[G::pU + ks * 64] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4 * HB; ++ks) {
                    pblk[G::pD1 + ks * 64] = 2.0f * r1[ks >> 2][ks & 3];
                    pblk[G::pD2 + ks * 64] = 2.0f * r2[ks >> 2][ks & 3];
                    pblk[G::pZ1 + ks * 64] = 0.f;
                    pblk[G::pZ2 + ks * 64] = 0.f;
                }
                if (q == 0) a.ahat[(size_t)a.N * (a.ntile16 * 16) + k] = 1.f;
            }
            // (EllipticSolver: T = +inf and no time input; t_N then only counts the active steps -- as m dt with ONE rounding, so that
            //  round(t_N / dt) is exact however long the trajectory ran: the committor notebook runs N = 5000)
            if (kvalid && q == 0) { a.VN[k] = VN; a.YN[k] = Y; a.tN[k] = (Tend == __builtin_inff()) ? (float)msteps * dt : t; }
            if (kvalid) {
#pragma unroll
                for (int b = 0; b < DBI; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * b + 4 * r + q;
                        if (f < D) a.XN[(size_t)k * D + f] = X[b][r];
                    }
            }
        }
    }
    // active-step count (integer, order independent)
    for (int off = 1; off < 64; off <<= 1) nact += __shfl_xor(nact, off);
    if (lane == 0 && nact) atomicAdd(a.kcount, nact);
}

// =======================================================================================
// Backward kernel: grad_theta sum_samples [ a V + w (s u^) . grad_x V ]
//   forward tangent (stored, times w):  z1' = w z1^,  h1' = d1 z1',  z2' = w z2^,  h2' = d2 z2'
//   adjoints:  gz2' = w3h2 d2                      (adjoint of the tangent pre-activation z2')
//              gz2  = a gz2' + w3h2 phi''(z2) z2'
//              gh1' = w3h1 + W2h gz2' ; gh1 = a w3h1 + W2h gz2
//              gz1' = gh1' d1 ;  gz1 = gh1 d1 + gh1' phi''(z1) z1'
//   gradients: dW2 = [x0,h1]^T gz2 + [x0',h1']^T gz2' ; dW1 = x0^T gz1 + x0'^T gz1' ;
//              dW3 = a [x0,h1,h2] + [x0',h1',h2'] ; db2 = gz2 ; db1 = gz1 ; db3 = a
// Workgroup = 4 waves, rounds of 4 sample blocks, weight-gradient tiles split over the waves as in
// hjb_bwd_kernel; the A operands (x0, x0', h1, h1') come straight from the stored images in
// feature-on-lane form, the B operands (adjoints) through per-wave LDS exchange tiles.
// =======================================================================================
template <int D, int H>
__global__ __launch_bounds__(256, 2) void gen_bwd_kernel(const GenArgs a) {
    PSP_COND_EXIT(a);
    using G = GGeo<D, H>;
    constexpr int DI = G::DI, DBI = G::DBI, HB = G::HB, KSH = G::KSH;
    constexpr int WH = G::WH, WD = G::WD, NIB = G::NIB, NRX = G::NRX, NRH = G::NRH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int wh = (WH == 1) ? 0 : wave % WH, wd = (WD == 1) ? 0 : wave / WH;
    const int lofsF = image_lane_offset_F(lane);
    const int qq = lane >> 4, col = lane & 15;
    const float* __restrict__ P = a.params;

    stage_aop(lds + G::gW2hr, HB, KSH, tid, nthr, [&](int row, int c2) {
        return (row < H && c2 < H) ? P[G::oW2 + (DI + row) * H + c2] : 0.f; });
    __syncthreads();
    // w3 (h1 / h2 parts) in T layout, straight from the parameter vector (32 L1-resident loads per round)
    auto w3_T = [&](int base, int m, int o0) {          // o0 = opaque zero: keeps the loads inside the round loop
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int f = 16 * m + 4 * r + q; v[r] = f < H ? P[base + o0 + (f < H ? f : 0)] : 0.f; }
        return v;
    };
    float* exch = lds + G::gEx;
    float* my_ex = exch + wave * (G::EXT * 256);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // weight-gradient accumulators of this wave (rows x columns): dW2 x-rows, dW2 h-rows, dW1
    f32x4 acc2x[NRX][NIB], acc2h[NRH][NIB], acc1[NRX][NIB];
    float bs2[NIB], bs1[NIB], g3x[NRX], g3h1[NRH];
    f32x4 g3h2T[HB];                                    // dW3 (h2 part) in T layout, reduced over lanes at the end
    float g3b = 0.f;
#pragma unroll
    for (int s = 0; s < NRX; ++s) { g3x[s] = 0.f;
#pragma unroll
        for (int t = 0; t < NIB; ++t) { acc2x[s][t] = zero4; acc1[s][t] = zero4; } }
#pragma unroll
    for (int s = 0; s < NRH; ++s) { g3h1[s] = 0.f;
#pragma unroll
        for (int t = 0; t < NIB; ++t) acc2h[s][t] = zero4; }
#pragma unroll
    for (int t = 0; t < NIB; ++t) { bs2[t] = 0.f; bs1[t] = 0.f; }
#pragma unroll
    for (int m = 0; m < HB; ++m) g3h2T[m] = zero4;

    int cbc[NIB], rxc[NRX], rhc[NRH];
#pragma unroll
    for (int t = 0; t < NIB; ++t) cbc[t] = ((wh + WH * t) < HB ? (wh + WH * t) : HB - 1) * 256;
#pragma unroll
    for (int s = 0; s < NRX; ++s) rxc[s] = ((wd + WD * s) < DBI ? (wd + WD * s) : DBI - 1) * 256;
#pragma unroll
    for (int s = 0; s < NRH; ++s) rhc[s] = ((wd + WD * s) < HB ? (wd + WD * s) : HB - 1) * 256;

    const int Kpad = a.ntile16 * 16;
    const long long nblk = (long long)(a.N + 1) * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    for (long long round = blockIdx.x; round < nround; round += gridDim.x) {
        const long long rb = round * 4;
        // L2 touch-prefetch (one dword per 128-B line): wave w pulls block w's X / U / d1 / z1^ images
        // (read in P2/P3 by every wave) and its own block of the NEXT round (d1, d2, z1^, z2^ for P1).
        float touch[6];
        {
            const long long xb0 = rb + wave;
            const float* xt = a.path + (size_t)(xb0 < nblk ? xb0 : nblk - 1) * (size_t)G::PB;
            constexpr int NA = G::pD2;                       // X, U, d1 images are contiguous: [0, pD2)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int o = lane * 32 + i * 2048; touch[i] = xt[o < NA ? o : 0]; }
            touch[4] = xt[G::pZ1 + ((lane * 32 < 4 * HB * 64) ? lane * 32 : 0)];
            const long long nb0 = (round + gridDim.x) * 4 + wave;
            const float* nt = a.path + (size_t)(nb0 < nblk ? nb0 : nblk - 1) * (size_t)G::PB + G::pD1;
            touch[5] = nt[(lane * 32 < 16 * HB * 64) ? lane * 32 : 0];
        }
        // ------------------------------------------------------------------ P1: adjoints of the own block
        {
            f32x4 gz1[HB], gz1t[HB];
            const long long blk0 = rb + wave;
            const bool bvalid = blk0 < nblk;
            const long long blk = bvalid ? blk0 : nblk - 1;
            const int n = (int)(blk / a.ntile16), t16 = (int)(blk % a.ntile16);
            const int k = t16 * 16 + j;
            const bool kvalid = bvalid && k < a.K_local;
            const float* pb = a.path + (size_t)blk * (size_t)G::PB + lane;
            const bool fin = (n == a.N);
            // wY / wV / ahat are zero-padded to 16*ntile16 entries: plain loads, no branch around them
            const size_t wofs = a.per_sample ? (size_t)n * Kpad : 0;            // per-sample mode: wY is (N+1, Kpad), ahat IS the coefficient
            const float wy = a.wY[wofs + k], wv = a.per_sample ? 0.f : a.wV[k], ah = a.ahat[(size_t)n * Kpad + k];
            const float wsv = (bvalid && !fin) ? wy : 0.f;                             // weight of the tangent part
            const float av = bvalid ? (a.per_sample ? ah : (fin ? wv : wy * ah)) : 0.f;
            const int o0 = opaque_i(0);
            f32x4 gz2[HB], gz2t[HB];
            {
                f32x4 d2[HB], z2t[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d2[m][r] = pb[G::pD2 + (4 * m + r) * 64];
                        z2t[m][r] = wsv * pb[G::pZ2 + (4 * m + r) * 64];
                    }
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    const f32x4 w3h2 = w3_T(G::oW3 + DI + H, m, o0);
                    gz2t[m] = w3h2 * d2[m];
                    gz2[m] = av * gz2t[m] + w3h2 * step2(d2[m]) * z2t[m];
                    // dW3 (h2 part): a h2 + h2'   with h2 = (d2/2)^2, h2' = d2 z2'
                    g3h2T[m] += av * (0.25f * d2[m] * d2[m]) + d2[m] * z2t[m];
                }
            }
            g3b += (q == 0) ? av : 0.f;
#pragma unroll
            for (int m = 0; m < HB; ++m) { gz1t[m] = w3_T(G::oW3 + DI, m, o0); gz1[m] = av * gz1t[m]; }
            gemm_T<HB, KSH, HB>(gz1t, lds + G::gW2hr, gz2t, lane);      // gh1' = w3h1 + W2h gz2'
            gemm_T<HB, KSH, HB>(gz1, lds + G::gW2hr, gz2, lane);        // gh1  = a w3h1 + W2h gz2
            {   // d1 / z1^ are L2-resident (touch-prefetched one round ahead): fetch them only now
                f32x4 d1[HB], z1t[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d1[m][r] = pb[G::pD1 + (4 * m + r) * 64];
                        z1t[m][r] = wsv * pb[G::pZ1 + (4 * m + r) * 64];
                    }
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    gz1[m] = gz1[m] * d1[m] + gz1t[m] * step2(d1[m]) * z1t[m];
                    gz1t[m] = gz1t[m] * d1[m];
                }
            }
            if (!kvalid) {
#pragma unroll
                for (int m = 0; m < HB; ++m) { gz2[m] = zero4; gz2t[m] = zero4; gz1[m] = zero4; gz1t[m] = zero4; }
            }
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                tile_put(my_ex + m * 256, gz2[m], lane);
                tile_put(my_ex + (HB + m) * 256, gz2t[m], lane);
                tile_put(my_ex + (2 * HB + m) * 256, gz1[m], lane);
                tile_put(my_ex + (3 * HB + m) * 256, gz1t[m], lane);
            }
        }
        __syncthreads();
        // ------------------------------------------------------------------ P2: all weight gradients
        // One loop over the 4 sample blocks feeds dW2 (x- and h-rows) and dW1 from a single load of the
        // X / U images per block.  Two-stage operand pipeline: while the x-row MFMAs of block sb issue,
        // its h-row images (d1, z1^) are in flight; while the h-row MFMAs issue, X / U of block sb+1 are.
        {
            auto blk_of = [&](int sb) {                            // provably wave-uniform 32-bit block index
                const long long c0 = rb + sb;
                return (size_t)__builtin_amdgcn_readfirstlane((int)(c0 < nblk ? c0 : nblk - 1));
            };
            f32x4 xbuf[NRX], ubuf[NRX], dbuf[NRH], zbuf[NRH];
            {
                const float* sp = a.path + (size_t)blk_of(0) * (size_t)G::PB;
#pragma unroll
                for (int s2 = 0; s2 < NRX; ++s2) {
                    xbuf[s2] = image_get_F(sp + G::pX + rxc[s2], lofsF);
                    ubuf[s2] = image_get_F(sp + G::pU + rxc[s2], lofsF);
                }
            }
#pragma unroll 1
            for (int sb = 0; sb < 4; ++sb) {
                const bool sval = (rb + sb) < nblk;
                const int cb = (int)blk_of(sb);
                const int n = cb / a.ntile16, t16 = cb % a.ntile16;
                const float* sp = a.path + (size_t)cb * (size_t)G::PB;
                const float* ex = exch + sb * (G::EXT * 256);
                // stage A loads: h-row images of this block
#pragma unroll
                for (int s2 = 0; s2 < NRH; ++s2) {
                    dbuf[s2] = image_get_F(sp + G::pD1 + rhc[s2], lofsF);
                    zbuf[s2] = image_get_F(sp + G::pZ1 + rhc[s2], lofsF);
                }
                // per-sample weights of the 4 samples this lane sees in F layout (samples 4q'..4q'+3)
                const bool fin = (n == a.N);
                const int k4 = t16 * 16 + 4 * qq;
                const f32x4 wy4 = *reinterpret_cast<const f32x4*>(a.wY + (a.per_sample ? (size_t)n * Kpad : 0) + k4);
                const f32x4 wv4 = a.per_sample ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.wV + k4);
                const f32x4 ah4 = *reinterpret_cast<const f32x4*>(a.ahat + (size_t)n * Kpad + k4);
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 w4 = (sval && !fin) ? wy4 : z4;
                const f32x4 a4 = sval ? (a.per_sample ? ah4 : (fin ? wv4 : wy4 * ah4)) : z4;
                f32x4 bz2[NIB], bz2t[NIB], bz1[NIB], bz1t[NIB];
#pragma unroll
                for (int t = 0; t < NIB; ++t) {
                    bz2[t] = tile_get(ex + cbc[t], lane);
                    bz2t[t] = tile_get(ex + HB * 256 + cbc[t], lane);
                    bz1[t] = tile_get(ex + 2 * HB * 256 + cbc[t], lane);
                    bz1t[t] = tile_get(ex + 3 * HB * 256 + cbc[t], lane);
                    bs2[t] += hsum4(bz2[t]);
                    bs1[t] += hsum4(bz1[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < NRX; ++s2) {                          // x-rows: A = x0, x0' = w U
                    const f32x4 x0 = xbuf[s2];
                    const f32x4 xt = w4 * ubuf[s2];
                    g3x[s2] += hsum4(a4 * x0 + xt);
#pragma unroll
                    for (int t = 0; t < NIB; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc2x[s2][t] = mfma16(x0[r], bz2[t][r], acc2x[s2][t]);
                            acc1[s2][t] = mfma16(x0[r], bz1[t][r], acc1[s2][t]);
                            acc2x[s2][t] = mfma16(xt[r], bz2t[t][r], acc2x[s2][t]);
                            acc1[s2][t] = mfma16(xt[r], bz1t[t][r], acc1[s2][t]);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // stage B loads: x-row images of the next block (xbuf / ubuf are free now)
                    const float* spn = a.path + (size_t)blk_of(sb < 3 ? sb + 1 : sb) * (size_t)G::PB;
#pragma unroll
                    for (int s2 = 0; s2 < NRX; ++s2) {
                        xbuf[s2] = image_get_F(spn + G::pX + rxc[s2], lofsF);
                        ubuf[s2] = image_get_F(spn + G::pU + rxc[s2], lofsF);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < NRH; ++s2) {                          // h-rows: A = h1 = (d1/2)^2, h1' = d1 w z1^
                    const f32x4 d1 = dbuf[s2];
                    const f32x4 h1 = 0.25f * d1 * d1;
                    const f32x4 ht = d1 * (w4 * zbuf[s2]);
                    g3h1[s2] += hsum4(a4 * h1 + ht);
#pragma unroll
                    for (int t = 0; t < NIB; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc2h[s2][t] = mfma16(h1[r], bz2[t][r], acc2h[s2][t]);
                            acc2h[s2][t] = mfma16(ht[r], bz2t[t][r], acc2h[s2][t]);
                        }
                }
            }
        }
        asm volatile("" :: "v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]), "v"(touch[4]), "v"(touch[5]));
        __syncthreads();
    }

    // ---- write-out: tile (rb, cb): lane (col, qq), reg rr <-> dW[16 rb + 4 qq + rr][16 cb + col]   (weights are (in, out))
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
#pragma unroll
    for (int s = 0; s < NRX; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) {
            const int rbk = wd + WD * s, cbk = wh + WH * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * rbk + 4 * qq + rr, jo = 16 * cbk + col;
                if (rbk < DBI && cbk < HB && i < DI && jo < H) {
                    gp[G::oW2 + i * H + jo] = acc2x[s][t][rr];
                    gp[G::oW1 + i * H + jo] = acc1[s][t][rr];
                }
            }
        }
#pragma unroll
    for (int s = 0; s < NRH; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) {
            const int rbk = wd + WD * s, cbk = wh + WH * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * rbk + 4 * qq + rr, jo = 16 * cbk + col;
                if (rbk < HB && cbk < HB && i < H && jo < H) gp[G::oW2 + (DI + i) * H + jo] = acc2h[s][t][rr];
            }
        }
    // biases: lane = output feature in F layout; sum over q', lanes q' == 0 of the wd == 0 waves write
#pragma unroll
    for (int t = 0; t < NIB; ++t) {
        const float v2 = qsum(bs2[t]), v1 = qsum(bs1[t]);
        const int f = 16 * (wh + WH * t) + col;
        if (wd == 0 && qq == 0 && (wh + WH * t) < HB && f < H) { gp[G::ob2 + f] = v2; gp[G::ob1 + f] = v1; }
    }
    // dW3: x and h1 parts live on lane = feature (F layout) in the wh == 0 waves; h2 part and b3 in T layout
#pragma unroll
    for (int s = 0; s < NRX; ++s) {
        const float v = qsum(g3x[s]);
        const int f = 16 * (wd + WD * s) + col;
        if (wh == 0 && qq == 0 && (wd + WD * s) < DBI && f < DI) gp[G::oW3 + f] = v;
    }
#pragma unroll
    for (int s = 0; s < NRH; ++s) {
        const float v = qsum(g3h1[s]);
        const int f = 16 * (wd + WD * s) + col;
        if (wh == 0 && qq == 0 && (wd + WD * s) < HB && f < H) gp[G::oW3 + DI + f] = v;
    }
    // T-layout partial sums: reduce over the 16 trajectory lanes, then over the 4 waves through LDS
    __syncthreads();
    float* red = lds;                                   // [4 waves][HB*16 + 1]
#pragma unroll
    for (int m = 0; m < HB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = g3h2T[m][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (j == 0) red[wave * (HB * 16 + 1) + 16 * m + 4 * r + q] = v;
        }
    {
        float v = g3b;
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if (lane == 0) red[wave * (HB * 16 + 1) + HB * 16] = v;
    }
    __syncthreads();
    for (int f = tid; f < HB * 16 + 1; f += nthr) {
        const float v = (red[f] + red[(HB * 16 + 1) + f]) + (red[2 * (HB * 16 + 1) + f] + red[3 * (HB * 16 + 1) + f]);
        if (f < H) gp[G::oW3 + DI + H + f] = v;
        else if (f == HB * 16) gp[G::ob3] = v;
    }
}


#ifndef PSP_GEN_X3_RING
#define PSP_GEN_X3_RING 2         // depth of the split-product consumers' row-operand ring (3 spills 32 dwords under the 256-register cap)
#endif
#ifndef PSP_GEN_BF16_RING
#define PSP_GEN_BF16_RING 4       // depth of the bf16 consumers' row-operand ring (A/B: -DPSP_GEN_BF16_RING=3 is round 1's)
#endif
// =======================================================================================
// Backward kernel, role-specialised variant (same scheme as hjb_bwd2_kernel).
// One 8-wave workgroup per CU, rounds of 4 sample blocks, one barrier per round, double-buffered LDS exchange.
//   producers (waves 0-3): the adjoint panels gz2, gz2', gz1, gz1' of their own block (two register-chained
//       64-MFMA products + the element-wise adjoint algebra) -> exchange buffer of the NEXT round; the T-layout parts
//       of dW3 (h2 rows) and db3 as running register sums;
//   consumers (waves 4-7): wave c owns column block(s) c of dW1 / dW2 for ALL row blocks, so its B operands are just
//       four exchange tiles per block; the row operands (X, U and d1, z1^ image tiles, feature-on-lane, straight from
//       the path store) run through a 3-deep register ring, one row tile per slot group:
//           x rows:  dW2x += x^T gz2 + (w U)^T gz2',   dW1 += x^T gz1 + (w U)^T gz1'     (16 MFMAs per tile)
//           h rows:  dW2h += h1^T gz2 + h1'^T gz2',     h1 = (d1/2)^2, h1' = d1 w z1^    ( 8 MFMAs per tile)
//       biases from the exchange tiles, the feature-on-lane parts of dW3 from the row tiles (rows split over the waves).
// =======================================================================================
// X3 (psp_gen_config.mlp_dtype = PSP_MLP_F16X3, fp32 path store, shared trajectory weights): the consumers' weight-gradient
// outer products as split f16 products over PAIRS of sample blocks (the bf16 consumers' structure with hi / lo packs: three
// v_mfma_f32_16x16x32_f16 per fp32-grade product instead of eight fp32 MFMAs).  The whole kernel is linear in the trajectory weights
// (~1 / K: below the f16 normal range), so they are scaled by a power of two that maps the largest |wY|, |wV| (one scan of the two
// K-vectors per workgroup) to [2^7, 2^8) and the partial gradient is scaled back.  Every weighted operand then sits in the upper
// f16 range, where the UNSCALED residual lo = f16(x - hi) is a normal number (for the unweighted O(1) operands a subnormal lo
// costs at most 3e-8 absolute: fp32's own epsilon), so a.b = hi_a hi_b + hi_a lo_b + lo_a hi_b runs on ONE accumulator with no
// 2048 anywhere.  Range: weighted adjoints stay below 65504 as long as the network factors (w3 phi', W2 products) stay below 256.
// (Tried and dropped, round 4: stamps (tools/r4/gen_stamps.py) show the X3 consumers at 20.2 k cycles of work per round and the
//  producers at 9.3 k + 11.4 k of barrier wait.  Moving the h rows of dW2 (4 of the consumers' 11 row items per block) to the
//  producer waves -- producer p for column block p, on the buffer the consumers are reading, a four-slot ring of their own --
//  is correct on all 172 value-net tests and levels the roles (17.5 k / 21.1 k), but the backward goes 4.25 -> 4.33-4.44 ms: a
//  producer and a consumer wave share each SIMD, and what looked like idle producer time is what lets the consumers run at the
//  CU's full load / issue rate.  The sum of the work per CU binds, not the longer role.)
// (Tried and dropped: the role cut of hjb_bwd3_kernel -- producers writing the four adjoint panels pre-split as pair images,
//  consumers owning ROW items for all column blocks so that each row operand is loaded and split by one wave only: 656 instead of
//  2 111 VALU instructions per round in the consumers, correct on all 59 tests, but 20 accumulator tiles + three row items'
//  packs + their landing registers do not fit 256 registers: 57 spilled dwords in the consumer loop, 5.3 ms against 4.47.)
template <int D, int H, bool BF16 = false, bool X3 = false>
__global__ __launch_bounds__(512) void gen_bwd2_kernel(const GenArgs a) {
    PSP_COND_EXIT(a);
    GradCheck<X3> gchk;                                      // backward side of the range guard (hjb_kernels.h)
    static_assert(!(BF16 && X3), "one matrix-product mode");
    using G = GGeo<D, H>;
    constexpr int DI = G::DI, DBI = G::DBI, HB = G::HB, KSH = G::KSH, EXT = G::EXT;
    constexpr int WHc = (HB >= 4) ? 4 : (HB >= 2 ? 2 : 1), WDc = 4 / WHc;      // consumers: columns first
    constexpr int NIB = cdiv(HB, WHc), NRX = cdiv(DBI, WDc), NRH = cdiv(HB, WDc), NROW = NRX + NRH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const bool producer = wave < 4;
    const int sub = wave & 3;
    const int qq = lane >> 4, col = lane & 15;
    // bf16 mode reads the bf16-pair path block (GGeo::q*): block stride PB16, F-layout lane offset inside a pair image, and
    // the v_perm selector that turns the lane's half of a dword into an fp32 (features 4r+q with r >= 2 sit in the high halves)
    constexpr unsigned PBx = BF16 ? (unsigned)G::PB16 : (unsigned)G::PB;
    const unsigned lofsU = BF16 ? (unsigned)(64 * ((col >> 2) & 1) + 16 * (col & 3) + 4 * qq) : (unsigned)image_lane_offset_F(lane);
    const unsigned selF = (col >> 3) ? 0x03020c0cu : 0x01000c0cu;
    const float* __restrict__ P = a.params;

    gen_stage<(X3 ? 2 : (BF16 ? 1 : 0)), KSH, HB>(lds + G::gW2hr, HB, tid, nthr, [&](int row, int c2) {
        return (row < H && c2 < H) ? P[G::oW2 + (DI + row) * H + c2] : 0.f; });
    __syncthreads();
    // (the split table pads the contraction to whole 32-feature steps: larger than the fp32 table unless H is a multiple of 32)
    constexpr int TBL = X3 ? SplitGeo<KSH, HB>::floats(HB) : G::gEx - G::gW2hr;
    float* bufs = lds + G::gW2hr + TBL;               // [2 buffers][4 blocks][EXT tiles][256]
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int Kpad = a.ntile16 * 16;
    float gs = 1.0f, ginv = 1.0f;                     // X3: power-of-two scale of the trajectory weights and its inverse
    if constexpr (X3) {
        float am = 0.f;
        for (int k0 = tid; k0 < Kpad; k0 += nthr) am = fmaxf(am, fmaxf(fabsf(a.wY[k0]), fabsf(a.wV[k0])));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) am = fmaxf(am, __shfl_xor(am, o));
        if (lane == 0) bufs[wave] = am;
        __syncthreads();
        am = fmaxf(fmaxf(fmaxf(bufs[0], bufs[1]), fmaxf(bufs[2], bufs[3])), fmaxf(fmaxf(bufs[4], bufs[5]), fmaxf(bufs[6], bufs[7])));
        const unsigned e = (__float_as_uint(am) >> 23) & 0xFFu;
        if (e >= 8u && e <= 249u) { gs = __uint_as_float((261u - e) << 23); ginv = __uint_as_float((e - 7u) << 23); }
        __syncthreads();                              // (the slots are part of the exchange area)
    }
    const long long nblk = (long long)(a.N + 1) * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    const int R = (int)((nround - blockIdx.x + gridDim.x - 1) / gridDim.x);
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;

    if (producer) {
        auto w3_T = [&](int base, int m, int o0) {          // o0 = opaque zero: keeps the loads inside the round loop
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int f = 16 * m + 4 * r + q; v[r] = f < H ? P[base + o0 + (f < H ? f : 0)] : 0.f; }
            return v;
        };
        f32x4 g3h2T[HB];                                // dW3 (h2 part) in T layout, reduced over lanes at the end
        float g3b = 0.f;
#pragma unroll
        for (int m = 0; m < HB; ++m) g3h2T[m] = zero4;
        // the four images of the own block (d2, z2^, d1, z1^; T layout) are requested one iteration ahead
        f32x4 pd2[HB], pz2[HB], pd1[HB], pz1[HB];
        auto prefetch_block = [&](int it2) __attribute__((always_inline)) {
            const long long b0 = ((long long)blockIdx.x + (long long)it2 * gridDim.x) * 4 + sub;
            const float* pb = a.path + (size_t)(b0 < nblk ? b0 : nblk - 1) * (size_t)PBx + lane;
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                pd2[m] = image_get_T<BF16>(pb, G::pD2, G::qD2, m);
                pz2[m] = image_get_T<BF16>(pb, G::pZ2, G::qZ2, m);
                pd1[m] = image_get_T<BF16>(pb, G::pD1, G::qD1, m);
                pz1[m] = image_get_T<BF16>(pb, G::pZ1, G::qZ1, m);
            }
        };
        prefetch_block(0);
#ifdef PSP_STAMPS
        unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int it = 0; it <= R; ++it) {
            PSP_STAMP(tp0);
            if (it < R) {
                const long long round = blockIdx.x + (long long)it * gridDim.x;
                const long long rb = round * 4;
                float* my_ex = bufs + ((it & 1) * 4 + sub) * (EXT * 256);
                // L2 touch (one dword per 128-B line) of the row images the consumers read next iteration
                float touch[5];
                {
                    const long long xb0 = rb + sub;
                    const float* xt = a.path + (size_t)(xb0 < nblk ? xb0 : nblk - 1) * (size_t)PBx;
                    constexpr int NA = BF16 ? G::qD2 : G::pD2;   // X, U, d1 images are contiguous: [0, pD2)
                    constexpr int oZ1 = BF16 ? G::qZ1 : G::pZ1, nZ1 = (BF16 ? 2 : 4) * HB * 64;
                    // measured in round 2 (A/B builds): the touch is worth 8 % to the bf16 kernel, whose consumers wait for loads
                    // (3.38 against 3.67 ms), and COSTS the fp32 kernel 4.5 % (6.31 against 6.02 ms: MFMA-bound, its consumers'
                    // 3-deep ring already covers an L2 hit; the touches only add requests) -- so only the bf16 kernel touches
                    if constexpr (BF16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const int o = lane * 32 + i * 2048; touch[i] = xt[o < NA ? o : 0]; }
                        touch[4] = xt[oZ1 + ((lane * 32 < nZ1) ? lane * 32 : 0)];
                    } else {
#pragma unroll
                        for (int i = 0; i < 5; ++i) touch[i] = 0.f;
                        (void)xt; (void)NA; (void)oZ1; (void)nZ1;
                    }
                }
                        // ---------------------------------------------------------- adjoints of the own block
                {
                    f32x4 gz1[HB], gz1t[HB];
                    const long long blk0 = rb + sub;
                    const bool bvalid = blk0 < nblk;
                    const long long blk = bvalid ? blk0 : nblk - 1;
                    const int n = (int)(blk / a.ntile16), t16 = (int)(blk % a.ntile16);
                    const int k = t16 * 16 + j;
                    const bool kvalid = bvalid && k < a.K_local;
                    const float* pb = a.path + (size_t)blk * (size_t)PBx + lane;
                    const bool fin = (n == a.N);
                    // wY / wV / ahat are zero-padded to 16*ntile16 entries: plain loads, no branch around them
                    const size_t wofs = a.per_sample ? (size_t)n * Kpad : 0;            // per-sample mode: wY is (N+1, Kpad), ahat IS the coefficient
            const float wy = gs * a.wY[wofs + k], wv = a.per_sample ? 0.f : gs * a.wV[k], ah = a.ahat[(size_t)n * Kpad + k];
                    const float wsv = (bvalid && !fin) ? wy : 0.f;                             // weight of the tangent part
                    const float av = bvalid ? (a.per_sample ? ah : (fin ? wv : wy * ah)) : 0.f;
                    const int o0 = opaque_i(0);
                    f32x4 gz2[HB], gz2t[HB];
                    {
                        f32x4 d2[HB], z2t[HB];
        #pragma unroll
                        for (int m = 0; m < HB; ++m) {
                            d2[m] = image_get_T<BF16>(pb, G::pD2, G::qD2, m);
                            z2t[m] = wsv * image_get_T<BF16>(pb, G::pZ2, G::qZ2, m);
                        }
        #pragma unroll
                        for (int m = 0; m < HB; ++m) {
                            const f32x4 w3h2 = w3_T(G::oW3 + DI + H, m, o0);
                            gz2t[m] = w3h2 * d2[m];
                            gz2[m] = av * gz2t[m] + w3h2 * step2(d2[m]) * z2t[m];
                            // dW3 (h2 part): a h2 + h2'   with h2 = (d2/2)^2, h2' = d2 z2'
                            g3h2T[m] += av * (0.25f * d2[m] * d2[m]) + d2[m] * z2t[m];
                        }
                    }
                    g3b += (q == 0) ? av : 0.f;
        #pragma unroll
                    for (int m = 0; m < HB; ++m) { gz1t[m] = w3_T(G::oW3 + DI, m, o0); gz1[m] = av * gz1t[m]; }
                    gen_gemm<(X3 ? 2 : (BF16 ? 1 : 0)), HB, KSH, HB>(gz1t, lds + G::gW2hr, gz2t, lane);      // gh1' = w3h1 + W2h gz2'
                    gen_gemm<(X3 ? 2 : (BF16 ? 1 : 0)), HB, KSH, HB>(gz1, lds + G::gW2hr, gz2, lane);        // gh1  = a w3h1 + W2h gz2
                    {   // d1 / z1^ are L2-resident (touch-prefetched one round ahead): fetch them only now
                        f32x4 d1[HB], z1t[HB];
        #pragma unroll
                        for (int m = 0; m < HB; ++m) {
                            d1[m] = image_get_T<BF16>(pb, G::pD1, G::qD1, m);
                            z1t[m] = wsv * image_get_T<BF16>(pb, G::pZ1, G::qZ1, m);
                        }
        #pragma unroll
                        for (int m = 0; m < HB; ++m) {
                            gz1[m] = gz1[m] * d1[m] + gz1t[m] * step2(d1[m]) * z1t[m];
                            gz1t[m] = gz1t[m] * d1[m];
                        }
                    }
                    prefetch_block(it + 1);               // next round's images: a whole iteration of lead
                    if (!kvalid) {
#pragma unroll
                        for (int m = 0; m < HB; ++m) { gz2[m] = zero4; gz2t[m] = zero4; gz1[m] = zero4; gz1t[m] = zero4; }
                    }
        #pragma unroll
                    for (int m = 0; m < HB; ++m) {
                        tile_put(my_ex + m * 256, gz2[m], lane);
                        tile_put(my_ex + (HB + m) * 256, gz2t[m], lane);
                        tile_put(my_ex + (2 * HB + m) * 256, gz1[m], lane);
                        tile_put(my_ex + (3 * HB + m) * 256, gz1t[m], lane);
                    }
                }

                asm volatile("" :: "v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]), "v"(touch[4]));
            }
            PSP_STAMP(tp1);
            __syncthreads();                              // swap the exchange buffers (pairs with the consumer loop)
            PSP_STAMP(tp2);
            PSP_ACC(0, tp1, tp0); PSP_ACC(1, tp2, tp1);
        }
#ifdef PSP_STAMPS
        if (a.dbg && lane == 0) {
            stamps[7] = (unsigned long long)R;
            for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + (tid >> 6)) * 8 + i] = stamps[i];
        }
#endif
        // T-layout partial sums of dW3 (h2 rows) and db3: reduce over the 16 trajectory lanes, then over the 4 producers
        float* red = bufs;                                // [4][HB*16 + 1]; the exchange area is free after the last barrier
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = ginv * g3h2T[m][r];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                if (j == 0) red[sub * (HB * 16 + 1) + 16 * m + 4 * r + q] = v;
            }
        {
            float v = ginv * g3b;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (lane == 0) red[sub * (HB * 16 + 1) + HB * 16] = v;
        }
        __syncthreads();                                  // pairs with the consumers' barrier before the dW3 write-out
        return;
    }
    // ==================================================================================== consumers
    const int wh = (WHc == 1) ? 0 : sub % WHc, wd = (WDc == 1) ? 0 : sub / WHc;
    f32x4 acc2x[NRX][NIB], acc2h[NRH][NIB], acc1[NRX][NIB];
    float bs2[NIB], bs1[NIB], g3r[NROW];
#pragma unroll
    for (int s = 0; s < NRX; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) { acc2x[s][t] = zero4; acc1[s][t] = zero4; }
#pragma unroll
    for (int s = 0; s < NRH; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) acc2h[s][t] = zero4;
#pragma unroll
    for (int t = 0; t < NIB; ++t) { bs2[t] = 0.f; bs1[t] = 0.f; }
#pragma unroll
    for (int i = 0; i < NROW; ++i) g3r[i] = 0.f;
    int cbc[NIB];
#pragma unroll
    for (int t = 0; t < NIB; ++t) cbc[t] = ((wh + WHc * t) < HB ? (wh + WHc * t) : HB - 1) * 256;
    // row item i of a block: i < NRX -> x rows (images X, U), else h rows (images d1, z1^); image offsets of the pair
    constexpr int oX = BF16 ? G::qX : G::pX, oU = BF16 ? G::qU : G::pU, oD1 = BF16 ? G::qD1 : G::pD1, oZ1c = BF16 ? G::qZ1 : G::pZ1;
    constexpr int BLK = BF16 ? 128 : 256;              // dwords of one 16-feature block of an image
    auto row_ofs0 = [&](int i) { return i < NRX ? oX + ((wd + WDc * i) < DBI ? (wd + WDc * i) : DBI - 1) * BLK
                                               : oD1 + ((wd + WDc * (i - NRX)) < HB ? (wd + WDc * (i - NRX)) : HB - 1) * BLK; };
    auto row_ofs1 = [&](int i) { return i < NRX ? oU + ((wd + WDc * i) < DBI ? (wd + WDc * i) : DBI - 1) * BLK
                                               : oZ1c + ((wd + WDc * (i - NRX)) < HB ? (wd + WDc * (i - NRX)) : HB - 1) * BLK; };
    const int nblk_i = (int)nblk;
    auto blk_at = [&](long long c0) __attribute__((always_inline)) {
        const int c = (c0 < (long long)nblk_i) ? (int)c0 : nblk_i - 1;
        return __builtin_amdgcn_readfirstlane(c);
    };
    typedef const __attribute__((address_space(1))) float* gptr_t;
    auto get_F = [&](int blk, int ofs) __attribute__((always_inline)) {
        // fp32: block address opaque before the offset is added (sgpr_block_addr: no hoisted per-offset bases, 17 -> 0 spilled
        // SGPRs); the bf16 kernel sits on its 256-register cap and the same form costs it 10-14 spilled VGPRs, so it keeps
        // the single sum
        unsigned long long addr;
        if constexpr (BF16) {
            addr = (unsigned long long)a.path + 4ull * ((unsigned long long)blk * PBx + (unsigned)ofs);
            asm volatile("" : "+s"(addr));
        } else {
            addr = sgpr_block_addr(a.path, (unsigned long long)blk, (unsigned)PBx, (unsigned)ofs);
        }
#if defined(PSP_GEN_ABLATE) && (PSP_GEN_ABLATE & 1)
        f32x4 v = {1.f, 1.f, 1.f, 1.f};                     // diagnostic build: consumers skip their path-store loads
        (void)addr;
#else
        f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((gptr_t)addr + lofsU);
#endif
        if constexpr (BF16) {                              // four samples of this lane's feature: its half of each dword -> fp32
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u = __float_as_uint(v[e]);
                v[e] = __uint_as_float(__builtin_amdgcn_perm(u, u, selF));
            }
        }
        return v;
    };
    f32x4 ra[3], rb_[3];                               // 3-deep ring of row-operand pairs
    f32x4 w4n = zero4, a4n = zero4;                    // per-sample weights of the NEXT block (4 samples per lane)
    auto load_weights = [&](long long c0) __attribute__((always_inline)) {
        const bool sval = c0 < nblk;
        const int cb = blk_at(c0);
        const int n = cb / a.ntile16, t16 = cb % a.ntile16;
        const bool fin = (n == a.N);
        const int k4 = t16 * 16 + 4 * qq;
        const f32x4 wy4 = *reinterpret_cast<const f32x4*>(a.wY + (a.per_sample ? (size_t)n * Kpad : 0) + k4);
        const f32x4 wv4 = a.per_sample ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.wV + k4);
        const f32x4 ah4 = *reinterpret_cast<const f32x4*>(a.ahat + (size_t)n * Kpad + k4);
        w4n = (sval && !fin) ? wy4 : zero4;
        a4n = sval ? (a.per_sample ? ah4 : (fin ? wv4 : wy4 * ah4)) : zero4;
    };
    if constexpr (X3) {
        // split f16 outer products over pairs of sample blocks (the structure of the bf16 consumers below; fp32 path store)
        auto split2 = [&](const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) __attribute__((always_inline)) {
            split8u(u0, u1, hi, lo);                      // hi = f16(x), lo = f16(x - hi): the unscaled residual (header comment)
        };
        auto weights_of = [&](long long c0, f32x4& w4, f32x4& a4) __attribute__((always_inline)) {
            const bool sval = c0 < nblk;
            const int cb = blk_at(c0);
            const int n = cb / a.ntile16, t16 = cb % a.ntile16;
            const bool fin = (n == a.N);
            const int k4 = t16 * 16 + 4 * qq;
            const f32x4 wy4 = gs * *reinterpret_cast<const f32x4*>(a.wY + k4);
            const f32x4 wv4 = gs * *reinterpret_cast<const f32x4*>(a.wV + k4);
            const f32x4 ah4 = *reinterpret_cast<const f32x4*>(a.ahat + (size_t)n * Kpad + k4);
            w4 = (sval && !fin) ? wy4 : zero4;
            a4 = sval ? (fin ? wv4 : wy4 * ah4) : zero4;
        };
        constexpr int RB = PSP_GEN_X3_RING;
        constexpr int NROWP = cdiv(NROW, RB) * RB;
        f32x4 ra0[RB], rb0[RB], ra1[RB], rb1[RB];
        auto unit_blk = [&](int u, int which) __attribute__((always_inline)) {
            const long long rbu = ((long long)blockIdx.x + (long long)(u >> 1) * gridDim.x) * 4 + 2 * (u & 1) + which;
            return rbu;
        };
        f32x4 w40n, a40n, w41n, a41n;
        {
            const int c0 = blk_at(unit_blk(0, 0)), c1 = blk_at(unit_blk(0, 1));
            weights_of(unit_blk(0, 0), w40n, a40n);
            weights_of(unit_blk(0, 1), w41n, a41n);
#pragma unroll
            for (int i = 0; i < RB - 1 && i < NROW; ++i) {
                ra0[i] = get_F(c0, row_ofs0(i)); rb0[i] = get_F(c0, row_ofs1(i));
                ra1[i] = get_F(c1, row_ofs0(i)); rb1[i] = get_F(c1, row_ofs1(i));
            }
        }
        __syncthreads();                                      // pairs with producer iteration 0
#ifdef PSP_STAMPS
        unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int it = 1; it <= R; ++it) {
            PSP_STAMP(tc0);
            const float* exch = bufs + ((it - 1) & 1) * 4 * (EXT * 256);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int u = 2 * (it - 1) + p;
                const float* ex0 = exch + (2 * p) * (EXT * 256);
                const float* ex1 = ex0 + EXT * 256;
                const int cb0 = blk_at(unit_blk(u, 0)), cb1 = blk_at(unit_blk(u, 1));
                const int nb0 = blk_at(unit_blk(u + 1, 0)), nb1 = blk_at(unit_blk(u + 1, 1));   // next unit (clamped past the end)
                const f32x4 w40 = w40n, a40 = a40n, w41 = w41n, a41 = a41n;
                weights_of(unit_blk(u + 1, 0), w40n, a40n);   // consumed a whole pair later
                weights_of(unit_blk(u + 1, 1), w41n, a41n);
                // B operands: the adjoint panels of the pair as (hi, 2048 hi, lo) packs
                f16x8 z2h[NIB], z2l[NIB], z2th[NIB], z2tl[NIB], z1h[NIB], z1l[NIB], z1th[NIB], z1tl[NIB];
#pragma unroll
                for (int t = 0; t < NIB; ++t) {
                    const f32x4 u0 = tile_get(ex0 + cbc[t], lane), u1 = tile_get(ex1 + cbc[t], lane);
                    const f32x4 v0 = tile_get(ex0 + 2 * HB * 256 + cbc[t], lane), v1 = tile_get(ex1 + 2 * HB * 256 + cbc[t], lane);
                    bs2[t] += hsum4(u0) + hsum4(u1);
                    bs1[t] += hsum4(v0) + hsum4(v1);
                    split2(u0, u1, z2h[t], z2l[t]);
                    split2(v0, v1, z1h[t], z1l[t]);
                    split2(tile_get(ex0 + HB * 256 + cbc[t], lane), tile_get(ex1 + HB * 256 + cbc[t], lane), z2th[t], z2tl[t]);
                    split2(tile_get(ex0 + 3 * HB * 256 + cbc[t], lane), tile_get(ex1 + 3 * HB * 256 + cbc[t], lane), z1th[t], z1tl[t]);
                }
#pragma unroll
                for (int i = 0; i < NROWP; ++i) {
                    const int slot = i % RB;
                    {   // request the item RB - 1 ahead: same pair, or the first items of the next one
                        const int jn = i + RB - 1;
                        const int ns = jn % RB;
                        if (jn < NROW) {
                            ra0[ns] = get_F(cb0, row_ofs0(jn)); rb0[ns] = get_F(cb0, row_ofs1(jn));
                            ra1[ns] = get_F(cb1, row_ofs0(jn)); rb1[ns] = get_F(cb1, row_ofs1(jn));
                        } else if (jn >= NROWP && jn - NROWP < NROW) {
                            ra0[ns] = get_F(nb0, row_ofs0(jn - NROWP)); rb0[ns] = get_F(nb0, row_ofs1(jn - NROWP));
                            ra1[ns] = get_F(nb1, row_ofs0(jn - NROWP)); rb1[ns] = get_F(nb1, row_ofs1(jn - NROWP));
                        }
                    }
                    if (i < NROW) {
                        f32x4 A00, A10, A01, A11;
                        if (i < NRX) {
                            A00 = ra0[slot]; A10 = w40 * rb0[slot];
                            A01 = ra1[slot]; A11 = w41 * rb1[slot];
                        } else {
                            const f32x4 d10 = ra0[slot], d11 = ra1[slot];
                            A00 = 0.25f * d10 * d10; A10 = d10 * (w40 * rb0[slot]);
                            A01 = 0.25f * d11 * d11; A11 = d11 * (w41 * rb1[slot]);
                        }
                        if (i % WHc == wh) g3r[i] += hsum4(a40 * A00 + A10) + hsum4(a41 * A01 + A11);
                        f16x8 P0h, P0l, P1h, P1l;
                        split2(A00, A01, P0h, P0l);
                        split2(A10, A11, P1h, P1l);
                        // PA . pz = hi_a hi_b + hi_a lo_b + lo_a hi_b on one accumulator
                        auto fma3 = [&](f32x4& acc, const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl)
                                        __attribute__((always_inline)) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
                        };
#pragma unroll
                        for (int t = 0; t < NIB; ++t) {
                            if (i < NRX) {
                                fma3(acc2x[i][t], P0h, P0l, z2h[t], z2l[t]);
                                fma3(acc1[i][t], P0h, P0l, z1h[t], z1l[t]);
                                fma3(acc2x[i][t], P1h, P1l, z2th[t], z2tl[t]);
                                fma3(acc1[i][t], P1h, P1l, z1th[t], z1tl[t]);
                            } else {
                                fma3(acc2h[i - NRX][t], P0h, P0l, z2h[t], z2l[t]);
                                fma3(acc2h[i - NRX][t], P1h, P1l, z2th[t], z2tl[t]);
                            }
                        }
                    }
                }
            }
            PSP_STAMP(tc1);
            __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
            PSP_STAMP(tc2);
            PSP_ACC(0, tc1, tc0); PSP_ACC(1, tc2, tc1);
        }
#ifdef PSP_STAMPS
        if (a.dbg && lane == 0) {
            stamps[7] = (unsigned long long)R;
            for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + (tid >> 6)) * 8 + i] = stamps[i];
        }
#endif
    } else if constexpr (BF16) {
        // bf16 outer products (v_mfma_f32_16x16x32_bf16): the 32-deep k-step is the sample index of TWO sample blocks --
        // lane (row / col, qq) holds samples 4qq..4qq+3 of either block in both operands, so a k-step is a pack of the two
        // blocks' registers on both sides (the sum over k does not care which slot a sample sits in).  One MFMA replaces
        // eight fp32 ones; the consumers turn from MFMA-bound into a stream over the path store.
        auto pack2 = [&](const f32x4& u0, const f32x4& u1) __attribute__((always_inline)) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (__bf16)u0[e]; v[4 + e] = (__bf16)u1[e]; }
            return v;
        };
        auto weights_of = [&](long long c0, f32x4& w4, f32x4& a4) __attribute__((always_inline)) {
            const bool sval = c0 < nblk;
            const int cb = blk_at(c0);
            const int n = cb / a.ntile16, t16 = cb % a.ntile16;
            const bool fin = (n == a.N);
            const int k4 = t16 * 16 + 4 * qq;
            const f32x4 wy4 = *reinterpret_cast<const f32x4*>(a.wY + (a.per_sample ? (size_t)n * Kpad : 0) + k4);
            const f32x4 wv4 = a.per_sample ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.wV + k4);
            const f32x4 ah4 = *reinterpret_cast<const f32x4*>(a.ahat + (size_t)n * Kpad + k4);
            w4 = (sval && !fin) ? wy4 : zero4;
            a4 = sval ? (a.per_sample ? ah4 : (fin ? wv4 : wy4 * ah4)) : zero4;
        };
        // Row-operand ring that runs THROUGH pair and round boundaries (round 2).  The path store is an input: the blocks a
        // workgroup will visit are known in advance, so the loads of the next pair's first items are requested while the last
        // items of the current pair are still being multiplied, and the per-sample weights of a pair a whole pair ahead.
        // Measured before: the ring was refilled from empty at every pair and the weight loads were issued BEHIND the refill
        // (vmcnt retires in order), so the first item of every pair waited a full memory latency -- without the consumers'
        // loads the kernel ran 1.89 instead of 3.56 ms.  Slots stay compile-time: items are counted modulo NROWP, the item
        // count padded to a multiple of the ring depth (the padding item loads and multiplies nothing).
        constexpr int RB = PSP_GEN_BF16_RING;
        constexpr int NROWP = cdiv(NROW, RB) * RB;
        f32x4 ra0[RB], rb0[RB], ra1[RB], rb1[RB];
        // block ids of pair unit u (two per round): round blockIdx.x + (u / 2) gridDim.x, blocks 2 (u % 2), 2 (u % 2) + 1
        auto unit_blk = [&](int u, int which) __attribute__((always_inline)) {
            const long long rbu = ((long long)blockIdx.x + (long long)(u >> 1) * gridDim.x) * 4 + 2 * (u & 1) + which;
            return rbu;
        };
        f32x4 w40n, a40n, w41n, a41n;
        {
            const int c0 = blk_at(unit_blk(0, 0)), c1 = blk_at(unit_blk(0, 1));
            weights_of(unit_blk(0, 0), w40n, a40n);
            weights_of(unit_blk(0, 1), w41n, a41n);
#pragma unroll
            for (int i = 0; i < RB - 1 && i < NROW; ++i) {
                ra0[i] = get_F(c0, row_ofs0(i)); rb0[i] = get_F(c0, row_ofs1(i));
                ra1[i] = get_F(c1, row_ofs0(i)); rb1[i] = get_F(c1, row_ofs1(i));
            }
        }
        __syncthreads();                                      // pairs with producer iteration 0
        for (int it = 1; it <= R; ++it) {
            const float* exch = bufs + ((it - 1) & 1) * 4 * (EXT * 256);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int u = 2 * (it - 1) + p;
                const float* ex0 = exch + (2 * p) * (EXT * 256);
                const float* ex1 = ex0 + EXT * 256;
                const int cb0 = blk_at(unit_blk(u, 0)), cb1 = blk_at(unit_blk(u, 1));
                const int nb0 = blk_at(unit_blk(u + 1, 0)), nb1 = blk_at(unit_blk(u + 1, 1));   // next unit (clamped past the end)
                const f32x4 w40 = w40n, a40 = a40n, w41 = w41n, a41 = a41n;
                weights_of(unit_blk(u + 1, 0), w40n, a40n);   // consumed a whole pair later
                weights_of(unit_blk(u + 1, 1), w41n, a41n);
                bf16x8 pz2[NIB], pz2t[NIB], pz1[NIB], pz1t[NIB];
#pragma unroll
                for (int t = 0; t < NIB; ++t) {
                    const f32x4 u0 = tile_get(ex0 + cbc[t], lane), u1 = tile_get(ex1 + cbc[t], lane);
                    const f32x4 v0 = tile_get(ex0 + 2 * HB * 256 + cbc[t], lane), v1 = tile_get(ex1 + 2 * HB * 256 + cbc[t], lane);
                    bs2[t] += hsum4(u0) + hsum4(u1);
                    bs1[t] += hsum4(v0) + hsum4(v1);
                    pz2[t] = pack2(u0, u1);
                    pz1[t] = pack2(v0, v1);
                    pz2t[t] = pack2(tile_get(ex0 + HB * 256 + cbc[t], lane), tile_get(ex1 + HB * 256 + cbc[t], lane));
                    pz1t[t] = pack2(tile_get(ex0 + 3 * HB * 256 + cbc[t], lane), tile_get(ex1 + 3 * HB * 256 + cbc[t], lane));
                }
#pragma unroll
                for (int i = 0; i < NROWP; ++i) {
                    const int slot = i % RB;
                    {   // request the item RB - 1 ahead: same pair, or the first items of the next one
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int jn = i + RB - 1;
                        const int ns = jn % RB;
                        if (jn < NROW) {
                            ra0[ns] = get_F(cb0, row_ofs0(jn)); rb0[ns] = get_F(cb0, row_ofs1(jn));
                            ra1[ns] = get_F(cb1, row_ofs0(jn)); rb1[ns] = get_F(cb1, row_ofs1(jn));
                        } else if (jn >= NROWP && jn - NROWP < NROW) {
                            ra0[ns] = get_F(nb0, row_ofs0(jn - NROWP)); rb0[ns] = get_F(nb0, row_ofs1(jn - NROWP));
                            ra1[ns] = get_F(nb1, row_ofs0(jn - NROWP)); rb1[ns] = get_F(nb1, row_ofs1(jn - NROWP));
                        }
                    }
                    if (i < NROW) {
                        f32x4 A00, A10, A01, A11;
                        if (i < NRX) {
                            A00 = ra0[slot]; A10 = w40 * rb0[slot];
                            A01 = ra1[slot]; A11 = w41 * rb1[slot];
                        } else {
                            const f32x4 d10 = ra0[slot], d11 = ra1[slot];
                            A00 = 0.25f * d10 * d10; A10 = d10 * (w40 * rb0[slot]);
                            A01 = 0.25f * d11 * d11; A11 = d11 * (w41 * rb1[slot]);
                        }
                        if (i % WHc == wh) g3r[i] += hsum4(a40 * A00 + A10) + hsum4(a41 * A01 + A11);
                        const bf16x8 PA0 = pack2(A00, A01), PA1 = pack2(A10, A11);
#pragma unroll
                        for (int t = 0; t < NIB; ++t) {
                            if (i < NRX) {
                                acc2x[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA0, pz2[t], acc2x[i][t], 0, 0, 0);
                                acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA0, pz1[t], acc1[i][t], 0, 0, 0);
                                acc2x[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA1, pz2t[t], acc2x[i][t], 0, 0, 0);
                                acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA1, pz1t[t], acc1[i][t], 0, 0, 0);
                            } else {
                                acc2h[i - NRX][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA0, pz2[t], acc2h[i - NRX][t], 0, 0, 0);
                                acc2h[i - NRX][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA1, pz2t[t], acc2h[i - NRX][t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
        }
    } else {
    // prologue: first two row items and the weights of this workgroup's first block
    {
        const int b0 = blk_at((long long)blockIdx.x * 4);
        ra[0] = get_F(b0, row_ofs0(0)); rb_[0] = get_F(b0, row_ofs1(0));
        if (NROW > 1) { ra[1] = get_F(b0, row_ofs0(1)); rb_[1] = get_F(b0, row_ofs1(1)); }
        load_weights((long long)blockIdx.x * 4);
    }
    __syncthreads();                                      // pairs with producer iteration 0
    for (int it = 1; it <= R; ++it) {
        const long long rb = ((long long)blockIdx.x + (long long)(it - 1) * gridDim.x) * 4;
        const float* exch = bufs + ((it - 1) & 1) * 4 * (EXT * 256);
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const float* ex = exch + sb * (EXT * 256);
            const int cb = blk_at(rb + sb);
            // next block (within the round, or the first block of this workgroup's next round)
            const long long cn0 = (sb < 3) ? rb + sb + 1 : rb + 4LL * gridDim.x;
            const int cnx = blk_at(cn0);
            const f32x4 w4 = w4n, a4 = a4n;
            f32x4 bz2[NIB], bz2t[NIB], bz1[NIB], bz1t[NIB];
#pragma unroll
            for (int t = 0; t < NIB; ++t) {
                bz2[t] = tile_get(ex + cbc[t], lane);
                bz2t[t] = tile_get(ex + HB * 256 + cbc[t], lane);
                bz1[t] = tile_get(ex + 2 * HB * 256 + cbc[t], lane);
                bz1t[t] = tile_get(ex + 3 * HB * 256 + cbc[t], lane);
            }
            load_weights(cn0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NIB; ++t) { bs2[t] += hsum4(bz2[t]); bs1[t] += hsum4(bz1[t]); }
#pragma unroll
            for (int i = 0; i < NROW; ++i) {
                const int slot = (sb * NROW + i) % 3;
                // request the row item two ahead (same block, or the next block's first items)
                {
                    const int i2 = i + 2;
                    const int nslot = (sb * NROW + i2) % 3;
                    if (i2 < NROW) { ra[nslot] = get_F(cb, row_ofs0(i2)); rb_[nslot] = get_F(cb, row_ofs1(i2)); }
                    else { ra[nslot] = get_F(cnx, row_ofs0(i2 - NROW)); rb_[nslot] = get_F(cnx, row_ofs1(i2 - NROW)); }
                }
                f32x4 A0, A1;
                if (i < NRX) { A0 = ra[slot]; A1 = w4 * rb_[slot]; }                         // x0, x0' = w U
                else { const f32x4 d1 = ra[slot]; A0 = 0.25f * d1 * d1; A1 = d1 * (w4 * rb_[slot]); }   // h1, h1'
                if (i % WHc == wh) g3r[i] += hsum4(a4 * A0 + A1);                             // dW3 rows, split over waves
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NIB; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (i < NRX) {
                            acc2x[i][t] = mfma16(A0[r], bz2[t][r], acc2x[i][t]);
                            acc1[i][t] = mfma16(A0[r], bz1[t][r], acc1[i][t]);
                            acc2x[i][t] = mfma16(A1[r], bz2t[t][r], acc2x[i][t]);
                            acc1[i][t] = mfma16(A1[r], bz1t[t][r], acc1[i][t]);
                        } else {
                            acc2h[i - NRX][t] = mfma16(A0[r], bz2[t][r], acc2h[i - NRX][t]);
                            acc2h[i - NRX][t] = mfma16(A1[r], bz2t[t][r], acc2h[i - NRX][t]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the ring's slot phase is static: re-home the two pairs prefetched for the next round to slots 0 and 1
        if ((4 * NROW) % 3 != 0) {
            const f32x4 t0a = ra[(4 * NROW) % 3], t0b = rb_[(4 * NROW) % 3];
            const f32x4 t1a = ra[(4 * NROW + 1) % 3], t1b = rb_[(4 * NROW + 1) % 3];
            ra[0] = t0a; rb_[0] = t0b; ra[1] = t1a; rb_[1] = t1b;
        }
        __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
    }

    }
    // ---- write-out: tile (rbk, cbk): lane (col, qq), reg rr <-> dW[16 rbk + 4 qq + rr][16 cbk + col]   (weights are (in, out))
    const float osc = ginv;                            // X3: un-scale the weights' power of two (1 otherwise)
#pragma unroll
    for (int s = 0; s < NRX; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) {
            const int rbk = wd + WDc * s, cbk = wh + WHc * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * rbk + 4 * qq + rr, jo = 16 * cbk + col;
                if (rbk < DBI && cbk < HB && i < DI && jo < H) {
                    { const float gv_ = osc * acc2x[s][t][rr]; gp[G::oW2 + i * H + jo] = gv_; gchk.see(gv_); }
                    { const float gv_ = osc * acc1[s][t][rr]; gp[G::oW1 + i * H + jo] = gv_; gchk.see(gv_); }
                }
            }
        }
#pragma unroll
    for (int s = 0; s < NRH; ++s)
#pragma unroll
        for (int t = 0; t < NIB; ++t) {
            const int rbk = wd + WDc * s, cbk = wh + WHc * t;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * rbk + 4 * qq + rr, jo = 16 * cbk + col;
                if (rbk < HB && cbk < HB && i < H && jo < H) { const float gv_ = osc * acc2h[s][t][rr]; gp[G::oW2 + (DI + i) * H + jo] = gv_; gchk.see(gv_); }
            }
        }
#pragma unroll
    for (int t = 0; t < NIB; ++t) {
        const float v2 = ginv * qsum(bs2[t]), v1 = ginv * qsum(bs1[t]);
        const int f = 16 * (wh + WHc * t) + col;
        if (wd == 0 && qq == 0 && (wh + WHc * t) < HB && f < H) { { const float gv_ = v2; gp[G::ob2 + f] = gv_; gchk.see(gv_); } { const float gv_ = v1; gp[G::ob1 + f] = gv_; gchk.see(gv_); } }
    }
    // dW3, x and h1 rows: lane = feature; row item i belongs to the wave with i % WHc == wh
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const float v = ginv * qsum(g3r[i]);
        if (i % WHc == wh && qq == 0) {
            if (i < NRX) {
                const int f = 16 * (wd + WDc * i) + col;
                if ((wd + WDc * i) < DBI && f < DI) { const float gv_ = v; gp[G::oW3 + f] = gv_; gchk.see(gv_); }
            } else {
                const int f = 16 * (wd + WDc * (i - NRX)) + col;
                if ((wd + WDc * (i - NRX)) < HB && f < H) { const float gv_ = v; gp[G::oW3 + DI + f] = gv_; gchk.see(gv_); }
            }
        }
    }
    __syncthreads();                                      // pairs with the producers' barrier after their LDS write
    {
        const float* red = bufs;
        for (int f = tid - 256; f < HB * 16 + 1; f += 256) {
            const float v = (red[f] + red[(HB * 16 + 1) + f]) + (red[2 * (HB * 16 + 1) + f] + red[3 * (HB * 16 + 1) + f]);
            if (f < H) { const float gv_ = v; gp[G::oW3 + DI + H + f] = gv_; gchk.see(gv_); }
            else if (f == HB * 16) { const float gv_ = v; gp[G::ob3] = gv_; gchk.see(gv_); }
        }
    }
    gchk.raise(a.cond);
}

struct GenInstance {
    int d, H, n_params, path_floats_per_block;
    int path_dwords_per_block16;     // bf16-pair path block (mlp_dtype == PSP_MLP_BF16)
    int (*fwd_lds_bytes)();
    int (*bwd_lds_bytes)();
    hipError_t (*launch_fwd)(const GenArgs&, int grid, int block, hipStream_t);
    hipError_t (*launch_bwd)(const GenArgs&, int grid, int block, hipStream_t);
    int (*bwd2_lds_bytes)();
    hipError_t (*launch_bwd2)(const GenArgs&, int grid, hipStream_t);   // role-specialised variant, 512 threads
    hipError_t (*launch_fwd_bf16)(const GenArgs&, int grid, int block, hipStream_t);   // value-net products on bf16 MFMA
    hipError_t (*launch_bwd2_bf16)(const GenArgs&, int grid, hipStream_t);             // adjoint products + weight-gradient outer products on bf16 MFMA
    int (*fwd_x3_lds_bytes)();                                                         // split-product forward (PSP_MLP_F16X3)
    hipError_t (*launch_fwd_x3)(const GenArgs&, int grid, int block, hipStream_t);
    hipError_t (*launch_bwd2_x3)(const GenArgs&, int grid, hipStream_t);               // split-product weight gradients (shared weights only)
};

template <int D, int H>
struct GenLaunch {
    using G = GGeo<D, H>;
    static int fwd_lds() { return G::fwd_lds_floats() * 4; }
    static int bwd_lds() { return G::bwd_lds_floats() * 4; }
    static int bwd2_lds() { return (G::gEx + 2 * 4 * G::EXT * 256) * 4; }
    static hipError_t bwd2(const GenArgs& a, int grid, hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_bwd2_kernel<D, H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bwd2_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_bwd2_kernel<D, H>), dim3(grid), dim3(512), bwd2_lds(), s, a);
        return hipGetLastError();
    }
    static hipError_t bwd2_bf16(const GenArgs& a, int grid, hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_bwd2_kernel<D, H, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bwd2_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_bwd2_kernel<D, H, true>), dim3(grid), dim3(512), bwd2_lds(), s, a);
        return hipGetLastError();
    }
    template <bool BF16, bool PHILOX>
    static hipError_t fwd_as(const GenArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_fwd_kernel<D, H, BF16, PHILOX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fwd_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_fwd_kernel<D, H, BF16, PHILOX>), dim3(grid), dim3(block), fwd_lds(), s, a);
        return hipGetLastError();
    }
    static hipError_t fwd(const GenArgs& a, int grid, int block, hipStream_t s) {
        return a.noise_mode == NOISE_PHILOX ? fwd_as<false, true>(a, grid, block, s) : fwd_as<false, false>(a, grid, block, s);
    }
    // specialised problem kinds (gen_fwd_kernel SPECK): unbounded, non-adaptive, path store on, no per-step value output
    static int spec_kind(const GenArgs& a) {
        if (!spec_enabled() || a.noise_mode != NOISE_PHILOX || a.domain_kind != DOM_NONE || a.adaptive || a.store_path != 1 || a.Vsteps != nullptr) return -1;
        if (a.drift_kind == DRIFT_DWELL && a.h_kind == GH_QUAD) return DRIFT_DWELL | (GH_QUAD << 4);
        if (a.drift_kind == DRIFT_ZERO && a.h_kind == GH_ALLEN_CAHN) return DRIFT_ZERO | (GH_ALLEN_CAHN << 4);
        return -1;
    }
    template <bool BF16, bool X3, int SPECK>
    static hipError_t fwd_spec(const GenArgs& a, int grid, int block, hipStream_t s) {
        const int bytes = X3 ? G::fwd_x3_lds_floats() * 4 : fwd_lds();
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_fwd_kernel<D, H, BF16, true, X3, SPECK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_fwd_kernel<D, H, BF16, true, X3, SPECK>), dim3(grid), dim3(block), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t fwd_bf16(const GenArgs& a, int grid, int block, hipStream_t s) {
        const int sk = spec_kind(a);
        if (sk == (DRIFT_DWELL | (GH_QUAD << 4))) return fwd_spec<true, false, DRIFT_DWELL | (GH_QUAD << 4)>(a, grid, block, s);
        if (sk == (DRIFT_ZERO | (GH_ALLEN_CAHN << 4))) return fwd_spec<true, false, DRIFT_ZERO | (GH_ALLEN_CAHN << 4)>(a, grid, block, s);
        return a.noise_mode == NOISE_PHILOX ? fwd_as<true, true>(a, grid, block, s) : fwd_as<true, false>(a, grid, block, s);
    }
    static int bwd2_x3_lds() { return (G::gW2hr + SplitGeo<G::KSH, G::HB>::floats(G::HB) + 2 * 4 * G::EXT * 256) * 4; }
    static hipError_t bwd2_x3(const GenArgs& a, int grid, hipStream_t s) {
        if (bwd2_x3_lds() > 160 * 1024) return bwd2(a, grid, s);       // (split table does not fit: the fp32-MFMA kernel, same results)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_bwd2_kernel<D, H, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bwd2_x3_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_bwd2_kernel<D, H, false, true>), dim3(grid), dim3(512), bwd2_x3_lds(), s, a);
        return hipGetLastError();
    }
    static int fwd_x3_lds() { return G::fwd_x3_lds_floats() * 4; }
    template <bool PHILOX>
    static hipError_t fwd_x3_as(const GenArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_fwd_kernel<D, H, false, PHILOX, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fwd_x3_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_fwd_kernel<D, H, false, PHILOX, true>), dim3(grid), dim3(block), fwd_x3_lds(), s, a);
        return hipGetLastError();
    }
    static hipError_t fwd_x3(const GenArgs& a, int grid, int block, hipStream_t s) {
        const int sk = spec_kind(a);
        if (sk == (DRIFT_DWELL | (GH_QUAD << 4))) return fwd_spec<false, true, DRIFT_DWELL | (GH_QUAD << 4)>(a, grid, block, s);
        if (sk == (DRIFT_ZERO | (GH_ALLEN_CAHN << 4))) return fwd_spec<false, true, DRIFT_ZERO | (GH_ALLEN_CAHN << 4)>(a, grid, block, s);
        return a.noise_mode == NOISE_PHILOX ? fwd_x3_as<true>(a, grid, block, s) : fwd_x3_as<false>(a, grid, block, s);
    }
#ifdef PSP_LEGACY_BWD
    // gen_bwd_kernel: superseded by gen_bwd2_kernel; diagnostic builds only (-DPSP_LEGACY_BWD + PSP_BWD_VARIANT=1)
    static hipError_t bwd(const GenArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_bwd_kernel<D, H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gen_bwd_kernel<D, H>), dim3(grid), dim3(block), bwd_lds(), s, a);
        return hipGetLastError();
    }
    static constexpr auto legacy_bwd = &bwd;
#else
    static constexpr hipError_t (*legacy_bwd)(const GenArgs&, int, int, hipStream_t) = nullptr;
#endif
    static GenInstance instance() {
        return GenInstance{D, H, G::P, G::PB, G::PB16, &fwd_lds, &bwd_lds, &fwd, legacy_bwd, &bwd2_lds, &bwd2, &fwd_bf16, &bwd2_bf16,
                           &fwd_x3_lds, &fwd_x3, &bwd2_x3};
    }
};

}  // namespace psp

#define PSP_DEFINE_GEN_INSTANCE(D_, H_) \
    extern "C" psp::GenInstance psp_gen_instance_##D_##_##H_() { return psp::GenLaunch<D_, H_>::instance(); }
#define PSP_DECLARE_GEN_INSTANCE(D_, H_) extern "C" psp::GenInstance psp_gen_instance_##D_##_##H_();
