// hjbq_kernels.h -- quad-trajectory forward rollout for the SMALLEST trajectory counts (K <= 4 x number of CUs).
//
// hjbs_fwd_kernel gives a 16-trajectory tile to one CU: K = 1024 (BASELINE configs[1]) is 64 tiles on a chip with 256 CUs,
// and a step still costs 173 v_mfma_f32_16x16x4_f32 per SIMD (5.5 k cycles of matrix pipe) behind four barriers.  Here a
// workgroup owns FOUR trajectories, so K = 1024 is 256 workgroups -- the whole chip -- and the products run on
// v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4 outer products, 8 cycles, same flop rate as the 16x16x4 form):
//   * the ACTIVATIONS are the A operand: lane (kk, i) of ONE register holds the value of input feature kk of this wave for
//     trajectory i, and CBSZ = 4 / ABID = kk broadcasts block kk to all 16 blocks -- no replication, no transposes;
//   * the WEIGHTS are the B operand: lane (blk, c) holds W[64 slab + 4 blk + c][feature kk]; they are loaded once and stay
//     in registers (104 per lane at d = 100, H = 64);
//   * output register i of lane (blk, c) is row 64 slab + 4 blk + c of the product for trajectory i.
// Every product is split over the eight waves along its CONTRACTION index (wave w owns the state features of the groups
// (b, q) = w, w + 8, ... -- a group is the four features 16 b + 4 r + q that share one Philox call -- and the hidden units
// 2 HB w ... 2 HB w + 2 HB - 1); the eight partial products meet in LDS, where the owner lane (kk, i) of the NEXT product's
// input adds them up (eight ds_read_b32), applies bias / tanh / the Euler update and has its A operand.  Element-wise work,
// Philox and the path store are done once per (feature, trajectory) by the owner lane.  Per step and SIMD: 2 x 104 MFMAs x 8
// cycles = 1.7 k cycles of matrix pipe, four barriers.
// Same algebra, reference lines, Philox counters and path-store format as hjb_fwd_kernel / hjbs_fwd_kernel (the backward
// kernels are unchanged: a quad writes its four columns of the 16-trajectory block); summation orders differ by a few ulp.
#pragma once
#include "hjb_kernels.h"

namespace psp {

template <int D, int H>
struct GeoQ {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB;
    static constexpr int W = 8;                          // waves per workgroup
    static constexpr int NG = 4 * DB;                    // feature groups (b, q): features 16 b + 4 r + q, r = 0..3
    static constexpr int NS = cdiv(NG, W);               // group slots per wave
    static constexpr int KO = 4 * NS;                    // owned state features per wave: block kk = 4 s + r of the register
    static constexpr int KHo = 2 * HB;                   // owned hidden units per wave (16 HB / 8)
    static constexpr int SD = cdiv(W * KO, 64);          // 64-row slabs of a product with d output rows (rows in OWNER order)
    static constexpr int RD = 64 * SD, RH = 64;          // rows of the partial-product buffers
    static constexpr bool fits = KO <= 16 && HB <= 4;    // one A-operand register per vector
    static constexpr int NP = cdiv(NG, 16);              // noise-producer waves (64 Philox calls each: 16 groups x 4 trajectories)
    static constexpr int NT = 64 * (W + NP);             // threads per workgroup
    // LDS (floats): partial products [wave][row][trajectory], final reduction scratch [wave][quantity][trajectory],
    // Brownian increments [parity][group][trajectory][r]
    static constexpr int pH1 = 0, pDR = pH1 + W * RH * 4, pH2 = pDR + W * RD * 4, pZ = pH2 + W * RH * 4,
                         pBV = pZ + W * RD * 4, fRed = pBV + W * RD * 4, fXi = fRed + W * 16, XIB = 16 * NP * 16,
                         lds_floats = fXi + 2 * XIB;
};

// acc[i] (lane (blk, c)) += act[lane (KK, i)] * w[lane (blk, c)]
template <int KK>
__device__ __forceinline__ f32x4 mfma4(float act, float w, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(act, w, acc, 4, KK, 0);
}
__device__ __forceinline__ float bsum16(float v) {       // sum over the 16 blocks of a wave (trajectory = lane & 3 fixed)
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// FAST: on-device noise, no u_L2 log, no time-feature table (decided at launch).  The time loop of that instance has no
// vector-memory LOAD: a load in a wave-uniform branch costs the common path an `s_waitcnt vmcnt(0)` at the join, and vmcnt
// counts in order, so the wave would wait for its own path-store writes three times per step.
template <int D, int H, int FAST_>
__global__ __launch_bounds__((GeoQ<D, H>::NT)) void hjbq_fwd_kernel(const HjbArgs a) {
    // FAST_ = 2: the problem switches of the LLGC configurations as compile-time constants (hjb_kernels.h, hjb_fwd_kernel)
    constexpr bool FAST = FAST_ != 0, SPEC = FAST_ == 2;
    const int k_drift = SPEC ? (int)DRIFT_DENSE : a.drift_kind, k_sigma = SPEC ? (int)SIGMA_DENSE : a.sigma_kind;
    const int k_run = SPEC ? (int)RUN_ZERO : a.runcost_kind, k_loss = SPEC ? (int)LOSS_LOGVAR : a.loss_kind;
    const bool k_adaptive = SPEC ? true : (a.adaptive != 0);
    const uint32_t iter_now = a.iter_dev ? *a.iter_dev : a.iter;       // wave-uniform scalar load
    using G = Geo<D, H>;
    using Q = GeoQ<D, H>;
    constexpr int NG = Q::NG, NS = Q::NS, KO = Q::KO, KHo = Q::KHo, SD = Q::SD, RD = Q::RD, RH = Q::RH, W = Q::W;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave_id >= Q::W;               // noise-producer waves (see the time loop); they run the prologue as wave 0
    const int wave = producer ? 0 : wave_id;
    const int blk = lane >> 2, tr = lane & 3;            // activation owner: (feature slot kk = blk, trajectory tr)
    const int wrow = lane;                               // weight holder: row 4 blk + c of a slab, c = lane & 3
    const float* __restrict__ P = a.params;
    const float dt = a.dt, sqdt = a.sqdt;
    const bool denseA = k_drift == DRIFT_DENSE, denseB = k_sigma == SIGMA_DENSE;

    // ---- the state feature and the hidden unit this lane owns
    const int s = blk >> 2, r = blk & 3;
    const int gam = wave + W * s;                        // group (b, q)
    const int fb = gam >> 2, fq = gam & 3;
    const int f = 16 * fb + 4 * r + fq;
    const bool gvalid = blk < KO && gam < NG;            // a row of the path-store images exists (padded features hold zeros)
    const bool fvalid = gvalid && f < D;
    const int fc = fvalid ? f : 0;
    const int m = KHo * wave + blk;
    const bool hslot = blk < KHo;                        // m < 16 HB: a row of the h1 / h2 images exists
    const bool mvalid = hslot && m < H;
    const int mc = mvalid ? m : 0;
    const float b3f = fvalid ? P[G::ob3 + fc] : 0.f;
    const float vdrf = (fvalid && (k_drift == DRIFT_DIAG || k_drift == DRIFT_DWELL)) ? a.drift[fc] : 0.f;
    const float vrunf = (fvalid && k_run == RUN_DIAGQ) ? a.runcost[fc] : 0.f;
    const float vtermf = fvalid ? a.term[fc] : 0.f;
    const float b1m = mvalid ? P[G::ob1 + mc] : 0.f;
    const float w1tm = mvalid ? P[G::oW1 + mc * (D + 1)] : 0.f;
    const float b2m = mvalid ? P[G::ob2 + mc] : 0.f;

    // ---- this wave's weight slices as B operands: lane `wrow` of fragment (slab sl, owned input kk) holds W[64 sl + wrow][kk-th input].
    //      A lane reads a COLUMN of each matrix (stride = a row): straight from global memory that is 64 cache lines per load
    //      and 104 loads per lane (20 us, an eighth of the kernel at N = 50).  Each matrix is staged once through LDS instead
    //      (coalesced copy, odd row stride so that the column reads are free of bank conflicts), in the area the partial
    //      products use later.
    float w1q[KO], aq[SD][KO], bq[SD][KO], w2q[KHo], w3q[SD][KHo];
    // Output rows of the d-row products (dt A x, W3 h2, B v) are numbered in OWNER order, rho = KO * owner wave + slot kk: the
    // 16 owner lanes-blocks of a wave then read 64 consecutive words of each partial (row = feature made the four-feature stride
    // of a group hit 8 of the 32 LDS banks: an 8-way conflict on every read of phases D and E)
    auto row_feature = [&](int rho, bool& valid) __attribute__((always_inline)) {
        const int w = rho / KO, kk = rho - w * KO;
        const int g = w + W * (kk >> 2);
        const int fr = 16 * (g >> 2) + 4 * (kk & 3) + (g & 3);
        valid = w < W && g < NG && fr < D;
        return valid ? fr : 0;
    };
    bool rowok[SD];
    int rowf[SD];
#pragma unroll
    for (int sl = 0; sl < SD; ++sl) rowf[sl] = row_feature(64 * sl + wrow, rowok[sl]);
    constexpr int LSD = D | 1, LSH = H | 1;              // LDS row strides (floats)
    static_assert((D > H ? D : H) * LSD <= Q::fRed && (D > H ? D : H) * LSH <= Q::fRed, "staging area");
    auto stage = [&](const float* __restrict__ M, int R, int C, int gstride, int lstride) __attribute__((always_inline)) {
        __syncthreads();                                 // the previous matrix has been read
        for (int idx = tid; idx < R * C; idx += Q::NT) {
            const int row = idx / C, col = idx - row * C;
            lds[row * lstride + col] = M[row * gstride + col];
        }
        __syncthreads();
    };
    int fko[KO];                                         // owned state features (clamped) and their validity, per slot
    bool oko[KO];
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
        const int g = wave + W * (kk >> 2);
        const int fk = 16 * (g >> 2) + 4 * (kk & 3) + (g & 3);
        oko[kk] = g < NG && fk < D;
        fko[kk] = oko[kk] ? fk : 0;
    }
    stage(P + G::oW1 + 1, H, D, D + 1, LSD);            // W1 without its time column
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
        const bool in = oko[kk] && wrow < H;
        const float v = lds[(in ? wrow : 0) * LSD + fko[kk]];
        w1q[kk] = in ? v : 0.f;
    }
#pragma unroll
    for (int sl = 0; sl < SD; ++sl)
#pragma unroll
        for (int kk = 0; kk < KO; ++kk) { aq[sl][kk] = 0.f; bq[sl][kk] = 0.f; }
    if (denseA) {                                        // (wave-uniform: the pointer means something else otherwise)
        stage(a.drift, D, D, D, LSD);
#pragma unroll
        for (int sl = 0; sl < SD; ++sl)
#pragma unroll
            for (int kk = 0; kk < KO; ++kk) {
                const bool in = oko[kk] && rowok[sl];
                const float v = lds[rowf[sl] * LSD + fko[kk]];
                aq[sl][kk] = in ? dt * v : 0.f;
            }
    }
    if (denseB) {
        stage(a.sigma, D, D, D, LSD);
#pragma unroll
        for (int sl = 0; sl < SD; ++sl)
#pragma unroll
            for (int kk = 0; kk < KO; ++kk) {
                const bool in = oko[kk] && rowok[sl];
                const float v = lds[rowf[sl] * LSD + fko[kk]];
                bq[sl][kk] = in ? v : 0.f;
            }
    }
    stage(P + G::oW2, H, H, H, LSH);
#pragma unroll
    for (int kk = 0; kk < KHo; ++kk) {
        const int mk = KHo * wave + kk;
        const bool in = mk < H && wrow < H;
        const float v = lds[(in ? wrow : 0) * LSH + (mk < H ? mk : 0)];
        w2q[kk] = in ? v : 0.f;
    }
    stage(P + G::oW3, D, H, H, LSH);
#pragma unroll
    for (int sl = 0; sl < SD; ++sl)
#pragma unroll
        for (int kk = 0; kk < KHo; ++kk) {
            const int mk = KHo * wave + kk;
            const bool in = mk < H && rowok[sl];
            const float v = lds[rowf[sl] * LSH + (mk < H ? mk : 0)];
            w3q[sl][kk] = in ? v : 0.f;
        }
    __syncthreads();                                     // the staging area becomes the partial-product buffers

    // ---- the four trajectories of this workgroup: columns 4 c .. 4 c + 3 of 16-trajectory tile t16
    const int t16 = blockIdx.x >> 2, j16 = 4 * (blockIdx.x & 3) + tr;
    const int k = t16 * 16 + j16;
    const bool kvalid = k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    // float offsets of the owned elements inside a path block (register-image layout of Geo: row 4 b + r, column 16 q + j)
    const int offX = G::pX + (4 * fb + r) * 64 + 16 * fq + j16;
    const int offXi = G::pXi + (4 * fb + r) * 64 + 16 * fq + j16;
    const int offH = (4 * (m >> 4) + ((m >> 2) & 3)) * 64 + 16 * (m & 3) + j16;
    const bool storing = a.store_path != 0;
    const float store_cxi = (a.store_path == 3) ? 0.f : 1.f;                    // image in the xi slot: c_xi xi + c_z Z
    const float store_cz = (a.store_path == 3) ? 1.f : (a.store_path == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));

    float x = 0.f;                                       // X_0 (solver.py:365-367)
    {
        const float v = a.x0[(size_t)(kvalid ? k : 0) * a.x0_stride + fc];
        x = (fvalid && kvalid) ? v : 0.f;
    }
    float Yw = 0.f, Fw = 0.f, ULw = 0.f;                 // this lane's partial of Y, of the running-cost integral and of u_L2

    float* partH1 = lds + Q::pH1;
    float* partDR = lds + Q::pDR;
    float* partH2 = lds + Q::pH2;
    float* partZ = lds + Q::pZ;
    float* partBV = lds + Q::pBV;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // own slot of the partial buffers (lane `wrow` writes rows 64 sl + wrow, all four trajectories: one ds_write_b128)
    f32x4* myH1 = reinterpret_cast<f32x4*>(partH1) + wave * RH + wrow;
    f32x4* myH2 = reinterpret_cast<f32x4*>(partH2) + wave * RH + wrow;
    f32x4* myDR = reinterpret_cast<f32x4*>(partDR) + wave * RD + wrow;
    f32x4* myZ = reinterpret_cast<f32x4*>(partZ) + wave * RD + wrow;
    f32x4* myBV = reinterpret_cast<f32x4*>(partBV) + wave * RD + wrow;
    // owner reads: element (row, tr) of each wave's partial
    const int rdH = mc * 4 + tr, rdD = (KO * wave + (blk < KO ? blk : 0)) * 4 + tr;

    // ---- Brownian increments.  Philox + Box-Muller are ~150 VALU instructions whichever lanes are active; done by every
    //      wave for its own features they cost 1.0 k of the 6.1 k cycles of a step (two waves per SIMD, nothing to hide them
    //      behind).  NP extra waves produce them instead, one step AHEAD: lane (blk, i) of producer p makes the call of group
    //      16 p + blk for trajectory i (same counters as every other forward kernel) and leaves its four values in LDS; the
    //      producers only meet the others at the barriers, so their instructions fill issue slots the latency-bound phases
    //      leave empty.  Supplied noise (parity runs) is loaded by the owner lanes as before.
    float* xibuf = lds + Q::fXi;
    const bool philox = FAST || a.noise_mode == NOISE_PHILOX;
    const bool has_uref = !FAST && a.uref != nullptr;
    auto produce = [&](int n) __attribute__((always_inline)) {
        const int gp = 16 * (wave_id - W) + blk;
        if (gp < NG) {
            const f32x4 z4 = philox_block(kglob, (uint32_t)n, (uint32_t)gp, iter_now, a.seed_lo, a.seed_hi);
            *reinterpret_cast<f32x4*>(xibuf + (n & 1) * Q::XIB + (gp * 4 + tr) * 4) = z4;
        }
    };
    if (producer) {
        __builtin_amdgcn_s_setprio(0);                   // fill idle issue slots only
        if (philox) produce(0);
        __syncthreads();
#pragma unroll 1
        for (int n = 0; n < a.N; ++n) {
            __syncthreads();                             // (1) every owner has read the increments of step n - 1
            if (philox && n + 1 < a.N) produce(n + 1);
            __syncthreads();                             // (2)
            __syncthreads();                             // (3)
            if (denseB) __syncthreads();                 // (4)
        }
        __syncthreads();                                 // the final reduction's barrier
        return;
    }
    __syncthreads();                                     // increments of step 0 are in LDS
    __builtin_amdgcn_s_setprio(2);                       // ahead of the producer wave that shares this SIMD
    const float* xird = xibuf + (gam < NG ? (gam * 4 + tr) * 4 + r : 0);
#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#pragma unroll 1
    for (int n = 0; n < a.N; ++n) {
        PSP_STAMP(qs0);
        float tn = (float)n * dt;
        if constexpr (!FAST) { if (a.tfeat) tn = a.tfeat[n]; }
        float* pblk = a.path + ((size_t)n * a.ntile16 + t16) * (size_t)G::PB;
        // ---- A: partial products of W1 x (h1 pre-activation) over the own state features
        if (storing && gvalid) pblk[offX] = x;
        {
            f32x4 h1a = zero4, h1b = zero4;              // two accumulators: half the dependent-chain length
            static_for<0, KO>([&](auto kk) {
                if constexpr (decltype(kk)::value & 1) h1b = mfma4<decltype(kk)::value>(x, w1q[decltype(kk)::value], h1b);
                else h1a = mfma4<decltype(kk)::value>(x, w1q[decltype(kk)::value], h1a);
            });
            *myH1 = h1a + h1b;
        }
        PSP_STAMP(qa1);
        __syncthreads();
        PSP_STAMP(qb0);
        // The drift product dt A x needs X_n only and its result only at the end of the step: its MFMAs (2 SD per owned feature)
        // go in FRONT of phases B and C, where the matrix pipe would otherwise idle while the partial sums come back from LDS
        constexpr int SDB = (SD + 1) / 2;                // slabs of the drift product done in phase B, the rest in phase C
        auto drift_slabs = [&](auto lo, auto hi) __attribute__((always_inline)) {
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            if constexpr (HI > LO) {
                f32x4 dr[HI - LO][2];                    // two accumulators per slab (even / odd features)
#pragma unroll
                for (int sl = 0; sl < HI - LO; ++sl) { dr[sl][0] = zero4; dr[sl][1] = zero4; }
                static_for<0, KO>([&](auto kk) {
#pragma unroll
                    for (int sl = 0; sl < HI - LO; ++sl)
                        dr[sl][decltype(kk)::value & 1] =
                            mfma4<decltype(kk)::value>(x, aq[LO + sl][decltype(kk)::value], dr[sl][decltype(kk)::value & 1]);
                });
#pragma unroll
                for (int sl = 0; sl < HI - LO; ++sl) myDR[64 * (LO + sl)] = dr[sl][0] + dr[sl][1];
            }
        };
        // ---- B: h1 = tanh(W1 [t, x] + b1) for the own hidden units, partial products of W2 h1
        {
            float hp[W];
#pragma unroll
            for (int p = 0; p < W; ++p) hp[p] = partH1[p * RH * 4 + rdH];
            if (denseA) drift_slabs(std::integral_constant<int, 0>{}, std::integral_constant<int, SDB>{});
            float h1 = fmaf(tn, w1tm, b1m);
#pragma unroll
            for (int p = 0; p < W; ++p) h1 += hp[p];
            h1 = mvalid ? tanh_f32(h1) : 0.f;             // padded rows of the images hold zeros
            if (storing && hslot) pblk[G::pH1 + offH] = h1;
            f32x4 h2a = zero4, h2b = zero4;              // two accumulators: the chain is only KHo long
            static_for<0, KHo>([&](auto kk) {
                if constexpr (decltype(kk)::value & 1) h2b = mfma4<decltype(kk)::value>(h1, w2q[decltype(kk)::value], h2b);
                else h2a = mfma4<decltype(kk)::value>(h1, w2q[decltype(kk)::value], h2a);
            });
            *myH2 = h2a + h2b;
        }
        PSP_STAMP(qb1);
        __syncthreads();
        PSP_STAMP(qc0);
        // ---- C: h2 = tanh(W2 h1 + b2), partial products of W3 h2
        {
            float hp[W];
#pragma unroll
            for (int p = 0; p < W; ++p) hp[p] = partH2[p * RH * 4 + rdH];
            if (denseA) drift_slabs(std::integral_constant<int, SDB>{}, std::integral_constant<int, SD>{});
            float h2 = b2m;
#pragma unroll
            for (int p = 0; p < W; ++p) h2 += hp[p];
            h2 = mvalid ? tanh_f32(h2) : 0.f;
            if (storing && hslot) pblk[G::pH2 + offH] = h2;
            f32x4 zp[SD];
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) zp[sl] = zero4;
            static_for<0, KHo>([&](auto kk) {
#pragma unroll
                for (int sl = 0; sl < SD; ++sl) zp[sl] = mfma4<decltype(kk)::value>(h2, w3q[sl][decltype(kk)::value], zp[sl]);
            });
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) myZ[64 * sl] = zp[sl];
        }
        // Brownian increment of the own feature: left in LDS by the producer waves a step ago (or loaded: parity runs)
        float xi = xird[(n & 1) * Q::XIB];
        if constexpr (!FAST) { if (!philox) xi = a.xi[((size_t)(n + 1) * a.K_local + (kvalid ? k : 0)) * D + fc]; }
        xi = (fvalid && (philox || kvalid)) ? xi : 0.f;
        PSP_STAMP(qc1);
        __syncthreads();
        PSP_STAMP(qd0);
        // ---- D: Z = W3 h2 + b3 for the own feature, row-sum terms, increment v, partial products of B v
        float Z = b3f, drs = 0.f;
        {
            float zp[W], dp[W];
#pragma unroll
            for (int p = 0; p < W; ++p) zp[p] = partZ[p * RD * 4 + rdD];
            if (denseA) {
#pragma unroll
                for (int p = 0; p < W; ++p) dp[p] = partDR[p * RD * 4 + rdD];
            }
#pragma unroll
            for (int p = 0; p < W; ++p) Z += zp[p];
            if (denseA) {
#pragma unroll
                for (int p = 0; p < W; ++p) drs += dp[p];
            }
        }
        Z = fvalid ? Z : 0.f;
        if (storing && a.store_path != 4 && gvalid) pblk[offXi] = store_cxi * xi + store_cz * Z;     // 1: xi, 2: xi - sqrt(dt) Z, 3: Z (hjb_fwd_kernel)
        float UL = 0.f;
        if (has_uref) {                                  // u_L2 logging: |-Z_n - u*(t_n)|^2 (solver.py:491-494)
            const float e = fvalid ? Z + a.uref[(size_t)n * D + fc] : 0.f;
            UL = e * e;
        }
        const float v = k_adaptive ? (sqdt * xi - dt * Z) : (sqdt * xi);        // v = c dt + xi sqrt(dt)
        // X_{n+1} = X + b(X) dt + sigma v   (solver.py:471-472)
        float xn = x;
        if (denseA) xn += drs;
        else if (k_drift == DRIFT_DIAG) xn += dt * (vdrf * x);
        else if (k_drift == DRIFT_DWELL) xn -= dt * (4.0f * vdrf * (x * (x * x - 1.0f)));
        if (denseB) {
            f32x4 bv[SD];
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) bv[sl] = zero4;
            static_for<0, KO>([&](auto kk) {
#pragma unroll
                for (int sl = 0; sl < SD; ++sl) bv[sl] = mfma4<decltype(kk)::value>(v, bq[sl][decltype(kk)::value], bv[sl]);
            });
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) myBV[64 * sl] = bv[sl];
            PSP_STAMP(qd1);
            __syncthreads();
            PSP_STAMP(qe0);
            PSP_ACC(3, qd1, qd0);   // D: Z, v, sigma product
            PSP_ACC(5, qe0, qd1);   // wait at barrier 4
#pragma unroll
            for (int p = 0; p < W; ++p) xn += partBV[p * RD * 4 + rdD];
        } else if (k_sigma == SIGMA_SCALE) {
            xn += a.sigma_scale * v;
        } else {
            xn += v;
        }
        x = fvalid ? xn : 0.f;
        // running cost f(X_{n+1}) and this lane's share of the Y update (solver.py:477-478): linear in the row sums
        const float fX = vrunf * x * x;
        const float S = Z * Z;
        const float term = (k_loss == LOSS_RELENT) ? -(0.5f * S + fX) * dt          // Y carries -Zsum (hjb_fwd_kernel)
                           : (k_adaptive ? (fX - 0.5f * S) : (fX + 0.5f * S)) * dt + (Z * xi) * sqdt;
        Yw += term;
        Fw = fmaf(fX, dt, Fw);
        ULw = fmaf(UL, dt, ULw);
        PSP_STAMP(qs1);
        PSP_ACC(0, qa1, qs0);       // A: X store, W1 + drift products
        PSP_ACC(1, qb1, qb0);       // B: h1, W2 product
        PSP_ACC(2, qc1, qc0);       // C: h2, W3 product, Philox
        PSP_ACC(4, (qb0 - qa1) + (qc0 - qb1), (qc1 - qd0));   // waits at barriers 1-3
        PSP_ACC(6, qs1, qs0);       // whole step
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0) {
        stamps[7] = (unsigned long long)a.N;
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
    }
#endif

    // ---- terminal cost of the own feature; the lanes of a trajectory meet per wave (shuffles), the waves in LDS
    float g;
    if (a.term_kind == TERM_LINEAR) g = vtermf * x;
    else if (a.term_kind == TERM_DIAGQ) g = vtermf * x * x;
    else g = vtermf * (x - 1.0f) * (x - 1.0f);
    g = fvalid ? g : 0.f;
    if (a.XN && kvalid && fvalid) a.XN[(size_t)k * D + f] = x;
    const float Yp = bsum16(Yw), Fp = bsum16(Fw), gp = bsum16(g), Up = bsum16(ULw);
    float* red = lds + Q::fRed;                          // [wave][Y, F, g, U][trajectory]
    if (blk == 0) {
        red[wave * 16 + tr] = Yp;
        red[wave * 16 + 4 + tr] = Fp;
        red[wave * 16 + 8 + tr] = gp;
        red[wave * 16 + 12 + tr] = Up;
    }
    __syncthreads();
    if (wave == 0) {
        float Ys = 0.f, F = 0.f, gt = 0.f, Us = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) {                    // fixed order: bitwise reproducible
            Ys += red[w * 16 + tr]; F += red[w * 16 + 4 + tr]; gt += red[w * 16 + 8 + tr]; Us += red[w * 16 + 12 + tr];
        }
        const float Y = (a.y0 ? a.y0[0] : 0.f) + Ys;
        const float Dk = Y - gt;
        const bool out = kvalid && blk == 0;
        if (out) {
            a.D[k] = Dk;
            if (a.Fint) a.Fint[k] = F;
            if (a.uref) a.ul2[k] = Us;
            if (a.Yout) a.Yout[k] = Y;
        }
        double sD = out ? (double)Dk : 0.0, sD2 = out ? (double)Dk * (double)Dk : 0.0;
        sD += __shfl_xor(sD, 1); sD += __shfl_xor(sD, 2);
        sD2 += __shfl_xor(sD2, 1); sD2 += __shfl_xor(sD2, 2);
        if (lane == 0) { a.fwd_partial[2 * blockIdx.x] = sD; a.fwd_partial[2 * blockIdx.x + 1] = sD2; }
    }
}

// =======================================================================================
// Quad-trajectory adjoint sweep (gradients through the state path at the smallest K): hjb_adj_kernel walks one
// 16-trajectory tile per wave, 626 MFMAs per step on ONE SIMD -- at K = 1024 that is 64 waves and 0.42 ms of the 0.66 ms
// iteration with the reference's default flags.  The recursion of hjba_kernels.h has the forward's shape with transposed
// matrices,
//     lambda' --B^T--> q,  gZ = coefW W - dt q  --W3^T--> (.)(1 - h2^2) = dz2 --W2^T--> (.)(1 - h1^2) = dz1 --W1x^T--> + lambda' + A^T lambda'
// so it runs on the machinery of hjbq_fwd_kernel: four trajectories per workgroup, products split over eight waves along the
// contraction index, partial products reduced by the owner lanes through LDS, four barriers per step.  The images of a step
// (W, h2, h1, X_n: one dword per owner lane) are requested a step ahead.
// =======================================================================================
template <int D, int H>
__global__ __launch_bounds__(512) void hjbq_adj_kernel(const HjbArgs a) {
    using G = Geo<D, H>;
    using Q = GeoQ<D, H>;
    constexpr int NG = Q::NG, KO = Q::KO, KHo = Q::KHo, SD = Q::SD, RD = Q::RD, RH = Q::RH, W = Q::W;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, tr = lane & 3;
    const int wrow = lane;
    const float* __restrict__ P = a.params;
    const float dt = a.dt, sqdt = a.sqdt, rsq = 1.0f / a.sqdt;
    const bool denseA = a.drift_kind == DRIFT_DENSE, denseB = a.sigma_kind == SIGMA_DENSE;

    // ---- ownership as in hjbq_fwd_kernel
    const int s = blk >> 2, r = blk & 3;
    const int gam = wave + W * s;
    const int fb = gam >> 2, fq = gam & 3;
    const int f = 16 * fb + 4 * r + fq;
    const bool gvalid = blk < KO && gam < NG;
    const bool fvalid = gvalid && f < D;
    const int fc = fvalid ? f : 0;
    const int m = KHo * wave + blk;
    const bool hslot = blk < KHo;
    const bool mvalid = hslot && m < H;
    const int mc = mvalid ? m : 0;
    const float vdrf = (fvalid && (a.drift_kind == DRIFT_DIAG || a.drift_kind == DRIFT_DWELL)) ? a.drift[fc] : 0.f;
    const float vrunf = (fvalid && a.runcost_kind == RUN_DIAGQ) ? a.runcost[fc] : 0.f;
    const float vtermf = fvalid ? a.term[fc] : 0.f;

    // ---- TRANSPOSED weight slices as B operands: lane `wrow` of fragment (slab, owned input kk) holds M^T[row][kk-th input]
    float w3Tq[KO], aTq[SD][KO], bTq[SD][KO], w2Tq[KHo], w1Tq[SD][KHo];
    auto row_feature = [&](int rho, bool& valid) __attribute__((always_inline)) {
        const int w = rho / KO, kk = rho - w * KO;
        const int g = w + W * (kk >> 2);
        const int fr = 16 * (g >> 2) + 4 * (kk & 3) + (g & 3);
        valid = w < W && g < NG && fr < D;
        return valid ? fr : 0;
    };
    bool rowok[SD];
    int rowf[SD];
#pragma unroll
    for (int sl = 0; sl < SD; ++sl) rowf[sl] = row_feature(64 * sl + wrow, rowok[sl]);
    constexpr int LSD = D | 1, LSH = H | 1;
    auto stage = [&](const float* __restrict__ M, int R, int C, int gstride, int lstride) __attribute__((always_inline)) {
        __syncthreads();
        for (int idx = tid; idx < R * C; idx += 512) {
            const int row = idx / C, col = idx - row * C;
            lds[row * lstride + col] = M[row * gstride + col];
        }
        __syncthreads();
    };
    int fko[KO];
    bool oko[KO];
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
        const int g = wave + W * (kk >> 2);
        const int fk = 16 * (g >> 2) + 4 * (kk & 3) + (g & 3);
        oko[kk] = g < NG && fk < D;
        fko[kk] = oko[kk] ? fk : 0;
    }
    stage(P + G::oW3, D, H, H, LSH);                     // W3 (d x H): W3^T[m][f] = W3[f][m]
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
        const bool in = oko[kk] && wrow < H;
        const float v = lds[fko[kk] * LSH + (in ? wrow : 0)];
        w3Tq[kk] = in ? v : 0.f;
    }
#pragma unroll
    for (int sl = 0; sl < SD; ++sl)
#pragma unroll
        for (int kk = 0; kk < KO; ++kk) { aTq[sl][kk] = 0.f; bTq[sl][kk] = 0.f; }
    if (denseA) {
        stage(a.drift, D, D, D, LSD);                    // (dt A)^T[row][f] = dt A[f][row]
#pragma unroll
        for (int sl = 0; sl < SD; ++sl)
#pragma unroll
            for (int kk = 0; kk < KO; ++kk) {
                const bool in = oko[kk] && rowok[sl];
                const float v = lds[fko[kk] * LSD + rowf[sl]];
                aTq[sl][kk] = in ? dt * v : 0.f;
            }
    }
    if (denseB) {
        stage(a.sigma, D, D, D, LSD);                    // B^T[row][f] = B[f][row]
#pragma unroll
        for (int sl = 0; sl < SD; ++sl)
#pragma unroll
            for (int kk = 0; kk < KO; ++kk) {
                const bool in = oko[kk] && rowok[sl];
                const float v = lds[fko[kk] * LSD + rowf[sl]];
                bTq[sl][kk] = in ? v : 0.f;
            }
    }
    stage(P + G::oW2, H, H, H, LSH);                     // W2^T[m][mk] = W2[mk][m]
#pragma unroll
    for (int kk = 0; kk < KHo; ++kk) {
        const int mk = KHo * wave + kk;
        const bool in = mk < H && wrow < H;
        const float v = lds[(mk < H ? mk : 0) * LSH + (in ? wrow : 0)];
        w2Tq[kk] = in ? v : 0.f;
    }
    stage(P + G::oW1 + 1, H, D, D + 1, LSD);            // W1 without its time column: W1x^T[row][mk] = W1[mk][1 + row]
#pragma unroll
    for (int sl = 0; sl < SD; ++sl)
#pragma unroll
        for (int kk = 0; kk < KHo; ++kk) {
            const int mk = KHo * wave + kk;
            const bool in = mk < H && rowok[sl];
            const float v = lds[(mk < H ? mk : 0) * LSD + rowf[sl]];
            w1Tq[sl][kk] = in ? v : 0.f;
        }
    __syncthreads();                                     // the staging area becomes the partial-product buffers

    const int t16 = blockIdx.x >> 2, j16 = 4 * (blockIdx.x & 3) + tr;
    const int k = t16 * 16 + j16;
    const bool kvalid = k < a.K_local;
    const int offX = G::pX + (4 * fb + r) * 64 + 16 * fq + j16;
    const int offXi = G::pXi + (4 * fb + r) * 64 + 16 * fq + j16;
    const int offH = (4 * (m >> 4) + ((m >> 2) & 3)) * 64 + 16 * (m & 3) + j16;
    const float mu = (kvalid && a.adj_mu) ? a.adj_mu[k] : 0.f;
    const float nu = (kvalid && a.adj_nu) ? a.adj_nu[k] : 0.f;
    const float coefW = (a.store_path == 3) ? nu * dt : mu * sqdt;
    const float wf = (mu + nu) * dt;                     // weight of grad f(X_{n+1})
    const float wT = a.adj_wT ? (kvalid ? a.adj_wT[k] : 0.f) : (nu - mu);      // weight of grad g(X_N) in lambda_N
    const bool need_x = a.runcost_kind == RUN_DIAGQ || a.drift_kind == DRIFT_DWELL;

    float* partU2 = lds + Q::pH1;                        // (the forward's buffers, same shapes)
    float* partA = lds + Q::pDR;
    float* partU1 = lds + Q::pH2;
    float* partL = lds + Q::pZ;
    float* partQ = lds + Q::pBV;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4* myU2 = reinterpret_cast<f32x4*>(partU2) + wave * RH + wrow;
    f32x4* myU1 = reinterpret_cast<f32x4*>(partU1) + wave * RH + wrow;
    f32x4* myA = reinterpret_cast<f32x4*>(partA) + wave * RD + wrow;
    f32x4* myL = reinterpret_cast<f32x4*>(partL) + wave * RD + wrow;
    f32x4* myQ = reinterpret_cast<f32x4*>(partQ) + wave * RD + wrow;
    const int rdH = mc * 4 + tr, rdD = (KO * wave + (blk < KO ? blk : 0)) * 4 + tr;

    // lambda_N = wT grad g(X_N);  X_N also serves grad f at the last step
    float xn1 = (fvalid && kvalid) ? a.XN[(size_t)k * D + fc] : 0.f;
    float lam;
    {
        float gg;
        if (a.term_kind == TERM_LINEAR) gg = vtermf;
        else if (a.term_kind == TERM_DIAGQ) gg = 2.0f * vtermf * xn1;
        else gg = 2.0f * vtermf * (xn1 - 1.0f);
        lam = fvalid ? wT * gg : 0.f;
    }
    // images of step n, requested during step n + 1 (one dword per owner lane and image)
    auto blk_ptr = [&](int nn) __attribute__((always_inline)) { return a.path + ((size_t)nn * a.ntile16 + t16) * (size_t)G::PB; };
    float wimg_n, xin_n = 0.f, h2_n, h1_n;
    {
        const float* pb = blk_ptr(a.N - 1);
        wimg_n = pb[gvalid ? offXi : 0];
        h2_n = pb[G::pH2 + (hslot ? offH : 0)];
        h1_n = pb[G::pH1 + (hslot ? offH : 0)];
        if (need_x) xin_n = pb[gvalid ? offX : 0];
    }

#pragma unroll 1
    for (int n = a.N - 1; n >= 0; --n) {
        float* pblk = blk_ptr(n);
        const float wimg = gvalid ? wimg_n : 0.f, xin = gvalid ? xin_n : 0.f;
        const float h2 = hslot ? h2_n : 0.f, h1 = hslot ? h1_n : 0.f;
        {
            const float* pb = blk_ptr(n > 0 ? n - 1 : 0);          // (step 0 re-reads its own block: unused)
            wimg_n = pb[gvalid ? offXi : 0];
            h2_n = pb[G::pH2 + (hslot ? offH : 0)];
            h1_n = pb[G::pH1 + (hslot ? offH : 0)];
            if (need_x) xin_n = pb[gvalid ? offX : 0];
        }
        // lambda' = lambda_{n+1} + (mu + nu) dt grad f(X_{n+1})
        const float lamp = fvalid ? lam + (2.0f * wf) * (vrunf * xn1) : 0.f;
        // ---- A: partial products of B^T lambda' and (dt A)^T lambda' over the own state features
        {
            f32x4 qp[SD][2], ap[SD][2];
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) { qp[sl][0] = zero4; qp[sl][1] = zero4; ap[sl][0] = zero4; ap[sl][1] = zero4; }
            if (denseB) {
                static_for<0, KO>([&](auto kk) {
#pragma unroll
                    for (int sl = 0; sl < SD; ++sl)
                        qp[sl][decltype(kk)::value & 1] =
                            mfma4<decltype(kk)::value>(lamp, bTq[sl][decltype(kk)::value], qp[sl][decltype(kk)::value & 1]);
                });
#pragma unroll
                for (int sl = 0; sl < SD; ++sl) myQ[64 * sl] = qp[sl][0] + qp[sl][1];
            }
            if (denseA) {
                static_for<0, KO>([&](auto kk) {
#pragma unroll
                    for (int sl = 0; sl < SD; ++sl)
                        ap[sl][decltype(kk)::value & 1] =
                            mfma4<decltype(kk)::value>(lamp, aTq[sl][decltype(kk)::value], ap[sl][decltype(kk)::value & 1]);
                });
#pragma unroll
                for (int sl = 0; sl < SD; ++sl) myA[64 * sl] = ap[sl][0] + ap[sl][1];
            }
        }
        if (denseA || denseB) __syncthreads();
        // ---- B: q = B^T lambda', gZ = coefW W - dt q (back into the xi slot as gZ / sqrt(dt)), partial products of W3^T gZ
        float qv, drl = 0.f;
        if (denseB) {
            qv = 0.f;
#pragma unroll
            for (int p = 0; p < W; ++p) qv += partQ[p * RD * 4 + rdD];
        } else if (a.sigma_kind == SIGMA_SCALE) {
            qv = a.sigma_scale * lamp;
        } else {
            qv = lamp;
        }
        if (denseA) {
#pragma unroll
            for (int p = 0; p < W; ++p) drl += partA[p * RD * 4 + rdD];
        }
        const float gz = fvalid ? coefW * wimg - dt * qv : 0.f;
        if (gvalid) pblk[offXi] = rsq * gz;
        {
            f32x4 ua = zero4, ub = zero4;
            static_for<0, KO>([&](auto kk) {
                if constexpr (decltype(kk)::value & 1) ub = mfma4<decltype(kk)::value>(gz, w3Tq[decltype(kk)::value], ub);
                else ua = mfma4<decltype(kk)::value>(gz, w3Tq[decltype(kk)::value], ua);
            });
            *myU2 = ua + ub;
        }
        __syncthreads();
        // ---- C: dz2 = (W3^T gZ)(1 - h2^2) for the own hidden units, partial products of W2^T dz2
        {
            float u2 = 0.f;
#pragma unroll
            for (int p = 0; p < W; ++p) u2 += partU2[p * RH * 4 + rdH];
            const float dz2 = mvalid ? u2 * (1.0f - h2 * h2) : 0.f;
            f32x4 ua = zero4, ub = zero4;
            static_for<0, KHo>([&](auto kk) {
                if constexpr (decltype(kk)::value & 1) ub = mfma4<decltype(kk)::value>(dz2, w2Tq[decltype(kk)::value], ub);
                else ua = mfma4<decltype(kk)::value>(dz2, w2Tq[decltype(kk)::value], ua);
            });
            *myU1 = ua + ub;
        }
        __syncthreads();
        // ---- D: dz1 = (W2^T dz2)(1 - h1^2), partial products of W1x^T dz1
        {
            float u1 = 0.f;
#pragma unroll
            for (int p = 0; p < W; ++p) u1 += partU1[p * RH * 4 + rdH];
            const float dz1 = mvalid ? u1 * (1.0f - h1 * h1) : 0.f;
            f32x4 lp[SD];
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) lp[sl] = zero4;
            static_for<0, KHo>([&](auto kk) {
#pragma unroll
                for (int sl = 0; sl < SD; ++sl) lp[sl] = mfma4<decltype(kk)::value>(dz1, w1Tq[sl][decltype(kk)::value], lp[sl]);
            });
#pragma unroll
            for (int sl = 0; sl < SD; ++sl) myL[64 * sl] = lp[sl];
        }
        __syncthreads();
        // ---- E: lambda_n = lambda' + dt b'(X_n)^T lambda' + W1x^T dz1
        float ln = lamp;
#pragma unroll
        for (int p = 0; p < W; ++p) ln += partL[p * RD * 4 + rdD];
        if (denseA) ln += drl;
        else if (a.drift_kind == DRIFT_DIAG) ln += dt * (vdrf * lamp);
        else if (a.drift_kind == DRIFT_DWELL) ln -= dt * (4.0f * vdrf * ((3.0f * xin * xin - 1.0f) * lamp));   // b' = -4 kappa (3 x^2 - 1)
        lam = fvalid ? ln : 0.f;
        if (need_x) xn1 = xin;                           // X_n is X_{n+1} of the next (earlier) step
    }
}

template <int D, int H>
struct HjbqLaunch {
    static int lds_bytes() { return GeoQ<D, H>::fits ? GeoQ<D, H>::lds_floats * 4 : (1 << 30); }
    template <int FAST>
    static hipError_t fwd_as(const HjbArgs& a, int grid, hipStream_t s) {
        const int bytes = lds_bytes();
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbq_fwd_kernel<D, H, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbq_fwd_kernel<D, H, FAST>), dim3(grid), dim3(GeoQ<D, H>::NT), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t adj(const HjbArgs& a, int grid, hipStream_t s) {
        if constexpr (GeoQ<D, H>::fits) {
            const int bytes = lds_bytes();
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbq_adj_kernel<D, H>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbq_adj_kernel<D, H>), dim3(grid), dim3(512), bytes, s, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    static hipError_t fwd(const HjbArgs& a, int grid, hipStream_t s) {
        if constexpr (GeoQ<D, H>::fits) {
            const bool fast = a.noise_mode == NOISE_PHILOX && a.uref == nullptr && a.tfeat == nullptr;
            const bool spec = spec_enabled() && fast && a.drift_kind == DRIFT_DENSE && a.sigma_kind == SIGMA_DENSE && a.adaptive && a.runcost_kind == RUN_ZERO &&
                              a.loss_kind != LOSS_RELENT;
            return spec ? fwd_as<2>(a, grid, s) : fast ? fwd_as<1>(a, grid, s) : fwd_as<0>(a, grid, s);
        } else {
            return hipErrorInvalidValue;
        }
    }
};

}  // namespace psp
