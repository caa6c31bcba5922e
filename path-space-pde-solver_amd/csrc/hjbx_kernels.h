// hjbx_kernels.h -- split-product backward of the narrow HJB family (psp_hjb_config.mlp_dtype = PSP_MLP_F16X3).
//
// Same gradient as hjb_bwd2_kernel (hjb_kernels.h: analytic MLP backward over all (n, k) samples of the path store), with every
// product as three f16 MFMAs per fp32 product (gemm_Tx's operand split: x = hi + lo / 2048).  The fp32 kernel is bound by the
// consumers' weight-gradient MFMAs (352 of the 452 v_mfma_f32_16x16x4_f32 per round on every SIMD); here a weight-gradient
// tile contracts TWO 16-sample blocks per v_mfma_f32_16x16x32_f16 (k = 8 g + e: e < 4 block c0, e >= 4 block c1, sample
// 4 g + (e & 3)), three instructions per tile and pair instead of eight fp32 ones, and the roles are cut so that no operand is
// split twice:
//   PRODUCERS (waves 0-3, one sample block each per round): G = w sqrt(dt) image, dz2 = (W3^T G)(1 - h2^2),
//     dz1 = (W2^T dz2)(1 - h1^2) register-chained in T layout (gemm_Txp), all four bias sums (db3, db2, db1, time column), and
//     the three panels written ONCE, already split, as f16 images (hi | lo) of the PAIR: feature-major, per feature and sample
//     quad g the four samples of block c0 then of block c1 (ds_write_b16, conflict-free), so that
//   CONSUMERS (waves 4-7) get an A operand (feature tile x 32 samples of the pair) as ONE 16-byte LDS read per lane, no VALU;
//     their B operands are the feature-on-lane reads of the path store (h2 / h1 block w, X blocks w and w + 4), split on
//     arrival: 32 values per pair and wave.  Wave w owns dW3[:, w], dW2[:, w] and dW1[:, {w, w + 4}].  For dW3 and dW2 the B
//     operand is a tanh output (|h| <= 1), so 2048 hi_h is exact in f16 and main and correction terms share ONE accumulator
//     (2048 a.b = hi_a (2048 hi_b) + hi_a lo_b + lo_a hi_b); dW1 keeps two.
// Magnitudes: G (and with it dz2, dz1) carries the trajectory weights w_k ~ 1 / K -- far below the f16 normal range, where the
// hi part would lose its bits.  Each workgroup scales G by a power of two (exact; the gradient is linear in G) and scales its
// partial gradient back when it is written.  The scale maps  max_k |w_k| * sqrt(dt) * I  into [1, 2): max_k |w_k| from a scan of
// ALL of this rank's weights by every workgroup (K_local floats, L2-resident; round 3 -- a first-round estimate failed silently
// when the first tiles carried zero weights, and overflowed on a weight 3e4 x the first round's), and I = 8 when the stored image
// is the Brownian increment itself (adaptive forward process, store_path 1: |xi| < 6 from Box-Muller on 24-bit uniforms), else
// the largest |image| of the workgroup's first round (non-adaptive / attached images carry Z or the adjoint gZ; the stated
// range condition there: no later image entry above 3e4 x that maximum).  Below the scale the split keeps its accuracy down to
// 2^-36 of the maximum (hi subnormal, lo still exact), i.e. weights 1e5 x smaller than the largest lose nothing that matters.
// One barrier per round swaps the two exchange buffers, as in hjb_bwd2_kernel.  LDS: W3^T and W2^T split tables, 2 x 2 pairs
// of (2 D + 4 * 16 HB) * 64 bytes (d = 100, H = 64: 158 736 bytes).  Reference lines: solver.py:468-472 (what carries a
// gradient), function_space.py:190-195 (the net).
#pragma once
#include "hjb_kernels.h"

namespace psp {

template <int D, int H>
struct GeoX {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB, KSD = G::KSD, KSH = G::KSH;
    static constexpr int T3 = 0, T2 = T3 + SplitGeo<KSD, DB>::floats(HB), EX = T2 + SplitGeo<KSH, HB>::floats(HB);   // floats
    // exchange area of one PAIR of sample blocks, in halves: G (hi, lo: D features), dz2, dz1 (hi, lo: 16 HB features); element
    // (feature f, block c of the pair, sample s) at f * 32 + ((s >> 2) ^ ((f >> 2) & 3)) * 8 + c * 4 + (s & 3)  (chunk swizzle: see the producers)
    static constexpr int hG = 0, hGl = hG + D * 32, hZ2 = hGl + D * 32, hZ2l = hZ2 + HB * 512, hZ1 = hZ2l + HB * 512,
                         hZ1l = hZ1 + HB * 512, PAIRH = hZ1l + HB * 512;
    static constexpr int RS = 16 * DB + 3 * 16 * HB;                   // bias-sum slots per producer (G | dz2 | dz1 | t dz1)
    static constexpr int NX = cdiv(DB, 4);                             // X blocks per consumer wave
    static constexpr int NT = DB + 2 * HB;                             // A tiles per pair: G, dz2, dz1
    static_assert(4 * RS <= 4 * (PAIRH / 2), "bias sums reuse the exchange area");
    static constexpr int GS = EX + 4 * (PAIRH / 2);                    // 4 floats: the producers' largest |image| of round 0,
    static int lds_floats() { return GS + 12; }                        // 8 floats: per-wave largest |w_k| of the weight scan
};

template <int NB>
__device__ __forceinline__ void split_panel(const f32x4 (&v)[NB], f16x4 (&hi)[NB], f16x4 (&lo)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if ((NB & 1) && b == NB - 1) split4c(v[b], hi[b], lo[b]);       // (a trailing odd block may feed 16x16x16 MFMAs directly: hjb_kernels.h split4c)
        else split4(v[b], hi[b], lo[b]);
    }
}

// gemm_Tx with the input panel already split (hi / lo per 16-feature block; `last` = the fp32 value in[INB-1][0] for the exact
// trailing k-step of SplitGeo::ODD_F32)
template <int MB, int KS, int INB>
__device__ __forceinline__ void gemm_Txp(f32x4 (&acc)[MB], const float* wlds, const f16x4 (&hi)[INB], const f16x4 (&lo)[INB],
                                         float last, int lane) {
    using SG = SplitGeo<KS, INB>;
    constexpr int NS = SG::NS;
    lane = opaque_i(lane);
    f32x4 corr[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) corr[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NS > 0) {
        const f16x8* tbl = reinterpret_cast<const f16x8*>(wlds) + lane;
        constexpr int NU = NS * MB, CU = 1, NCH = cdiv(NU, CU);
        f16x8 ah[2][CU], al[2][CU];
#pragma unroll
        for (int kk = 0; kk < CU; ++kk)
            if (kk < NU) {
                ah[0][kk] = tbl[((kk % MB) * SG::per_mb + (kk / MB) * 512) / 4];
                al[0][kk] = tbl[((kk % MB) * SG::per_mb + (kk / MB) * 512) / 4 + 64];
            }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) {
#pragma unroll
                for (int kk = 0; kk < CU; ++kk) {
                    const int u = (c + 1) * CU + kk;
                    if (u < NU) {
                        ah[(c + 1) & 1][kk] = tbl[((u % MB) * SG::per_mb + (u / MB) * 512) / 4];
                        al[(c + 1) & 1][kk] = tbl[((u % MB) * SG::per_mb + (u / MB) * 512) / 4 + 64];
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < CU; ++kk) {
                const int u = c * CU + kk;
                if (u < NU) {
                    const int S = u / MB, mb = u % MB;
                    const f16x8 bh = __builtin_shufflevector(hi[2 * S], hi[2 * S + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const f16x8 bl = __builtin_shufflevector(lo[2 * S], lo[2 * S + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c & 1][kk], bh, acc[mb], 0, 0, 0);
                    corr[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c & 1][kk], bl, corr[mb], 0, 0, 0);
                    corr[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c & 1][kk], bh, corr[mb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(kFenceMask);
        }
    }
    if constexpr (SG::ODD_F32) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(wlds[mb * SG::per_mb + NS * 512 + lane], last, acc[mb]);
    }
    if constexpr (SG::ODD_H16) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const f16x4* p = reinterpret_cast<const f16x4*>(wlds + mb * SG::per_mb + NS * 512) + lane;
            const f16x4 a_hi = p[0], a_lo = p[64];
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, hi[INB - 1], acc[mb], 0, 0, 0);
            corr[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_hi, lo[INB - 1], corr[mb], 0, 0, 0);
            corr[mb] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_lo, hi[INB - 1], corr[mb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = acc[mb] + kSplitInv * corr[mb];
}

// REGEN: the xi image is regenerated from the Philox counters (psp_hjb_config.store_path 4) -- its own instance, so that the
// landing registers of the stored image and the deeper h2 / h1 lead of the regenerating schedule do not add up in one allocation
template <int D, int H, bool REGEN = false>
__global__ __launch_bounds__(512) void hjb_bwd3_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    GradCheck<true> gchk;                                      // backward side of the range guard (hjb_kernels.h)
    using G = Geo<D, H>;
    using X = GeoX<D, H>;
    constexpr int DB = G::DB, HB = G::HB, KSD = G::KSD, KSH = G::KSH, NX = X::NX, RS = X::RS, PAIRH = X::PAIRH, NT = X::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const bool producer = wave < 4;
    const int sub = wave & 3;                         // producer: block within the round; consumer: tile column
    const float* __restrict__ P = a.params;

    const double invK = 1.0 / (double)a.K_global;
    const float meanD = (a.loss_kind == LOSS_LOGVAR) ? (float)(a.sums[0] * invK) : 0.f;
    const float coef = (float)(2.0 * invK);
    {   // largest |w_k| of this rank (header comment); NaN entries drop out of fmaxf
        float wm = 0.f;
        for (int k = tid; k < a.K_local; k += nthr) {
            const float dk = a.D[k];
            wm = fmaxf(wm, fabsf(a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)));
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if (lane == 0) lds[X::GS + 4 + wave] = wm;
    }
    stage_aop_x3<KSD, DB>(lds + X::T3, HB, tid, nthr, [&](int row, int col) {      // W3^T and W2^T (producers)
        return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
    stage_aop_x3<KSH, HB>(lds + X::T2, HB, tid, nthr, [&](int row, int col) {
        return (row < H && col < H) ? P[G::oW2 + col * H + row] : 0.f; });
    __syncthreads();
    _Float16* exh = reinterpret_cast<_Float16*>(lds + X::EX);           // [2 buffers][2 pairs][PAIRH] halves
    // power-of-two scale of G (header comment): max |w| of the scan x sqrt(dt) x image bound
    const bool xi_image = a.adaptive != 0 && (a.store_path == 1 || a.store_path == 4);
    auto g_scale = [&](float& gs, float& ginv) __attribute__((always_inline)) {
        const float* slot = lds + X::GS;
        float imax = fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
        if (xi_image || !(imax >= 1.1754944e-38f)) imax = xi_image ? 8.0f : 1.0f;
        const float wmax = fmaxf(fmaxf(fmaxf(slot[4], slot[5]), fmaxf(slot[6], slot[7])),
                                 fmaxf(fmaxf(slot[8], slot[9]), fmaxf(slot[10], slot[11])));
        const float amax = wmax * a.sqdt * imax;
        const unsigned e = (__float_as_uint(amax) >> 23) & 0xFFu;
        const bool ok = e >= 1u && e <= 253u;                             // zero / subnormal / non-finite: no scaling
        gs = ok ? __uint_as_float((254u - e) << 23) : 1.0f;
        ginv = ok ? __uint_as_float(e << 23) : 1.0f;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sqdt = a.sqdt, dt = a.dt;
    constexpr bool regen = REGEN;                               // xi from the Philox counters instead of the path store
    const uint32_t iter_now = a.iter_dev ? *a.iter_dev : a.iter;
    const long long nblk = (long long)a.N * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    const int R = (int)((nround - blockIdx.x + gridDim.x - 1) / gridDim.x);   // rounds of this workgroup (>= 1)

    if (producer) {
        // ================================================================================ producers
        // (the producers' chain sets the pace of a round -- stamps: the consumers idle 15 - 40 % at the barrier; s_setprio 3 for
        // the producers, which share each SIMD with one consumer wave, measured 1.4 % slower: not kept)
        f32x4 sG[DB], sZ2[HB], sZ1[HB], sT1[HB];
#pragma unroll
        for (int b = 0; b < DB; ++b) sG[b] = zero4;
#pragma unroll
        for (int m = 0; m < HB; ++m) { sZ2[m] = zero4; sZ1[m] = zero4; sT1[m] = zero4; }
        auto own_block = [&](int it2) __attribute__((always_inline)) {
            const long long b0 = ((long long)blockIdx.x + (long long)it2 * gridDim.x) * 4 + sub;
            return b0 < nblk ? b0 : -1LL;
        };
        auto path_of = [&](int it2) __attribute__((always_inline)) {
            const long long b0 = own_block(it2);
            return a.path + (size_t)(b0 >= 0 ? b0 : nblk - 1) * (size_t)G::PB + lane;
        };
        // (the register budget -- 256, with 76 registers of bias sums -- has no room for a second set of landing registers, and a
        // spilled register's reload would wait behind every load in flight)
        f32x4 xin[DB], h2n[HB], h1n[HB];
        float dkn;
        {
            const long long b0 = own_block(0);
            const long long blk = b0 >= 0 ? b0 : nblk - 1;
            const float* pb = path_of(0);
            const int k0 = (int)(blk % a.ntile16) * 16 + j;
            dkn = a.D[k0 < a.K_local ? k0 : 0];
            // Stored xi (REGEN = false): h2 of the current block is requested at the top of its round (used behind the first
            // product); h1 (used behind the second product) and the xi image and weight of the NEXT round behind the first product,
            // when the split G panel is dead.  The stamped producer spends ~3 000 of its 9 300 cycles per round waiting for h2 / h1
            // and sets the pace of the kernel (the consumers idle 28 % at the barrier), but a whole round of lead for them next to
            // the xi landing registers spills inside the loop (tried in round 3: 6 - 19 dwords).
            // store_path 4 (REGEN): xi comes from the Philox counters at the top of each round (no landing registers, no latency)
            // and every panel is requested a whole round ahead -- the loads for the NEXT block stand right where the current
            // block's registers are consumed (h2 behind the first tanh', h1 behind the second): same registers, issue order =
            // consumption order (vmcnt is in order).
            if constexpr (regen) {
#pragma unroll
                for (int b = 0; b < DB; ++b) xin[b] = zero4;
            } else {
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xin[b][r] = pb[G::pXi + (4 * b + r) * 64];
            }
            if constexpr (regen) {
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2n[m][r] = pb[G::pH2 + (4 * m + r) * 64];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1n[m][r] = pb[G::pH1 + (4 * m + r) * 64];
            }
        }
        float gs, ginv;
        {
            const long long b0 = own_block(0);
            const int k0 = (int)((b0 >= 0 ? b0 : nblk - 1) % a.ntile16) * 16 + j;
            const bool kv = b0 >= 0 && k0 < a.K_local;
            float am = 0.f;
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) am = fmaxf(am, kv ? fabsf(xin[b][r]) : 0.f);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) am = fmaxf(am, __shfl_xor(am, o));
            if (lane == 0) lds[X::GS + sub] = am;
            __syncthreads();                              // (A) pairs with the consumers' barrier behind their first loads
            g_scale(gs, ginv);
        }
#ifdef PSP_STAMPS
        unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int it = 0; it <= R; ++it) {
            PSP_STAMP(tp0);
            if (it < R) {
                const long long blk0 = own_block(it);
                const bool bvalid = blk0 >= 0;
                const long long blk = bvalid ? blk0 : nblk - 1;
                const int t16 = (int)(blk % a.ntile16);
                const int k = t16 * 16 + j;
                const bool kvalid = bvalid && k < a.K_local;
                const float tn = (float)(blk / a.ntile16) * dt;
                // exchange image: element (feature f, block c of the pair, sample s) at f * 32 + (((s >> 2) ^ ((f >> 2) & 3)) * 8
                // + c * 4 + (s & 3): the 16-byte chunk of a sample quad is XOR-swizzled with the feature row, so that the consumers'
                // ds_read_b128 of rows i, i + 4, i + 8, i + 12 (64 bytes apart: the same 16 banks) fall on four different bank
                // groups (round 3: the unswizzled reads cost the kernel 27 % LDS bank-conflict cycles).  (f >> 2) & 3 = r here.
                _Float16* ex0 = exh + ((it & 1) * 2 + (sub >> 1)) * PAIRH + (sub & 1) * 4 + (j & 3);
                _Float16* exr[4] = {ex0 + (((j >> 2) ^ 0) * 8), ex0 + (((j >> 2) ^ 1) * 8), ex0 + (((j >> 2) ^ 2) * 8), ex0 + (((j >> 2) ^ 3) * 8)};
                const float* pn = path_of(it + 1);
                const float* pc = path_of(it);
                const float dkc = dkn;
                if constexpr (!regen) {
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h2n[m][r] = pc[G::pH2 + (4 * m + r) * 64];
                }
                // LOSS_WEIGHTS: the caller supplies w_k = dLoss/dY_k directly in the D argument
                const float wk = kvalid ? (a.loss_kind == LOSS_WEIGHTS ? dkc : coef * (dkc - meanD)) : 0.f;
                f32x4 Gt[DB];
                if constexpr (regen) {
                    const uint32_t kg = (uint32_t)(a.k_offset + k), nstep = (uint32_t)(blk / a.ntile16);
                    // (q through an opaque copy: the first Philox round of the call index 4 b + q is loop-invariant, and seven
                    // hoisted products per lane were spilled and reloaded every round -- behind every path load in flight)
                    const int qo = opaque_i(q);
#pragma unroll
                    for (int b = 0; b < DB; ++b) {                  // one Philox call at a time (their temporaries do not pile up)
                        f32x4 xi = philox_block(kg, nstep, (uint32_t)(4 * b + qo), iter_now, a.seed_lo, a.seed_hi);
                        if (16 * b + 16 > D) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (16 * b + 4 * r + q >= D) xi[r] = 0.f;
                        }
                        {
#pragma clang fp contract(off)
                            Gt[b] = (wk * sqdt * gs) * xi;
                            sG[b] += Gt[b];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    {
#pragma clang fp contract(off)
#pragma unroll
                        for (int b = 0; b < DB; ++b) {
                            Gt[b] = (wk * sqdt * gs) * xin[b];      // adaptive: the (Z + c) dt term cancels
                            sG[b] += Gt[b];
                        }
                    }
                }
                PSP_STAMP(tp1);
                PSP_ACC(0, tp1, tp0);                 // h2 loads issued, weights -> G
                f16x4 Gh[DB], Gl[DB];
                split_panel<DB>(Gt, Gh, Gl);
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * b + 4 * r + q;
                        if (16 * b + 16 <= D || f < D) {
                            exr[r][X::hG + f * 32] = Gh[b][r];
                            exr[r][X::hGl + f * 32] = Gl[b][r];
                        }
                    }
                PSP_STAMP(tp2);
                PSP_ACC(1, tp2, tp1);                 // G split + image write
                f32x4 dz2[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) dz2[m] = zero4;
                gemm_Txp<HB, KSD, DB>(dz2, lds + X::T3, Gh, Gl, Gt[DB - 1][0], lane);
                {
                    // (no contraction here: the two instances of this kernel must round dz2 and the bias sums alike -- one of them
                    // had folded the product into the sum, 1 ulp in db2)
#pragma clang fp contract(off)
#pragma unroll
                    for (int m = 0; m < HB; ++m) { dz2[m] = dz2[m] * (1.0f - h2n[m] * h2n[m]); sZ2[m] += dz2[m]; }
                }
                if constexpr (regen) {
                    __builtin_amdgcn_sched_barrier(0);              // (the loads below overwrite h2n: not before it is consumed)
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h2n[m][r] = pn[G::pH2 + (4 * m + r) * 64];
                } else {
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h1n[m][r] = pc[G::pH1 + (4 * m + r) * 64];
                    const long long n0 = own_block(it + 1);
                    const long long nb1 = n0 >= 0 ? n0 : nblk - 1;
                    const int k1 = (int)(nb1 % a.ntile16) * 16 + j;
                    dkn = a.D[k1 < a.K_local ? k1 : 0];
#pragma unroll
                    for (int b = 0; b < DB; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xin[b][r] = pn[G::pXi + (4 * b + r) * 64];
                }
                PSP_STAMP(tp3);
                PSP_ACC(2, tp3, tp2);                 // W3^T G, tanh', h1 loads, next xi (loads or Philox)
                f16x4 Zh[HB], Zl[HB];
                split_panel<HB>(dz2, Zh, Zl);
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        exr[r][X::hZ2 + (16 * m + 4 * r + q) * 32] = Zh[m][r];
                        exr[r][X::hZ2l + (16 * m + 4 * r + q) * 32] = Zl[m][r];
                    }
                f32x4 dz1[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) dz1[m] = zero4;
                gemm_Txp<HB, KSH, HB>(dz1, lds + X::T2, Zh, Zl, dz2[HB - 1][0], lane);
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int m = 0; m < HB; ++m) {
                        dz1[m] = dz1[m] * (1.0f - h1n[m] * h1n[m]);
                        sZ1[m] += dz1[m];
                        sT1[m] += tn * dz1[m];
                    }
                }
                if constexpr (regen) {
                    __builtin_amdgcn_sched_barrier(0);
                    const long long n0 = own_block(it + 1);
                    const long long nb1 = n0 >= 0 ? n0 : nblk - 1;
                    const int k1 = (int)(nb1 % a.ntile16) * 16 + j;
                    dkn = a.D[k1 < a.K_local ? k1 : 0];
#pragma unroll
                    for (int m = 0; m < HB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h1n[m][r] = pn[G::pH1 + (4 * m + r) * 64];
                }
                split_panel<HB>(dz1, Zh, Zl);
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        exr[r][X::hZ1 + (16 * m + 4 * r + q) * 32] = Zh[m][r];
                        exr[r][X::hZ1l + (16 * m + 4 * r + q) * 32] = Zl[m][r];
                    }
            }
            PSP_STAMP(tp4);
            PSP_ACC(3, tp4, tp0);                     // whole produce phase
            __syncthreads();                              // swap the exchange buffers (pairs with the consumer loop)
            PSP_STAMP(tp5);
            PSP_ACC(4, tp5, tp4);                     // barrier wait
            PSP_ACC(6, tp5, tp0);
        }
#ifdef PSP_STAMPS
        if (a.dbg && lane == 0) {
            stamps[7] = (unsigned long long)R;
            for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
        }
#endif
        // per-wave bias sums -> LDS (the exchange area is free after the last barrier); lane (j = 0, q), component r of block b
        // holds feature 16 b + 4 r + q
        float* red = lds + X::EX + sub * RS;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = ginv * jsumf(sG[b][r]);
                if (j == 0) red[16 * b + 4 * r + q] = v;
            }
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v2 = ginv * jsumf(sZ2[m][r]), v1 = ginv * jsumf(sZ1[m][r]), vt = ginv * jsumf(sT1[m][r]);
                if (j == 0) {
                    red[16 * DB + 16 * m + 4 * r + q] = v2;
                    red[16 * DB + 16 * HB + 16 * m + 4 * r + q] = v1;
                    red[16 * DB + 32 * HB + 16 * m + 4 * r + q] = vt;
                }
            }
        __syncthreads();                                  // pairs with the consumers' barrier before the bias write-out
        return;
    }
    // ==================================================================================== consumers
    const int ibw = sub < HB ? sub : HB - 1;              // hidden block of this wave's dW3 / dW2 column (clamped: discarded below)
    int obx[NX];
#pragma unroll
    for (int s = 0; s < NX; ++s) obx[s] = (sub + 4 * s) < DB ? (sub + 4 * s) : DB - 1;
    f32x4 a3[DB], a2[HB], a1[HB][NX], c1[HB][NX];     // dW3 / dW2: one chain (2048 x); dW1: main / correction chains
#pragma unroll
    for (int b = 0; b < DB; ++b) a3[b] = zero4;
#pragma unroll
    for (int m = 0; m < HB; ++m) {
        a2[m] = zero4;
#pragma unroll
        for (int s = 0; s < NX; ++s) { a1[m][s] = zero4; c1[m][s] = zero4; }
    }
    const int nblk_i = (int)nblk;                     // N * ntile16 < 2^31 is checked by the host
    auto blk_at = [&](long long c0) __attribute__((always_inline)) {
        const int c = (c0 < (long long)nblk_i) ? (int)c0 : nblk_i - 1;
        return __builtin_amdgcn_readfirstlane(c);
    };
    typedef const __attribute__((address_space(1))) float* gptr_t;
    auto sbase = [&](int blk, int ofs) __attribute__((always_inline)) {
        return (gptr_t)sgpr_block_addr(a.path, (unsigned long long)blk, (unsigned)G::PB, (unsigned)ofs);
    };
    const unsigned lofsU = (unsigned)image_lane_offset_F(lane);
    auto get_F = [&](gptr_t base) __attribute__((always_inline)) {
        return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + lofsU);
    };
    // landing registers of the pair one ahead: h2, h1 (block ibw), X (blocks obx) of the pair's two sample blocks
    f32x4 Lh2[2], Lh1[2], Lx[NX][2];
    auto issue = [&](int c0, int c1) __attribute__((always_inline)) {
        Lh2[0] = get_F(sbase(c0, G::pH2 + ibw * 256)); Lh2[1] = get_F(sbase(c1, G::pH2 + ibw * 256));
        Lh1[0] = get_F(sbase(c0, G::pH1 + ibw * 256)); Lh1[1] = get_F(sbase(c1, G::pH1 + ibw * 256));
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            Lx[s][0] = get_F(sbase(c0, G::pX + obx[s] * 256));
            Lx[s][1] = get_F(sbase(c1, G::pX + obx[s] * 256));
        }
    };
    auto pack = [&](const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) __attribute__((always_inline)) {
        split8(u0, u1, hi, lo);
    };
    // A operand of tile t of the pair (0 .. DB-1: G, then dz2, then dz1): lane (i, g) = feature 16 tile + i, samples 4 g .. 4 g + 3
    // of block c0, then of block c1 -- one 16-byte read each for hi and lo
    const int aofs = opaque_i((lane & 15) * 32 + (((lane >> 4) ^ ((lane >> 2) & 3)) * 8));   // (the producers' chunk swizzle)
    auto a_load = [&](const _Float16* e, int t, f16x8& Ah, f16x8& Al) __attribute__((always_inline)) {
        const int hi = t < DB ? X::hG + t * 512 : (t < DB + HB ? X::hZ2 + (t - DB) * 512 : X::hZ1 + (t - DB - HB) * 512);
        const int lo = t < DB ? X::hGl + t * 512 : (t < DB + HB ? X::hZ2l + (t - DB) * 512 : X::hZ1l + (t - DB - HB) * 512);
        Ah = *reinterpret_cast<const f16x8*>(e + hi + aofs);
        Al = *reinterpret_cast<const f16x8*>(e + lo + aofs);
    };
    const f16x8 k2048 = {(_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f,
                         (_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f};
    // one pair of sample blocks (exchange area e; B operands in the landing registers); then request pair (n0, n1)
    auto pair_phase = [&](const _Float16* e, int n0, int n1) __attribute__((always_inline)) {
        f16x8 Ah[3], Al[3];
        a_load(e, 0, Ah[0], Al[0]);
        a_load(e, 1, Ah[1], Al[1]);
        f16x8 h2h, h2l, h1h, h1l, xh[NX], xl[NX];
        pack(Lh2[0], Lh2[1], h2h, h2l);
        pack(Lh1[0], Lh1[1], h1h, h1l);
#pragma unroll
        for (int s = 0; s < NX; ++s) pack(Lx[s][0], Lx[s][1], xh[s], xl[s]);
        const f16x8 h2s = h2h * k2048, h1s = h1h * k2048;              // exact: |tanh| <= 1
        __builtin_amdgcn_sched_barrier(0);
        issue(n0, n1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 2 < NT) a_load(e, t + 2, Ah[(t + 2) % 3], Al[(t + 2) % 3]);
            const f16x8 ah = Ah[t % 3], al = Al[t % 3];
            if (t < DB) {                                               // 2048 dW3[t][ibw] += G[t] h2^T
                a3[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, h2s, a3[t], 0, 0, 0);
                a3[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, h2l, a3[t], 0, 0, 0);
                a3[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, h2h, a3[t], 0, 0, 0);
            } else if (t < DB + HB) {                                   // 2048 dW2[m][ibw] += dz2[m] h1^T
                const int m = t - DB;
                a2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, h1s, a2[m], 0, 0, 0);
                a2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, h1l, a2[m], 0, 0, 0);
                a2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, h1h, a2[m], 0, 0, 0);
            } else {                                                    // dW1[m][obx] += dz1[m] X^T
                const int m = t - DB - HB;
#pragma unroll
                for (int s = 0; s < NX; ++s) {
                    a1[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[s], a1[m][s], 0, 0, 0);
                    c1[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[s], c1[m][s], 0, 0, 0);
                    c1[m][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[s], c1[m][s], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(kFenceMask);
        }
    };

    {
        const long long rb = (long long)blockIdx.x * 4;
        issue(blk_at(rb), blk_at(rb + 1));                // first pair's operands, while the producers start
    }
    __syncthreads();                                      // (A) the producers' scale slots are written
    __syncthreads();                                      // pairs with producer iteration 0
#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int it = 1; it <= R; ++it) {
        PSP_STAMP(tc0);
        const long long rb = ((long long)blockIdx.x + (long long)(it - 1) * gridDim.x) * 4;
        const _Float16* exr = exh + ((it - 1) & 1) * 2 * PAIRH;
        const long long rn = rb + 4LL * gridDim.x;        // first block of this workgroup's next round
        pair_phase(exr, blk_at(rb + 2), blk_at(rb + 3));
        PSP_STAMP(tc1);
        pair_phase(exr + PAIRH, blk_at(rn), blk_at(rn + 1));
        PSP_STAMP(tc2);
        PSP_ACC(0, tc2, tc0);                         // both pairs
        PSP_ACC(1, tc1, tc0);                         // first pair
        __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
        PSP_STAMP(tc3);
        PSP_ACC(4, tc3, tc2);                         // barrier wait
        PSP_ACC(6, tc3, tc0);
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0) {
        stamps[7] = (unsigned long long)R;
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
    }
#endif
    float gs, ginv;
    g_scale(gs, ginv);
    const float inv1 = ginv, inv2 = ginv * kSplitInv;     // un-scale: G's power of two, and the 2048 of the one-chain tiles

    // ---- consumers write their tiles into the workgroup's partial gradient
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
    const int col = lane & 15, qq = lane >> 4;
    if (sub < HB) {
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const f32x4 v = inv2 * a3[b];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o3 = 16 * b + 4 * qq + rr, i3 = 16 * sub + col;
                if (o3 < D && i3 < H) { const float gv_ = v[rr]; gp[G::oW3 + o3 * H + i3] = gv_; gchk.see(gv_); }
            }
        }
#pragma unroll
        for (int m = 0; m < HB; ++m) {
            const f32x4 v = inv2 * a2[m];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o2 = 16 * m + 4 * qq + rr, i2 = 16 * sub + col;
                if (o2 < H && i2 < H) { const float gv_ = v[rr]; gp[G::oW2 + o2 * H + i2] = gv_; gchk.see(gv_); }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < HB; ++m)
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            const f32x4 v = inv1 * (a1[m][s] + kSplitInv * c1[m][s]);
            const int ob = sub + 4 * s;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o1 = 16 * m + 4 * qq + rr, i1 = 16 * ob + col;
                if (ob < DB && o1 < H && i1 < D) { const float gv_ = v[rr]; gp[G::oW1 + o1 * (D + 1) + 1 + i1] = gv_; gchk.see(gv_); }
            }
        }
    // bias gradients and the time column of dW1: fixed-order sum of the four producers' partial sums (LDS)
    __syncthreads();                                      // pairs with the producers' barrier after their LDS write
    {
        const float* red = lds + X::EX;
        const int ct = tid - 256;
        for (int f = ct; f < D; f += 256)
            { const float gv_ = (red[f] + red[RS + f]) + (red[2 * RS + f] + red[3 * RS + f]); gp[G::ob3 + f] = gv_; gchk.see(gv_); }
        for (int f = ct; f < H; f += 256) {
            const float* r2 = red + 16 * DB + f;
            { const float gv_ = (r2[0] + r2[RS]) + (r2[2 * RS] + r2[3 * RS]); gp[G::ob2 + f] = gv_; gchk.see(gv_); }
            const float* r1 = r2 + 16 * HB;
            { const float gv_ = (r1[0] + r1[RS]) + (r1[2 * RS] + r1[3 * RS]); gp[G::ob1 + f] = gv_; gchk.see(gv_); }
            const float* rt = r1 + 16 * HB;
            { const float gv_ = (rt[0] + rt[RS]) + (rt[2 * RS] + rt[3 * RS]); gp[G::oW1 + f * (D + 1)] = gv_; gchk.see(gv_); }
        }
    }
    gchk.raise(a.cond);
}

template <int D, int H>
struct HjbxLaunch {
    using X = GeoX<D, H>;
    static int lds_bytes() { return X::lds_floats() * 4; }
    static hipError_t bwd(const HjbArgs& a, int grid, hipStream_t s) {
        const int bytes = lds_bytes();
        if (a.store_path == 4) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjb_bwd3_kernel<D, H, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjb_bwd3_kernel<D, H, true>), dim3(grid), dim3(512), bytes, s, a);
            return hipGetLastError();
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjb_bwd3_kernel<D, H, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjb_bwd3_kernel<D, H, false>), dim3(grid), dim3(512), bytes, s, a);
        return hipGetLastError();
    }
};

}  // namespace psp
