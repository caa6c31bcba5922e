// hjbwx_kernels.h -- split-product version of hjbw_bwd2_kernel: the role-specialised backward of the wide family for d <= 256
// (round 4).  hjbw_bwd2_kernel is the one kernel left that is bound by the fp32 matrix pipe (82 % busy, 8 DB + 32 fp32 MFMAs of 32
// cycles per sample block and consumer wave).  Same roles, same exchange, same flush layout; what changes:
//   * producers: dz2 = (W3^T G)(1 - h2^2) through gemm_Tx on the split W3^T table (84 f16 MFMAs instead of 208 fp32 ones at d = 200);
//   * consumers: the four blocks of a round as TWO PAIRS -- every weight-gradient tile contracts the pair's 32 samples in three
//     v_mfma_f32_16x16x32_f16 on ONE accumulator (operands split with unscaled residuals, as in hjbw_bwd_x3_kernel and
//     hjbd_bwd_kernel<.., X3>): dW3 += xi^T (w h2), dW2 += dz2^T h1, dW1 += dz1^T X_n; dz1 = (W2^T dz2)(1 - h1^2) moves to the
//     producers (a second split table, 24 f16 MFMAs; the consumers kept its W2 block in 16 registers and ran it as 16 fp32 MFMAs);
//   * the trajectory weights carry a power of two that maps  max_k |w_k| sqrt(dt) 8  (one scan of D per workgroup) into [2^10, 2^11):
//     every weighted operand (G, dz2, dz1, w h2) then has normal f16 residuals; all accumulators are scaled back when they are
//     written (exact);
//   * the streamed tiles (h2, h1, then the xi and X_n tiles of both blocks of a pair) run through one register ring of RD item pairs
//     that continues across pair and round boundaries; a round's item count is padded to a multiple of RD by re-reads.
// Compiled in its own translation unit WITHOUT the SLP vectoriser (two-instruction operand split, hjb_kernels.h split8u): a consumer
// splits 2 DB + 7 packs of eight values per pair.  Range guard: partial gradients pass GradCheck like every x3 backward.
#pragma once
#include "hjbw_kernels.h"

#ifndef PSP_ABL_WX
#define PSP_ABL_WX 0   // measurement only (tools/r4): 1 = consumers without their MFMAs (operands stay live); 4 = producers without
                       // their xi reads (what fetching xi once could save at most: d = 200 1.85 -> 1.71 ms)
#endif

namespace psp {

template <int D, int H>
struct GeoBX {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB, KSD = G::KSD, KSH = G::KSH;
    static constexpr int tblF = SplitGeo<KSD, DB>::floats(HB);          // split W3^T table (producers)
    static constexpr int tbl2F = SplitGeo<KSH, HB>::floats(HB);         // split W2^T table (producers)
    static constexpr int oZ1 = 4 * HB * 64, oWts = 2 * 4 * HB * 64, EXQ = oWts + 64;   // per block: dz2 and dz1 k-step images, 16 weights
    static constexpr int bufs = tblF + tbl2F, oScan = bufs + 2 * 4 * EXQ, lds_floats = oScan + 16;
#ifdef PSP_WX_RD
    static constexpr int RD = PSP_WX_RD;                                // (measurement builds)
#else
    static constexpr int RD = DB <= 13 ? 10 : 8;                        // ring depth in item PAIRS (two f32x4 each): what the consumer's
#endif                                                                  // 4 (2 DB + HB) accumulator registers leave room for without spills
    static constexpr int NIP = 2 * DB + 2;                              // items of a pair: h2, h1, DB xi tiles, DB X tiles
    static constexpr int NIR = ((2 * NIP + RD - 1) / RD) * RD;          // items of a round, padded to a multiple of RD
    static constexpr int RS = 16 * DB + 3 * 16 * HB;                    // per-producer bias-sum slots: G | dz2 | dz1 | t dz1
    static_assert(4 * RS <= 2 * 4 * EXQ, "bias sums reuse the exchange area");
};

template <int D, int H>
__global__ __launch_bounds__(512) void hjbw_bwd2x_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    GradCheck<true> gchk;
    using G = Geo<D, H>;
    using BX = GeoBX<D, H>;
    constexpr int DB = BX::DB, HB = BX::HB, KSD = BX::KSD, KSH = BX::KSH, EXQ = BX::EXQ, RD = BX::RD, RS = BX::RS,
                  NIP = BX::NIP, NIR = BX::NIR;
    static_assert(HB == 4, "one consumer wave per hidden block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const bool producer = wave < 4;
    const int sub = wave & 3;
    const float* __restrict__ P = a.params;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    stage_aop_x3<KSD, DB>(lds, HB, tid, nthr, [&](int row, int col) {   // W3^T as a split A-operand table (producers)
        return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
    stage_aop_x3<KSH, HB>(lds + BX::tblF, HB, tid, nthr, [&](int row, int col) {   // W2^T likewise: dz1[i] = sum_o W2[o][i] dz2[o]
        return (row < H && col < H) ? P[G::oW2 + col * H + row] : 0.f; });
    float* bufs = lds + BX::bufs;                     // [2 buffers][4 blocks][EXQ]
    float* scan = lds + BX::oScan;

    const double invK = 1.0 / (double)a.K_global;
    const float meanD = (a.loss_kind == LOSS_LOGVAR) ? (float)(a.sums[0] * invK) : 0.f;
    const float coef = (float)(2.0 * invK);
    const float dt = a.dt;
    // ---- weight scale: the largest |w_k| of this rank's trajectories (every workgroup scans all of D: K_local floats)
    float gs, ginv;
    {
        float wm = 0.f;
        for (int k = tid; k < a.K_local; k += nthr) {
            const float dk = a.D[k];
            wm = fmaxf(wm, fabsf(a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)));
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if (lane == 0) scan[wave] = wm;
        __syncthreads();                                                // (also: the table is staged)
        float m8 = scan[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m8 = fmaxf(m8, scan[w]);
        const float amax = m8 * a.sqdt * 8.0f;
        const unsigned e = (__float_as_uint(amax) >> 23) & 0xFFu;
        const bool ok = e >= 11u && e <= 253u;                           // zero / tiny / non-finite weights: no scaling
        const float sc = ok ? __uint_as_float((264u - e) << 23) : 1.0f;  // largest |w| sqrt(dt) 8 -> [2^10, 2^11): hjbw_bwd_x3_kernel's scale
        ginv = ok ? __uint_as_float((e - 10u) << 23) : 1.0f;
        gs = a.sqdt * sc;
    }
    const long long nblk = (long long)a.N * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    const int R = (int)((nround - blockIdx.x + gridDim.x - 1) / gridDim.x);   // rounds of this workgroup (>= 1)

    if (producer) {
        // ================================================================================ producers
        f32x4 sG[DB], sZ2[HB], sZ1[HB], sT1[HB];
#pragma unroll
        for (int b = 0; b < DB; ++b) sG[b] = zero4;
#pragma unroll
        for (int m = 0; m < HB; ++m) { sZ2[m] = zero4; sZ1[m] = zero4; sT1[m] = zero4; }
        auto own_block = [&](int it2) __attribute__((always_inline)) {
            const long long b0 = ((long long)blockIdx.x + (long long)it2 * gridDim.x) * 4 + sub;
            return b0 < nblk ? b0 : -1LL;
        };
        f32x4 xin[DB];
        float dkn;
        {
            const long long b0 = own_block(0);
            const long long blk = b0 >= 0 ? b0 : nblk - 1;
            const float* pb = a.path + (size_t)blk * (size_t)G::PB + lane;
            const int k0 = (int)(blk % a.ntile16) * 16 + j;
            dkn = a.D[k0 < a.K_local ? k0 : 0];
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) xin[b][r] = pb[G::pXi + (4 * b + r) * 64];
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
                const float* pb = a.path + (size_t)blk * (size_t)G::PB + lane;
                float* ex = bufs + ((it & 1) * 4 + sub) * EXQ;
                const float dk = dkn;                 // LOSS_WEIGHTS: the caller supplies w_k = dLoss/dY_k directly in D
                const float wk = kvalid ? (a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)) : 0.f;
                const float wks = wk * gs;            // scaled
                f32x4 Gt[DB];
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    Gt[b] = wks * xin[b];             // adaptive: the (Z + c) dt term cancels; else the image holds xi + sqrt(dt) Z
                    sG[b] += Gt[b];
                }
                f32x4 h2[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[m][r] = pb[G::pH2 + (4 * m + r) * 64];
                if (q == 0) ex[BX::oWts + j] = wks;   // the consumers weight their h2 operand with it
                f32x4 dz2[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) dz2[m] = zero4;
                gemm_Tx<HB, KSD, DB, 1>(dz2, lds, Gt, lane);
#pragma unroll
                for (int m = 0; m < HB; ++m) { dz2[m] = dz2[m] * (1.0f - h2[m] * h2[m]); sZ2[m] += dz2[m]; }
#pragma unroll
                for (int ks = 0; ks < 4 * HB; ++ks) ex[ks * 64 + lane] = dz2[ks >> 2][ks & 3];
                __builtin_amdgcn_sched_barrier(0);        // (G is dead from here: its registers take h1 and dz1)
                {                                         // the next round's xi tiles: requested once G is dead (the dz1 half of
                                                          // the round and the barrier cover their latency), so that xin and G
                                                          // never hold 2 DB tiles at the same time
                    const long long n0 = own_block(it + 1);
                    const long long nblk1 = n0 >= 0 ? n0 : nblk - 1;
                    const float* pn = a.path + (size_t)nblk1 * (size_t)G::PB + lane;
                    const int k1 = (int)(nblk1 % a.ntile16) * 16 + j;
                    dkn = a.D[k1 < a.K_local ? k1 : 0];
#pragma unroll
                    for (int b = 0; b < DB; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xin[b][r] = (PSP_ABL_WX & 4) ? 0.25f * (float)(b + r) : pn[G::pXi + (4 * b + r) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 h1[HB], dz1[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[m][r] = pb[G::pH1 + (4 * m + r) * 64];
#pragma unroll
                for (int m = 0; m < HB; ++m) dz1[m] = zero4;
                gemm_Tx<HB, KSH, HB, 1>(dz1, lds + BX::tblF, dz2, lane);
                const float tn = (float)(blk / a.ntile16) * dt;
#pragma unroll
                for (int m = 0; m < HB; ++m) { dz1[m] = dz1[m] * (1.0f - h1[m] * h1[m]); sZ1[m] += dz1[m]; sT1[m] += tn * dz1[m]; }
#pragma unroll
                for (int ks = 0; ks < 4 * HB; ++ks) ex[BX::oZ1 + ks * 64 + lane] = dz1[ks >> 2][ks & 3];
            }
            PSP_STAMP(tp1);
            __syncthreads();                              // swap the exchange buffers (pairs with the consumer loop)
            PSP_STAMP(tp2);
            PSP_ACC(0, tp1, tp0); PSP_ACC(1, tp2, tp1);
        }
#ifdef PSP_STAMPS
        if (a.dbg && lane == 0) {
            stamps[7] = (unsigned long long)R;
            for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
        }
#endif
        float* red = bufs + sub * RS;                     // bias sums -> LDS (the exchange area is free after the last barrier)
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = jsumf(sG[b][r]);
                if (j == 0) red[16 * b + 4 * r + q] = v;
            }
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v2 = jsumf(sZ2[m][r]), v1 = jsumf(sZ1[m][r]), vt = jsumf(sT1[m][r]);
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
    const int ib = sub;                                   // hidden block of this wave
    f32x4 acc3[DB], acc1[DB], acc2[HB];
#pragma unroll
    for (int b = 0; b < DB; ++b) { acc3[b] = zero4; acc1[b] = zero4; }
#pragma unroll
    for (int m = 0; m < HB; ++m) acc2[m] = zero4;
    const int nblk_i = (int)nblk;                         // N * ntile16 < 2^31 is checked by the host
    auto blk_at = [&](long long c0) __attribute__((always_inline)) {
        const int c = (c0 < (long long)nblk_i) ? (int)c0 : nblk_i - 1;
        return __builtin_amdgcn_readfirstlane(c);
    };
    typedef const __attribute__((address_space(1))) float* gptr_t;
    const unsigned lofsU = (unsigned)image_lane_offset_F(lane);
    auto get_F = [&](int blk, int ofs) __attribute__((always_inline)) {
        gptr_t base = (gptr_t)sgpr_block_addr(a.path, (unsigned long long)blk, (unsigned)G::PB, (unsigned)ofs);
        return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + lofsU);
    };
    // item g of a round (0 .. NIR - 1; NIR and beyond: the next round): pair p = g / NIP, item i = g % NIP of the pair:
    //   0: h2 tiles (hidden block ib), 1: h1 tiles, 2 .. DB + 1: xi tiles, DB + 2 .. 2 DB + 1: X tiles; padding items re-read item 0
    f32x4 st0[RD], st1[RD];
    int bb[6];                                            // blocks of the round, and the first pair of the next round
    auto item_ofs = [&](int i) __attribute__((always_inline)) {
        return i == 0 ? G::pH2 + ib * 256 : (i == 1 ? G::pH1 + ib * 256 : (i < DB + 2 ? G::pXi + (i - 2) * 256 : G::pX + (i - DB - 2) * 256));
    };
    auto item_load = [&](auto gi) __attribute__((always_inline)) {
        constexpr int g = decltype(gi)::value;
        constexpr int gr = g % NIR, nxt = g / NIR;       // (nxt = 1: the next round)
        constexpr int p = (gr < 2 * NIP) ? gr / NIP : 0, i = (gr < 2 * NIP) ? gr % NIP : 0;
        static_assert(nxt == 0 || p == 0, "the ring reaches at most into the next round's first pair");
        const int c0 = nxt ? bb[4] : bb[2 * p], c1 = nxt ? bb[5] : bb[2 * p + 1];
        st0[g % RD] = get_F(c0, item_ofs(i));
        st1[g % RD] = get_F(c1, item_ofs(i));
    };
    auto pack = [&](const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) __attribute__((always_inline)) { split8u(u0, u1, hi, lo); };
    auto mma3 = [&](f32x4& acc, const f16x8& Ah, const f16x8& Al, const f16x8& Bh, const f16x8& Bl) __attribute__((always_inline)) {
#if (PSP_ABL_WX & 1)
        acc[0] += (float)Ah[0] + (float)Al[1] + (float)Bh[2] + (float)Bl[3]; return;   // ablation: operands stay live, no MFMA
#endif
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, acc, 0, 0, 0);
    };
    const int rb0 = blockIdx.x * 4;
    bb[0] = blk_at(rb0); bb[1] = blk_at((long long)rb0 + 1); bb[2] = bb[0]; bb[3] = bb[1]; bb[4] = bb[0]; bb[5] = bb[1];
    static_for<0, RD - 1>([&](auto gi) { item_load(gi); });             // (inside the first pair: RD - 1 <= NIP)
    static_assert(RD - 1 <= NIP, "prologue stays inside the first pair");
    __syncthreads();                                      // pairs with producer iteration 0
#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int it = 1; it <= R; ++it) {
        PSP_STAMP(tc0);
        const int rb = (blockIdx.x + (it - 1) * gridDim.x) * 4;
        const float* exch = bufs + ((it - 1) & 1) * 4 * EXQ;
        bb[0] = blk_at(rb); bb[1] = blk_at((long long)rb + 1); bb[2] = blk_at((long long)rb + 2); bb[3] = blk_at((long long)rb + 3);
        bb[4] = blk_at((long long)rb + 4LL * gridDim.x); bb[5] = blk_at((long long)rb + 4LL * gridDim.x + 1);
        static_for<0, 2>([&](auto pc) {
            constexpr int p = decltype(pc)::value, g0 = p * NIP;
            const float* ex0 = exch + (2 * p) * EXQ;
            const float* ex1 = exch + (2 * p + 1) * EXQ;
            const f32x4 w40 = *reinterpret_cast<const f32x4*>(ex0 + BX::oWts + 4 * q);   // weights of the lane's samples 4 q' .. 4 q' + 3
            const f32x4 w41 = *reinterpret_cast<const f32x4*>(ex1 + BX::oWts + 4 * q);
            // ---- item 0: h2 tiles -> the weighted B operand of layer 3
            item_load(std::integral_constant<int, g0 + 0 + RD - 1>{});
            f16x8 Bh2h, Bh2l;
            pack(st0[(g0 + 0) % RD] * w40, st1[(g0 + 0) % RD] * w41, Bh2h, Bh2l);
            // ---- item 1: h1 tiles
            item_load(std::integral_constant<int, g0 + 1 + RD - 1>{});
            const f32x4 oh10 = st0[(g0 + 1) % RD], oh11 = st1[(g0 + 1) % RD];
            f16x8 Bh1h, Bh1l;
            pack(oh10, oh11, Bh1h, Bh1l);
            // ---- the pair's dz1 tiles (hidden block ib) from the producers' images
            f16x8 A1h, A1l;
            pack(tile_get(ex0 + BX::oZ1 + ib * 256, lane), tile_get(ex1 + BX::oZ1 + ib * 256, lane), A1h, A1l);
            // ---- layer 2: dW2[:, ib] += dz2^T h1
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                f16x8 Ah, Al;
                pack(tile_get(ex0 + m * 256, lane), tile_get(ex1 + m * 256, lane), Ah, Al);
                mma3(acc2[m], Ah, Al, Bh1h, Bh1l);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- layer 3: dW3[:, ib] += xi^T (w h2)
            static_for<0, DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, g = g0 + 2 + i;
                item_load(std::integral_constant<int, g + RD - 1>{});
                f16x8 Ah, Al;
                pack(st0[g % RD], st1[g % RD], Ah, Al);
                mma3(acc3[i], Ah, Al, Bh2h, Bh2l);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- layer 1: dW1[ib, :] += dz1^T X_n
            static_for<0, DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, g = g0 + 2 + DB + i;
                item_load(std::integral_constant<int, g + RD - 1>{});
                f16x8 Bh, Bl;
                pack(st0[g % RD], st1[g % RD], Bh, Bl);
                mma3(acc1[i], A1h, A1l, Bh, Bl);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // padding items of the round (re-reads; keep the ring's slot arithmetic): their loads are issued, nothing consumes them
        static_for<2 * NIP, NIR>([&](auto gc) { item_load(std::integral_constant<int, decltype(gc)::value + RD - 1>{}); });
        // the ring now holds the first RD - 1 items of the next round's first pair
        PSP_STAMP(tc1);
        __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
        PSP_STAMP(tc2);
        PSP_ACC(0, tc1, tc0); PSP_ACC(1, tc2, tc1);
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0) {
        stamps[7] = (unsigned long long)R;
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
    }
#endif

    // ---- flush: same mapping as hjbw_bwd2_kernel (tile rows = 16 ob + 4 qq + rr, columns = 16 ib + col), scaled back
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
    const int col = lane & 15, qq = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < DB; ++ob)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o3 = 16 * ob + 4 * qq + rr, i3 = 16 * ib + col;
            if (o3 < D && i3 < H) { const float v = ginv * acc3[ob][rr]; gp[G::oW3 + o3 * H + i3] = v; gchk.see(v); }
            const int o1 = 16 * ib + 4 * qq + rr, i1 = 16 * ob + col;
            if (o1 < H && i1 < D) { const float v = ginv * acc1[ob][rr]; gp[G::oW1 + o1 * (D + 1) + 1 + i1] = v; gchk.see(v); }
        }
#pragma unroll
    for (int ob = 0; ob < HB; ++ob)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o2 = 16 * ob + 4 * qq + rr, i2 = 16 * ib + col;
            if (o2 < H && i2 < H) { const float v = ginv * acc2[ob][rr]; gp[G::oW2 + o2 * H + i2] = v; gchk.see(v); }
        }
    __syncthreads();                                      // pairs with the producers' barrier after their LDS write
    {
        const float* red = bufs;
        const int ct = tid - 256;
        for (int f = ct; f < D; f += 256) {
            const float v = ginv * ((red[f] + red[RS + f]) + (red[2 * RS + f] + red[3 * RS + f]));
            gp[G::ob3 + f] = v; gchk.see(v);
        }
        for (int f = ct; f < H; f += 256) {
            const float* r2 = red + 16 * DB + f;
            const float* r1 = r2 + 16 * HB;
            const float* rt = r1 + 16 * HB;
            const float v2 = ginv * ((r2[0] + r2[RS]) + (r2[2 * RS] + r2[3 * RS]));
            const float v1 = ginv * ((r1[0] + r1[RS]) + (r1[2 * RS] + r1[3 * RS]));
            const float vt = ginv * ((rt[0] + rt[RS]) + (rt[2 * RS] + rt[3 * RS]));
            gp[G::ob2 + f] = v2; gchk.see(v2);
            gp[G::ob1 + f] = v1; gchk.see(v1);
            gp[G::oW1 + f * (D + 1)] = vt; gchk.see(vt);
        }
    }
    gchk.raise(a.cond);
}

template <int D, int H>
struct HjbwxLaunch {
    using BX = GeoBX<D, H>;
    static constexpr bool kOk = (D <= 256) && (Geo<D, H>::HB == 4) && (BX::lds_floats * 4 <= 160 * 1024);
    static hipError_t bwd(const HjbArgs& a, int grid, hipStream_t s) {
        if constexpr (kOk) {
            const int bytes = BX::lds_floats * 4;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_bwd2x_kernel<D, H>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_bwd2x_kernel<D, H>), dim3(grid), dim3(512), bytes, s, a);
            return hipGetLastError();
        }
        return hipErrorNotSupported;
    }
};

}  // namespace psp
