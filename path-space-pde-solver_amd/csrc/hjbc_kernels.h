// hjbc_kernels.h -- cooperative forward kernel of the wide family for d > 256 (round 4).
//
// hjbw_fwd_kernel gives every wave one 16-trajectory tile and the whole d x d operand tables: at d = 500 a wave pulls 2.2 MB of
// A operands through the vector-memory path per step for 3 480 MFMAs, four waves per CU.  Measured on that kernel (same box,
// tools/r4): with every table read an L1 hit it is 3 % faster, with HALF the operand bytes 23 % faster -- the bytes per MFMA through
// the L1 -> register path (~110 B/clk per CU) bind it, not the L2 stream and not the matrix pipe (33 % busy).
//
// Here a workgroup of EIGHT waves (two per SIMD, <= 256 registers) owns TWO tiles, and the OUTPUT blocks are dealt out instead of
// the tiles: wave w computes state blocks 4 w .. 4 w + 3 of both tiles.  Every fetched 2 KiB operand block feeds six MFMAs (three
// per tile) instead of three, each table byte is fetched once per workgroup and step instead of once per tile, and a wave needs
// 4 x 2 accumulators instead of 32.  The B operands (the state / increment panels of both tiles as hi / lo f16 packs) live in
// two LDS regions that every wave reads; element-wise work (Euler update, Philox, tanh, path stores) follows block ownership, so
// every feature is touched by exactly one wave.  Four workgroup barriers per step:
//   A  state image complete            -> P12: x += (dt A) x (owned blocks) and, in the same k-loop, h1 block (w & 3) of tile (w >> 2)
//   B  h1 exchange complete            -> P3:  h2 block (w & 3) of tile (w >> 2)
//   C  h2 exchange complete            -> P4:  Z (owned blocks) = W3 h2 + b3, Philox, row-sum partials, increment image
//   D  increment image + partials      -> P5:  x += B v (owned blocks); waves 0 / 1 finish Y of tile 0 / 1; new state image
// Same arithmetic per element as hjbw_fwd_kernel<.., X3> (k-steps ascending, main and correction chains, acc + corr / 2048); the
// row sums |Z|^2, Z.xi and the terminal cost are summed per wave over its blocks and then over the eight waves in a fixed order --
// a different but deterministic fp32 summation order.  Path store, D, partial sums: the layouts of hjbw_fwd_kernel, so the
// backward kernels, the loss reduction and the range guard's fp32 twin (hjbw_fwd_kernel on the same grid, two waves per
// workgroup) are unchanged.  Serves: on-device noise, no running cost, no u_L2 log (make_plan: fwd_coop); everything else stays
// on hjbw_fwd_kernel.  Reference lines: those of hjb_fwd_kernel (solver.py:440-478).
#pragma once
#include "hjbw_kernels.h"

namespace psp {

template <int D, int H, int NT_>
struct GeoC {
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    static constexpr int DB = W::DB, HB = W::HB, KS8 = W::KS8;
    static constexpr int NWV = 8, NT = NT_;                         // waves, tiles per workgroup
    static_assert(NT == 2 || NT == 4, "two or four tiles per workgroup");
    // state blocks are dealt out in S-steps (pairs of blocks: one hi / lo pack of the images): NP S-steps = NB blocks per wave
    static constexpr int NP = cdiv(KS8, NWV), NB = 2 * NP;          // d = 500: 2 / 4;  d = 192 .. 256: 1 / 2
    static constexpr int NOWN = cdiv(KS8, NP);                      // waves that own state blocks (the others only run the hidden layers)
    static constexpr bool MAYPAD = (DB % NB) != 0;                  // odd block count: the last owner's last block is padding (zero, never stored)
    static constexpr int NH = NT * HB / NWV;                        // hidden-layer outputs per wave: block w & 3 of tiles (w >> 2) NH + i
    static constexpr int IMG8 = KS8 * 2 * 64;                       // f16x8 elements of one tile's hi / lo image
    // Where they fit: separate regions for the state and the increment image and for the h1 / h2 exchanges (four barriers per step).
    // Else (d = 500 with four tiles: 4 x 32 KiB per image): ONE image region and one exchange buffer, with a barrier before each is
    // overwritten (six).
    static constexpr bool TWO = (W::fImg + 2 * NT * IMG8 * 4 + 2 * NT * HB * 256 + 2 * NT * NWV * 16 + 64) * 4 <= 160 * 1024;
    // LDS (floats): the per-feature vectors of GeoW, the image region(s), the exchange(s), the partial sums
    static constexpr int cImg0 = W::fImg, cImg1 = TWO ? cImg0 + NT * IMG8 * 4 : cImg0, cH1 = cImg1 + NT * IMG8 * 4,
                         cH2 = TWO ? cH1 + NT * HB * 256 : cH1, cRed = cH2 + NT * HB * 256, lds_floats = cRed + 2 * NT * NWV * 16 + 64;
};

// acc[t][m] += T[S][b0 + m] . img[t][S] over all S-steps for the NB owned blocks of all NT tiles.  Rolled over S in pairs (static
// ring indices); the A operands of S-step S + 1 are requested before the MFMAs of S, the B operands (LDS) one tile pair ahead.
// `between(i)`: called once per S-step with i = 0 .. KS8 - 1 after the products of its first tile pair have been issued -- VALU work
// of the caller that runs in the MFMAs' shadow (the step's Philox calls).
struct NoCoopBetween { __device__ __forceinline__ void operator()(int) const {} };
template <int NB, int NT, int KS8, int LDT, bool UNROLLED = false, class BT = NoCoopBetween>
__device__ __forceinline__ void coop_gemm(f32x4 (&acc)[NT][NB], const float* __restrict__ tbl, const f16x8* img8, int img_tile_stride,
                                          int lane, BT between = BT(), bool padlast = false) {
    static_assert(NB % 2 == 0 && NT % 2 == 0, "blocks and tiles in pairs");
    constexpr int NPAIR = NT / 2;
    // B ring: two slots (a tile pair ahead) for two tiles; ONE for four tiles -- 16 accumulators more and a second slot spill
    // (the pair's operands are re-requested into the same registers right behind its products; the partner wave covers the wait)
    constexpr int BR = (NT == 2 || NB == 2) ? 2 : 1;                  // (d = 500, four tiles, two slots: 12.08 -> 12.33 ms; d = 256: 4.01 -> 3.98)
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    f16x8 ah[2][NB], al[2][NB], bh[BR][2], bl[BR][2];
    f32x4 corr[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < NB; ++m) corr[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (a padding block -- the last owner's last block when the block count is odd -- re-reads its neighbour: no access past the table;
    //  the caller zeroes its result)
    const unsigned padofs = padlast ? 0u : 128u;
    auto load_a = [&](int st, int S) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < NB; m += 2) {                               // fresh SGPR base every 4 KiB (two output blocks)
            gptr8_t tp = sgpr_ptr8(tbl + ((size_t)S * LDT + m) * 512);
            ah[st][m] = tp[ul]; al[st][m] = tp[64 + ul];
            if (m + 2 == NB) { ah[st][m + 1] = tp[padofs + ul]; al[st][m + 1] = tp[padofs + 64 + ul]; }
            else { ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = tp[192 + ul]; }
        }
    };
    auto load_b = [&](int st, int S, int pr) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bh[st][i] = img8[(2 * pr + i) * img_tile_stride + (2 * S) * 64];
            bl[st][i] = img8[(2 * pr + i) * img_tile_stride + (2 * S + 1) * 64];
        }
    };
    auto products = [&](int sa, int sb, int pr) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = 2 * pr + i;
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sa][m], bh[sb][i], acc[t][m], 0, 0, 0);
                corr[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sa][m], bl[sb][i], corr[t][m], 0, 0, 0);
                corr[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[sa][m], bh[sb][i], corr[t][m], 0, 0, 0);
            }
    };
    // units u = S * NPAIR + pr; B ring index u & 1 (static: an S pair holds an even number of units), A ring index S & 1
    load_a(0, 0);
    load_b(0, 0, 0);
    auto s_step = [&](int Sc, int h) __attribute__((always_inline)) {   // h = Sc & 1, as a constant
        const int Sn = Sc + 1 < KS8 ? Sc + 1 : KS8 - 1;                 // past the end: re-read the last step (unused)
        load_a((h + 1) & 1, Sn);
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            const int u = h * NPAIR + pr;                               // (parity of the unit within the S pair)
            if constexpr (BR == 2) {
                if (pr + 1 < NPAIR) load_b((u + 1) & 1, Sc, pr + 1);
                else load_b((u + 1) & 1, Sn, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            products(h & 1, BR == 2 ? (u & 1) : 0, pr);
            if (pr == 0) between(Sc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BR == 1) {
                if (pr + 1 < NPAIR) load_b(0, Sc, pr + 1);
                else load_b(0, Sn, 0);
            }
        }
    };
    if constexpr (UNROLLED) {                                           // (the caller's `between` needs the S-step as a constant)
#pragma unroll
        for (int S = 0; S < KS8; ++S) s_step(S, S & 1);
    } else {
#pragma unroll 1
        for (int S = 0; S + 1 < KS8; S += 2) { s_step(S, 0); s_step(S + 1, 1); }
        if constexpr (KS8 & 1) s_step(KS8 - 1, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[t][m] = acc[t][m] + kSplitInv * corr[t][m];
}

// hacc[i] += TH[S][hb] . img[t0 + i][S]: the first hidden layer's block of this wave for its NH tiles (K = d: the same state image)
template <int NH, int KS8, bool UNROLLED = false, class BT = NoCoopBetween>
__device__ __forceinline__ void coop_hidden(f32x4 (&hacc)[NH], const float* __restrict__ tblh, int LDH, const f16x8* img8,
                                            int img_tile_stride, int t0, int lane, BT between = BT()) {
    const unsigned ul = (unsigned)lane;
    tblh = opaque_base(tblh);
    f16x8 wh[2], wl[2], bh[2][NH], bl[2][NH];
    f32x4 corr[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) corr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int st, int S) __attribute__((always_inline)) {
        gptr8_t tp = sgpr_ptr8(tblh + (size_t)S * LDH * 512);
        wh[st] = tp[ul]; wl[st] = tp[64 + ul];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            bh[st][i] = img8[(t0 + i) * img_tile_stride + (2 * S) * 64];
            bl[st][i] = img8[(t0 + i) * img_tile_stride + (2 * S + 1) * 64];
        }
    };
    load(0, 0);
    auto s_step = [&](int Sc, int h) __attribute__((always_inline)) {   // h = Sc & 1, as a constant
        load((h + 1) & 1, Sc + 1 < KS8 ? Sc + 1 : KS8 - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            hacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[h], bh[h][i], hacc[i], 0, 0, 0);
            corr[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[h], bl[h][i], corr[i], 0, 0, 0);
            corr[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[h], bh[h][i], corr[i], 0, 0, 0);
        }
        between(Sc);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (UNROLLED) {
#pragma unroll
        for (int S = 0; S < KS8; ++S) s_step(S, S & 1);
    } else {
#pragma unroll 1
        for (int S = 0; S + 1 < KS8; S += 2) { s_step(S, 0); s_step(S + 1, 1); }
        if constexpr (KS8 & 1) s_step(KS8 - 1, 0);
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) hacc[i] = hacc[i] + kSplitInv * corr[i];
}

template <int D, int H, int NT_>
__global__ __launch_bounds__(512, 2) void hjbc_fwd_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    using C = GeoC<D, H, NT_>;
    constexpr int DB = C::DB, HB = C::HB, KS8 = C::KS8, NT = C::NT, NB = C::NB, NP = C::NP, NWV = C::NWV, NH = C::NH;
    const int k_drift = a.drift_kind, k_sigma = a.sigma_kind, k_loss = a.loss_kind, k_store = a.store_path;
    const bool k_adaptive = a.adaptive != 0;
    const uint32_t iter_now = a.iter_dev ? *a.iter_dev : a.iter;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;
    const float* __restrict__ T = a.tables;

    stage_vec(lds + W::vb1, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob1 + f] : 0.f; });
    stage_vec(lds + W::vw1t, HB, tid, nthr, [&](int f) { return f < H ? P[G::oW1 + f * (D + 1)] : 0.f; });
    stage_vec(lds + W::vb2, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob2 + f] : 0.f; });
    stage_vec(lds + W::vb3, DB, tid, nthr, [&](int f) { return f < D ? P[G::ob3 + f] : 0.f; });
    stage_vec(lds + W::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (k_drift == DRIFT_DIAG || k_drift == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + W::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });

    // ownership: state blocks b0 .. b0 + NB - 1 (S-steps s0 .. s0 + NP - 1) of all tiles; hidden block hb of tiles ht0 .. ht0 + NH - 1
    const bool owner = (C::NOWN == NWV) ? true : wave < C::NOWN;
    const int b0 = owner ? wave * NB : 0, s0 = owner ? wave * NP : 0;
    const bool padlast = C::MAYPAD && (b0 + NB > DB);                    // this wave's last block is padding: clamp its reads, zero its values, no stores
    auto blk = [&](int m) __attribute__((always_inline)) { return (C::MAYPAD && m == NB - 1 && padlast) ? b0 + m - 1 : b0 + m; };
    auto zero_pad = [&](f32x4 (&V)[NT][NB]) __attribute__((always_inline)) {
        if (C::MAYPAD && padlast) {
#pragma unroll
            for (int t = 0; t < NT; ++t) V[t][NB - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int hb = wave & 3, ht0 = (wave >> 2) * NH;
    int t16[NT], kk[NT];
    bool tvalid[NT], kvalid[NT];
    uint32_t kglob[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int raw = blockIdx.x * NT + t;
        tvalid[t] = raw < a.ntile16;                     // surplus tiles of the last workgroup run along on the last tile, store nothing
        t16[t] = tvalid[t] ? raw : a.ntile16 - 1;
        kk[t] = t16[t] * 16 + j;
        kvalid[t] = tvalid[t] && kk[t] < a.K_local;
        kglob[t] = (uint32_t)(a.k_offset + kk[t]);
    }
    const float dt = a.dt, sqdt = a.sqdt;
    f16x8* img0 = reinterpret_cast<f16x8*>(lds + C::cImg0) + lane;      // state image   [tile][S][hi | lo][64]
    f16x8* img1 = reinterpret_cast<f16x8*>(lds + C::cImg1) + lane;      // increment image (four tiles: the same region)
    f32x4* hx1 = reinterpret_cast<f32x4*>(lds + C::cH1) + lane;         // h1 exchange [tile][block][64] of f32x4 (T layout: 4 r per lane)
    f32x4* hx2 = reinterpret_cast<f32x4*>(lds + C::cH2) + lane;         // h2 exchange (four tiles: the same buffer)
    float* red = lds + C::cRed;                                         // [2][tile][wave][16]
    const float store_cxi = (k_store == 3) ? 0.f : 1.f;
    const float store_cz = (k_store == 3) ? 1.f : (k_store == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));

    // ---- X_0 (solver.py:365-367): owned blocks of all tiles, T layout
    f32x4 X[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * (b0 + m) + 4 * r + q;
                const float v = a.x0[(size_t)(kvalid[t] ? kk[t] : 0) * a.x0_stride + (f < D ? f : D - 1)];
                X[t][m][r] = (f < D && kvalid[t] && owner) ? v : 0.f;
            }
    auto write_image = [&](f16x8* img, const f32x4 (&V)[NT][NB]) __attribute__((always_inline)) {
        if (owner) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f16x8 ph, pl;
                    split_pack(V[t][2 * p], V[t][2 * p + 1], ph, pl);
                    img[t * C::IMG8 + (2 * (s0 + p)) * 64] = ph;
                    img[t * C::IMG8 + (2 * (s0 + p) + 1) * 64] = pl;
                }
        }
    };
    write_image(img0, X);
    float Y = a.y0 ? a.y0[0] : 0.f;                                      // (meaningful in waves 0 .. NT - 1: tile = wave)
    __syncthreads();

    const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;        // index by block * 4
    typedef __attribute__((address_space(1))) float* gwptr_t;
    const unsigned ul = (unsigned)lane;

#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#pragma unroll 1
    for (int n = 0; n < a.N; ++n) {
        PSP_STAMP(cs0);
        const float tn = (float)n * dt;
        const f32x4* vecs = opaque(vecs0);
        const int qn = opaque_i(q);
        const f32x4* vb1 = vecs + W::vb1 / 4;
        const f32x4* vw1t = vecs + W::vw1t / 4;
        const f32x4* vb2 = vecs + W::vb2 / 4;
        const f32x4* vb3 = vecs + W::vb3 / 4;
        const f32x4* vdr = vecs + W::vdr / 4;
        auto pbase = [&](int t, int ofs) __attribute__((always_inline)) {
            return (gwptr_t)sgpr_block_addr(a.path, (unsigned long long)n * a.ntile16 + t16[t], (unsigned)G::PB, (unsigned)ofs);
        };
        // ---- path store of X_n (owned blocks)
        if (k_store && owner) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (tvalid[t]) {
                    gwptr_t px = pbase(t, G::pX + b0 * 256);
#pragma unroll
                    for (int e = 0; e < 4 * NB; ++e)
                        if (!(C::MAYPAD && (e >> 2) == NB - 1 && padlast)) PSP_PATH_STORE(px + e * 64 + ul, X[t][e >> 2][e & 3]);
                }
        }
        // ---- P2: x += (dt A) x_n on the owned blocks; the step's Brownian increments are generated in the shadow of its MFMAs
        //      (one Philox call per S-step: NT NB calls, KS8 S-steps)
        // (two tiles: the 8 calls run in the shadow of P2's MFMAs; four tiles: 64 more live registers there spill -- the 16 calls run
        //  in the shadow of the first hidden layer's product, which comes AFTER P2 for that reason: both read the same state image)
        constexpr bool XI_EARLY = NT == 2;
        f32x4 xi[NT][NB];
        auto xi_call = [&](int c) __attribute__((always_inline)) {      // call c = t * NB + m (compile-time after unrolling)
            const int t = c / NB, m = c % NB, b = b0 + m;
            f32x4 v = philox_block(kglob[t], (uint32_t)n, (uint32_t)(4 * b + qn), iter_now, a.seed_lo, a.seed_hi);
            if (16 * (DB - 1) + 16 > D || C::MAYPAD) {                  // partial last block / padding block: keep padded features at zero
#pragma unroll
                for (int r = 0; r < 4; ++r) if (16 * b + 4 * r + q >= D) v[r] = 0.f;
            }
#if defined(PSP_XI_PIN) && PSP_XI_PIN
            asm volatile("" : "+v"(v));            // (A/B switch)
#endif
            // (NOT pinned with an empty asm: the compiler sinks part of this arithmetic towards its use in P4; pinning it here keeps 16 - 64
            //  more registers live across three phases and measured slower -- four tiles 11.9 -> 12.3 ms, two tiles unchanged: the
            //  "shadow" of the hidden layer's product is not free, the vector issue port is as busy there as in P4)
            xi[t][m] = v;
        };
        if (k_drift == DRIFT_DENSE && owner) {
            // calls [Sc NC / KS8, (Sc + 1) NC / KS8) of the NC = NT NB calls stand behind S-step Sc (the S loop is unrolled: constants)
            constexpr int NC = NT * NB;
            if constexpr (XI_EARLY) {
                coop_gemm<NB, NT, KS8, DB, true>(X, T + W::xA + (size_t)b0 * 512, img0, C::IMG8, lane, [&](int Sc) __attribute__((always_inline)) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (c >= Sc * NC / KS8 && c < (Sc + 1) * NC / KS8) xi_call(c);
                }, padlast);
            } else {
                coop_gemm<NB, NT, KS8, DB>(X, T + W::xA + (size_t)b0 * 512, img0, C::IMG8, lane, NoCoopBetween(), padlast);
            }
            zero_pad(X);
        } else if (owner) {
            if (k_drift == DRIFT_DIAG) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m) X[t][m] += dt * (vdr[blk(m) * 4] * X[t][m]);
            } else if (k_drift == DRIFT_DWELL) {                        // b = -4 kappa x (x^2 - 1), problems.py:311-315
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m)
                        X[t][m] -= dt * (4.0f * vdr[blk(m) * 4] * (X[t][m] * (X[t][m] * X[t][m] - 1.0f)));
            }
            if constexpr (XI_EARLY) {
#pragma unroll
                for (int c = 0; c < NT * NB; ++c) xi_call(c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        PSP_STAMP(cs1);
        // ---- P1: first hidden layer, block hb of this wave's NH tiles (function_space.py:190-195); tanh; exchange + path store
        {
            f32x4 h1[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i) h1[i] = vb1[hb * 4] + tn * vw1t[hb * 4];
            if constexpr (XI_EARLY) {
                coop_hidden<NH, KS8>(h1, T + W::xW1 + (size_t)hb * 512, HB, img0, C::IMG8, ht0, lane);
            } else {
                constexpr int NC = NT * NB;
                coop_hidden<NH, KS8, true>(h1, T + W::xW1 + (size_t)hb * 512, HB, img0, C::IMG8, ht0, lane, [&](int Sc) __attribute__((always_inline)) {
                    if (owner) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            if (c >= Sc * NC / KS8 && c < (Sc + 1) * NC / KS8) xi_call(c);
                    }
                });
            }
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                h1[i] = tanh4(h1[i]);
                hx1[((ht0 + i) * HB + hb) * 64] = h1[i];
                if (k_store && tvalid[ht0 + i]) {
                    gwptr_t ph = pbase(ht0 + i, G::pH1 + hb * 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) PSP_PATH_STORE(ph + r * 64 + ul, h1[i][r]);
                }
            }
        }
        // the A operands of the two small products (W2: block hb; W3: the owned blocks; two S-steps each) do not depend on this step's
        // activations: requested here, a barrier and a phase ahead of their use (P4 fetched them per tile, latency exposed every time)
        // (W3 early only where its 16 NB registers fit beside the increments: with four tiles of four blocks they are requested at the
        //  top of P4 instead -- d = 500: 12.08 -> 11.9 ms)
        constexpr bool W3_EARLY = !(NT == 4 && NB == 4);
        f16x8 w2h[2], w2l[2], w3h[2][NB], w3l[2][NB];
        auto load_w3 = [&]() __attribute__((always_inline)) {
            const float* t3 = opaque_base(T + W::xW3 + (size_t)b0 * 512);
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                if (owner) {
#pragma unroll
                    for (int m = 0; m < NB; m += 2) {
                        gptr8_t t3p = sgpr_ptr8(t3 + ((size_t)S * DB + m) * 512);
                        const unsigned o2 = (m + 2 == NB && padlast) ? 0u : 128u;
                        w3h[S][m] = t3p[ul]; w3l[S][m] = t3p[64 + ul];
                        w3h[S][m + 1] = t3p[o2 + ul]; w3l[S][m + 1] = t3p[o2 + 64 + ul];
                    }
                }
            }
        };
        {
            const float* t2 = opaque_base(T + W::xW2 + (size_t)hb * 512);
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                gptr8_t tp = sgpr_ptr8(t2 + (size_t)S * HB * 512);
                w2h[S] = tp[ul]; w2l[S] = tp[64 + ul];
            }
            if constexpr (W3_EARLY) load_w3();
        }
        PSP_STAMP(cs2);
        __syncthreads();                                                 // B: h1 exchange complete; every wave is done with the state image
        PSP_STAMP(cs3);
        // ---- P3: second hidden layer, block hb of this wave's tiles
        {
            f32x4 h2[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                f32x4 hin[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) hin[m] = hx1[((ht0 + i) * HB + m) * 64];
                f32x4 acc1 = vb2[hb * 4], corr1 = {0.f, 0.f, 0.f, 0.f};       // (the order of gemm_regs_x3: S ascending, main, then the two corrections)
#pragma unroll
                for (int S = 0; S < 2; ++S) {
                    f16x8 bh, bl;
                    split_pack(hin[2 * S], hin[2 * S + 1], bh, bl);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[S], bh, acc1, 0, 0, 0);
                    corr1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[S], bl, corr1, 0, 0, 0);
                    corr1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[S], bh, corr1, 0, 0, 0);
                }
                h2[i] = tanh4(acc1 + kSplitInv * corr1);
            }
            if constexpr (!C::TWO) __syncthreads();                      // B2: every wave has read h1 -- h2 goes into the same buffer
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                hx2[((ht0 + i) * HB + hb) * 64] = h2[i];
                if (k_store && tvalid[ht0 + i]) {
                    gwptr_t ph = pbase(ht0 + i, G::pH2 + hb * 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) PSP_PATH_STORE(ph + r * 64 + ul, h2[i][r]);
                }
            }
        }
        PSP_STAMP(cs4);
        __syncthreads();                                                 // C: h2 exchange complete
        PSP_STAMP(cs5);
        // ---- P4: Z = W3 h2 + b3 on the owned blocks, row sums |Z|^2 and Z.xi (solver.py:477-478),
        //      v = c dt + xi sqrt(dt) (c = -Z if adaptive, solver.py:451-456) -> increment image (dense sigma) or x += sigma v
        if (owner) {
            if constexpr (!W3_EARLY) load_w3();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 hin[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) hin[m] = hx2[(t * HB + m) * 64];
                f32x4 Z[NB], zc[NB];
#pragma unroll
                for (int m = 0; m < NB; ++m) { Z[m] = vb3[blk(m) * 4]; zc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int S = 0; S < 2; ++S) {
                    f16x8 bh, bl;
                    split_pack(hin[2 * S], hin[2 * S + 1], bh, bl);
#pragma unroll
                    for (int m = 0; m < NB; ++m) {
                        Z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3h[S][m], bh, Z[m], 0, 0, 0);
                        zc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3h[S][m], bl, zc[m], 0, 0, 0);
                        zc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3l[S][m], bh, zc[m], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < NB; ++m) Z[m] = Z[m] + kSplitInv * zc[m];
                if (C::MAYPAD && padlast) Z[NB - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                float S = 0.f, Pz = 0.f;
#pragma unroll
                for (int m = 0; m < NB; ++m) {
                    const int b = b0 + m;
                    const f32x4 xv = xi[t][m];
                    if (k_store && tvalid[t] && !(C::MAYPAD && m == NB - 1 && padlast)) {   // image in the xi slot: c_xi xi + c_z Z (see hjb_fwd_kernel)
                        gwptr_t pxi = pbase(t, G::pXi + b * 256);
                        const f32x4 wv = store_cxi * xv + store_cz * Z[m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) PSP_PATH_STORE(pxi + r * 64 + ul, wv[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S = fmaf(Z[m][r], Z[m][r], S);
                        Pz = fmaf(Z[m][r], xv[r], Pz);
                    }
                    xi[t][m] = k_adaptive ? (sqdt * xv - dt * Z[m]) : (sqdt * xv);      // the increment v replaces xi
                }
                S = qsum(S);
                Pz = qsum(Pz);
                if (q == 0) {
                    red[(t * NWV + wave) * 16 + j] = S;
                    red[NT * NWV * 16 + (t * NWV + wave) * 16 + j] = Pz;
                }
            }
            if (k_sigma == SIGMA_DENSE) {
                write_image(img1, xi);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m) X[t][m] += (k_sigma == SIGMA_SCALE ? a.sigma_scale : 1.0f) * xi[t][m];
            }
        }
        PSP_STAMP(cs6);
        __syncthreads();                                                 // D: increment image and partial sums complete
        PSP_STAMP(cs7);
        // ---- Y += (-h + Z.c) dt + Z.xi sqrt(dt): wave t for tile t, partials of the owning waves in a fixed order
        if (wave < NT) {
            float S = 0.f, Pz = 0.f;
#pragma unroll
            for (int w = 0; w < C::NOWN; ++w) {
                S += red[(wave * NWV + w) * 16 + j];
                Pz += red[NT * NWV * 16 + (wave * NWV + w) * 16 + j];
            }
            if (k_loss == LOSS_RELENT) {
                Y = Y - (0.5f * S) * dt;                                // Y carries -Zsum (hjb_fwd_kernel)
            } else {
                const float drift_y = k_adaptive ? (0.f - 0.5f * S) : (0.f + 0.5f * S);
                Y = Y + drift_y * dt + Pz * sqdt;
            }
        }
        // ---- P5: x += B v on the owned blocks, then the state image of the next step
        if (k_sigma == SIGMA_DENSE && owner)
            coop_gemm<NB, NT, KS8, DB>(X, T + W::xB + (size_t)b0 * 512, img1, C::IMG8, lane, NoCoopBetween(), padlast);
        zero_pad(X);
        if constexpr (!C::TWO) {
            PSP_STAMP(cs8a);
            __syncthreads();                                             // E: every wave is done with the increment image
            PSP_STAMP(cs8b);
            PSP_ACC(4, cs8b, cs8a);
        }
        write_image(img0, X);
        PSP_STAMP(cs8);
        __syncthreads();                                                 // A (of the next step): state image complete
        PSP_STAMP(cs9);
        PSP_ACC(0, cs1, cs0);   // X store + P2 (drift product [+ Philox])
        PSP_ACC(1, cs4, cs3);   // P3
        PSP_ACC(2, cs6, cs5);   // P4
        PSP_ACC(3, cs8, cs7);   // Y + P5 + image (four tiles: with barrier E)
        PSP_ACC(4, cs3, cs2); PSP_ACC(4, cs5, cs4); PSP_ACC(4, cs7, cs6); PSP_ACC(4, cs9, cs8);   // barriers B, C, D, A (+ E)
        PSP_ACC(5, cs2, cs1);   // P1 (h1 [+ Philox])
        PSP_ACC(6, cs9, cs0);
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0) {
        stamps[7] = (unsigned long long)a.N;
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
    }
#endif

    // ---- terminal cost g(X_N) and D = Y - g  (problems.py:49,164,334; solver.py:167-168)
    if (owner) {
        const f32x4* vterm = vecs0 + W::vterm / 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float g = 0.f;
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                if (C::MAYPAD && m == NB - 1 && padlast) continue;       // (a padding block has no terminal cost: (0 - 1)^2 tv is not zero)
                const f32x4 tv = vterm[blk(m) * 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = X[t][m][r];
                    if (a.term_kind == TERM_LINEAR) g = fmaf(tv[r], x, g);
                    else if (a.term_kind == TERM_DIAGQ) g = fmaf(tv[r] * x, x, g);
                    else g = fmaf(tv[r] * (x - 1.0f), (x - 1.0f), g);
                }
            }
            g = qsum(g);
            if (q == 0) red[(t * NWV + wave) * 16 + j] = g;
            if (a.XN && kvalid[t]) {
#pragma unroll
                for (int m = 0; m < NB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * (b0 + m) + 4 * r + q;
                        if (f < D) a.XN[(size_t)kk[t] * D + f] = X[t][m][r];
                    }
            }
        }
    }
    __syncthreads();
    double sD = 0.0, sD2 = 0.0;
    if (wave < NT) {
        float g = 0.f;
#pragma unroll
        for (int w = 0; w < C::NOWN; ++w) g += red[(wave * NWV + w) * 16 + j];
        const float Dk = Y - g;
        bool kv = false;
        int k = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) if (wave == t) { kv = kvalid[t]; k = kk[t]; }
        if (kv && q == 0) {
            a.D[k] = Dk;
            if (a.Fint) a.Fint[k] = 0.f;
            if (a.Yout) a.Yout[k] = Y;
            sD = (double)Dk; sD2 = (double)Dk * (double)Dk;
        }
    }
    sD = jsum(sD); sD2 = jsum(sD2);
    __syncthreads();
    double* redd = reinterpret_cast<double*>(lds + C::cRed);
    if (lane == 0 && wave < NT) { redd[2 * wave] = sD; redd[2 * wave + 1] = sD2; }
    __syncthreads();
    if (tid == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < NT; ++w) { t0 += redd[2 * w]; t1 += redd[2 * w + 1]; }
        a.fwd_partial[2 * blockIdx.x] = t0;
        a.fwd_partial[2 * blockIdx.x + 1] = t1;
    }
}

template <int D, int H>
struct HjbcLaunch {
    // tiles per workgroup: four when that still gives every CU a workgroup, else two
    static int lds_bytes(int nt) { return (nt == 4 ? GeoC<D, H, 4>::lds_floats : GeoC<D, H, 2>::lds_floats) * 4; }
    template <int NT>
    static hipError_t fwd_nt(const HjbArgs& a, int grid, hipStream_t s) {
        const int bytes = GeoC<D, H, NT>::lds_floats * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbc_fwd_kernel<D, H, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbc_fwd_kernel<D, H, NT>), dim3(grid), dim3(512), bytes, s, a);
        return hipGetLastError();
    }
    // grid = ceil(ntile16 / nt) workgroups of 512 threads; the x3 tables of hjbw_tables_kernel(.., 3)
    static hipError_t fwd(const HjbArgs& a, int grid, int nt, hipStream_t s) {
        hipError_t e = HjbwLaunch<D, H>::tables(a, 3, s);
        if (e != hipSuccess) return e;
        return nt == 4 ? fwd_nt<4>(a, grid, s) : fwd_nt<2>(a, grid, s);
    }
};

}  // namespace psp
