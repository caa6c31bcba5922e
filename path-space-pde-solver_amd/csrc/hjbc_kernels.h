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

template <int D, int H>
struct GeoC {
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    static constexpr int DB = W::DB, HB = W::HB, KS8 = W::KS8;
    static constexpr int NWV = 8, NT = 2, NB = 4, NP = 2;          // waves, tiles per workgroup, state blocks / S-steps per wave
    static_assert(DB % 4 == 0 && DB <= NWV * NB, "the cooperative forward deals out whole groups of four state blocks");
    static constexpr int NOWN = DB / NB;                            // waves that own state blocks (the others only run the hidden layers)
    static constexpr int IMG8 = KS8 * 2 * 64;                       // f16x8 elements of one tile's hi / lo image
    // LDS (floats): the per-feature vectors of GeoW, then two image regions (state, increment), the h1 / h2 exchanges and the partial sums
    static constexpr int cImg0 = W::fImg, cImg1 = cImg0 + NT * IMG8 * 4, cH1 = cImg1 + NT * IMG8 * 4, cH2 = cH1 + NT * HB * 256,
                         cRed = cH2 + NT * HB * 256, lds_floats = cRed + 2 * NT * NWV * 16 + 64;
};

// acc[t][m] += T[S][b0 + m] . img[t][S] over all S-steps for the NB owned blocks of both tiles; WITHH: in the same loop
// hacc += TH[S][hb] . img[htile][S] (the first hidden layer reads the same state image).  Rolled over S in pairs (static ring
// indices), operands of the next S-step requested before the MFMAs of this one.
template <int NB, int NT, int KS8, int LDT, bool WITHH, bool ALLOWN = false>
__device__ __forceinline__ void coop_gemm(f32x4 (&acc)[NT][NB], const float* __restrict__ tbl, bool owner_, f32x4& hacc,
                                          const float* __restrict__ tblh, int LDH, const f16x8* img8, int img_tile_stride, int htile,
                                          int lane) {
    static_assert(KS8 % 2 == 0 && NB % 2 == 0, "S-steps in pairs");
    const bool owner = ALLOWN ? true : owner_;                          // (d = 500: every wave owns blocks -- no scalar branches in the k-loop)
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    tblh = opaque_base(tblh);
    f16x8 ah[2][NB], al[2][NB], bh[2][NT], bl[2][NT];
    [[maybe_unused]] f16x8 wh[2], wl[2];
    f32x4 corr[NT][NB];
    f32x4 hcorr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < NB; ++m) corr[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int st, int S) __attribute__((always_inline)) {
        if (owner) {
#pragma unroll
            for (int m = 0; m < NB; m += 2) {                           // fresh SGPR base every 4 KiB (two output blocks)
                gptr8_t tp = sgpr_ptr8(tbl + ((size_t)S * LDT + m) * 512);
                ah[st][m] = tp[ul]; al[st][m] = tp[64 + ul];
                ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = tp[192 + ul];
            }
        }
        if constexpr (WITHH) {
            gptr8_t tp = sgpr_ptr8(tblh + (size_t)S * LDH * 512);
            wh[st] = tp[ul]; wl[st] = tp[64 + ul];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (owner || (WITHH && t == htile)) {
                bh[st][t] = img8[t * img_tile_stride + (2 * S) * 64];
                bl[st][t] = img8[t * img_tile_stride + (2 * S + 1) * 64];
            }
        }
    };
    auto products = [&](int st) __attribute__((always_inline)) {
        if (owner) {
#pragma unroll
            for (int m = 0; m < NB; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[st][m], bh[st][t], acc[t][m], 0, 0, 0);
                    corr[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[st][m], bl[st][t], corr[t][m], 0, 0, 0);
                    corr[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[st][m], bh[st][t], corr[t][m], 0, 0, 0);
                }
        }
        if constexpr (WITHH) {
            // the hidden block's tile: a bit select with a wave-uniform mask (a ?: here becomes a scalar branch inside the k-loop)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned msk = htile == 0 ? 0u : ~0u;
            const u32x4 h0 = __builtin_bit_cast(u32x4, bh[st][0]), h1 = __builtin_bit_cast(u32x4, bh[st][NT - 1]);
            const u32x4 l0 = __builtin_bit_cast(u32x4, bl[st][0]), l1 = __builtin_bit_cast(u32x4, bl[st][NT - 1]);
            const f16x8 xh = __builtin_bit_cast(f16x8, (h0 & ~msk) | (h1 & msk)), xl = __builtin_bit_cast(f16x8, (l0 & ~msk) | (l1 & msk));
            hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st], xh, hacc, 0, 0, 0);
            hcorr = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st], xl, hcorr, 0, 0, 0);
            hcorr = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[st], xh, hcorr, 0, 0, 0);
        }
    };
    load(0, 0);
#pragma unroll 1
    for (int S = 0; S < KS8; S += 2) {
        load(1, S + 1);
        __builtin_amdgcn_sched_barrier(0);
        products(0);
        __builtin_amdgcn_sched_barrier(0);
        load(0, S + 2 < KS8 ? S + 2 : KS8 - 1);                         // past the end: re-read the last step (unused)
        __builtin_amdgcn_sched_barrier(0);
        products(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (owner) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < NB; ++m) acc[t][m] = acc[t][m] + kSplitInv * corr[t][m];
    }
    if constexpr (WITHH) hacc = hacc + kSplitInv * hcorr;
}

template <int D, int H>
__global__ __launch_bounds__(512, 2) void hjbc_fwd_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    using C = GeoC<D, H>;
    constexpr int DB = C::DB, HB = C::HB, KS8 = C::KS8, NT = C::NT, NB = C::NB, NP = C::NP, NWV = C::NWV;
    static_assert(NT == 2, "two tiles per workgroup");
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

    // ownership: state blocks b0 .. b0 + 3 (S-steps s0, s0 + 1) of both tiles; hidden block hb of tile ht
    const bool owner = (C::NOWN == NWV) ? true : wave < C::NOWN;
    const int b0 = owner ? wave * NB : 0, s0 = owner ? wave * NP : 0;
    const int hb = wave & 3, ht = wave >> 2;
    int t16[NT], kk[NT];
    bool tvalid[NT], kvalid[NT];
    uint32_t kglob[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int raw = blockIdx.x * NT + t;
        tvalid[t] = raw < a.ntile16;                     // the surplus tile of the last workgroup runs along on the last tile, stores nothing
        t16[t] = tvalid[t] ? raw : a.ntile16 - 1;
        kk[t] = t16[t] * 16 + j;
        kvalid[t] = tvalid[t] && kk[t] < a.K_local;
        kglob[t] = (uint32_t)(a.k_offset + kk[t]);
    }
    const float dt = a.dt, sqdt = a.sqdt;
    f16x8* img0 = reinterpret_cast<f16x8*>(lds + C::cImg0) + lane;      // state image   [tile][S][hi | lo][64]
    f16x8* img1 = reinterpret_cast<f16x8*>(lds + C::cImg1) + lane;      // increment image
    f32x4* hx1 = reinterpret_cast<f32x4*>(lds + C::cH1) + lane;         // h1 exchange [tile][block][64] of f32x4 (T layout: 4 r per lane)
    f32x4* hx2 = reinterpret_cast<f32x4*>(lds + C::cH2) + lane;
    float* red = lds + C::cRed;                                         // [2][tile][wave][16]
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float store_cxi = (k_store == 3) ? 0.f : 1.f;
    const float store_cz = (k_store == 3) ? 1.f : (k_store == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));

    // ---- X_0 (solver.py:365-367): owned blocks of both tiles, T layout
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
    float Y = a.y0 ? a.y0[0] : 0.f;                                      // (meaningful in waves 0 / 1: tile = wave)
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
                    for (int e = 0; e < 4 * NB; ++e) PSP_PATH_STORE(px + e * 64 + ul, X[t][e >> 2][e & 3]);
                }
        }
        // ---- P12: x += (dt A) x_n on the owned blocks; first hidden layer, block hb of tile ht (function_space.py:190-195)
        f32x4 h1 = vb1[hb * 4] + tn * vw1t[hb * 4];
        if (k_drift == DRIFT_DENSE) {
            coop_gemm<NB, NT, KS8, DB, true, C::NOWN == NWV>(X, T + W::xA + (size_t)b0 * 512, owner, h1, T + W::xW1 + (size_t)hb * 512, HB, img0, C::IMG8,
                                              ht, lane);
        } else {
            f32x4 none[NT][NB];
            coop_gemm<NB, NT, KS8, DB, true>(none, T + W::xA, false, h1, T + W::xW1 + (size_t)hb * 512, HB, img0, C::IMG8, ht, lane);
            if (k_drift == DRIFT_DIAG) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m) X[t][m] += dt * (vdr[(b0 + m) * 4] * X[t][m]);
            } else if (k_drift == DRIFT_DWELL) {                        // b = -4 kappa x (x^2 - 1), problems.py:311-315
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m)
                        X[t][m] -= dt * (4.0f * vdr[(b0 + m) * 4] * (X[t][m] * (X[t][m] * X[t][m] - 1.0f)));
            }
        }
        PSP_STAMP(cs1);
        h1 = tanh4(h1);
        hx1[(ht * HB + hb) * 64] = h1;
        if (k_store && tvalid[ht]) {
            gwptr_t ph = pbase(ht, G::pH1 + hb * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) PSP_PATH_STORE(ph + r * 64 + ul, h1[r]);
        }
        PSP_STAMP(cs2);
        __syncthreads();                                                 // B
        PSP_STAMP(cs3);
        // ---- P3: second hidden layer, block hb of tile ht
        {
            f32x4 hin[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) hin[m] = hx1[(ht * HB + m) * 64];
            f32x4 h2[1] = {vb2[hb * 4]};
            gemm_regs_x3<1, HB, HB>(h2, T + W::xW2 + (size_t)hb * 512, hin, lane);
            h2[0] = tanh4(h2[0]);
            hx2[(ht * HB + hb) * 64] = h2[0];
            if (k_store && tvalid[ht]) {
                gwptr_t ph = pbase(ht, G::pH2 + hb * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r) PSP_PATH_STORE(ph + r * 64 + ul, h2[0][r]);
            }
        }
        PSP_STAMP(cs4);
        __syncthreads();                                                 // C
        PSP_STAMP(cs5);
        // ---- P4: Z = W3 h2 + b3 on the owned blocks, Brownian increment, row sums |Z|^2 and Z.xi (solver.py:477-478),
        //      v = c dt + xi sqrt(dt) (c = -Z if adaptive, solver.py:451-456) -> increment image (dense sigma) or x += sigma v
        f32x4 V[NT][NB];
        if (owner) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 hin[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) hin[m] = hx2[(t * HB + m) * 64];
                f32x4 Z[NB];
#pragma unroll
                for (int m = 0; m < NB; ++m) Z[m] = vb3[(b0 + m) * 4];
                gemm_regs_x3<NB, HB, DB>(Z, T + W::xW3 + (size_t)b0 * 512, hin, lane);
                float S = 0.f, Pz = 0.f;
#pragma unroll
                for (int m = 0; m < NB; ++m) {
                    const int b = b0 + m;
                    f32x4 xi = philox_block(kglob[t], (uint32_t)n, (uint32_t)(4 * b + qn), iter_now, a.seed_lo, a.seed_hi);
                    if (16 * (DB - 1) + 16 > D) {                       // partial last block: keep padded features at zero
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (16 * b + 4 * r + q >= D) xi[r] = 0.f;
                    }
                    if (k_store && tvalid[t]) {                         // image in the xi slot: c_xi xi + c_z Z (see hjb_fwd_kernel)
                        gwptr_t pxi = pbase(t, G::pXi + b * 256);
                        const f32x4 wv = store_cxi * xi + store_cz * Z[m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) PSP_PATH_STORE(pxi + r * 64 + ul, wv[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S = fmaf(Z[m][r], Z[m][r], S);
                        Pz = fmaf(Z[m][r], xi[r], Pz);
                    }
                    V[t][m] = k_adaptive ? (sqdt * xi - dt * Z[m]) : (sqdt * xi);
                }
                S = qsum(S);
                Pz = qsum(Pz);
                if (q == 0) {
                    red[(t * NWV + wave) * 16 + j] = S;
                    red[NT * NWV * 16 + (t * NWV + wave) * 16 + j] = Pz;
                }
            }
            if (k_sigma == SIGMA_DENSE) {
                write_image(img1, V);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < NB; ++m) X[t][m] += (k_sigma == SIGMA_SCALE ? a.sigma_scale : 1.0f) * V[t][m];
            }
        }
        PSP_STAMP(cs6);
        __syncthreads();                                                 // D
        PSP_STAMP(cs7);
        // ---- Y += (-h + Z.c) dt + Z.xi sqrt(dt): waves 0 / 1 for tile 0 / 1, partials of the owning waves in a fixed order
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
        if (k_sigma == SIGMA_DENSE) {
            f32x4 nohid = zero4;
            coop_gemm<NB, NT, KS8, DB, false, C::NOWN == NWV>(X, T + W::xB + (size_t)b0 * 512, owner, nohid, T, 0, img1, C::IMG8, 0, lane);
        }
        write_image(img0, X);
        PSP_STAMP(cs8);
        __syncthreads();                                                 // A (of the next step)
        PSP_STAMP(cs9);
        PSP_ACC(0, cs1, cs0);   // X store + P12
        PSP_ACC(1, cs4, cs3);   // P3
        PSP_ACC(2, cs6, cs5);   // P4
        PSP_ACC(3, cs8, cs7);   // Y + P5 + image
        PSP_ACC(4, cs3, cs2); PSP_ACC(4, cs5, cs4); PSP_ACC(4, cs7, cs6); PSP_ACC(4, cs9, cs8);   // the four barriers
        PSP_ACC(5, cs2, cs1);   // tanh + h1 exchange / store
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
                const f32x4 tv = vterm[(b0 + m) * 4];
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
        const bool kv = wave == 0 ? kvalid[0] : kvalid[NT - 1];
        const int k = wave == 0 ? kk[0] : kk[NT - 1];
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
    using C = GeoC<D, H>;
    static int lds_bytes() { return C::lds_floats * 4; }
    // grid = ceil(ntile16 / 2) workgroups of 512 threads; the x3 tables of hjbw_tables_kernel(.., 3)
    static hipError_t fwd(const HjbArgs& a, int grid, hipStream_t s) {
        hipError_t e = HjbwLaunch<D, H>::tables(a, 3, s);
        if (e != hipSuccess) return e;
        const int bytes = C::lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbc_fwd_kernel<D, H>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbc_fwd_kernel<D, H>), dim3(grid), dim3(512), bytes, s, a);
        return hipGetLastError();
    }
};

}  // namespace psp
