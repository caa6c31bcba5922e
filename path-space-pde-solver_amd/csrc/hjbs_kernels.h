// hjbs_kernels.h -- feature-split forward rollout for SMALL trajectory counts.
//
// hjb_fwd_kernel gives every wave a whole 16-trajectory tile: with K = 1024 that is 64 waves on a chip with 1024 SIMDs,
// and the N time steps of a tile are sequential (36 k cycles each).  Here the W (4 or 8) waves of a workgroup share ONE tile
// and split every product by OUTPUT block (wave w owns state blocks w, w+W, ... and hidden blocks w, w+W, ...):
//   * a wave's slice of all five weight tables (W1, W2, W3, dt A, B rows of its blocks) is a few hundred A-operand
//     fragments -- they are loaded once and stay IN REGISTERS for the whole kernel; the time loop issues no table loads;
//   * activations travel through four small LDS images (X_n, h1, h2, v: the B operands, one dword per lane and k-step),
//     four workgroup barriers per step;
//   * Philox / tanh / row sums are done for the own blocks only (no redundancy);
//   * Y_N = sum over steps of terms LINEAR in the per-step row sums (|Z|^2, Z.xi, f(X)), so each wave accumulates its
//     own partial Y over the features it owns and the four partials are added once, after the last step.
// Same algebra, reference lines, Philox counters and path-store format as hjb_fwd_kernel (the backward kernels are
// unchanged); the summation order of Y differs (per-wave partials), which moves D by a few ulp.
#pragma once
#include "hjb_kernels.h"

namespace psp {

template <int D, int H>
struct GeoS {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB, KP = 4 * DB, KH = 4 * HB;
    static constexpr int KD = G::KSD;                                // k-steps of a d-deep contraction that carry real features (<= KP)
    // waves per workgroup (= per tile).  More than four state or hidden blocks: EIGHT waves, two per SIMD -- the matrix-pipe
    // time of a step is the same (the pipe is per SIMD), but one wave's Philox / tanh / LDS round trips and barrier waits
    // now run under the other wave's MFMAs, and a wave owns half the blocks (half the dependent-chain length per phase)
    static constexpr int W = (DB > 4 || HB > 4) ? 8 : 4;
    static constexpr int NBo = cdiv(DB, W), NHo = cdiv(HB, W);       // owned state / hidden blocks per wave
    // LDS (floats): per-feature vectors, the four images, cross-wave reduction scratch
    static constexpr int vb1 = 0, vw1t = vb1 + HB * 16, vb2 = vw1t + HB * 16, vb3 = vb2 + HB * 16,
                         vdr = vb3 + DB * 16, vrun = vdr + DB * 16, vterm = vrun + DB * 16,
                         iX = vterm + DB * 16, iH1 = iX + KP * 64, iH2 = iH1 + KH * 64, iV = iH2 + KH * 64,
                         fRed = iV + KP * 64, lds_floats = fRed + W * 64;
};

// FAST: on-device noise, no u_L2 log, no time-feature table (decided at launch).  The time loop of that instance has no
// vector-memory LOAD: a load in a wave-uniform branch leaves an `s_waitcnt vmcnt(0)` at the join on the common path, and
// vmcnt counts in order, so every step waited for its own path-store writes (hjbq_kernels.h has the same split).
template <int D, int H, int FAST_>
__global__ __launch_bounds__((64 * GeoS<D, H>::W)) void hjbs_fwd_kernel(const HjbArgs a) {
    // FAST_ = 2: the problem switches of the LLGC configurations as compile-time constants (hjb_kernels.h, hjb_fwd_kernel)
    constexpr bool FAST = FAST_ != 0, SPEC = FAST_ == 2;
    const int k_drift = SPEC ? (int)DRIFT_DENSE : a.drift_kind, k_sigma = SPEC ? (int)SIGMA_DENSE : a.sigma_kind;
    const int k_run = SPEC ? (int)RUN_ZERO : a.runcost_kind, k_loss = SPEC ? (int)LOSS_LOGVAR : a.loss_kind;
    const bool k_adaptive = SPEC ? true : (a.adaptive != 0);
    const uint32_t iter_now = a.iter_dev ? *a.iter_dev : a.iter;       // wave-uniform scalar load
    using G = Geo<D, H>;
    using S_ = GeoS<D, H>;
    constexpr int DB = S_::DB, HB = S_::HB, KP = S_::KP, KH = S_::KH, NBo = S_::NBo, NHo = S_::NHo, W = S_::W, KD = S_::KD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;
    const bool hasH = wave < HB, hasS = wave < DB;     // owns at least one hidden / state block (wave-uniform)

    stage_vec(lds + S_::vb1, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob1 + f] : 0.f; });
    stage_vec(lds + S_::vw1t, HB, tid, nthr, [&](int f) { return f < H ? P[G::oW1 + f * (D + 1)] : 0.f; });
    stage_vec(lds + S_::vb2, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob2 + f] : 0.f; });
    stage_vec(lds + S_::vb3, DB, tid, nthr, [&](int f) { return f < D ? P[G::ob3 + f] : 0.f; });
    stage_vec(lds + S_::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (k_drift == DRIFT_DIAG || k_drift == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + S_::vrun, DB, tid, nthr, [&](int f) {
        return (f < D && k_run == RUN_DIAGQ) ? a.runcost[f] : 0.f; });
    stage_vec(lds + S_::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });

    // ---- this wave's weight slices as A-operand fragments: lane (i, q) of fragment (block mb, k-step ks) holds
    //      W[16 mb + rowmap(i)][4 ks + q]  (rowmap as in stage_aop)
    const int ri = 4 * ((lane & 15) & 3) + ((lane & 15) >> 2);
    float w1r[NHo][KD], w2r[NHo][KH], w3r[NBo][KH], ar[NBo][KD], br[NBo][KD];
    const bool denseA = k_drift == DRIFT_DENSE, denseB = k_sigma == SIGMA_DENSE;
#pragma unroll
    for (int io = 0; io < NHo; ++io) {
        const int row = 16 * (wave + W * io) + ri;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const int col = 4 * ks + q;
            w1r[io][ks] = (row < H && col < D) ? P[G::oW1 + row * (D + 1) + 1 + col] : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            const int col = 4 * ks + q;
            w2r[io][ks] = (row < H && col < H) ? P[G::oW2 + row * H + col] : 0.f;
        }
    }
#pragma unroll
    for (int io = 0; io < NBo; ++io) {
        const int row = 16 * (wave + W * io) + ri;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            const int col = 4 * ks + q;
            w3r[io][ks] = (row < D && col < H) ? P[G::oW3 + row * H + col] : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const int col = 4 * ks + q;
            const bool in = row < D && col < D;
            ar[io][ks] = (denseA && in) ? a.dt * a.drift[row * D + col] : 0.f;
            br[io][ks] = (denseB && in) ? a.sigma[row * D + col] : 0.f;
        }
    }
    __syncthreads();

    const int t16 = blockIdx.x;                        // one tile per workgroup
    const int k = t16 * 16 + j;
    const bool kvalid = k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    const float dt = a.dt, sqdt = a.sqdt;
    float* imgX = lds + S_::iX;
    float* imgH1 = lds + S_::iH1;
    float* imgH2 = lds + S_::iH2;
    float* imgV = lds + S_::iV;
    const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;               // index by block * 4
    const f32x4* vterm = vecs0 + S_::vterm / 4;

    f32x4 X[NBo];                                      // own blocks of X_0 (solver.py:365-367), T layout
#pragma unroll
    for (int io = 0; io < NBo; ++io)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * (wave + W * io) + 4 * r + q;
            const float v = a.x0[(size_t)(kvalid ? k : 0) * a.x0_stride + (f < D ? f : D - 1)];
            X[io][r] = (f < D && kvalid) ? v : 0.f;
        }
    float Yw = 0.f, Fw = 0.f, ULw = 0.f;               // this wave's partial of Y, of the running-cost integral and of u_L2
    const float store_cxi = (a.store_path == 3) ? 0.f : 1.f;                    // image in the xi slot: c_xi xi + c_z Z
    const float store_cz = (a.store_path == 3) ? 1.f : (a.store_path == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));

#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#pragma unroll 1
    for (int n = 0; n < a.N; ++n) {
        PSP_STAMP(ss0);
        float tn = (float)n * dt;
        if constexpr (!FAST) { if (a.tfeat) tn = a.tfeat[n]; }
        const f32x4* vecs = opaque(vecs0);             // re-read the small vectors each step (no hoisting)
        const int qn = opaque_i(q);
        const f32x4* vb1 = vecs + S_::vb1 / 4;
        const f32x4* vw1t = vecs + S_::vw1t / 4;
        const f32x4* vb2 = vecs + S_::vb2 / 4;
        const f32x4* vb3 = vecs + S_::vb3 / 4;
        const f32x4* vdr = vecs + S_::vdr / 4;
        const f32x4* vrun = vecs + S_::vrun / 4;
        float* pblk = a.path + ((size_t)n * a.ntile16 + t16) * (size_t)G::PB + lane;
        // ---- P0: own blocks of X_n -> image (B operand of the W1 and drift products) and path store
#pragma unroll
        for (int io = 0; io < NBo; ++io) {
            const int sb = wave + W * io;
            if (sb < DB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) imgX[(4 * sb + r) * 64 + lane] = X[io][r];
                if (a.store_path) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pblk[(G::pX / 64 + 4 * sb + r) * 64] = X[io][r];
                }
            }
        }
        __syncthreads();
        PSP_STAMP(ss1);
        // ---- P1: own hidden blocks of h1 = tanh(W1 [t, x] + b1) and the drift part of own state blocks, one k-loop
        f32x4 h1[NHo];
#pragma unroll
        for (int io = 0; io < NHo; ++io) {
            const int hb = (wave + W * io) < HB ? (wave + W * io) : HB - 1;
            h1[io] = vb1[hb * 4] + tn * vw1t[hb * 4];
        }
        // (wave-uniform tests stay OUTSIDE the unrolled k-loops: inside, every k-step became a branch with accumulator moves
        //  and hazard nops around it).  With eight waves some own no hidden block / no state block: they skip the product
        //  instead of multiplying zero weights (the matrix pipe is shared by the two waves of a SIMD), and a wave WITHOUT a
        //  hidden block defers its drift product to phase P2, where the pipe only carries the short W2 product
        if (hasH) {
            float bx[KD];
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) bx[ks] = imgX[ks * 64 + lane];
            if (denseA && hasS) {
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) {
#pragma unroll
                    for (int io = 0; io < NHo; ++io) h1[io] = mfma16(w1r[io][ks], bx[ks], h1[io]);
#pragma unroll
                    for (int io = 0; io < NBo; ++io) X[io] = mfma16(ar[io][ks], bx[ks], X[io]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KD; ++ks)
#pragma unroll
                    for (int io = 0; io < NHo; ++io) h1[io] = mfma16(w1r[io][ks], bx[ks], h1[io]);
            }
        }
        if (k_drift == DRIFT_DIAG) {
#pragma unroll
            for (int io = 0; io < NBo; ++io) {
                const int sb = (wave + W * io) < DB ? (wave + W * io) : DB - 1;
                X[io] += dt * (vdr[sb * 4] * X[io]);
            }
        } else if (k_drift == DRIFT_DWELL) {
#pragma unroll
            for (int io = 0; io < NBo; ++io) {
                const int sb = (wave + W * io) < DB ? (wave + W * io) : DB - 1;
                X[io] -= dt * (4.0f * vdr[sb * 4] * (X[io] * (X[io] * X[io] - 1.0f)));
            }
        }
        if (hasH)
#pragma unroll
        for (int io = 0; io < NHo; ++io) {
            const int hb = wave + W * io;
            h1[io] = tanh4(h1[io]);
            if (hb < HB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) imgH1[(4 * hb + r) * 64 + lane] = h1[io][r];
                if (a.store_path) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pblk[(G::pH1 / 64 + 4 * hb + r) * 64] = h1[io][r];
                }
            }
        }
        __syncthreads();
        PSP_STAMP(ss2);
        // ---- P2: own hidden blocks of h2 = tanh(W2 h1 + b2)
        f32x4 h2[NHo];
#pragma unroll
        for (int io = 0; io < NHo; ++io) {
            const int hb = (wave + W * io) < HB ? (wave + W * io) : HB - 1;
            h2[io] = vb2[hb * 4];
        }
        if (!hasH && hasS && denseA) {                 // deferred from P1: the X_n image stays valid until the end of P4
            float bx[KD];
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) bx[ks] = imgX[ks * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < KD; ++ks)
#pragma unroll
                for (int io = 0; io < NBo; ++io) X[io] = mfma16(ar[io][ks], bx[ks], X[io]);
        }
        if (hasH) {
            float bh[KH];
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) bh[ks] = imgH1[ks * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                for (int io = 0; io < NHo; ++io) h2[io] = mfma16(w2r[io][ks], bh[ks], h2[io]);
        }
        if (hasH)
#pragma unroll
        for (int io = 0; io < NHo; ++io) {
            const int hb = wave + W * io;
            h2[io] = tanh4(h2[io]);
            if (hb < HB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) imgH2[(4 * hb + r) * 64 + lane] = h2[io][r];
                if (a.store_path) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pblk[(G::pH2 / 64 + 4 * hb + r) * 64] = h2[io][r];
                }
            }
        }
        __syncthreads();
        PSP_STAMP(ss3);
        // ---- P3: own state blocks of Z = W3 h2 + b3, Brownian increment, row-sum partials, increment panel v
        f32x4 Z[NBo];
#pragma unroll
        for (int io = 0; io < NBo; ++io) {
            const int sb = (wave + W * io) < DB ? (wave + W * io) : DB - 1;
            Z[io] = vb3[sb * 4];
        }
        if (hasS) {
            float bh[KH];
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) bh[ks] = imgH2[ks * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                for (int io = 0; io < NBo; ++io) Z[io] = mfma16(w3r[io][ks], bh[ks], Z[io]);
        }
        float S = 0.f, Pz = 0.f, UL = 0.f;
#pragma unroll
        for (int io = 0; io < NBo; ++io) {
            const int sb = wave + W * io;
            if (sb < DB) {
                f32x4 xi;
                if (FAST || a.noise_mode == NOISE_PHILOX) {
                    xi = philox_block(kglob, (uint32_t)n, (uint32_t)(4 * sb + qn), iter_now, a.seed_lo, a.seed_hi);
                } else {
                    const float* xrow = a.xi + ((size_t)(n + 1) * a.K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * sb + 4 * r + q;
                        const float v = xrow[f < D ? f : D - 1];
                        xi[r] = (f < D && kvalid) ? v : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) if (16 * sb + 4 * r + q >= D) xi[r] = 0.f;
                if (a.store_path && a.store_path != 4) {   // 1: xi, 2: xi - sqrt(dt) Z, 3: Z, 4: no image (see hjb_fwd_kernel)
                    const f32x4 wv = store_cxi * xi + store_cz * Z[io];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pblk[(G::pXi / 64 + 4 * sb + r) * 64] = wv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S = fmaf(Z[io][r], Z[io][r], S);
                    Pz = fmaf(Z[io][r], xi[r], Pz);
                }
                if (!FAST && a.uref) {                 // u_L2 logging: |-Z_n - u*(t_n)|^2 (solver.py:491-494)
                    const float* ur = a.uref + (size_t)n * D;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * sb + 4 * r + q;
                        const float e = (f < D) ? Z[io][r] + ur[f < D ? f : D - 1] : 0.f;
                        UL = fmaf(e, e, UL);
                    }
                }
                const f32x4 v = k_adaptive ? (sqdt * xi - dt * Z[io]) : (sqdt * xi);     // v = c dt + xi sqrt(dt)
                if (denseB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) imgV[(4 * sb + r) * 64 + lane] = v[r];
                } else if (k_sigma == SIGMA_SCALE) {
                    X[io] += a.sigma_scale * v;
                } else {
                    X[io] += v;
                }
            }
        }
        PSP_STAMP(ss4);
        if (denseB) {
            __syncthreads();
            // ---- P4: X += B v for the own state blocks
            if (hasS) {
                float bv[KD];
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) bv[ks] = imgV[ks * 64 + lane];
#pragma unroll
                for (int ks = 0; ks < KD; ++ks)
#pragma unroll
                    for (int io = 0; io < NBo; ++io) X[io] = mfma16(br[io][ks], bv[ks], X[io]);
            }
        }
        // ---- running cost f(X_{n+1}) over the own blocks and the wave's partial of the Y update (solver.py:477-478):
        //      Y += (f -/+ 0.5 |Z|^2) dt + Z.xi sqrt(dt) is linear in the three row sums
        float fX = 0.f;
        if (k_run == RUN_DIAGQ) {
#pragma unroll
            for (int io = 0; io < NBo; ++io) {
                const int sb = wave + W * io;
                if (sb < DB) {
                    const f32x4 pv = vrun[sb * 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) fX = fmaf(pv[r] * X[io][r], X[io][r], fX);
                }
            }
        }
        const float term = (k_loss == LOSS_RELENT) ? -(0.5f * S + fX) * dt        // Y carries -Zsum (hjb_fwd_kernel)
                           : (k_adaptive ? (fX - 0.5f * S) : (fX + 0.5f * S)) * dt + Pz * sqdt;
        Yw += term;
        Fw = fmaf(fX, dt, Fw);
        ULw = fmaf(UL, dt, ULw);
        PSP_STAMP(ss5);
        PSP_ACC(0, ss1, ss0);   // P0: X image + store + barrier
        PSP_ACC(1, ss2, ss1);   // P1: W1 + drift product, tanh, h1 image + barrier
        PSP_ACC(2, ss3, ss2);   // P2: W2 product, tanh, h2 image + barrier
        PSP_ACC(3, ss4, ss3);   // P3: W3 product, Philox, v image
        PSP_ACC(4, ss5, ss4);   // P4: barrier + sigma product + costs
        PSP_ACC(6, ss5, ss0);
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0) {
        stamps[7] = (unsigned long long)a.N;
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
    }
#endif

    // ---- terminal cost over the own blocks; the four partials of a trajectory meet in LDS
    float g = 0.f;
#pragma unroll
    for (int io = 0; io < NBo; ++io) {
        const int sb = wave + W * io;
        if (sb < DB) {
            const f32x4 tv = vterm[sb * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = X[io][r];
                if (a.term_kind == TERM_LINEAR) g = fmaf(tv[r], x, g);
                else if (a.term_kind == TERM_DIAGQ) g = fmaf(tv[r] * x, x, g);
                else g = fmaf(tv[r] * (x - 1.0f), (x - 1.0f), g);
            }
            if (a.XN && kvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * sb + 4 * r + q;
                    if (f < D) a.XN[(size_t)k * D + f] = X[io][r];
                }
            }
        }
    }
    const float Yp = qsum(Yw), Fp = qsum(Fw), gp = qsum(g), Up = qsum(ULw);
    float* red = lds + S_::fRed;                       // [wave][j]: Y, then F, then g partials
    __syncthreads();                                   // the images are dead; fRed is separate, but keep the phases apart
    if (q == 0) red[wave * 64 + j] = Yp;
    if (q == 1) red[wave * 64 + 16 + j] = Fp;
    if (q == 2) red[wave * 64 + 32 + j] = gp;
    if (q == 3) red[wave * 64 + 48 + j] = Up;
    __syncthreads();
    if (wave == 0) {
        float Ys = 0.f, F = 0.f, gt = 0.f, Us = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) {                  // fixed order: bitwise reproducible
            Ys += red[w * 64 + j]; F += red[w * 64 + 16 + j]; gt += red[w * 64 + 32 + j]; Us += red[w * 64 + 48 + j];
        }
        const float Y = (a.y0 ? a.y0[0] : 0.f) + Ys;
        const float Dk = Y - gt;
        if (kvalid && q == 0) {
            a.D[k] = Dk;
            if (a.Fint) a.Fint[k] = F;
            if (a.uref) a.ul2[k] = Us;
            if (a.Yout) a.Yout[k] = Y;
        }
        double sD = (kvalid && q == 0) ? (double)Dk : 0.0, sD2 = (kvalid && q == 0) ? (double)Dk * (double)Dk : 0.0;
        sD = jsum(sD); sD2 = jsum(sD2);
        if (lane == 0) { a.fwd_partial[2 * blockIdx.x] = sD; a.fwd_partial[2 * blockIdx.x + 1] = sD2; }
    }
}

template <int D, int H>
struct HjbsLaunch {
    static int lds_bytes() { return GeoS<D, H>::lds_floats * 4; }
    template <int FAST>
    static hipError_t fwd_as(const HjbArgs& a, int grid, hipStream_t s) {
        const int bytes = lds_bytes();
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbs_fwd_kernel<D, H, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbs_fwd_kernel<D, H, FAST>), dim3(grid), dim3(64 * GeoS<D, H>::W), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t fwd(const HjbArgs& a, int grid, hipStream_t s) {
        const bool fast = a.noise_mode == NOISE_PHILOX && a.uref == nullptr && a.tfeat == nullptr;
        const bool spec = spec_enabled() && fast && a.drift_kind == DRIFT_DENSE && a.sigma_kind == SIGMA_DENSE && a.adaptive && a.runcost_kind == RUN_ZERO &&
                          a.loss_kind != LOSS_RELENT;
        return spec ? fwd_as<2>(a, grid, s) : fast ? fwd_as<1>(a, grid, s) : fwd_as<0>(a, grid, s);
    }
};

}  // namespace psp
