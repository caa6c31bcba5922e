// hjbd_kernels.h -- forward rollout with a DenseNet control (function_space.py:116-140: dense-concat layers,
// relu(.)^2, weights stored (in, out)), for the two places the reference uses one as the control of Solver:
//   * time_approx='outer' (solver.py:88, the constructor default): N nets DenseNet(d -> d), one per time step,
//     Z_n = z_n[n](X_n) (solver.py:352-353)                                   -> per_step = 1, time_input = 0
//   * a DenseNet swapped into z_n with time_approx='inner' (notebook extension point, solver.py:142-162):
//     DenseNet(d+1 -> d) on [t, X] with time as input column 0 (solver.py:355) -> per_step = 0, time_input = 1
// Per step:  z1 = W1^T u + b1, r1 = relu(z1), h1 = r1^2;  z2 = W2^T [u, h1] + b2, r2 = relu(z2), h2 = r2^2;
//            Z = W3^T [u, h1, h2] + b3      (u = X_n or [t_n, X_n]),
// then the same Euler-Maruyama / Y update as hjbw_fwd_kernel (solver.py:471-478).
//
// Built on the wide family's streaming products (hjbw_kernels.h): one wave per SIMD owns a 16-trajectory tile, all
// A-operand tables live in global memory (L2-resident, written per call by hjbd_tables_kernel), so "one weight set
// per time step" is nothing but a table pointer that advances with n.  The tables are filled from the REAL
// parameter layout with zero padding up to the compiled (D, H) instance, and the time column of every layer is
// folded into per-step bias vectors (b + t_n W[0, :]) by the tables kernel: the host keeps no index map.
// What the backward pass needs -- X_n and the image of the Brownian increment -- is written ROW-MAJOR with the real
// widths ((N, K, d) each): the parameter gradient of this family is formed by library GEMMs on those flat batches
// (plan_dense_native.py), not by a hand-written kernel.
#pragma once
#include "hjbw_kernels.h"

namespace psp {

struct DnetArgs {
    HjbArgs h;                 // problem, noise, outputs (params = the first parameter set; path / tables unused here)
    float* tbl;                // table region (DGeo::table_floats(...) floats)
    float* px;                 // (N, K_local, d_real) X_n          (store_path)
    float* pxi;                // (N, K_local, d_real) image of xi_{n+1}: xi, or xi + sqrt(dt) Z for a non-adaptive process
    float* pr1;                // optional (N, K_local, h_real) relu(z1) and relu(z2): saves the gradient pass their recomputation
    float* pr2;
    float* pimg;               // optional: the same four quantities as T-layout register images, N x ntile16 blocks of DGeo::PBI
                               // floats (what hjbd_bwd_kernel reads; written instead of the row-major stores when set)
    const float* wts;          // backward: per-trajectory weights dL/dD_k, zero padded to 16 * ntile16
    float* partial;            // backward: [N * slices][DGeo::PP] partial gradients in the padded (instance) layout
    int slices;                // backward: work items per time step (tiles of a step are split into this many slices)
    int d_real, h_real;        // the net's real input / hidden widths (the instance is zero padded above them)
    int time_input;            // 1: input is [t, x] (time = column 0); 0: input is x
    int per_step;              // 1: N consecutive parameter sets, one per time step
};

__device__ __forceinline__ f32x4 relu4d(f32x4 v) {
    f32x4 o;
    o[0] = fmaxf(v[0], 0.f); o[1] = fmaxf(v[1], 0.f); o[2] = fmaxf(v[2], 0.f); o[3] = fmaxf(v[3], 0.f);
    return o;
}

template <int D, int H>
struct DGeo {
    static constexpr int DB = cdiv(D, 16), HB = cdiv(H, 16), KP = 4 * DB;
    static_assert(H % 16 == 0 && D % 16 == 0, "DenseNet-control instances are whole blocks (real sizes are runtime values)");
    // one weight set, k-step-major [ks][blocks][64]:  [W1 | W2x] over x (2 HB blocks), W2h over h1, W3x over x, [W3h1 ; W3h2] over h
    static constexpr int tW12 = 0, tW2h = tW12 + KP * 2 * HB * 64, tW3x = tW2h + 4 * HB * HB * 64,
                         tW3h1 = tW3x + KP * DB * 64, tW3h2 = tW3h1 + 4 * HB * DB * 64,
                         // d <= 128: every product over the X image in ONE pass -- [W1 | W2x | W3x | dt A] as NXB output blocks
                         // per k-step (the drift matrix is copied into every step's set: 45 KB a step at d = 112)
                         MERGE = (D <= 128) ? 1 : 0, NXB = 2 * HB + 2 * DB, tXall = tW3h2 + 4 * HB * DB * 64,
                         set_floats = tXall + MERGE * KP * NXB * 64;
    // split-product forward (hjbd_fwd_kernel<.., X3>): the same tables as S-step-major hi / lo f16 images (table_fill_x3), 512 floats
    // per (32-feature step, 16-row block); KS8 / KH8 steps over the state / the hidden units
    static constexpr int KS8 = cdiv(DB, 2), KH8 = cdiv(HB, 2);
    static constexpr int xW12 = 0, xW2h = xW12 + KS8 * 2 * HB * 512, xW3x = xW2h + KH8 * HB * 512, xW3h1 = xW3x + KS8 * DB * 512,
                         xW3h2 = xW3h1 + KH8 * DB * 512, xXall = xW3h2 + KH8 * DB * 512,
                         set_floats_x3 = xXall + MERGE * KS8 * NXB * 512;
    static constexpr int oA_x = 0, oB_x = oA_x + KS8 * DB * 512, oSets_x = oB_x + KS8 * DB * 512;
    static constexpr int IMGX = KS8 * 512;
    // per-step bias vectors (time column folded in), T-layout order [block][q][r]
    static constexpr int v1 = 0, v2 = v1 + HB * 16, v3 = v2 + HB * 16, vec_floats = v3 + DB * 16;
    // region: [dt A][B][sets ...][vectors of step 0 .. N-1]
    static constexpr int oA = 0, oB = oA + KP * DB * 64, oSets = oB + KP * DB * 64;
    static __host__ __device__ long long table_floats(int N, int per_step) {
        return (long long)oSets + (long long)(per_step ? N : 1) * set_floats + (long long)N * vec_floats;
    }
    // LDS (floats): problem vectors, reduction scratch, two images per wave (X_n and the increment panel v)
    static constexpr int vdr = 0, vrun = vdr + DB * 16, vterm = vrun + DB * 16, fRed = vterm + DB * 16,
                         fImg = fRed + 64, IMG = KP * 64, lds_floats = fImg + 4 * 2 * IMG, lds_floats_x3 = fImg + 4 * 2 * IMGX;
    // (Round 3 tried the wide family's shared table stream -- gemm_img_x3s: each 8 KiB chunk of a step's table fetched once per
    //  workgroup -- for the two long products of the split forward at d <= 128, with the increment image aliased onto the X image
    //  so that two workgroups and their stages still fit a CU: correct on all 44 dense-control tests, but the stage's operand
    //  ring spills 45 instead of 22 dwords under the 256-register cap and the forward went 3.11 -> 3.31 ms.  Not kept.)
    // image block of one (step, 16-trajectory tile) for the backward kernel: X_n, relu(z1), relu(z2), xi image
    static constexpr int pX = 0, pR1 = pX + KP * 64, pR2 = pR1 + 4 * HB * 64, pXi = pR2 + 4 * HB * 64, PBI = pXi + KP * 64;
    // padded gradient layout of one work item (instance sizes, no time rows): W1 (D x H), b1, W2 ((D+H) x H), b2, W3 ((D+2H) x D), b3
    static constexpr int gW1 = 0, gb1 = gW1 + D * H, gW2 = gb1 + H, gb2 = gW2 + (D + H) * H, gW3 = gb2 + H,
                         gb3 = gW3 + (D + 2 * H) * D, PP = gb3 + D;
    // backward LDS (floats): three A-operand tables of the step's net, then the double-buffered dz2 / dz1 exchange
    static constexpr int bW3h2 = 0, bW3h1 = bW3h2 + HB * KP * 64, bW2h = bW3h1 + HB * KP * 64, bEx = bW2h + HB * 4 * HB * 64,
                         EXT = 2 * HB, bwd_lds_floats = bEx + 2 * 4 * EXT * 256;
    // real flat parameter layout of one net (registration order W1,b1,W2,b2,W3,b3; weights (in, out))
    static __host__ __device__ long long n_params(int d, int h, int time_input) {
        const long long di = d + (time_input ? 1 : 0);
        return di * h + h + (di + h) * h + h + (di + 2 * h) * d + d;
    }
};

// adjoint = 1: the TRANSPOSED orientations the adjoint sweep (hjbd_adj_kernel) contracts with, in the same region and with
// the same offsets / shapes as the forward set:  tW12 <- [W3h2^T | W3h1^T] (rows: hidden units, over the d outputs),
// tW2h <- W2h^T, tW3x <- W3x^T, tW3h1 <- W2x^T, tW3h2 <- W1^T (rows: state components, over the hidden units), oA <- (dt A)^T,
// oB <- B^T.  (The forward tables are dead once the rollout has finished; the backward kernel stages its own from the parameters.)
template <int D, int H>
__global__ __launch_bounds__(256) void hjbd_tables_kernel(const DnetArgs a, int adjoint) {
    PSP_COND_EXIT(a.h);
    using W = DGeo<D, H>;
    const HjbArgs& h = a.h;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gs = (long long)gridDim.x * blockDim.x;
    const int d = a.d_real, hh = a.h_real, to = a.time_input ? 1 : 0, di = d + to;
    const long long P = W::n_params(d, hh, a.time_input);
    const long long oW1 = 0, ob1 = (long long)di * hh, oW2 = ob1 + hh, ob2 = oW2 + (long long)(di + hh) * hh,
                    oW3 = ob2 + hh, ob3 = oW3 + (long long)(di + 2 * hh) * d;
    float* T = a.tbl;
    if (adjoint == 2) {                                // split-product forward tables (same sources as the fp32 forward set below)
        auto fill = [&](float* dstf, int MB, int NS, auto src) { table_fill_x3(dstf, MB, NS, gtid, gs, src); };
        if (h.drift_kind == DRIFT_DENSE) {
            const float dt = h.dt;
            const float* __restrict__ A = h.drift;
            fill(T + W::oA_x, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? dt * A[row * D + col] : 0.f; });
        }
        if (h.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = h.sigma;
            fill(T + W::oB_x, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? B[row * D + col] : 0.f; });
        }
        const int nsets = a.per_step ? h.N : 1;
        for (int s = 0; s < nsets; ++s) {
            const float* __restrict__ Pp = h.params + (long long)s * P;
            float* Ts = T + W::oSets_x + (long long)s * W::set_floats_x3;
            fill(Ts + W::xW12, 2 * W::HB, W::KS8, [&](int row, int col) {      // rows: [W1 outputs | W2 outputs]
                const int o = row < 16 * W::HB ? row : row - 16 * W::HB;
                if (o >= hh || col >= d) return 0.f;
                return row < 16 * W::HB ? Pp[oW1 + (long long)(to + col) * hh + o] : Pp[oW2 + (long long)(to + col) * hh + o]; });
            fill(Ts + W::xW2h, W::HB, W::KH8, [&](int row, int col) {
                return (row < hh && col < hh) ? Pp[oW2 + (long long)(di + col) * hh + row] : 0.f; });
            fill(Ts + W::xW3x, W::DB, W::KS8, [&](int row, int col) {
                return (row < d && col < d) ? Pp[oW3 + (long long)(to + col) * d + row] : 0.f; });
            fill(Ts + W::xW3h1, W::DB, W::KH8, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW3 + (long long)(di + col) * d + row] : 0.f; });
            fill(Ts + W::xW3h2, W::DB, W::KH8, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW3 + (long long)(di + hh + col) * d + row] : 0.f; });
            if (W::MERGE) {
                const bool denseA = h.drift_kind == DRIFT_DENSE;
                const float dt = h.dt;
                const float* __restrict__ A = h.drift;
                fill(Ts + W::xXall, W::NXB, W::KS8, [&](int row, int col) {
                    const int blk = row >> 4, o = row & 15;
                    if (blk < W::HB) { const int u = 16 * blk + o; return (u < hh && col < d) ? Pp[oW1 + (long long)(to + col) * hh + u] : 0.f; }
                    if (blk < 2 * W::HB) { const int u = 16 * (blk - W::HB) + o; return (u < hh && col < d) ? Pp[oW2 + (long long)(to + col) * hh + u] : 0.f; }
                    if (blk < 2 * W::HB + W::DB) { const int f = 16 * (blk - 2 * W::HB) + o; return (f < d && col < d) ? Pp[oW3 + (long long)(to + col) * d + f] : 0.f; }
                    const int f = 16 * (blk - 2 * W::HB - W::DB) + o;
                    return (denseA && f < D && col < D) ? dt * A[f * D + col] : 0.f; });
            }
        }
        // per-step bias vectors behind the sets, as in the fp32 layout
        float* V = T + W::oSets_x + (long long)nsets * W::set_floats_x3;
        const long long nv = (long long)h.N * W::vec_floats;
        for (long long idx = gtid; idx < nv; idx += gs) {
            const int n = (int)(idx / W::vec_floats), e = (int)(idx % W::vec_floats);
            const float* __restrict__ Pp = h.params + (long long)(a.per_step ? n : 0) * P;
            const float tn = h.tfeat ? h.tfeat[n] : (float)n * h.dt;
            const int which = e < W::v2 ? 0 : (e < W::v3 ? 1 : 2);
            const int loc = e - (which == 0 ? W::v1 : (which == 1 ? W::v2 : W::v3));
            const int f = 16 * (loc >> 4) + 4 * (loc & 3) + ((loc >> 2) & 3);
            float v = 0.f;
            if (which == 0 && f < hh) v = Pp[ob1 + f] + (to ? tn * Pp[oW1 + f] : 0.f);
            if (which == 1 && f < hh) v = Pp[ob2 + f] + (to ? tn * Pp[oW2 + f] : 0.f);
            if (which == 2 && f < d) v = Pp[ob3 + f] + (to ? tn * Pp[oW3 + f] : 0.f);
            V[idx] = v;
        }
        return;
    }
    if (adjoint == 3) {                                // adjoint sweep, split-product tables (the orientations of adjoint == 1 below)
        auto fill = [&](float* dstf, int MB, int NS, auto src) { table_fill_x3(dstf, MB, NS, gtid, gs, src); };
        if (h.drift_kind == DRIFT_DENSE) {
            const float dt = h.dt;
            const float* __restrict__ A = h.drift;
            fill(T + W::oA_x, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? dt * A[col * D + row] : 0.f; });
        }
        if (h.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = h.sigma;
            fill(T + W::oB_x, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? B[col * D + row] : 0.f; });
        }
        const int nsets = a.per_step ? h.N : 1;
        for (int s = 0; s < nsets; ++s) {
            const float* __restrict__ Pp = h.params + (long long)s * P;
            float* Ts = T + W::oSets_x + (long long)s * W::set_floats_x3;
            fill(Ts + W::xW12, 2 * W::HB, W::KS8, [&](int row, int col) {     // rows: [h2 units | h1 units]
                const int u = row < 16 * W::HB ? row : row - 16 * W::HB;
                if (u >= hh || col >= d) return 0.f;
                return row < 16 * W::HB ? Pp[oW3 + (long long)(di + hh + u) * d + col] : Pp[oW3 + (long long)(di + u) * d + col]; });
            fill(Ts + W::xW2h, W::HB, W::KH8, [&](int row, int col) {
                return (row < hh && col < hh) ? Pp[oW2 + (long long)(di + row) * hh + col] : 0.f; });
            fill(Ts + W::xW3x, W::DB, W::KS8, [&](int row, int col) {
                return (row < d && col < d) ? Pp[oW3 + (long long)(to + row) * d + col] : 0.f; });
            fill(Ts + W::xW3h1, W::DB, W::KH8, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW2 + (long long)(to + row) * hh + col] : 0.f; });
            fill(Ts + W::xW3h2, W::DB, W::KH8, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW1 + (long long)(to + row) * hh + col] : 0.f; });
        }
        return;
    }
    if (adjoint) {
        if (h.drift_kind == DRIFT_DENSE) {
            const float dt = h.dt;
            const float* __restrict__ A = h.drift;
            table_fill(T + W::oA, W::DB, W::KP, gtid, gs, [&](int row, int col) {
                return (row < D && col < D) ? dt * A[col * D + row] : 0.f; });
        }
        if (h.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = h.sigma;
            table_fill(T + W::oB, W::DB, W::KP, gtid, gs, [&](int row, int col) {
                return (row < D && col < D) ? B[col * D + row] : 0.f; });
        }
        const int nsets = a.per_step ? h.N : 1;
        for (int s = 0; s < nsets; ++s) {
            const float* __restrict__ Pp = h.params + (long long)s * P;
            float* Ts = T + W::oSets + (long long)s * W::set_floats;
            table_fill(Ts + W::tW12, 2 * W::HB, W::KP, gtid, gs, [&](int row, int col) {     // rows: [h2 units | h1 units]
                const int u = row < 16 * W::HB ? row : row - 16 * W::HB;
                if (u >= hh || col >= d) return 0.f;
                return row < 16 * W::HB ? Pp[oW3 + (long long)(di + hh + u) * d + col] : Pp[oW3 + (long long)(di + u) * d + col]; });
            table_fill(Ts + W::tW2h, W::HB, 4 * W::HB, gtid, gs, [&](int row, int col) {
                return (row < hh && col < hh) ? Pp[oW2 + (long long)(di + row) * hh + col] : 0.f; });
            table_fill(Ts + W::tW3x, W::DB, W::KP, gtid, gs, [&](int row, int col) {
                return (row < d && col < d) ? Pp[oW3 + (long long)(to + row) * d + col] : 0.f; });
            table_fill(Ts + W::tW3h1, W::DB, 4 * W::HB, gtid, gs, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW2 + (long long)(to + row) * hh + col] : 0.f; });
            table_fill(Ts + W::tW3h2, W::DB, 4 * W::HB, gtid, gs, [&](int row, int col) {
                return (row < d && col < hh) ? Pp[oW1 + (long long)(to + row) * hh + col] : 0.f; });
        }
        return;
    }
    if (h.drift_kind == DRIFT_DENSE) {
        const float dt = h.dt;
        const float* __restrict__ A = h.drift;
        table_fill(T + W::oA, W::DB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < D && col < D) ? dt * A[row * D + col] : 0.f; });
    }
    if (h.sigma_kind == SIGMA_DENSE) {
        const float* __restrict__ B = h.sigma;
        table_fill(T + W::oB, W::DB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < D && col < D) ? B[row * D + col] : 0.f; });
    }
    const int nsets = a.per_step ? h.N : 1;
    for (int s = 0; s < nsets; ++s) {
        const float* __restrict__ Pp = h.params + (long long)s * P;
        float* Ts = T + W::oSets + (long long)s * W::set_floats;
        table_fill(Ts + W::tW12, 2 * W::HB, W::KP, gtid, gs, [&](int row, int col) {      // rows: [W1 outputs | W2 outputs]
            const int o = row < 16 * W::HB ? row : row - 16 * W::HB;
            if (o >= hh || col >= d) return 0.f;
            return row < 16 * W::HB ? Pp[oW1 + (long long)(to + col) * hh + o] : Pp[oW2 + (long long)(to + col) * hh + o]; });
        table_fill(Ts + W::tW2h, W::HB, 4 * W::HB, gtid, gs, [&](int row, int col) {
            return (row < hh && col < hh) ? Pp[oW2 + (long long)(di + col) * hh + row] : 0.f; });
        table_fill(Ts + W::tW3x, W::DB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < d && col < d) ? Pp[oW3 + (long long)(to + col) * d + row] : 0.f; });
        table_fill(Ts + W::tW3h1, W::DB, 4 * W::HB, gtid, gs, [&](int row, int col) {
            return (row < d && col < hh) ? Pp[oW3 + (long long)(di + col) * d + row] : 0.f; });
        table_fill(Ts + W::tW3h2, W::DB, 4 * W::HB, gtid, gs, [&](int row, int col) {
            return (row < d && col < hh) ? Pp[oW3 + (long long)(di + hh + col) * d + row] : 0.f; });
        if (W::MERGE) {
            const bool denseA = h.drift_kind == DRIFT_DENSE;
            const float dt = h.dt;
            const float* __restrict__ A = h.drift;
            table_fill(Ts + W::tXall, W::NXB, W::KP, gtid, gs, [&](int row, int col) {
                const int blk = row >> 4, o = row & 15;
                if (blk < W::HB) { const int u = 16 * blk + o; return (u < hh && col < d) ? Pp[oW1 + (long long)(to + col) * hh + u] : 0.f; }
                if (blk < 2 * W::HB) { const int u = 16 * (blk - W::HB) + o; return (u < hh && col < d) ? Pp[oW2 + (long long)(to + col) * hh + u] : 0.f; }
                if (blk < 2 * W::HB + W::DB) { const int f = 16 * (blk - 2 * W::HB) + o; return (f < d && col < d) ? Pp[oW3 + (long long)(to + col) * d + f] : 0.f; }
                const int f = 16 * (blk - 2 * W::HB - W::DB) + o;
                return (denseA && f < D && col < D) ? dt * A[f * D + col] : 0.f; });
        }
    }
    // per-step bias vectors b + t_n W[0, :]  (time input = column 0 of every layer's input block; solver.py:355)
    float* V = T + W::oSets + (long long)nsets * W::set_floats;
    const long long nv = (long long)h.N * W::vec_floats;
    for (long long idx = gtid; idx < nv; idx += gs) {
        const int n = (int)(idx / W::vec_floats), e = (int)(idx % W::vec_floats);
        const float* __restrict__ Pp = h.params + (long long)(a.per_step ? n : 0) * P;
        const float tn = h.tfeat ? h.tfeat[n] : (float)n * h.dt;
        const int which = e < W::v2 ? 0 : (e < W::v3 ? 1 : 2);
        const int loc = e - (which == 0 ? W::v1 : (which == 1 ? W::v2 : W::v3));
        const int f = 16 * (loc >> 4) + 4 * (loc & 3) + ((loc >> 2) & 3);       // [block][q][r] -> feature 16 b + 4 r + q
        float v = 0.f;
        if (which == 0 && f < hh) v = Pp[ob1 + f] + (to ? tn * Pp[oW1 + f] : 0.f);
        if (which == 1 && f < hh) v = Pp[ob2 + f] + (to ? tn * Pp[oW2 + f] : 0.f);
        if (which == 2 && f < d) v = Pp[ob3 + f] + (to ? tn * Pp[oW3 + f] : 0.f);
        V[idx] = v;
    }
}

// X3 (psp_hjb_config.mlp_dtype = PSP_MLP_F16X3): every product as split f16 products (gemm_img_x3 / gemm_regs_x3, hjbw_kernels.h) on
// the tables of hjbd_tables_kernel(.., 2); the two images of a wave hold hi / lo packs
// SPEC (round 4): dense drift, dense sigma, adaptive process, no running cost, Philox noise, not the relative-entropy loss as
// compile-time constants -- the LLGC training launch of time_approx='outer' (hjb_kernels.h, hjb_fwd_kernel FAST_ = 2)
template <int D, int H, bool X3 = false, bool SPEC = false>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void hjbd_fwd_kernel(const DnetArgs da) {   // d <= 128: two workgroups per CU
    PSP_COND_EXIT(da.h);
    using W = DGeo<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP;
    constexpr int oSets = X3 ? W::oSets_x : W::oSets, SETF = X3 ? W::set_floats_x3 : W::set_floats, IMGF = X3 ? W::IMGX : W::IMG;
    [[maybe_unused]] const f32x4 zero4x = {0.f, 0.f, 0.f, 0.f};
    const HjbArgs& a = da.h;
    const int k_drift = SPEC ? (int)DRIFT_DENSE : a.drift_kind, k_sigma = SPEC ? (int)SIGMA_DENSE : a.sigma_kind;
    const int k_run = SPEC ? (int)RUN_ZERO : a.runcost_kind, k_loss = SPEC ? (int)LOSS_LOGVAR : a.loss_kind;
    const int k_noise = SPEC ? (int)NOISE_PHILOX : a.noise_mode;
    const bool k_adaptive = SPEC ? true : (a.adaptive != 0);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ T = da.tbl;
    const int dr = da.d_real;

    stage_vec(lds + W::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (k_drift == DRIFT_DIAG || k_drift == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + W::vrun, DB, tid, nthr, [&](int f) {
        return (f < D && k_run == RUN_DIAGQ) ? a.runcost[f] : 0.f; });
    stage_vec(lds + W::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });
    __syncthreads();

    const int t16raw = blockIdx.x * nwave + wave;
    const bool wave_valid = t16raw < a.ntile16;       // surplus waves of the last workgroup run along on the last tile
    const int t16 = wave_valid ? t16raw : a.ntile16 - 1;
    const int k = t16 * 16 + j;
    const bool kvalid = wave_valid && k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    const float dt = a.dt, sqdt = a.sqdt;
    float* imgX = lds + W::fImg + wave * 2 * IMGF;    // this wave's image of X_n   [KP][64] (X3: hi / lo packs)
    float* imgV = imgX + IMGF;                        // ... and of the increment panel v (dense sigma)
    [[maybe_unused]] f16x8* imgX8 = reinterpret_cast<f16x8*>(imgX) + lane;
    [[maybe_unused]] f16x8* imgV8 = reinterpret_cast<f16x8*>(imgV) + lane;
    const bool store = a.store_path && kvalid;
    // image in the xi slot: c_xi xi + c_z Z.  store_path 1: xi, or xi + sqrt(dt) Z for a non-adaptive process; 2: xi - sqrt(dt) Z
    // and 3: Z for the adjoint sweep (attached forward process / relative entropy), as in hjb_fwd_kernel
    const float store_cxi = (a.store_path == 3) ? 0.f : 1.f;
    const float store_cz = (a.store_path == 3) ? 1.f : (a.store_path == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));
    const int nsets_m1 = da.per_step ? a.N - 1 : 0;
    const float* Vbase = T + oSets + (long long)(nsets_m1 + 1) * SETF;

    double sD = 0.0, sD2 = 0.0;
    {
        const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;
        const f32x4* vterm = vecs0 + W::vterm / 4;
        f32x4 X[DB];                                   // X_0 (solver.py:365-367) in T layout
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * b + 4 * r + q;
                const float v = a.x0[(size_t)(kvalid ? k : 0) * a.x0_stride + (f < D ? f : D - 1)];
                X[b][r] = (f < D && kvalid) ? v : 0.f;
            }
        float Y = a.y0 ? a.y0[0] : 0.f;
        float Fsum = 0.f;

#pragma unroll 1
        for (int n = 0; n < a.N; ++n) {
            const f32x4* vecs = opaque(vecs0);
            const int qn = opaque_i(q);
            const f32x4* vdr = vecs + W::vdr / 4;
            const f32x4* vrun = vecs + W::vrun / 4;
            const float* Ts = T + oSets + (long long)(da.per_step ? n : 0) * SETF;     // this step's weight set
            const f32x4* Vn = reinterpret_cast<const f32x4*>(Vbase + (long long)n * W::vec_floats) + qn;
            const size_t row = ((size_t)n * a.K_local + (kvalid ? k : 0));
            // ---- X_n: LDS image (B operand of every product over x) and the row-major store for the backward pass
            if constexpr (X3) {
#pragma unroll
                for (int S = 0; S < W::KS8; ++S) {
                    f16x8 ph, pl;
                    split_pack(X[2 * S], (2 * S + 1 < DB) ? X[(2 * S + 1 < DB) ? 2 * S + 1 : 0] : zero4x, ph, pl);
                    imgX8[(2 * S) * 64] = ph; imgX8[(2 * S + 1) * 64] = pl;
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) imgX[ks * 64 + lane] = X[ks >> 2][ks & 3];
            }
            float* iblk = da.pimg ? da.pimg + ((size_t)n * a.ntile16 + t16) * (size_t)W::PBI + lane : nullptr;
            const bool store_img = a.store_path && wave_valid && da.pimg != nullptr;
            if (store_img) {
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) iblk[W::pX + ks * 64] = X[ks >> 2][ks & 3];
            } else if (store) {
                float* px = da.px + row * dr;
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * b + 4 * r + q;
                        if (f < dr) px[f] = X[b][r];
                    }
            }
            // ---- z1 and the x-part of z2 in one pass over the image (function_space.py:133-140); for d <= 128 the x-part of Z
            //      and the drift product ride in the same pass (one operand ring instead of four short ones)
            f32x4 z12[2 * HB];
            f32x4 Zx[W::MERGE ? DB : 1];
            f32x4 Xn[DB];
            if constexpr (W::MERGE) {
                f32x4 xa[W::NXB];
#pragma unroll
                for (int m = 0; m < HB; ++m) { xa[m] = Vn[(W::v1 / 16 + m) * 4]; xa[HB + m] = Vn[(W::v2 / 16 + m) * 4]; }
#pragma unroll
                for (int b = 0; b < DB; ++b) { xa[2 * HB + b] = Vn[(W::v3 / 16 + b) * 4]; xa[2 * HB + DB + b] = X[b]; }
                if constexpr (X3) {
                    // (output blocks in groups of at most 8: a correction chain of 24 registers under the 256-register cap)
                    if (k_drift == DRIFT_DENSE) gemm_img_x3<W::NXB, W::KS8, W::NXB, 8>(xa, Ts + W::xXall, imgX, lane);
                    else gemm_img_x3<2 * HB + DB, W::KS8, W::NXB, 8>(reinterpret_cast<f32x4 (&)[2 * HB + DB]>(xa), Ts + W::xXall, imgX, lane);
                } else {
                    if (k_drift == DRIFT_DENSE) gemm_img<W::NXB, KP, W::NXB>(xa, Ts + W::tXall, imgX, lane);
                    else gemm_img<2 * HB + DB, KP, W::NXB>(reinterpret_cast<f32x4 (&)[2 * HB + DB]>(xa), Ts + W::tXall, imgX, lane);
                }
#pragma unroll
                for (int m = 0; m < 2 * HB; ++m) z12[m] = xa[m];
#pragma unroll
                for (int b = 0; b < DB; ++b) { Zx[b] = xa[2 * HB + b]; Xn[b] = xa[2 * HB + DB + b]; }
            } else {
#pragma unroll
                for (int m = 0; m < HB; ++m) { z12[m] = Vn[(W::v1 / 16 + m) * 4]; z12[HB + m] = Vn[(W::v2 / 16 + m) * 4]; }
                if constexpr (X3) gemm_img_x3<2 * HB, W::KS8>(z12, Ts + W::xW12, imgX, lane);
                else gemm_img<2 * HB, KP>(z12, Ts + W::tW12, imgX, lane);
#pragma unroll
                for (int b = 0; b < DB; ++b) Xn[b] = X[b];
                if (k_drift == DRIFT_DENSE) {
                    if constexpr (X3) gemm_img_x3<DB, W::KS8>(Xn, T + W::oA_x, imgX, lane);
                    else gemm_img<DB, KP>(Xn, T + W::oA, imgX, lane);
                }
            }
            // ---- drift part of X_{n+1} (solver.py:471): the dense product is done above
            if (k_drift == DRIFT_DENSE) {
            } else if (k_drift == DRIFT_DIAG) {
#pragma unroll
                for (int b = 0; b < DB; ++b) Xn[b] += dt * (vdr[b * 4] * X[b]);
            } else if (k_drift == DRIFT_DWELL) {
#pragma unroll
                for (int b = 0; b < DB; ++b) Xn[b] -= dt * (4.0f * vdr[b * 4] * (X[b] * (X[b] * X[b] - 1.0f)));
            }
            f32x4 h1[HB], h2[HB];
            const bool store_r = store && da.pr1 != nullptr && !store_img;
            const int hr = da.h_real;
            auto put_r = [&](float* base, int m, const f32x4& r) __attribute__((always_inline)) {
                float* pr = base + row * hr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 16 * m + 4 * e + q;
                    if (f < hr) pr[f] = r[e];
                }
            };
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                const f32x4 r = X3 ? relu4n(z12[m]) : relu4d(z12[m]);       // (X3: NaN-propagating, gen_kernels.h)
                h1[m] = r * r;
                if (store_r) put_r(da.pr1, m, r);
                if (store_img) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) iblk[W::pR1 + (4 * m + e) * 64] = r[e];
                }
            }
            {
                f32x4 z2[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) z2[m] = z12[HB + m];
                if constexpr (X3) gemm_regs_x3<HB, HB>(z2, Ts + W::xW2h, h1, lane);
                else gemm_regs<HB, 4 * HB, HB>(z2, Ts + W::tW2h, h1, lane);
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    const f32x4 r = X3 ? relu4n(z2[m]) : relu4d(z2[m]);
                    h2[m] = r * r;
                    if (store_r) put_r(da.pr2, m, r);
                    if (store_img) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) iblk[W::pR2 + (4 * m + e) * 64] = r[e];
                    }
                }
            }
            // ---- control output four state blocks at a time: Z_g = W3x[g] x + W3h1[g] h1 + W3h2[g] h2 + b3
            float S = 0.f, Pz = 0.f;
            auto z_group = [&](auto nbc, int g) __attribute__((always_inline)) {
                constexpr int NB = decltype(nbc)::value;
                f32x4 Zg[NB];
                if constexpr (W::MERGE) {
#pragma unroll
                    for (int m = 0; m < NB; ++m) Zg[m] = Zx[4 * g + m];
                } else {
#pragma unroll
                    for (int m = 0; m < NB; ++m) Zg[m] = Vn[(W::v3 / 16 + 4 * g + m) * 4];
                    if constexpr (X3) gemm_img_x3<NB, W::KS8, DB>(Zg, Ts + W::xW3x + 4 * g * 512, imgX, lane);
                    else gemm_img<NB, KP, DB>(Zg, Ts + W::tW3x + 4 * g * 64, imgX, lane);
                }
                if constexpr (X3) {
                    gemm_regs_x3<NB, HB, DB>(Zg, Ts + W::xW3h1 + 4 * g * 512, h1, lane);
                    gemm_regs_x3<NB, HB, DB>(Zg, Ts + W::xW3h2 + 4 * g * 512, h2, lane);
                } else {
                    gemm_regs<NB, 4 * HB, HB, DB>(Zg, Ts + W::tW3h1 + 4 * g * 64, h1, lane);
                    gemm_regs<NB, 4 * HB, HB, DB>(Zg, Ts + W::tW3h2 + 4 * g * 64, h2, lane);
                }
                [[maybe_unused]] f32x4 vg[4] = {zero4x, zero4x, zero4x, zero4x};
#pragma unroll
                for (int m = 0; m < NB; ++m) {
                    const int b = 4 * g + m;
                    f32x4 xi;
                    if (k_noise == NOISE_PHILOX) {
                        xi = philox_block(kglob, (uint32_t)n, (uint32_t)(4 * b + qn), a.iter, a.seed_lo, a.seed_hi);
                    } else {
                        const float* xrow = a.xi + ((size_t)(n + 1) * a.K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * b + 4 * r + q;
                            xi[r] = xrow[f < D ? f : D - 1];
                        }
                    }
                    // padding carries no noise.  (Only blocks that CONTAIN padding are masked, behind a wave-uniform test: the 4 DB
                    // per-lane masks of the unconditional form were hoisted out of the time loop as 2 x 4 DB SGPRs, spilled to VGPR
                    // lanes, and the VGPRs that made room for them were reloaded from scratch in every step)
                    if (16 * b + 16 > dr) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (16 * b + 4 * r + q >= dr) xi[r] = 0.f;
                    }
                    if (!kvalid) xi = zero4x;
                    if (store_img) {
                        const f32x4 wv = store_cxi * xi + store_cz * Zg[m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) iblk[W::pXi + (4 * b + r) * 64] = wv[r];
                    } else if (store) {
                        float* pxi = da.pxi + row * dr;
                        const f32x4 wv = store_cxi * xi + store_cz * Zg[m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * b + 4 * r + q;
                            if (f < dr) pxi[f] = wv[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S = fmaf(Zg[m][r], Zg[m][r], S);
                        Pz = fmaf(Zg[m][r], xi[r], Pz);
                    }
                    const f32x4 v = k_adaptive ? (sqdt * xi - dt * Zg[m]) : (sqdt * xi);
                    if (k_sigma == SIGMA_DENSE) {
                        if constexpr (X3) vg[m] = v;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) imgV[(4 * b + r) * 64 + lane] = v[r];
                        }
                    } else if (k_sigma == SIGMA_SCALE) {
                        Xn[b] += a.sigma_scale * v;
                    } else {
                        Xn[b] += v;
                    }
                }
                if constexpr (X3) {                    // increment panel of this group as hi / lo packs (two S-steps per group)
                    if (k_sigma == SIGMA_DENSE) {
#pragma unroll
                        for (int s2 = 0; s2 < (NB + 1) / 2; ++s2) {
                            f16x8 ph, pl;
                            split_pack(vg[2 * s2], vg[2 * s2 + 1], ph, pl);
                            imgV8[(2 * (2 * g + s2)) * 64] = ph; imgV8[(2 * (2 * g + s2) + 1) * 64] = pl;
                        }
                    }
                }
            };
#pragma unroll
            for (int g = 0; g < DB / 4; ++g) z_group(std::integral_constant<int, 4>{}, g);
            if constexpr (DB % 4 != 0) z_group(std::integral_constant<int, DB % 4>{}, DB / 4);
            S = qsum(S);
            Pz = qsum(Pz);
            if (k_sigma == SIGMA_DENSE) {                                                // X += B v
                if constexpr (X3) gemm_img_x3<DB, W::KS8>(Xn, T + W::oB_x, imgV, lane);
                else gemm_img<DB, KP>(Xn, T + W::oB, imgV, lane);
            }
#pragma unroll
            for (int b = 0; b < DB; ++b) X[b] = Xn[b];
            // ---- running cost f(X_{n+1}) and Y update (solver.py:477-478)
            float fX = 0.f;
            if (k_run == RUN_DIAGQ) {
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const f32x4 pv = vrun[b * 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) fX = fmaf(pv[r] * X[b][r], X[b][r], fX);
                }
                fX = qsum(fX);
            }
            if (k_loss == LOSS_RELENT) {
                Y = Y - (0.5f * S + fX) * dt;           // Y carries -Zsum (hjb_fwd_kernel): D = -(Zsum + g), loss = -mean D
            } else {
                const float drift_y = k_adaptive ? (fX - 0.5f * S) : (fX + 0.5f * S);
                Y = Y + drift_y * dt + Pz * sqdt;
            }
            Fsum = fmaf(fX, dt, Fsum);
        }

        // ---- terminal cost g(X_N) and D = Y - g  (problems.py:49,164,334; solver.py:167-168)
        float g = 0.f;
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const f32x4 tv = vterm[b * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = X[b][r];
                if (a.term_kind == TERM_LINEAR) g = fmaf(tv[r], x, g);
                else if (a.term_kind == TERM_DIAGQ) g = fmaf(tv[r] * x, x, g);
                else g = fmaf(tv[r] * (x - 1.0f), (x - 1.0f), g);
            }
        }
        g = qsum(g);
        const float Dk = Y - g;
        if (kvalid && q == 0) a.D[k] = Dk;
        if (a.Fint && kvalid && q == 0) a.Fint[k] = Fsum;
        if (a.Yout && kvalid && q == 0) a.Yout[k] = Y;
        if (a.XN && kvalid) {
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * b + 4 * r + q;
                    if (f < D) a.XN[(size_t)k * D + f] = X[b][r];
                }
        }
        if (kvalid && q == 0) { sD = (double)Dk; sD2 = (double)Dk * (double)Dk; }
    }
    sD = jsum(sD); sD2 = jsum(sD2);
    double* red = reinterpret_cast<double*>(lds + W::fRed);
    if (lane == 0) { red[2 * wave] = sD; red[2 * wave + 1] = sD2; }
    __syncthreads();
    if (tid == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < nwave; ++w) { t0 += red[2 * w]; t1 += red[2 * w + 1]; }
        a.fwd_partial[2 * blockIdx.x] = t0;
        a.fwd_partial[2 * blockIdx.x + 1] = t1;
    }
}

// =======================================================================================
// Reverse-time adjoint sweep for a DenseNet control (gradients THROUGH the state path: adaptive_forward_process=True with
// detach_forward=False -- the reference's default flags, solver.py:451-469 -- and the relative-entropy loss, :179-180,
// 484-486).  Same recursion as hjb_adj_kernel / hjbw_adj_kernel (hjba_kernels.h has the derivation):
//     lambda'  = lambda_{n+1} + (mu + nu) dt grad f(X_{n+1})
//     gZ_n     = coefW W_n - dt B^T lambda'        (W_n = the image the forward left in the xi slot: store_path 2 or 3)
//     lambda_n = lambda' + dt b'(X_n)^T lambda' + J_n^T gZ_n
// with the Jacobian of the dense-concat net  Z = W3^T [x, h1, h2] + b3,  h = relu(z)^2:
//     dz2 = (W3h2^T gZ) 2 r2,   dz1 = (W3h1^T gZ + W2h^T dz2) 2 r1,   J^T gZ = W3x^T gZ + W2x^T dz2 + W1^T dz1
// from the relu images the forward stored.  gZ_n / sqrt(dt) overwrites the xi slot; hjbd_bwd_kernel then runs with unit
// weights.  One wave per 16-trajectory tile like the forward; the step's TRANSPOSED tables come from hjbd_tables_kernel
// (adjoint = 1) in the forward's table region.
// =======================================================================================
// X3: split f16 products on the transposed split sets of hjbd_tables_kernel(.., 3); images as hi / lo packs; trajectory weights scaled
// per wave by a power of two and the image written back scaled back (hjb_adj_kernel<.., X3>)
template <int D, int H, bool X3 = false>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void hjbd_adj_kernel(const DnetArgs da) {
    PSP_COND_EXIT(da.h);
    using W = DGeo<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP;
    const HjbArgs& a = da.h;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ T = da.tbl;

    stage_vec(lds + W::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (a.drift_kind == DRIFT_DIAG || a.drift_kind == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + W::vrun, DB, tid, nthr, [&](int f) {
        return (f < D && a.runcost_kind == RUN_DIAGQ) ? a.runcost[f] : 0.f; });
    stage_vec(lds + W::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });
    __syncthreads();

    const int t16 = blockIdx.x * nwave + wave;
    if (t16 >= a.ntile16) return;                      // no workgroup barriers below
    const int k = t16 * 16 + j;
    const bool kvalid = k < a.K_local;
    const float dt = a.dt, sqdt = a.sqdt;
    float mu = (kvalid && a.adj_mu) ? a.adj_mu[k] : 0.f;
    float nu = (kvalid && a.adj_nu) ? a.adj_nu[k] : 0.f;
    float wT_in = a.adj_wT ? (kvalid ? a.adj_wT[k] : 0.f) : (nu - mu);
    float ginv = 1.0f;
    if constexpr (X3) {
        float am = fmaxf(fmaxf(fabsf(mu), fabsf(nu)), fabsf(wT_in));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) am = fmaxf(am, __shfl_xor(am, o));
        const unsigned e = (__float_as_uint(am) >> 23) & 0xFFu;
        if (e >= 1u && e <= 253u) {
            const float gsc = __uint_as_float((254u - e) << 23);
            ginv = __uint_as_float(e << 23);
            mu *= gsc; nu *= gsc; wT_in *= gsc;
        }
    }
    const float rsq = ginv / a.sqdt;
    [[maybe_unused]] const f32x4 zero4x = {0.f, 0.f, 0.f, 0.f};
    constexpr int oSets = X3 ? W::oSets_x : W::oSets, SETF = X3 ? W::set_floats_x3 : W::set_floats;
    const float coefW = (a.store_path == 3) ? nu * dt : mu * sqdt;
    const float wf = (mu + nu) * dt;
    const float wT = wT_in;                                                     // weight of grad g(X_N) in lambda_N
    float* img = lds + W::fImg + wave * 2 * (X3 ? W::IMGX : W::IMG);
    [[maybe_unused]] f16x8* img8 = reinterpret_cast<f16x8*>(img) + lane;
    const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned ul = (unsigned)lane;
    typedef __attribute__((address_space(1))) float* gwptr_t;

    f32x4 lam[DB];                                     // lambda_N = wT grad g(X_N)
    {
        const f32x4* vterm = vecs0 + W::vterm / 4;
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const f32x4 tv = vterm[b * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * b + 4 * r + q;
                const float x = (f < D && kvalid) ? a.XN[(size_t)k * D + f] : 0.f;
                float gg;
                if (a.term_kind == TERM_LINEAR) gg = tv[r];
                else if (a.term_kind == TERM_DIAGQ) gg = 2.0f * tv[r] * x;
                else gg = 2.0f * tv[r] * (x - 1.0f);
                lam[b][r] = wT * gg;
            }
        }
    }

#pragma unroll 1
    for (int n = a.N - 1; n >= 0; --n) {
        const f32x4* vecs = opaque(vecs0);
        const f32x4* vdr = vecs + W::vdr / 4;
        const f32x4* vrun = vecs + W::vrun / 4;
        const float* Ts = T + oSets + (long long)(da.per_step ? n : 0) * SETF;     // this step's transposed set
        auto pbase = [&](int nn, int ofs) __attribute__((always_inline)) {
            return (gwptr_t)sgpr_block_addr(da.pimg, (unsigned long long)nn * a.ntile16 + t16, (unsigned)W::PBI, (unsigned)ofs);
        };
        // lambda' = lambda_{n+1} + (mu + nu) dt grad f(X_{n+1});  X_{n+1} from the next image block (or X_N)
        if (a.runcost_kind == RUN_DIAGQ) {
            const int nx = n + 1 < a.N ? n + 1 : n;
#pragma unroll
            for (int b = 0; b < DB; ++b) {
                gwptr_t px = pbase(nx, W::pX + b * 256);
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * b + 4 * r + q;
                    const float xp = px[r * 64 + ul];
                    const float xn = (f < D && kvalid) ? a.XN[(size_t)k * D + (f < D ? f : 0)] : 0.f;
                    x[r] = (n + 1 < a.N) ? xp : xn;
                }
                lam[b] += (2.0f * wf) * (vrun[b * 4] * x);
            }
        }
        if constexpr (X3) {
#pragma unroll
            for (int S = 0; S < W::KS8; ++S) {
                f16x8 ph, pl;
                split_pack(lam[2 * S], (2 * S + 1 < DB) ? lam[(2 * S + 1 < DB) ? 2 * S + 1 : 0] : zero4x, ph, pl);
                img8[(2 * S) * 64] = ph; img8[(2 * S + 1) * 64] = pl;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KP; ++ks) img[ks * 64 + lane] = lam[ks >> 2][ks & 3];
        }
        // q = B^T lambda'
        f32x4 qv[DB];
        if (a.sigma_kind == SIGMA_DENSE) {
#pragma unroll
            for (int b = 0; b < DB; ++b) qv[b] = zero4;
            if constexpr (X3) gemm_img_x3<DB, W::KS8>(qv, T + W::oB_x, img, lane);
            else gemm_img<DB, KP>(qv, T + W::oB, img, lane);
        } else if (a.sigma_kind == SIGMA_SCALE) {
#pragma unroll
            for (int b = 0; b < DB; ++b) qv[b] = a.sigma_scale * lam[b];
        } else {
#pragma unroll
            for (int b = 0; b < DB; ++b) qv[b] = lam[b];
        }
        // lambda += dt b'(X_n)^T lambda'   (in place; the image still holds lambda')
        if (a.drift_kind == DRIFT_DENSE) {
            if constexpr (X3) gemm_img_x3<DB, W::KS8>(lam, T + W::oA_x, img, lane);
            else gemm_img<DB, KP>(lam, T + W::oA, img, lane);
        } else if (a.drift_kind == DRIFT_DIAG) {
#pragma unroll
            for (int b = 0; b < DB; ++b) lam[b] += dt * (vdr[b * 4] * lam[b]);
        } else if (a.drift_kind == DRIFT_DWELL) {
#pragma unroll
            for (int b = 0; b < DB; ++b) {
                gwptr_t px = pbase(n, W::pX + b * 256);
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = px[r * 64 + ul];
                lam[b] -= dt * (4.0f * vdr[b * 4] * ((3.0f * x * x - 1.0f) * lam[b]));
            }
        }
        // gZ_n: back into the xi slot (as gZ / sqrt(dt)) and into the image (B operand of the transposed products)
        [[maybe_unused]] f32x4 gzp = zero4x;           // X3: gZ of the even block of a pair, until its odd partner is formed
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            gwptr_t pw = pbase(n, W::pXi + b * 256);
            f32x4 w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = pw[r * 64 + ul];
            const f32x4 gz = coefW * w - dt * qv[b];
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[r * 64 + ul] = rsq * gz[r];
            if constexpr (X3) {
                if ((b & 1) == 0 && b + 1 < DB) gzp = gz;
                else {
                    f16x8 ph, pl;
                    if (b & 1) split_pack(gzp, gz, ph, pl);
                    else split_pack(gz, zero4x, ph, pl);          // odd block count: the last step's upper half stays zero
                    img8[(2 * (b >> 1)) * 64] = ph; img8[(2 * (b >> 1) + 1) * 64] = pl;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) img[(4 * b + r) * 64 + lane] = gz[r];
            }
        }
        // adjoints of the hidden layers
        f32x4 u[2 * HB], dz2[HB], dz1[HB];
#pragma unroll
        for (int m = 0; m < 2 * HB; ++m) u[m] = zero4;
        if constexpr (X3) gemm_img_x3<2 * HB, W::KS8>(u, Ts + W::xW12, img, lane);
        else gemm_img<2 * HB, KP>(u, Ts + W::tW12, img, lane);            // [W3h2^T gZ | W3h1^T gZ]
#pragma unroll
        for (int m = 0; m < HB; ++m) {
            f32x4 r2;
#pragma unroll
            for (int e = 0; e < 4; ++e) r2[e] = pbase(n, W::pR2)[(4 * m + e) * 64 + ul];
            dz2[m] = u[m] * (2.0f * r2);
            dz1[m] = u[HB + m];
        }
        if constexpr (X3) gemm_regs_x3<HB, HB>(dz1, Ts + W::xW2h, dz2, lane);
        else gemm_regs<HB, 4 * HB, HB>(dz1, Ts + W::tW2h, dz2, lane);      // + W2h^T dz2
#pragma unroll
        for (int m = 0; m < HB; ++m) {
            f32x4 r1;
#pragma unroll
            for (int e = 0; e < 4; ++e) r1[e] = pbase(n, W::pR1)[(4 * m + e) * 64 + ul];
            dz1[m] = dz1[m] * (2.0f * r1);
        }
        // lambda_n += J_n^T gZ_n
        if constexpr (X3) {
            gemm_img_x3<DB, W::KS8>(lam, Ts + W::xW3x, img, lane);
            gemm_regs_x3<DB, HB>(lam, Ts + W::xW3h1, dz2, lane);
            gemm_regs_x3<DB, HB>(lam, Ts + W::xW3h2, dz1, lane);
        } else {
            gemm_img<DB, KP>(lam, Ts + W::tW3x, img, lane);               // W3x^T gZ
            gemm_regs<DB, 4 * HB, HB>(lam, Ts + W::tW3h1, dz2, lane);     // W2x^T dz2
            gemm_regs<DB, 4 * HB, HB>(lam, Ts + W::tW3h2, dz1, lane);     // W1^T dz1
        }
    }
}

// =======================================================================================
// Backward kernel of the DenseNet control: parameter gradient of sum_{n,k} G_n[k] . Z_n(X_n[k]),  G = w_k sqrt(dt) image
// (dL/dZ_n for a detached forward process), per time step when every step has its own net.
// One 4-wave workgroup per CU (one wave per SIMD: the 512-entry register file holds a wave's 30-odd accumulator tiles next
// to the column operands), rounds of 4 sample blocks, double-buffered LDS exchange, one barrier per round.  A work item =
// (step n, slice of its 16-trajectory tiles); its accumulators are flushed to partial[item] in the padded layout
// (reduced over the slices, and mapped to the real shapes, by the host).  Per round every wave
//   (1) T layout, own block:  G = w sqrt(dt) image;  dz2 = (W3h2 G) 2 r2;  dz1 = (W3h1 G + W2h dz2) 2 r1  (three register-
//       chained products from LDS tables of the step's net) -> dz2 / dz1 tiles in the exchange buffer;  barrier;
//   (2) feature-on-lane, all four blocks:  wave c owns the row items i = c (mod 4) of [x blocks | h1 blocks | h2 blocks]; row
//       operands are reads of the stored images (h = r^2 on the fly), column operands the G tiles (the xi image read the
//       same way, times w sqrt(dt)) and the exchanged dz2 / dz1 tiles:
//           x rows:  dW3x += x^T G,  dW2x += x^T dz2,  dW1 += x^T dz1      h1 rows:  dW3h1 += h1^T G,  dW2h += h1^T dz2
//           h2 rows: dW3h2 += h2^T G;   bias sums from the column tiles (wave 0).
// =======================================================================================
// PASS 0: everything in one launch.  Wider instances split the COLUMNS over two launches so that a wave's accumulators fit
// its registers: PASS 1 = the G columns (dW3 and db3; needs no adjoint panels), PASS 2 = the dz2 / dz1 columns (dW2, dW1,
// db2, db1; x and h1 rows only).  Both read the same images and write disjoint parts of partial[item].
// X3 (round 3; psp_hjb_config.mlp_dtype = PSP_MLP_F16X3 on a detached adaptive run, i.e. the stored image is the Brownian increment):
// the weight-gradient outer products of phase (2) -- three quarters of the kernel's fp32 MFMAs -- contract PAIRS of sample blocks
// on v_mfma_f32_16x16x32_f16 (k = 8 g + e: e < 4 block 2 p, e >= 4 block 2 p + 1, sample 4 g + (e & 3)): three instructions per
// (row item, column tile, pair) instead of eight fp32 ones.  Operands are split as x = hi + lo with the UNSCALED residual
// lo = f16(x - hi) on ONE accumulator (a.b = hi.hi + hi.lo + lo.hi): the weight-carrying column tiles (G, dz2, dz1 ~ 1 / K) are
// scaled by a power of two that maps  max_k |w_k| sqrt(dt) 8  (a scan of all weights; |xi| < 6) into [2^6, 2^7), where their
// residuals are normal f16 numbers; the unweighted row operands (x, h = r^2) are O(1), where a subnormal residual costs at
// most 3e-8 absolute, fp32's own epsilon.  Stated range: the network factors between G and dz2 / dz1 stay below 256 and
// h = r^2 below 65504.  The partial gradient is scaled back when it is written (exact).  Phase (1) runs its three adjoint
// products through gemm_Tx on split tables of the same size as the fp32 ones.
template <int D, int H, int PASS = 0, bool X3 = false>
__global__ __launch_bounds__(256) void hjbd_bwd_kernel(const DnetArgs da) {
    PSP_COND_EXIT(da.h);
    GradCheck<X3> gchk;                                      // backward side of the range guard (hjb_kernels.h)
    using W = DGeo<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP, EXT = W::EXT;
    constexpr bool PH1 = PASS != 1;                                 // adjoint panels needed
    constexpr bool GCOL = PASS != 2;                                // G columns handled here
    constexpr int C0 = (PASS == 2) ? DB : 0;                        // first column (global numbering: G tiles, then dz2, dz1)
    constexpr int NC = (PASS == 0) ? DB + 2 * HB : (PASS == 1 ? DB : 2 * HB);
    constexpr int NRI = (PASS == 2) ? DB + HB : DB + 2 * HB, NR = cdiv(NRI, 4);
    const HjbArgs& a = da.h;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, qq = lane >> 4, col = lane & 15;
    const unsigned lofsU = (unsigned)image_lane_offset_F(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int d = da.d_real, hh = da.h_real, to = da.time_input ? 1 : 0, di = d + to;
    const long long Pset = W::n_params(d, hh, da.time_input);
    const long long oW2 = (long long)di * hh + hh, oW3 = oW2 + (long long)(di + hh) * hh + hh;
    float* bufs = lds + W::bEx;
    const int S = da.slices, n_items = a.N * S;
    float gs = a.sqdt;
    float ginv = 1.0f;
    if constexpr (X3) {
        float wm = 0.f;
        const int Kpad = a.ntile16 * 16;
        for (int k = tid; k < Kpad; k += nthr) wm = fmaxf(wm, fabsf(da.wts[k]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if (lane == 0) bufs[sub] = wm;
        __syncthreads();
        const float amax = fmaxf(fmaxf(bufs[0], bufs[1]), fmaxf(bufs[2], bufs[3])) * a.sqdt * 8.0f;
        const unsigned e = (__float_as_uint(amax) >> 23) & 0xFFu;
        const bool ok = e >= 7u && e <= 253u;                     // zero / tiny / non-finite weights: no scaling
        const float sc = ok ? __uint_as_float((260u - e) << 23) : 1.0f;
        ginv = ok ? __uint_as_float((e - 6u) << 23) : 1.0f;
        gs = a.sqdt * sc;
    }
    typedef const __attribute__((address_space(1))) float* gptr_t;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int n = item / S, sl = item % S;
        const int t_lo = (int)((long long)a.ntile16 * sl / S), t_hi = (int)((long long)a.ntile16 * (sl + 1) / S);
        const int R = (t_hi - t_lo + 3) / 4;
        // ---- the step's net as A-operand tables, real layout -> zero padded
        const float* __restrict__ P = a.params + (long long)(da.per_step ? n : 0) * Pset;
        __syncthreads();                                  // the previous item's readers are done with tables and exchange
        if constexpr (PH1) {
        auto w3h2 = [&](int row, int c2) { return (row < hh && c2 < d) ? P[oW3 + (long long)(di + hh + row) * d + c2] : 0.f; };
        auto w3h1 = [&](int row, int c2) { return (row < hh && c2 < d) ? P[oW3 + (long long)(di + row) * d + c2] : 0.f; };
        auto w2h = [&](int row, int c2) { return (row < hh && c2 < hh) ? P[oW2 + (long long)(di + row) * hh + c2] : 0.f; };
        if constexpr (X3) {                               // split tables: the same bytes as the fp32 ones (SplitGeo, hjb_kernels.h)
            static_assert(SplitGeo<KP, DB>::floats(HB) == HB * KP * 64 && SplitGeo<4 * HB, HB>::floats(HB) == HB * 4 * HB * 64, "split tables keep the fp32 carve");
            stage_aop_x3<KP, DB>(lds + W::bW3h2, HB, tid, nthr, w3h2);
            stage_aop_x3<KP, DB>(lds + W::bW3h1, HB, tid, nthr, w3h1);
            stage_aop_x3<4 * HB, HB>(lds + W::bW2h, HB, tid, nthr, w2h);
        } else {
            stage_aop(lds + W::bW3h2, HB, KP, tid, nthr, w3h2);
            stage_aop(lds + W::bW3h1, HB, KP, tid, nthr, w3h1);
            stage_aop(lds + W::bW2h, HB, 4 * HB, tid, nthr, w2h);
        }
        }
        __syncthreads();
        const float* img_n = da.pimg + (size_t)n * a.ntile16 * (size_t)W::PBI;
        auto get_F = [&](int t16, int ofs) __attribute__((always_inline)) {
            const unsigned long long addr = sgpr_block_addr(img_n, (unsigned long long)t16, (unsigned)W::PBI, (unsigned)ofs);
            return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((gptr_t)addr + lofsU);
        };
        f32x4 acc[NR][NC];
        float bs[NC];
#pragma unroll
        for (int li = 0; li < NR; ++li)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[li][c] = zero4;
#pragma unroll
        for (int c = 0; c < NC; ++c) bs[c] = 0.f;

        // phase-1 operands of the own block (T layout), requested one round ahead: one wave per SIMD, nothing else hides them
        f32x4 pG[PH1 ? DB : 1], pR2[PH1 ? HB : 1], pR1[PH1 ? HB : 1];
        float pwk = 0.f;
        auto request1 = [&](int it2) __attribute__((always_inline)) {
            if constexpr (PH1) {
                const int t0 = t_lo + 4 * it2 + sub;
                const bool bvalid = t0 < t_hi;
                const int t16 = bvalid ? t0 : t_hi - 1;
                const float* pb = img_n + (size_t)t16 * (size_t)W::PBI + lane;
                const int k = t16 * 16 + j;
                pwk = (bvalid && k < a.K_local) ? da.wts[k] * gs : 0.f;
#pragma unroll
                for (int b = 0; b < DB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pG[b][r] = pb[W::pXi + (4 * b + r) * 64];
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pR2[m][r] = pb[W::pR2 + (4 * m + r) * 64]; pR1[m][r] = pb[W::pR1 + (4 * m + r) * 64]; }
            }
        };
        request1(0);
#pragma unroll 1
        for (int it = 0; it < R; ++it) {
            float* exch = bufs + (it & 1) * 4 * (EXT * 256);
            if constexpr (PH1) {   // ---- (1) adjoint panels of the own block, T layout (operands requested a round ahead)
                f32x4 G[DB];
#pragma unroll
                for (int b = 0; b < DB; ++b) G[b] = pwk * pG[b];
                f32x4 dz2[HB], dz1[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) { dz2[m] = zero4; dz1[m] = zero4; }
                if constexpr (X3) gemm_Tx<HB, KP, DB, 1>(dz2, lds + W::bW3h2, G, lane);
                else gemm_T<HB, KP, DB>(dz2, lds + W::bW3h2, G, lane);            // dh2 = W3h2 G
#pragma unroll
                for (int m = 0; m < HB; ++m) dz2[m] = dz2[m] * (2.0f * pR2[m]);
                if constexpr (X3) {
                    gemm_Tx<HB, KP, DB, 1>(dz1, lds + W::bW3h1, G, lane);
                    gemm_Tx<HB, 4 * HB, HB, 1>(dz1, lds + W::bW2h, dz2, lane);
                } else {
                    gemm_T<HB, KP, DB>(dz1, lds + W::bW3h1, G, lane);             // dh1 = W3h1 G + W2h dz2
                    gemm_T<HB, 4 * HB, HB>(dz1, lds + W::bW2h, dz2, lane);
                }
#pragma unroll
                for (int m = 0; m < HB; ++m) dz1[m] = dz1[m] * (2.0f * pR1[m]);
                float* my_ex = exch + sub * (EXT * 256);
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    tile_put(my_ex + m * 256, dz2[m], lane);
                    tile_put(my_ex + (HB + m) * 256, dz1[m], lane);
                }
            }
            __syncthreads();                              // (the buffer written two rounds ago is free: every wave has passed
                                                          //  the previous round's barrier after reading it)
            request1(it + 1 < R ? it + 1 : it);
            if constexpr (X3) {
                // ---- (2, split products) pairs of sample blocks.  ONE set of raw-operand registers: the operands of a pair are split
                // into f16 packs first, then the next pair is requested into the same registers and the MFMAs of this pair issue
                f32x4 gx[2][GCOL ? DB : 1], ax[2][NR], wv[2];
                auto requestp = [&](int p) __attribute__((always_inline)) {
#pragma unroll
                    for (int hh2 = 0; hh2 < 2; ++hh2) {
                        const int t0 = t_lo + 4 * it + 2 * p + hh2;
                        const bool bvalid = t0 < t_hi;
                        const int t16 = __builtin_amdgcn_readfirstlane(bvalid ? t0 : t_hi - 1);
                        f32x4 w4 = *reinterpret_cast<const f32x4*>(da.wts + t16 * 16 + 4 * qq);
                        wv[hh2] = bvalid ? w4 * gs : zero4;
                        if constexpr (GCOL) {
#pragma unroll
                            for (int b = 0; b < DB; ++b) gx[hh2][b] = get_F(t16, W::pXi + b * 256);
                        }
#pragma unroll
                        for (int li = 0; li < NR; ++li) {
                            const int i0 = sub + 4 * li, i = i0 < NRI ? i0 : NRI - 1;
                            const int ofs = (i < DB) ? W::pX + i * 256
                                          : ((i < DB + HB) ? W::pR1 + (i - DB) * 256 : W::pR2 + (i - DB - HB) * 256);
                            ax[hh2][li] = get_F(t16, ofs);
                        }
                    }
                };
                auto pack_split = [&](const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) __attribute__((always_inline)) {
                    split8u(u0, u1, hi, lo);
                };
                requestp(0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float* ex0 = exch + (2 * p) * (EXT * 256);
                    const float* ex1 = exch + (2 * p + 1) * (EXT * 256);
                    // (a block beyond the slice: its weights are zero, its exchange tiles hold zeros -- phase (1) wrote them with
                    //  weight zero --, and the clamped row operands it re-reads meet those zero columns only)
                    // row packs first (they meet both column groups), then the G columns, then -- after the next pair's raw operands
                    // have been requested into the registers the G columns came from -- the exchanged dz2 / dz1 columns: at most
                    // DB + NR pack pairs are live at once (all NC + NR of them spilled 130 registers)
                    f16x8 Ah[NR], Al[NR];
#pragma unroll
                    for (int li = 0; li < NR; ++li) {
                        const int i = sub + 4 * li;
                        if (i < DB) pack_split(ax[0][li], ax[1][li], Ah[li], Al[li]);
                        else pack_split(ax[0][li] * ax[0][li], ax[1][li] * ax[1][li], Ah[li], Al[li]);      // h = r^2
                    }
                    auto products = [&](auto c0c, auto ncc, const auto& Bh, const auto& Bl) __attribute__((always_inline)) {
                        constexpr int CB = decltype(c0c)::value, NB = decltype(ncc)::value;      // local columns CB .. CB + NB - 1
#pragma unroll
                        for (int li = 0; li < NR; ++li) {
                            const int i = sub + 4 * li;        // row item (wave-uniform): x block, h1 block or h2 block
                            if (i < NRI) {
                                const int ncol = (i < DB) ? DB + 2 * HB : ((i < DB + HB) ? DB + HB : DB);
#pragma unroll
                                for (int c = 0; c < NB; ++c)
                                    if (C0 + CB + c < ncol) {
                                        acc[li][CB + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[li], Bh[c], acc[li][CB + c], 0, 0, 0);
                                        acc[li][CB + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[li], Bl[c], acc[li][CB + c], 0, 0, 0);
                                        acc[li][CB + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[li], Bh[c], acc[li][CB + c], 0, 0, 0);
                                    }
                            }
                        }
                    };
                    if constexpr (GCOL) {
                        f16x8 Bh[DB], Bl[DB];
#pragma unroll
                        for (int b = 0; b < DB; ++b) {
                            const f32x4 c0 = wv[0] * gx[0][b], c1 = wv[1] * gx[1][b];
                            if (sub == 0) bs[b] += hsum4(c0) + hsum4(c1);
                            pack_split(c0, c1, Bh[b], Bl[b]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (p < 1) requestp(p + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        products(std::integral_constant<int, 0>{}, std::integral_constant<int, DB>{}, Bh, Bl);
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                        if (p < 1) requestp(p + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (PH1) {
                        f16x8 Bh[2 * HB], Bl[2 * HB];
#pragma unroll
                        for (int m = 0; m < 2 * HB; ++m) {
                            const f32x4 c0 = tile_get(ex0 + m * 256, lane), c1 = tile_get(ex1 + m * 256, lane);
                            if (sub == 0) bs[DB - C0 + m] += hsum4(c0) + hsum4(c1);
                            pack_split(c0, c1, Bh[m], Bl[m]);
                        }
                        products(std::integral_constant<int, DB - C0>{}, std::integral_constant<int, 2 * HB>{}, Bh, Bl);
                    }
                }
            } else {
            // ---- (2) weight-gradient outer products, feature on lane.  One wave per SIMD: nothing else hides the latency of the
            //      image reads, so the raw operands of block sb + 1 are requested before the MFMAs of block sb issue
            f32x4 gx[2][GCOL ? DB : 1], ax[2][NR], wv[2];
            auto request = [&](int buf, int sb) __attribute__((always_inline)) {
                const int t0 = t_lo + 4 * it + sb;
                const bool bvalid = t0 < t_hi;
                const int t16 = __builtin_amdgcn_readfirstlane(bvalid ? t0 : t_hi - 1);
                f32x4 w4 = *reinterpret_cast<const f32x4*>(da.wts + t16 * 16 + 4 * qq);
                wv[buf] = bvalid ? w4 * gs : zero4;
                if constexpr (GCOL) {
#pragma unroll
                    for (int b = 0; b < DB; ++b) gx[buf][b] = get_F(t16, W::pXi + b * 256);
                }
#pragma unroll
                for (int li = 0; li < NR; ++li) {
                    const int i0 = sub + 4 * li, i = i0 < NRI ? i0 : NRI - 1;
                    const int ofs = (i < DB) ? W::pX + i * 256
                                  : ((i < DB + HB) ? W::pR1 + (i - DB) * 256 : W::pR2 + (i - DB - HB) * 256);
                    ax[buf][li] = get_F(t16, ofs);
                }
            };
            request(0, 0);
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                const int cur = sb & 1;
                if (sb < 3) request(cur ^ 1, sb + 1);
                __builtin_amdgcn_sched_barrier(0);
                const float* ex = exch + sb * (EXT * 256);
                f32x4 ct[NC];                              // local column c <-> global column C0 + c
                if constexpr (GCOL) {
#pragma unroll
                    for (int b = 0; b < DB; ++b) ct[b] = wv[cur] * gx[cur][b];
                }
                if constexpr (PH1) {
#pragma unroll
                    for (int m = 0; m < 2 * HB; ++m) ct[DB - C0 + m] = tile_get(ex + m * 256, lane);
                }
                if (sub == 0) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) bs[c] += hsum4(ct[c]);
                }
#pragma unroll
                for (int li = 0; li < NR; ++li) {
                    const int i = sub + 4 * li;            // row item (wave-uniform): x block, h1 block or h2 block
                    if (i < NRI) {
                        const f32x4 raw = ax[cur][li];
                        const f32x4 A = (i < DB) ? raw : raw * raw;             // h = r^2
                        const int ncol = (i < DB) ? DB + 2 * HB : ((i < DB + HB) ? DB + HB : DB);   // h2 rows meet G only, h1 rows G and dz2
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            if (C0 + c < ncol) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[li][c] = mfma16(A[r], ct[c][r], acc[li][c]);
                            }
                    }
                }
            }
            }
        }
        // ---- flush the item: tile (row item i, column tile c): lane (col, qq), reg rr <-> dW[16 rb + 4 qq + rr][16 cb + col]
        float* gp = da.partial + (size_t)item * (size_t)W::PP;
#pragma unroll
        for (int li = 0; li < NR; ++li) {
            const int i = sub + 4 * li;
            if (i < NRI) {
                const int rbase = (i < DB) ? 16 * i : ((i < DB + HB) ? D + 16 * (i - DB) : D + H + 16 * (i - DB - HB));
#pragma unroll
                for (int cl = 0; cl < NC; ++cl) {
                    const int c = C0 + cl;
                    const bool used = (i < DB) || (i < DB + HB ? c < DB + HB : c < DB);
                    if (used) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int row = rbase + 4 * qq + rr;
                            const float v = ginv * acc[li][cl][rr];      // (X3: the power-of-two scale of the column tiles, taken back)
                            if (c < DB) { const float gv_ = v; gp[W::gW3 + row * D + 16 * c + col] = gv_; gchk.see(gv_); }                          // . G
                            else if (c < DB + HB) { const float gv_ = v; gp[W::gW2 + row * H + 16 * (c - DB) + col] = gv_; gchk.see(gv_); }          // . dz2
                            else { const float gv_ = v; gp[W::gW1 + row * H + 16 * (c - DB - HB) + col] = gv_; gchk.see(gv_); }                      // . dz1 (x rows only)
                        }
                    }
                }
            }
        }
        if (sub == 0) {                                   // bias gradients: column sums over the lane's samples, then over qq
#pragma unroll
            for (int cl = 0; cl < NC; ++cl) {
                const int c = C0 + cl;
                float v = ginv * bs[cl];
                v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                if (qq == 0) {
                    if (c < DB) { const float gv_ = v; gp[W::gb3 + 16 * c + col] = gv_; gchk.see(gv_); }
                    else if (c < DB + HB) { const float gv_ = v; gp[W::gb2 + 16 * (c - DB) + col] = gv_; gchk.see(gv_); }
                    else { const float gv_ = v; gp[W::gb1 + 16 * (c - DB - HB) + col] = gv_; gchk.see(gv_); }
                }
            }
        }
    }
    gchk.raise(da.h.cond);
}

// host-side launch table entry of this family
struct DnetInstance {
    int d, H;
    int lds_bytes;
    int set_floats, vec_floats, shared_floats;
    hipError_t (*launch_fwd)(const DnetArgs&, int grid, hipStream_t);
    int image_block_floats, partial_floats, bwd_lds_bytes;
    int bwd_passes;            // launches of hjbd_bwd_kernel: 1, 2 (columns split so that a wave's accumulator tiles fit its
                               // registers) or 0 (not covered: the library-GEMM formulation is used instead)
    hipError_t (*launch_bwd)(const DnetArgs&, int grid, hipStream_t);
    hipError_t (*launch_adj)(const DnetArgs&, int grid, hipStream_t);      // adjoint sweep (transposed tables + hjbd_adj_kernel)
    int lds_bytes_x3;                                                      // split-product forward (PSP_MLP_F16X3)
    hipError_t (*launch_fwd_x3)(const DnetArgs&, int grid, hipStream_t);
    hipError_t (*launch_adj_x3)(const DnetArgs&, int grid, hipStream_t);
    hipError_t (*launch_bwd_x3)(const DnetArgs&, int grid, hipStream_t);   // split-product weight-gradient outer products
};

template <int D, int H>
struct DnetLaunch {
    using W = DGeo<D, H>;
    static hipError_t adj(const DnetArgs& a, int grid, hipStream_t s) {
        hipLaunchKernelGGL((hjbd_tables_kernel<D, H>), dim3(512), dim3(256), 0, s, a, 1);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int bytes = W::lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_adj_kernel<D, H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_adj_kernel<D, H>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t adj_x3(const DnetArgs& a, int grid, hipStream_t s) {
        hipLaunchKernelGGL((hjbd_tables_kernel<D, H>), dim3(512), dim3(256), 0, s, a, 3);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int bytes = W::lds_floats_x3 * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_adj_kernel<D, H, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_adj_kernel<D, H, true>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t fwd(const DnetArgs& a, int grid, hipStream_t s) {
        hipLaunchKernelGGL((hjbd_tables_kernel<D, H>), dim3(512), dim3(256), 0, s, a, 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int bytes = W::lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_fwd_kernel<D, H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_fwd_kernel<D, H>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t fwd_x3(const DnetArgs& a, int grid, hipStream_t s) {
        hipLaunchKernelGGL((hjbd_tables_kernel<D, H>), dim3(512), dim3(256), 0, s, a, 2);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int bytes = W::lds_floats_x3 * 4;
        const HjbArgs& h = a.h;
        if (spec_enabled() && h.noise_mode == NOISE_PHILOX && h.drift_kind == DRIFT_DENSE && h.sigma_kind == SIGMA_DENSE && h.adaptive && h.runcost_kind == RUN_ZERO &&
            h.loss_kind != LOSS_RELENT && h.uref == nullptr && h.tfeat == nullptr) {            // the LLGC training launch: SPEC
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_fwd_kernel<D, H, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbd_fwd_kernel<D, H, true, true>), dim3(grid), dim3(256), bytes, s, a);
            return hipGetLastError();
        }
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_fwd_kernel<D, H, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_fwd_kernel<D, H, true>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static constexpr int NALL = W::DB + 2 * W::HB;
    static constexpr int kTiles0 = cdiv(NALL, 4) * NALL;                                      // accumulator tiles per wave, one launch
    static constexpr int kTiles1 = cdiv(NALL, 4) * W::DB, kTiles2 = cdiv(W::DB + W::HB, 4) * 2 * W::HB;   // ... split by columns
    static constexpr int kMaxTiles = 48;                  // beyond this the 512-entry register file spills heavily
    static constexpr int kPasses = kTiles0 <= kMaxTiles ? 1 : ((kTiles1 <= kMaxTiles && kTiles2 <= kMaxTiles) ? 2 : 0);
    template <int PASS>
    static hipError_t bwd_pass(const DnetArgs& a, int grid, hipStream_t s) {
        const int bytes = W::bwd_lds_floats * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_bwd_kernel<D, H, PASS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_bwd_kernel<D, H, PASS>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t bwd(const DnetArgs& a, int grid, hipStream_t s) {
        if constexpr (kPasses == 1) {
            return bwd_pass<0>(a, grid, s);
        } else if constexpr (kPasses == 2) {
            hipError_t e = bwd_pass<1>(a, grid, s);
            return e != hipSuccess ? e : bwd_pass<2>(a, grid, s);
        } else {
            return hipErrorNotSupported;
        }
    }
    template <int PASS>
    static hipError_t bwd_pass_x3(const DnetArgs& a, int grid, hipStream_t s) {
        const int bytes = W::bwd_lds_floats * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbd_bwd_kernel<D, H, PASS, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbd_bwd_kernel<D, H, PASS, true>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t bwd_x3(const DnetArgs& a, int grid, hipStream_t s) {       // split-product outer products (detached adaptive runs)
        if constexpr (kPasses == 1) {
            return bwd_pass_x3<0>(a, grid, s);
        } else if constexpr (kPasses == 2) {
            hipError_t e = bwd_pass_x3<1>(a, grid, s);
            return e != hipSuccess ? e : bwd_pass_x3<2>(a, grid, s);
        } else {
            return hipErrorNotSupported;
        }
    }
    static DnetInstance instance() {
        // (set / shared table sizes: the larger of the fp32 and the split layouts -- the caller allocates one region for both)
        return DnetInstance{D, H, W::lds_floats * 4, W::set_floats > W::set_floats_x3 ? W::set_floats : W::set_floats_x3, W::vec_floats,
                            W::oSets > W::oSets_x ? W::oSets : W::oSets_x, &fwd,
                            W::PBI, W::PP, W::bwd_lds_floats * 4, kPasses, &bwd, &adj, W::lds_floats_x3 * 4, &fwd_x3, &adj_x3, &bwd_x3};
    }
};

}  // namespace psp

#define PSP_DEFINE_DNET_INSTANCE(D_, H_) \
    extern "C" psp::DnetInstance psp_dnet_instance_##D_##_##H_() { return psp::DnetLaunch<D_, H_>::instance(); }
#define PSP_DECLARE_DNET_INSTANCE(D_, H_) extern "C" psp::DnetInstance psp_dnet_instance_##D_##_##H_();
