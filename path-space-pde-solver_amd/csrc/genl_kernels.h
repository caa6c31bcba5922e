// genl_kernels.h -- GeneralSolver / EllipticSolver rollout for value nets of ANY depth: V = DenseNet(d [+ 1] -> 1, arch = [H_1 .. H_L]),
// 1 <= L <= 4 hidden layers of up to 128 units (reference function_space.py:116-140: layer i sees the concatenation of the
// input and of all earlier hidden outputs, activation relu^2).  The (d, H)-templated kernels of gen_kernels.h keep their tables
// in LDS and their activations in registers, which fixes them to two hidden layers of at most 64; the nets the reference's
// diffusion-loss notebooks actually train are deeper and wider (Allen-Cahn.ipynb:72: arch = [110, 110, 50] at d = 100 -- the
// one configuration with a published timing, 0.31-0.35 s per iteration at K = 200, N = 25).  This family takes the shapes at
// RUN time:
//   * activations live in per-wave LDS images in T layout (k-step image: element (ks, lane = j + 16 q) = feature 4 ks + q of
//     sample j, hjb_kernels.h), every segment of the dense concatenation padded to whole 16-feature blocks;
//   * weights live in global memory (L2-resident: <= 1 MB) as pre-permuted A-operand tables built per call by
//     genl_tables_kernel, forward orientation (out^T = W^T in^T) and reverse orientation (g_in = W g_out) per layer;
//   * every product is a rolled loop of v_mfma_f32_16x16x4_f32 over the k-steps of the input image, two output blocks at a
//     time (tables are k-quad-major: one 16-byte load per lane straight from L2 feeds four MFMAs).
// One wave owns a 16-trajectory tile for all N steps (the time axis is sequential); the workgroup is that one wave, so a
// small batch (K = 200: 13 tiles) still spreads over 13 CUs and nothing needs a barrier beyond wave-level LDS ordering.
//   genl_fwd_kernel   per step: V(X, t), grad_x V by the reverse sweep, masked Euler-Maruyama step, Y update (h sees V(X, t) and
//                     the state BEFORE the move; exit tests of the bounded domains; solver.py:1091-1160 / :730-790) -- the same
//                     step as gen_fwd_kernel; keeps per sample only (x, t), the tangent direction s u^ and the coefficient a^.
//   genl_adj_kernel   per sample, in parallel over all (n, k): recomputes the activations and their tangent along s u^, runs
//                     the adjoint sweep of  a V + w d/d(s u^) V  and leaves a_i, a_i', zbar_i, zbar_i' for the weight-gradient
//                     GEMMs  dW_i = A_i^T Zbar_i + A_i'^T Zbar_i'  (plain library GEMMs over the sample axis, plan side).
// Derivation (per sample; a_0 = [x, t], z_i = W_i^T a_{i-1} + b_i, h_i = relu(z_i)^2, a_i = [a_{i-1}, h_i], V = w^T a_L + b):
//   tangent along u:  a_0' = [u, 0], z_i' = W_i^T a_{i-1}', h_i' = 2 relu(z_i) z_i', V' = w^T a_L'
//   adjoint of S = a V + w V':  abar_L = a w, abar_L' = w w;  for i = L..1:
//       zbar_i' = abar_h' * 2 relu(z_i),   zbar_i = abar_h * 2 relu(z_i) + abar_h' * 2 [z_i > 0] z_i',
//       abar_{i-1} = abar_a + W_i zbar_i,  abar_{i-1}' = abar_a' + W_i zbar_i';
//   dW_i = a_{i-1} zbar_i^T + a_{i-1}' zbar_i'^T, db_i = zbar_i, dw = a a_L + w a_L', db = a.
#pragma once
#include "gen_kernels.h"

namespace psp {

constexpr int GENL_MAXL = 4;        // hidden layers
constexpr int GENL_MAXDB = 7;       // input blocks (d + 1 <= 112)
constexpr int GENL_MAXHB = 8;       // hidden blocks per layer (H <= 128)

struct GenlArgs {
    GenArgs g;                      // problem, noise, outputs, weights: same meaning as in gen_kernels.h
    const float* tables;            // table region (built by genl_tables_kernel)
    float* tables_w;                // same pointer, writable (tables kernel)
    int d, D0, has_time, L;
    int H[GENL_MAXL], HB[GENL_MAXL];
    int off[GENL_MAXL + 1];         // block offset of segment s in the padded concatenation (s = 0: the input); off[L] + HB[L-1] = TB
    int roff[GENL_MAXL + 1];        // real feature offset of segment s in a_L
    int oW[GENL_MAXL + 1], ob[GENL_MAXL + 1];   // flat parameter offsets: W_i (in_i x H_i), b_i; index L: the output layer (in_L x 1), b
    int TB, DB0;
    long long tF[GENL_MAXL], tR[GENL_MAXL];      // float offsets of the forward / reverse tables
    long long vB[GENL_MAXL], vW;                 // float offsets of the staged bias vectors / output-layer vector
    // adjoint kernel: ROW-MAJOR outputs (sample, padded feature), one block of 16 samples after the other
    float* outA; float* outAd; float* outZb; float* outZdb;
    float* out_av;                  // (16 per block) coefficient a of V, (16 per block) weight w of the tangent part
    float* out_wy;
    long long blk0, blk1;           // sample blocks [blk0, blk1) of (N + 1) * ntile16 handled by this launch
    int HBsum;                      // sum of HB[i]
};

// padded feature index -> real index inside the concatenation a (or -1: padding)
__device__ __forceinline__ int genl_real_feature(const GenlArgs& a, int pf) {
    const int pb = pf >> 4;
    int s = 0;
#pragma unroll
    for (int i = 1; i <= GENL_MAXL; ++i) if (i <= a.L && pb >= a.off[i]) s = i;
    const int c = pf - 16 * a.off[s];
    const int width = (s == 0) ? a.D0 : a.H[s - 1];
    return c < width ? a.roff[s] + c : -1;
}

// A-operand tables + staged vectors from the flat parameters (DenseNet registration order W_1, b_1, .., W_out, b_out; weights (in, out))
__global__ __launch_bounds__(256) void genl_tables_kernel(const GenlArgs a) {
    PSP_COND_EXIT(a.g);
    const float* __restrict__ P = a.g.params;
    float* T = a.tables_w;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gn = (long long)gridDim.x * blockDim.x;
    for (int i = 0; i < a.L; ++i) {
        const int Hi = a.H[i], HBi = a.HB[i];
        const int inb = a.off[i] + (i == 0 ? a.DB0 : a.HB[i - 1]);      // input blocks of layer i = off[i + 1]
        const int KSin = 4 * inb, KSh = 4 * HBi;
        // forward: [mb][ks / 4][lane][ks & 3] (one 16-byte load per lane = the A operands of four consecutive k-steps),
        // row = 16 mb + rowmap(lane & 15) (an output unit), k = 4 ks + q (a padded input feature)
        for (long long idx = gtid; idx < (long long)HBi * KSin * 64; idx += gn) {
            const int lane = (int)((idx >> 2) & 63);
            const long long t = idx >> 8;                           // (mb, ks / 4)
            const int ks = 4 * (int)(t % (KSin / 4)) + (int)(idx & 3), mb = (int)(t / (KSin / 4));
            const int ii = lane & 15, q = lane >> 4;
            const int row = 16 * mb + 4 * (ii & 3) + (ii >> 2);
            const int rf = genl_real_feature(a, 4 * ks + q);
            T[a.tF[i] + idx] = (row < Hi && rf >= 0) ? P[a.oW[i] + rf * Hi + row] : 0.f;
        }
        // reverse: [ob][ks / 4][lane][ks & 3], row = 16 ob + rowmap (a padded input feature), k = 4 ks + q (an output unit)
        for (long long idx = gtid; idx < (long long)inb * KSh * 64; idx += gn) {
            const int lane = (int)((idx >> 2) & 63);
            const long long t = idx >> 8;
            const int ks = 4 * (int)(t % (KSh / 4)) + (int)(idx & 3), ob = (int)(t / (KSh / 4));
            const int ii = lane & 15, q = lane >> 4;
            const int rf = genl_real_feature(a, 16 * ob + 4 * (ii & 3) + (ii >> 2));
            const int col = 4 * ks + q;
            T[a.tR[i] + idx] = (col < Hi && rf >= 0) ? P[a.oW[i] + rf * Hi + col] : 0.f;
        }
        // bias in T-layout vector staging: [(b * 4 + q) * 4 + r] <- v(16 b + 4 r + q)
        for (long long idx = gtid; idx < (long long)HBi * 16; idx += gn) {
            const int r = (int)(idx & 3), q = (int)((idx >> 2) & 3), b = (int)(idx >> 4);
            const int f = 16 * b + 4 * r + q;
            T[a.vB[i] + idx] = f < Hi ? P[a.ob[i] + f] : 0.f;
        }
    }
    for (long long idx = gtid; idx < (long long)a.TB * 16; idx += gn) {       // output layer (in_L x 1) over the padded concatenation
        const int r = (int)(idx & 3), q = (int)((idx >> 2) & 3), b = (int)(idx >> 4);
        const int rf = genl_real_feature(a, 16 * b + 4 * r + q);
        T[a.vW + idx] = rf >= 0 ? P[a.oW[a.L] + rf] : 0.f;
    }
}

// Work split inside the workgroup: GENL_NW waves share the tile's LDS images; every product is cut by OUTPUT block -- wave w takes
// the pairs of blocks {2 w, 2 w + 1}, {2 (w + NW), ..}, .. -- and a barrier stands between a layer and the next.  The rolled
// k-loops are unrolled eight deep so that sixteen table operands (L2 latency ~1 us) are in flight per wave.
constexpr int GENL_NW = 4;
// acc[m] (m < nb <= 2 output blocks starting at the table pointer) += Table . image over KS k-steps (KS a multiple of 4: whole
// 16-feature blocks).  One 16-byte table load per lane feeds four MFMAs; four quads are requested per trip, i.e. 8 KiB of
// operands in flight per wave and block -- the products are bound by the L2 latency of these loads, not by the matrix pipe
__device__ __forceinline__ void genl_gemm2(f32x4 (&acc)[2], const float* __restrict__ tbl, int KS, int nb, const float* img, int lane) {
    const f32x4* t0 = reinterpret_cast<const f32x4*>(tbl) + lane;
    const f32x4* t1 = t0 + (nb > 1 ? (size_t)(KS / 4) * 64 : 0);
#pragma unroll 4
    for (int k4 = 0; k4 < KS / 4; ++k4) {
        const f32x4 a0 = t0[(size_t)k4 * 64], a1 = t1[(size_t)k4 * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float b = img[(4 * k4 + r) * 64 + lane];
            acc[0] = mfma16(a0[r], b, acc[0]);
            acc[1] = mfma16(a1[r], b, acc[1]);        // (nb == 1: a second copy of block 0, discarded by the caller)
        }
    }
}
// the same with two images sharing the table operands (value and tangent passes, adjoint and tangent-adjoint passes)
__device__ __forceinline__ void genl_gemm2x2(f32x4 (&acc)[2], f32x4 (&acd)[2], const float* __restrict__ tbl, int KS, int nb,
                                             const float* img, const float* imgd, int lane) {
    const f32x4* t0 = reinterpret_cast<const f32x4*>(tbl) + lane;
    const f32x4* t1 = t0 + (nb > 1 ? (size_t)(KS / 4) * 64 : 0);
#pragma unroll 4
    for (int k4 = 0; k4 < KS / 4; ++k4) {
        const f32x4 a0 = t0[(size_t)k4 * 64], a1 = t1[(size_t)k4 * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float b = img[(4 * k4 + r) * 64 + lane], bd = imgd[(4 * k4 + r) * 64 + lane];
            acc[0] = mfma16(a0[r], b, acc[0]); acd[0] = mfma16(a0[r], bd, acd[0]);
            acc[1] = mfma16(a1[r], b, acc[1]); acd[1] = mfma16(a1[r], bd, acd[1]);
        }
    }
}
__device__ __forceinline__ void img_put(float* img, int blk, const f32x4& v, int lane) {     // T-layout block -> k-steps 4 blk .. 4 blk + 3
    float* p = img + blk * 256 + lane;
    p[0] = v[0]; p[64] = v[1]; p[128] = v[2]; p[192] = v[3];
}
__device__ __forceinline__ f32x4 img_get(const float* img, int blk, int lane) {
    const float* p = img + blk * 256 + lane;
    f32x4 v;
    v[0] = p[0]; v[1] = p[64]; v[2] = p[128]; v[3] = p[192];
    return v;
}
__device__ __forceinline__ f32x4 vec_get(const float* __restrict__ vec, int blk, int q) {   // staged vector: [(b * 4 + q) * 4 + r]
    return *reinterpret_cast<const f32x4*>(vec + (blk * 4 + q) * 4);
}
__device__ __forceinline__ void wave_sync() { __syncthreads(); }      // the workgroup's waves share the LDS images

// ---- value net at the point held in image A (blocks 0 .. DB0 - 1 filled): fills the hidden segments of A (h_i) and R (relu(z_i));
// returns V.  Padded rows / features carry zero weights and biases, so they stay exactly zero.
__device__ __forceinline__ float genl_value(const GenlArgs& a, float* A, float* R, int lane, int q, int wave) {
    const float* __restrict__ T = a.tables;
    for (int i = 0; i < a.L; ++i) {
        const int HBi = a.HB[i];
        const int seg = a.off[i + 1];                                // first block of this layer's output segment = its input blocks
        const int KSin = 4 * seg;
        for (int mb0 = 2 * wave; mb0 < HBi; mb0 += 2 * GENL_NW) {
            const int nb = (HBi - mb0) < 2 ? 1 : 2;
            f32x4 acc[2];
            acc[0] = vec_get(T + a.vB[i], mb0, q);
            acc[1] = vec_get(T + a.vB[i], mb0 + nb - 1, q);
            genl_gemm2(acc, T + a.tF[i] + (size_t)mb0 * KSin * 64, KSin, nb, A, lane);
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < nb) {
                    const f32x4 r = relu4(acc[m]);
                    img_put(R, seg + mb0 + m, r, lane);
                    img_put(A, seg + mb0 + m, r * r, lane);
                }
        }
        wave_sync();                                                 // the next layer reads what this one wrote
    }
    float v = 0.f;                                                   // (every wave forms the same sum)
    for (int b = 0; b < a.TB; ++b) v = dot4(vec_get(T + a.vW, b, q), img_get(A, b, lane), v);
    return qsum(v) + a.g.params[a.ob[a.L]];
}

// ---- grad of V w.r.t. the input segment by the reverse sweep: G (TB blocks) <- w; for i = L..1: gz = G_h * 2 relu(z),
// G[0 .. seg) += W_i gz.  On return blocks 0 .. DB0 - 1 of G hold grad_{[x, t]} V.
__device__ __forceinline__ void genl_input_gradient(const GenlArgs& a, const float* R, float* G, float* GZ, int lane, int q, int wave) {
    const float* __restrict__ T = a.tables;
    for (int b = wave; b < a.TB; b += GENL_NW) img_put(G, b, vec_get(T + a.vW, b, q), lane);
    wave_sync();
    for (int i = a.L - 1; i >= 0; --i) {
        const int seg = a.off[i + 1];                                // first block of h_i; also the number of input blocks of layer i
        const int HBi = a.HB[i], KSh = 4 * HBi;
        for (int m = wave; m < HBi; m += GENL_NW) img_put(GZ, m, img_get(G, seg + m, lane) * (2.0f * img_get(R, seg + m, lane)), lane);
        wave_sync();
        for (int ob0 = 2 * wave; ob0 < seg; ob0 += 2 * GENL_NW) {
            const int nb = (seg - ob0) < 2 ? 1 : 2;
            f32x4 acc[2];
            acc[0] = img_get(G, ob0, lane);
            acc[1] = img_get(G, ob0 + nb - 1, lane);
            genl_gemm2(acc, T + a.tR[i] + (size_t)ob0 * KSh * 64, KSh, nb, GZ, lane);
#pragma unroll
            for (int m = 0; m < 2; ++m) if (m < nb) img_put(G, ob0 + m, acc[m], lane);
        }
        wave_sync();
    }
}

// LDS: A, R, G (TB blocks each) + GZ (GENL_MAXHB blocks), 1 KiB per block
__host__ __device__ inline int genl_fwd_lds_bytes(int TB) { return (3 * TB + GENL_MAXHB) * 1024; }

__global__ __launch_bounds__(64 * GENL_NW) void genl_fwd_kernel(const GenlArgs ga) {
    PSP_COND_EXIT(ga.g);
    const GenArgs& a = ga.g;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* A = lds;
    float* R = A + ga.TB * 256;
    float* G = R + ga.TB * 256;
    float* GZ = G + ga.TB * 256;
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool w0 = wave == 0;                                       // every wave carries the tile's state; wave 0 writes the outputs
    const int D = ga.d, DB0 = ga.DB0;
    const int t16 = blockIdx.x;
    const int k = t16 * 16 + j;
    const bool kvalid = k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    const float dt = a.dt, sqdt = a.sqdt, sig = a.sigma_scale, Tend = a.T;
    const int TBq = D >> 4, TRq = (D & 15) >> 2, TQq = D & 3;        // position of the time input (feature index D) in the T layout
    unsigned long long nact = 0;
    for (int i = threadIdx.x; i < (3 * ga.TB + GENL_MAXHB) * 256; i += 64 * GENL_NW) lds[i] = 0.f;
    wave_sync();

    f32x4 X[GENL_MAXDB];
#pragma unroll
    for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * b + 4 * r + q;
            const float v = (b < DB0) ? a.x0[(size_t)(kvalid ? k : 0) * D + (f < D ? f : D - 1)] : 0.f;
            X[b][r] = (f < D && kvalid) ? v : 0.f;
        }
    float t = (kvalid && ga.has_time) ? a.t0[k] : 0.f;
    bool stopped = !kvalid;
    float Y = 0.f;
    auto put_time = [&](float tv) {
        if (ga.has_time) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b == TBq && r == TRq && q == TQq) X[b][r] = tv;
        }
    };
    auto put_state = [&]() {
        if (w0) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b) if (b < DB0) img_put(A, b, X[b], lane);
        }
        wave_sync();
    };
    put_time(t);
    const float* vdr = a.drift;                                      // (d) kappa / diagonal of A, read per block below
    const size_t PBL = (size_t)2 * DB0 * 256;                        // path block: X image, U image

    for (int n = 0; n < a.N; ++n) {
        put_state();
        const float Vnow = genl_value(ga, A, R, lane, q, wave);
        if (n == 0) Y = Vnow;                                        // solver.py:1081 / :721
        genl_input_gradient(ga, R, G, GZ, lane, q, wave);
        const float alivef = stopped ? 0.f : 1.f;
        auto noise_block = [&](int b) __attribute__((always_inline)) {
            f32x4 xi;
            if (a.noise_mode == NOISE_PHILOX) {
                xi = philox_block((uint32_t)opaque_i((int)kglob), (uint32_t)n, (uint32_t)(4 * b + opaque_i(q)), a.iter, a.seed_lo, a.seed_hi);
            } else {
                const float* xrow = a.xi + ((size_t)n * a.K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int f = 16 * b + 4 * r + q; xi[r] = xrow[f < D ? f : D - 1]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) xi[r] = ((16 * b + 4 * r + q) < D && kvalid) ? xi[r] : 0.f;
            return xi;
        };
        auto z_block = [&](int b) __attribute__((always_inline)) {       // Z = sigma^T grad_x V, sigma = s I (solver.py:1104)
            const f32x4 gx = img_get(G, b, lane);
            f32x4 Z;
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[r] = ((16 * b + 4 * r + q) < D) ? sig * gx[r] : 0.f;
            return Z;
        };
        auto drift_vec = [&](int b) __attribute__((always_inline)) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int f = 16 * b + 4 * r + q; v[r] = (f < D && a.drift_kind != DRIFT_ZERO) ? vdr[f] : 0.f; }
            return v;
        };
        auto move_block = [&](int b, const f32x4& Z, const f32x4& xi) __attribute__((always_inline)) {
            const f32x4 cdt = a.adaptive ? (-dt) * Z : 0.f * Z;
            f32x4 drift = 0.f * Z;
            if (a.drift_kind == DRIFT_DWELL) drift = -(4.0f * drift_vec(b) * (X[b] * (X[b] * X[b] - 1.0f)));
            else if (a.drift_kind == DRIFT_DIAG) drift = drift_vec(b) * X[b];
            return (drift * dt + sig * cdt + (sig * sqdt) * xi) * alivef;
        };
        float rr = 0.f;
        if (a.domain_kind == DOM_SPHERE || a.h_kind >= GH_EXPBALL_LIN) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b < DB0 && (16 * b + 4 * r + q) < D) rr = fmaf(X[b][r], X[b][r], rr);
            rr = qsum(rr);
        }
        bool inside = true;
        if (a.domain_kind == DOM_SPHERE) {
            inside = sqrtf(rr) < a.dom_a;                            // the state BEFORE the move (:1121)
        } else if (a.domain_kind >= DOM_BOX) {                       // the boxes test the PROPOSAL (:1126-1129)
            float n_out = 0.f, n_le = 0.f;
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
                if (b < DB0) {
                    const f32x4 xi = noise_block(b);
                    const f32x4 Xp = X[b] + move_block(b, z_block(b), xi);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((16 * b + 4 * r + q) < D) {
                            const bool lo_ok = a.domain_kind != DOM_BOX || Xp[r] >= a.dom_a, hi_ok = Xp[r] <= a.dom_b;
                            n_out += (lo_ok && hi_ok) ? 0.f : 1.f;
                            n_le += hi_ok ? 1.f : 0.f;
                        }
                }
            n_out = qsum(n_out); n_le = qsum(n_le);
            inside = a.domain_kind == DOM_BOX_UPPER_ANY ? n_le > 0.f : n_out == 0.f;
        }
        const bool in_time = inside && (t + dt) <= Tend;             // new_selection (:1119-1131), fp32
        const bool act = in_time && !stopped;
        const float actf = act ? 1.f : 0.f;
        float S = 0.f, Pz = 0.f;
        float* pblk = a.path + ((size_t)n * a.ntile16 + t16) * PBL + lane;
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
            if (b < DB0) {
                const f32x4 xi = noise_block(b);
                const f32x4 Z = z_block(b);
#pragma unroll
                for (int r = 0; r < 4; ++r) { S = fmaf(Z[r], Z[r], S); Pz = fmaf(Z[r], xi[r], Pz); }
                const f32x4 cdt = a.adaptive ? (-dt) * Z : 0.f * Z;
                f32x4 u = sqdt * xi + cdt;                           // u^ = act ((-h_z + c) dt + xi sqrt(dt)), -h_z = Z for h = -|z|^2 / 2
                if (a.h_kind == GH_QUAD) u += dt * Z;
                const f32x4 U = (actf * sig) * u;
                const f32x4 step = move_block(b, Z, xi);
                if (a.store_path && w0) {                            // the sample point is the state BEFORE the move
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pblk[(4 * b + r) * 64] = X[b][r];
                        pblk[(size_t)DB0 * 256 + (4 * b + r) * 64] = U[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool fx = (16 * b + 4 * r + q) < D;
                    X[b][r] = (fx && act) ? X[b][r] + step[r] : X[b][r];
                }
            }
        S = qsum(S); Pz = qsum(Pz);
        float minus_h = 0.f, hy = 0.f;                               // Y update (solver.py:1141-1142): h sees V(X, t), not the running Y
        if (a.h_kind == GH_QUAD) minus_h = 0.5f * S;
        else if (a.h_kind == GH_ALLEN_CAHN) { minus_h = -(Vnow - Vnow * Vnow * Vnow); hy = 1.0f - 3.0f * Vnow * Vnow; }
        else if (a.h_kind >= GH_EXPBALL_LIN) {
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
        if (a.store_path && w0 && q == 0) a.ahat[(size_t)n * (a.ntile16 * 16) + k] = (n == 0 ? 1.f : 0.f) - hy * dt * actf;
        t = t + dt * actf;
        put_time(t);
        if (act && q == 0) ++nact;
        stopped = stopped || !in_time;
        wave_sync();                                                 // the G image is read above and rewritten by the next step
    }
    // final point: V(X_N, t_N) (solver.py:1163 / :799) as an extra value-only sample
    put_state();
    const float VN = genl_value(ga, A, R, lane, q, wave);
    if (a.store_path && w0) {
        float* pblk = a.path + ((size_t)a.N * a.ntile16 + t16) * PBL + lane;
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
            if (b < DB0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { pblk[(4 * b + r) * 64] = X[b][r]; pblk[(size_t)DB0 * 256 + (4 * b + r) * 64] = 0.f; }
            }
        if (q == 0) a.ahat[(size_t)a.N * (a.ntile16 * 16) + k] = 1.f;
    }
    if (kvalid && w0 && q == 0) { a.VN[k] = VN; a.YN[k] = Y; a.tN[k] = t; }
    if (kvalid && w0) {
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * b + 4 * r + q;
                if (b < DB0 && f < D) a.XN[(size_t)k * D + f] = X[b][r];
            }
    }
    for (int o = 1; o < 64; o <<= 1) nact += __shfl_xor(nact, o);
    if (w0 && lane == 0 && nact) atomicAdd(a.kcount, nact);
}

// =======================================================================================
// Adjoint kernel: per block of 16 samples (n, tile) of the path store.
// LDS: A / Abar (TB), Ad / Abard (TB), R (TB), Zd (TB), two k-step staging images of GENL_MAXHB blocks.
// =======================================================================================
__host__ __device__ inline int genl_adj_lds_bytes(int TB) { return (4 * TB + 2 * GENL_MAXHB) * 1024; }

__global__ __launch_bounds__(64 * GENL_NW) void genl_adj_kernel(const GenlArgs ga) {
    PSP_COND_EXIT(ga.g);
    const GenArgs& a = ga.g;
    const float* __restrict__ T = ga.tables;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int TB = ga.TB, DB0 = ga.DB0;
    float* A = lds;                   // a, then abar
    float* Ad = A + TB * 256;         // a', then abar'
    float* R = Ad + TB * 256;         // relu(z_i) at the hidden segments
    float* Zd = R + TB * 256;         // z_i' at the hidden segments
    float* S1 = Zd + TB * 256;        // staging: zbar_i
    float* S2 = S1 + GENL_MAXHB * 256;  // staging: zbar_i'
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int Kpad = a.ntile16 * 16;
    const size_t PBL = (size_t)2 * DB0 * 256;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long blk = ga.blk0 + blockIdx.x; blk < ga.blk1; blk += gridDim.x) {
        const int n = (int)(blk / a.ntile16), t16 = (int)(blk % a.ntile16);
        const int k = t16 * 16 + j;
        const bool fin = (n == a.N);
        const float wy = a.wY[k], wv = a.wV[k], ah = a.ahat[(size_t)n * Kpad + k];
        const bool sval = k < a.K_local;
        const float av = sval ? (fin ? wv : wy * ah) : 0.f;          // coefficient of grad_theta V
        const float ws = (sval && !fin) ? wy : 0.f;                  // weight of the tangent part
        const float* pb = a.path + (size_t)blk * PBL + lane;
        const size_t ob = (size_t)(blk - ga.blk0);
        wave_sync();                                                 // the previous block's sweep has finished with the images
        for (int b = DB0 + wave; b < TB; b += GENL_NW) { img_put(R, b, zero4, lane); img_put(Zd, b, zero4, lane); }
        for (int ks = wave; ks < 4 * DB0; ks += GENL_NW) { A[ks * 64 + lane] = pb[ks * 64]; Ad[ks * 64 + lane] = pb[(size_t)DB0 * 256 + ks * 64]; }
        wave_sync();
        // ---- recompute: z_i, z_i' (shared table operands), h_i = r^2, h_i' = 2 r z_i'
        for (int i = 0; i < ga.L; ++i) {
            const int seg = ga.off[i + 1];
            const int KSin = 4 * seg, HBi = ga.HB[i];
            for (int mb0 = 2 * wave; mb0 < HBi; mb0 += 2 * GENL_NW) {
                const int nb = (HBi - mb0) < 2 ? 1 : 2;
                f32x4 acc[2], acd[2];
                acc[0] = vec_get(T + ga.vB[i], mb0, q); acc[1] = vec_get(T + ga.vB[i], mb0 + nb - 1, q);
                acd[0] = zero4; acd[1] = zero4;
                genl_gemm2x2(acc, acd, T + ga.tF[i] + (size_t)mb0 * KSin * 64, KSin, nb, A, Ad, lane);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (m < nb) {
                        const f32x4 r = relu4(acc[m]);
                        img_put(R, seg + mb0 + m, r, lane);
                        img_put(Zd, seg + mb0 + m, acd[m], lane);
                        img_put(A, seg + mb0 + m, r * r, lane);
                        img_put(Ad, seg + mb0 + m, (2.0f * r) * acd[m], lane);
                    }
            }
            wave_sync();
        }
        // ---- the activations leave for the weight-gradient GEMMs: ROW-MAJOR (sample, padded feature), 16 rows of 16 TB floats per
        // block.  Wave w writes the rows j = w, w + NW, ..: lane l carries the four features of k-step l (and l + 64) of that
        // sample -- four LDS reads 16 lanes apart -- as one 16-byte store: a row leaves as contiguous 1 KiB pieces.
        {
            const int KSa = 4 * TB;
            // Every row carries one more 16-float block: (1, 0, ..) behind a, zeros behind a' -- the ones column makes the
            // bias gradients (column sums of zbar) a row of the same GEMM that forms the weight gradients.
            for (int jr = wave; jr < 16; jr += GENL_NW) {
                float* rowA = ga.outA + (ob * 16 + jr) * (size_t)(16 * TB + 16);
                float* rowD = ga.outAd + (ob * 16 + jr) * (size_t)(16 * TB + 16);
                for (int ks = lane; ks < KSa; ks += 64) {
                    const float* pa = A + ks * 64 + jr;
                    const float* pd = Ad + ks * 64 + jr;
                    *reinterpret_cast<f32x4*>(rowA + 4 * ks) = f32x4{pa[0], pa[16], pa[32], pa[48]};
                    *reinterpret_cast<f32x4*>(rowD + 4 * ks) = f32x4{pd[0], pd[16], pd[32], pd[48]};
                }
                if (lane < 4) {
                    *reinterpret_cast<f32x4*>(rowA + 16 * TB + 4 * lane) = f32x4{lane == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(rowD + 16 * TB + 4 * lane) = zero4;
                }
            }
            if (wave == 0) {                                         // last block of the zbar rows: (a, 0, ..) and (w, 0, ..)
                float* zr = ga.outZb + (ob * 16 + j) * (size_t)(16 * ga.HBsum + 16) + 16 * ga.HBsum + 4 * q;
                float* zd = ga.outZdb + (ob * 16 + j) * (size_t)(16 * ga.HBsum + 16) + 16 * ga.HBsum + 4 * q;
                *reinterpret_cast<f32x4*>(zr) = f32x4{q == 0 ? av : 0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(zd) = f32x4{q == 0 ? ws : 0.f, 0.f, 0.f, 0.f};
                if (q == 0) { ga.out_av[ob * 16 + j] = av; ga.out_wy[ob * 16 + j] = ws; }
            }
        }
        wave_sync();
        // ---- adjoint sweep: abar = a w_out, abar' = w w_out over the whole concatenation
        for (int b = wave; b < TB; b += GENL_NW) {
            const f32x4 w = vec_get(T + ga.vW, b, q);
            img_put(A, b, av * w, lane);
            img_put(Ad, b, ws * w, lane);
        }
        wave_sync();
        int zoff = ga.HBsum;                                           // block offset of layer i inside the Zbar images
        for (int i = ga.L - 1; i >= 0; --i) {
            const int seg = ga.off[i + 1];
            const int HBi = ga.HB[i], KSh = 4 * HBi;
            zoff -= HBi;
            for (int m = wave; m < HBi; m += GENL_NW) {
                const f32x4 r = img_get(R, seg + m, lane), zd = img_get(Zd, seg + m, lane);
                const f32x4 gh = img_get(A, seg + m, lane), ghd = img_get(Ad, seg + m, lane);
                const f32x4 zbd = ghd * (2.0f * r);
                const f32x4 zb = gh * (2.0f * r) + ghd * (step2(r) * zd);
                img_put(S1, m, zb, lane); img_put(S2, m, zbd, lane);
            }
            wave_sync();
            // zbar_i, zbar_i' of the block, row-major at column 16 zoff of the (sample, 16 HBsum) matrices
            for (int jr = wave; jr < 16; jr += GENL_NW) {
                float* rowZ = ga.outZb + (ob * 16 + jr) * (size_t)(16 * ga.HBsum + 16) + 16 * zoff;
                float* rowZd = ga.outZdb + (ob * 16 + jr) * (size_t)(16 * ga.HBsum + 16) + 16 * zoff;
                if (lane < KSh) {
                    const float* p1 = S1 + lane * 64 + jr;
                    const float* p2 = S2 + lane * 64 + jr;
                    *reinterpret_cast<f32x4*>(rowZ + 4 * lane) = f32x4{p1[0], p1[16], p1[32], p1[48]};
                    *reinterpret_cast<f32x4*>(rowZd + 4 * lane) = f32x4{p2[0], p2[16], p2[32], p2[48]};
                }
            }
            if (i > 0) {                                             // (the input segment's adjoint is not needed: no input gradient)
                // hidden segments below layer i only; two products with the same reverse table: abar += W zbar, abar' += W zbar'
                for (int ob0 = DB0 + 2 * wave; ob0 < seg; ob0 += 2 * GENL_NW) {
                    const int nb = (seg - ob0) < 2 ? 1 : 2;
                    f32x4 acc[2], acd[2];
                    acc[0] = img_get(A, ob0, lane); acc[1] = img_get(A, ob0 + nb - 1, lane);
                    acd[0] = img_get(Ad, ob0, lane); acd[1] = img_get(Ad, ob0 + nb - 1, lane);
                    genl_gemm2x2(acc, acd, T + ga.tR[i] + (size_t)ob0 * KSh * 64, KSh, nb, S1, S2, lane);
#pragma unroll
                    for (int m = 0; m < 2; ++m) if (m < nb) { img_put(A, ob0 + m, acc[m], lane); img_put(Ad, ob0 + m, acd[m], lane); }
                }
                wave_sync();
            }
        }
    }
}

}  // namespace psp
