// genl_kernels.h -- GeneralSolver / EllipticSolver rollout for value nets of ANY depth: V = dense-concat net (d [+ 1] -> 1,
// arch = [H_1 .. H_L]), 1 <= L <= 4 hidden layers of up to 128 units (reference function_space.py:116-140 DenseNet: layer i sees
// the concatenation of the input and of all earlier hidden outputs, activation relu^2; :143-158 DenseNet_tanh: tanh, nn.Linear
// weights; `Committor function.ipynb` cell 1: tanh^2).  The (d, H)-templated kernels of gen_kernels.h keep their tables in LDS
// and their activations in registers, which fixes them to two hidden layers of at most 64; the nets the reference's
// diffusion-loss notebooks actually train are deeper and wider (Allen-Cahn.ipynb:72: arch = [110, 110, 50] at d = 100;
// Committor function.ipynb: [d + 10, d, d, d] with tanh^2).  This family takes shapes and activation at RUN time:
//   * activations live in per-tile LDS images in T layout (k-step image: element (ks, lane = j + 16 q) = feature 4 ks + q of
//     sample j, hjb_kernels.h), every segment of the dense concatenation padded to whole 16-feature blocks;
//   * weights live in global memory (L2-resident: <= 1 MB) as pre-permuted A-operand tables built per call by
//     genl_tables_kernel, forward orientation (out^T = W^T in^T) and reverse orientation (g_in = W g_out) per layer;
//   * every product is a rolled, register-double-buffered loop of v_mfma_f32_16x16x4_f32 over the k-steps of the input image
//     (tables are k-quad-major: one 16-byte load per lane straight from L2 feeds four MFMAs per image).
// NW waves share a 16-trajectory tile (NW = 8: products cut by OUTPUT block, the hidden block hb of the concatenation belongs to
// wave hb % NW, which keeps that block's r = relu(z) / tanh(z) and tangent z' in REGISTERS -- the LDS holds only what a product
// reads as its B operand; NW = 1 for small nets: no barriers at all, many tiles per CU).
//   genl_fwd_kernel   per step: V(X, t), grad_x V by the reverse sweep, masked Euler-Maruyama step, Y update (h sees V(X, t) and
//                     the state BEFORE the move; exit tests of the bounded domains; solver.py:1091-1160 / :730-790) -- the same
//                     step as gen_fwd_kernel; keeps per sample only (x, t), the tangent direction s u^ and the coefficient a^.
//                     A tile whose trajectories have all stopped leaves the time loop (solver.py:1093-1097 / :742-744; the
//                     committor notebook runs N = 5000 with exits after 500 .. 1600 steps) and records its step count.
//   genl_bwd_kernel   per sample block, in parallel over all executed (n, tile): recomputes the activations and their tangent
//                     along s u^, runs the adjoint sweep of  a V + w d/d(s u^) V,  and accumulates EVERY parameter gradient in
//                     registers: the weight tiles dW_i (16 inputs x 16 units) as MFMA outer products over the 16 samples of the
//                     block straight from the LDS images (feature-on-lane reads), biases and the output layer as lane sums.
//                     Per-workgroup partial gradients, summed in a fixed order by reduce_grad_kernel.
// Derivation (per sample; a_0 = [x, t], z_i = W_i^T a_{i-1} + b_i, r_i = relu(z_i) | tanh(z_i), h_i = phi(r_i), a_i = [a_{i-1}, h_i],
// V = w^T a_L + b;  phi(r) = r^2 (relu^2, tanh^2) or r (tanh);  phi1 = dh/dz, phi2 = d^2h/dz^2 as functions of r):
//   tangent along u:  a_0' = [u, 0], z_i' = W_i^T a_{i-1}', h_i' = phi1(r_i) z_i', V' = w^T a_L'
//   adjoint of S = a V + w V':  abar_L = a w, abar_L' = w w;  for i = L..1:
//       zbar_i' = abar_h' phi1(r_i),   zbar_i = abar_h phi1(r_i) + abar_h' phi2(r_i) z_i',
//       abar_{i-1} = abar_a + W_i zbar_i,  abar_{i-1}' = abar_a' + W_i zbar_i';
//   dW_i = a_{i-1} zbar_i^T + a_{i-1}' zbar_i'^T, db_i = zbar_i, dw = a a_L + w a_L', db = a.
#pragma once
#include "gen_kernels.h"

namespace psp {

constexpr int GENL_MAXL = 4;        // hidden layers
constexpr int GENL_MAXDB = 7;       // input blocks (d + 1 <= 112)
constexpr int GENL_MAXHB = 8;       // hidden blocks per layer (H <= 128)
enum { GACT_RELU2 = 0, GACT_TANH2 = 1, GACT_TANH = 2 };

// per-NW register-array bounds: hidden blocks per wave (HBsum <= 32; NW = 1 only for HBsum <= 8), concatenation blocks per wave
// (TB <= 39; NW = 1 only for TB <= 16)
template <int NW> struct GenlGeo {
    static constexpr int MAXSLOT = (NW == 1) ? 8 : (32 + NW - 1) / NW;
    // weight-gradient tiles per wave and launch group = accumulator quads of the backward kernel: eight waves of 32 hold the 256
    // tiles of the Allen-Cahn notebook net
    static constexpr int MAXT = 32;
};

struct GenlArgs {
    GenArgs g;                      // problem, noise, outputs, weights: same meaning as in gen_kernels.h
    const float* tables;            // table region (built by genl_tables_kernel)
    float* tables_w;                // same pointer, writable (tables kernel)
    int d, D0, has_time, L;
    int H[GENL_MAXL], HB[GENL_MAXL];
    int off[GENL_MAXL + 1];         // block offset of segment s in the padded concatenation (s = 0: the input); off[L] + HB[L-1] = TB
    int roff[GENL_MAXL + 1];        // real feature offset of segment s in a_L
    int inw[GENL_MAXL + 1];         // real input width of layer i (i = L: the output layer)
    int oW[GENL_MAXL + 1], ob[GENL_MAXL + 1];   // flat parameter offsets: W_i, b_i; index L: the output layer (in_L x 1), b
    int TB, DB0;
    long long tF[GENL_MAXL], tR[GENL_MAXL];      // float offsets of the forward / reverse tables
    long long vB[GENL_MAXL], vW;                 // float offsets of the staged bias vectors / output-layer vector
    int HBsum;                      // sum of HB[i]
    int act;                        // GACT_*
    int linear_layout;              // 1: weights stored (out, in) (nn.Linear: DenseNet_tanh) instead of (in, out)
    int* nexec;                     // (ntile16) time steps the tile executed before all of its trajectories had stopped
    // backward
    float* gpart;                   // (gridDim.x, P) partial gradients
    int n_tiles;                    // weight-gradient tiles: sum_i off[i + 1] * HB[i]
    int tcum[GENL_MAXL + 1];        // tiles of the layers below i
    long long P;
    long long table_floats;         // size of the table region (TLDS instances copy it into LDS)
    int time_first;                 // 1: the net's input is [t, x] (Solver's value-function ansatz, solver.py:338) -- the kernels keep the
                                    // time in their LAST input row; only the parameter index map differs
    float time_scale;               // the net sees time_scale * t (value-function ansatz: the step index n = t / dt, solver.py:336, 439)
};

// padded feature index -> real index inside the concatenation a (or -1: padding)
// (AP: pointer to the arguments -- a generic one in the tables kernel, the kernel-argument segment in the rollout kernels)
template <class AP>
__device__ __forceinline__ int genl_real_feature(AP a, int pf) {
    const int pb = pf >> 4;
    int s = 0;
#pragma unroll
    for (int i = 1; i <= GENL_MAXL; ++i) if (i <= a->L && pb >= a->off[i]) s = i;
    const int c = pf - 16 * a->off[s];
    const int width = (s == 0) ? a->D0 : a->H[s - 1];
    if (c >= width) return -1;
    if (s == 0 && a->time_first) return c == a->d ? 0 : c + 1;       // kernel rows [x, t] <-> parameter rows [t, x]
    return a->roff[s] + c;
}
// weight multiplier of a padded input feature: the time row carries time_scale (tables and gradients alike)
template <class AP>
__device__ __forceinline__ float genl_feature_scale(AP a, int pf) {
    return (a->has_time && a->time_scale != 1.0f && pf == a->d) ? a->time_scale : 1.0f;
}
// flat parameter index of W_i[input feature rf][unit u] (i = L: the output layer, u = 0)
template <class AP>
__device__ __forceinline__ int genl_w_index(AP a, int i, int rf, int u) {
    const int Hi = (i < a->L) ? a->H[i] : 1;
    return a->oW[i] + (a->linear_layout ? u * a->inw[i] + rf : rf * Hi + u);
}

// A-operand tables + staged vectors from the flat parameters (registration order W_1, b_1, .., W_out, b_out)
__global__ __launch_bounds__(256) void genl_tables_kernel(const GenlArgs a) {
    PSP_COND_EXIT(a.g);
    const float* __restrict__ P = a.g.params;
    float* T = a.tables_w;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gn = (long long)gridDim.x * blockDim.x;
    for (int i = 0; i < a.L; ++i) {
        const int Hi = a.H[i], HBi = a.HB[i];
        const int inb = a.off[i + 1];                                   // input blocks of layer i
        const int KSin = 4 * inb, KSh = 4 * HBi;
        // forward: [mb][ks / 4][lane][ks & 3] (one 16-byte load per lane = the A operands of four consecutive k-steps),
        // row = 16 mb + rowmap(lane & 15) (an output unit), k = 4 ks + q (a padded input feature)
        for (long long idx = gtid; idx < (long long)HBi * KSin * 64; idx += gn) {
            const int lane = (int)((idx >> 2) & 63);
            const long long t = idx >> 8;                           // (mb, ks / 4)
            const int ks = 4 * (int)(t % (KSin / 4)) + (int)(idx & 3), mb = (int)(t / (KSin / 4));
            const int ii = lane & 15, q = lane >> 4;
            const int row = 16 * mb + 4 * (ii & 3) + (ii >> 2);
            const int rf = genl_real_feature(&a, 4 * ks + q);
            T[a.tF[i] + idx] = (row < Hi && rf >= 0) ? genl_feature_scale(&a, 4 * ks + q) * P[genl_w_index(&a, i, rf, row)] : 0.f;
        }
        // reverse: [ob][ks / 4][lane][ks & 3], row = 16 ob + rowmap (a padded input feature), k = 4 ks + q (an output unit)
        for (long long idx = gtid; idx < (long long)inb * KSh * 64; idx += gn) {
            const int lane = (int)((idx >> 2) & 63);
            const long long t = idx >> 8;
            const int ks = 4 * (int)(t % (KSh / 4)) + (int)(idx & 3), ob = (int)(t / (KSh / 4));
            const int ii = lane & 15, q = lane >> 4;
            const int pfr = 16 * ob + 4 * (ii & 3) + (ii >> 2);
            const int rf = genl_real_feature(&a, pfr);
            const int col = 4 * ks + q;
            T[a.tR[i] + idx] = (col < Hi && rf >= 0) ? genl_feature_scale(&a, pfr) * P[genl_w_index(&a, i, rf, col)] : 0.f;
        }
        // bias in T-layout vector staging: [(b * 4 + q) * 4 + r] <- v(16 b + 4 r + q)
        for (long long idx = gtid; idx < (long long)HBi * 16; idx += gn) {
            const int r = (int)(idx & 3), q = (int)((idx >> 2) & 3), b = (int)(idx >> 4);
            const int f = 16 * b + 4 * r + q;
            T[a.vB[i] + idx] = f < Hi ? P[a.ob[i] + f] : 0.f;
        }
    }
    for (long long idx = gtid; idx < (long long)a.TB * 16; idx += gn) {       // output layer over the padded concatenation
        const int r = (int)(idx & 3), q = (int)((idx >> 2) & 3), b = (int)(idx >> 4);
        const int rf = genl_real_feature(&a, 16 * b + 4 * r + q);
        T[a.vW + idx] = rf >= 0 ? genl_feature_scale(&a, 16 * b + 4 * r + q) * P[a.oW[a.L] + rf] : 0.f;
    }
}

typedef const GenlArgs* KArgs;
typedef const GenArgs* KGen;

// ---- activation as functions of the stored r (relu(z) or tanh(z)); `act` is a kernel argument: uniform branches
__device__ __forceinline__ f32x4 gact_r(int act, f32x4 z) { return act == GACT_RELU2 ? relu4(z) : tanh4(z); }
__device__ __forceinline__ f32x4 gact_h(int act, f32x4 r) { return act == GACT_TANH ? r : r * r; }
__device__ __forceinline__ f32x4 gact_h1(int act, f32x4 r) {
    if (act == GACT_RELU2) return 2.0f * r;
    const f32x4 s = 1.0f - r * r;
    return act == GACT_TANH ? s : (2.0f * r) * s;
}
__device__ __forceinline__ f32x4 gact_h2(int act, f32x4 r) {
    if (act == GACT_RELU2) return step2(r);
    const f32x4 s = 1.0f - r * r;
    return act == GACT_TANH ? (-2.0f * r) * s : (2.0f * s) * (1.0f - 3.0f * (r * r));
}

// acc += Table . image over KS k-steps (KS a multiple of 4: whole 16-feature blocks) for ONE output block.  One 16-byte table
// load per lane feeds four MFMAs; the loads run a whole chunk of GENL_U quads ahead of the MFMAs that consume them (two register
// buffers): the products are fed from L2 (latency ~1 us), and a chunk of eight quads is 32 (64 with two images) MFMAs deep.
template <int GENL_U>
__device__ __forceinline__ void genl_gemm1(f32x4& acc, const float* __restrict__ tbl, int KS, const float* img, int lane) {
    const f32x4* t0 = reinterpret_cast<const f32x4*>(tbl) + lane;
    const int n4 = KS / 4;
    f32x4 cur[GENL_U], nxt[GENL_U];
#pragma unroll
    for (int u = 0; u < GENL_U; ++u) { cur[u] = f32x4{0.f, 0.f, 0.f, 0.f}; if (u < n4) cur[u] = t0[(size_t)u * 64]; }
    for (int base = 0; base < n4; base += GENL_U) {
#pragma unroll
        for (int u = 0; u < GENL_U; ++u) { const int k4 = base + GENL_U + u; nxt[u] = cur[u]; if (k4 < n4) nxt[u] = t0[(size_t)k4 * 64]; }
#pragma unroll
        for (int u = 0; u < GENL_U; ++u)
            if (base + u < n4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mfma16(cur[u][r], img[(4 * (base + u) + r) * 64 + lane], acc);
            }
#pragma unroll
        for (int u = 0; u < GENL_U; ++u) cur[u] = nxt[u];
    }
}
// the same with two images sharing the table operands (value and tangent passes, adjoint and tangent-adjoint passes)
template <int GENL_U>
__device__ __forceinline__ void genl_gemm1x2(f32x4& acc, f32x4& acd, const float* __restrict__ tbl, int KS, const float* img,
                                             const float* imgd, int lane) {
    const f32x4* t0 = reinterpret_cast<const f32x4*>(tbl) + lane;
    const int n4 = KS / 4;
    f32x4 cur[GENL_U], nxt[GENL_U];
#pragma unroll
    for (int u = 0; u < GENL_U; ++u) { cur[u] = f32x4{0.f, 0.f, 0.f, 0.f}; if (u < n4) cur[u] = t0[(size_t)u * 64]; }
    for (int base = 0; base < n4; base += GENL_U) {
#pragma unroll
        for (int u = 0; u < GENL_U; ++u) { const int k4 = base + GENL_U + u; nxt[u] = cur[u]; if (k4 < n4) nxt[u] = t0[(size_t)k4 * 64]; }
#pragma unroll
        for (int u = 0; u < GENL_U; ++u)
            if (base + u < n4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (4 * (base + u) + r) * 64 + lane;
                    acc = mfma16(cur[u][r], img[o], acc);
                    acd = mfma16(cur[u][r], imgd[o], acd);
                }
            }
#pragma unroll
        for (int u = 0; u < GENL_U; ++u) cur[u] = nxt[u];
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
// the tile's waves share the LDS images; one wave per tile needs no barrier (its LDS accesses are ordered)
template <int NW> __device__ __forceinline__ void tile_sync() { if constexpr (NW > 1) __syncthreads(); }

// layer index of hidden block hb (hidden blocks counted over the layers: hb = off[i + 1] - DB0 + mb) -- uniform
__device__ __forceinline__ int genl_layer_of(KArgs a, int hb) {
    int i = 0;
#pragma unroll
    for (int l = 1; l < GENL_MAXL; ++l) if (l < a->L && hb >= a->off[l + 1] - a->DB0) i = l;
    return i;
}

// ---- value net at the point held in image A (blocks 0 .. DB0 - 1 filled): fills the hidden segments of A (h_i); the wave keeps
// r_i of ITS hidden blocks in Rr (slot s <-> hidden block wave + NW s); returns V.  Padded rows / features carry zero weights
// and biases, so they stay exactly zero.
template <int NW>
__device__ __forceinline__ float genl_value(KArgs a, const float* __restrict__ T, float* A, f32x4 (&Rr)[GenlGeo<NW>::MAXSLOT], int lane, int q, int wave) {
    for (int i = 0; i < a->L; ++i) {
        const int HBi = a->HB[i];
        const int seg = a->off[i + 1];                                // first block of this layer's output segment = its input blocks
        const int KSin = 4 * seg, hoff = seg - a->DB0;
#pragma unroll
        for (int s = 0; s < GenlGeo<NW>::MAXSLOT; ++s) {
            const int mb = wave + NW * s - hoff;
            if (mb >= 0 && mb < HBi) {
                f32x4 acc = vec_get(T + a->vB[i], mb, q);
                genl_gemm1<8>(acc, T + a->tF[i] + (size_t)mb * KSin * 64, KSin, A, lane);
                const f32x4 r = gact_r(a->act, acc);
                Rr[s] = r;
                img_put(A, seg + mb, gact_h(a->act, r), lane);
            }
        }
        tile_sync<NW>();                                             // the next layer reads what this one wrote
    }
    float v = 0.f;                                                   // (every wave forms the same sum)
    for (int b = 0; b < a->TB; ++b) v = dot4(vec_get(T + a->vW, b, q), img_get(A, b, lane), v);
    return qsum(v) + a->g.params[a->ob[a->L]];
}

// ---- grad of V w.r.t. the input segment by the reverse sweep: G (TB blocks) <- w; for i = L..1: G_h *= phi1(r) in place (= gz),
// G[0 .. seg) += W_i gz.  On return blocks 0 .. DB0 - 1 of G hold grad_{[x, t]} V.
template <int NW>
__device__ __forceinline__ void genl_input_gradient(KArgs a, const float* __restrict__ T, const f32x4 (&Rr)[GenlGeo<NW>::MAXSLOT], float* G,
                                                    int lane, int q, int wave) {
    for (int b = wave; b < a->TB; b += NW) img_put(G, b, vec_get(T + a->vW, b, q), lane);
    tile_sync<NW>();
    for (int i = a->L - 1; i >= 0; --i) {
        const int seg = a->off[i + 1];                                // first block of h_i; also the number of input blocks of layer i
        const int HBi = a->HB[i], KSh = 4 * HBi, hoff = seg - a->DB0;
#pragma unroll
        for (int s = 0; s < GenlGeo<NW>::MAXSLOT; ++s) {
            const int mb = wave + NW * s - hoff;
            if (mb >= 0 && mb < HBi) img_put(G, seg + mb, img_get(G, seg + mb, lane) * gact_h1(a->act, Rr[s]), lane);
        }
        tile_sync<NW>();
        for (int ob = wave; ob < seg; ob += NW) {
            f32x4 acc = img_get(G, ob, lane);
            genl_gemm1<8>(acc, T + a->tR[i] + (size_t)ob * KSh * 64, KSh, G + seg * 256, lane);
            img_put(G, ob, acc, lane);
        }
        tile_sync<NW>();
    }
}

// LDS: A, G (TB blocks each), 1 KiB per block
__host__ __device__ inline int genl_fwd_lds_bytes(int TB) { return 2 * TB * 1024; }

// (NW = 4: two workgroups per CU at 256 registers a wave -- the step chain of a tile is bound by the L2 latency of its table
//  operands and by its barriers, not by the matrix pipe, so two tiles in flight per CU are worth more than eight waves on one
//  once the batch fills the chip; NW = 8 for small batches, where the latency of ONE tile is what counts)
// TLDS (one-wave instances of small nets): the whole table region is copied into LDS behind the images at kernel start -- the step
// chain of a small net is a sequence of ~20 tiny products, each of which otherwise opens with the full L2 latency of its first
// operands (`Committor function.ipynb`'s net: 29 KB of tables, 15 us per step from L2)
template <int NW, bool TLDS = false>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 1 : 2) void genl_fwd_kernel(const GenlArgs ga_) {
    PSP_COND_EXIT(ga_.g);
    static_assert(!TLDS || NW == 1, "LDS-resident tables: one-wave instances only");
    const KArgs ga = &ga_;
    const KGen a = &ga->g;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* A = lds;
    float* G = A + ga->TB * 256;
    const float* __restrict__ T = ga->tables;
    if constexpr (TLDS) {
        float* Tl = lds + 2 * ga->TB * 256;
        for (long long i = threadIdx.x; i < ga->table_floats / 4; i += 64 * NW)
            reinterpret_cast<f32x4*>(Tl)[i] = reinterpret_cast<const f32x4*>(ga->tables)[i];
        T = Tl;
    }
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool w0 = wave == 0;                                       // every wave carries the tile's state; wave 0 writes the outputs
    const int D = ga->d, DB0 = ga->DB0;
    const int t16 = blockIdx.x;
    const int k = t16 * 16 + j;
    const bool kvalid = k < a->K_local;
    const uint32_t kglob = (uint32_t)(a->k_offset + k);
    const float dt = a->dt, sqdt = a->sqdt, sig = a->sigma_scale, Tend = a->T;
    const int TBq = D >> 4, TRq = (D & 15) >> 2, TQq = D & 3;        // position of the time input (feature index D) in the T layout
    unsigned long long nact = 0;
    for (int i = threadIdx.x; i < 2 * ga->TB * 256; i += 64 * NW) lds[i] = 0.f;
    tile_sync<NW>();

    f32x4 X[GENL_MAXDB];
    f32x4 Rr[GenlGeo<NW>::MAXSLOT];
#pragma unroll
    for (int s = 0; s < GenlGeo<NW>::MAXSLOT; ++s) Rr[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * b + 4 * r + q;
            const float v = (b < DB0) ? a->x0[(size_t)(kvalid ? k : 0) * D + (f < D ? f : D - 1)] : 0.f;
            X[b][r] = (f < D && kvalid) ? v : 0.f;
        }
    float t = (kvalid && ga->has_time) ? a->t0[k] : 0.f;
    bool stopped = !kvalid;
    float Y = 0.f;
    int msteps = 0;                                                  // active steps of this trajectory (exact)
    auto put_time = [&](float tv) {
        if (ga->has_time) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b == TBq && r == TRq && opaque_i(q) == TQq) X[b][r] = tv;
        }
    };
    auto put_state = [&]() {
        if (w0) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b) if (b < DB0) img_put(A, b, X[b], lane);
        }
        tile_sync<NW>();
    };
    put_time(t);
    const float* vdr = a->drift;                                      // (d) kappa / diagonal of A, read per block below
    const size_t PBL = (size_t)2 * DB0 * 256;                        // path block: X image, U image

    int nex = a->N;
    for (int n = 0; n < a->N; ++n) {
        // every trajectory of the tile frozen: nothing changes any more (solver.py:1093-1097 / :742-744 leave the loop); all the
        // waves of the tile carry the same state, so the verdict is uniform over the workgroup
        if (__builtin_amdgcn_ballot_w64(!stopped) == 0ull) { nex = n; break; }
        // (feature masks (16 b + 4 r + q < D) are invariant over the time loop: hoisted, they are ~60 SGPR pairs, all spilled to VGPR
        //  lanes; an opaque copy of q per step keeps them as one v_cmp where they are used)
        const int qv = opaque_i(q);
        put_state();
        const float Vnow = genl_value<NW>(ga, T, A, Rr, lane, q, wave);
        if (n == 0) Y = Vnow;                                        // solver.py:1081 / :721
        if (a->Vsteps && w0 && q == 0) {                             // Solver's value-function ansatz: sum_n (Y_n(X_n) - Y)^2 (solver.py:438-440)
            a->Vsteps[(size_t)n * (a->ntile16 * 16) + k] = Vnow;
            a->Ysteps[(size_t)n * (a->ntile16 * 16) + k] = Y;
        }
        genl_input_gradient<NW>(ga, T, Rr, G, lane, q, wave);
        const float alivef = stopped ? 0.f : 1.f;
        auto noise_block = [&](int b) __attribute__((always_inline)) {
            f32x4 xi;
            if (a->noise_mode == NOISE_PHILOX) {
                xi = philox_block((uint32_t)opaque_i((int)kglob), (uint32_t)n, (uint32_t)(4 * b + opaque_i(q)), a->iter, a->seed_lo, a->seed_hi);
            } else {
                const float* xrow = a->xi + ((size_t)n * a->K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int f = 16 * b + 4 * r + qv; xi[r] = xrow[f < D ? f : D - 1]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) xi[r] = ((16 * b + 4 * r + qv) < D && kvalid) ? xi[r] : 0.f;
            return xi;
        };
        auto z_block = [&](int b) __attribute__((always_inline)) {       // Z = sigma^T grad_x V, sigma = s I (solver.py:1104)
            const f32x4 gx = img_get(G, b, lane);
            f32x4 Z;
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[r] = ((16 * b + 4 * r + qv) < D) ? sig * gx[r] : 0.f;
            return Z;
        };
        auto drift_vec = [&](int b) __attribute__((always_inline)) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int f = 16 * b + 4 * r + qv; v[r] = (f < D && a->drift_kind != DRIFT_ZERO) ? vdr[f] : 0.f; }
            return v;
        };
        auto move_block = [&](int b, const f32x4& Z, const f32x4& xi) __attribute__((always_inline)) {
            const f32x4 cdt = a->adaptive ? (-dt) * Z : 0.f * Z;
            f32x4 drift = 0.f * Z;
            if (a->drift_kind == DRIFT_DWELL) drift = -(4.0f * drift_vec(b) * (X[b] * (X[b] * X[b] - 1.0f)));
            else if (a->drift_kind == DRIFT_DIAG) drift = drift_vec(b) * X[b];
            return (drift * dt + sig * cdt + (sig * sqdt) * xi) * alivef;
        };
        float rr = 0.f;
        if (a->domain_kind == DOM_SPHERE || a->domain_kind == DOM_ANNULUS || a->h_kind >= GH_EXPBALL_LIN) {
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b < DB0 && (16 * b + 4 * r + qv) < D) rr = fmaf(X[b][r], X[b][r], rr);
            rr = qsum(rr);
        }
        bool inside = true;
        if (a->domain_kind == DOM_SPHERE) {
            inside = sqrtf(rr) < a->dom_a;                            // the state BEFORE the move (:1121)
        } else if (a->domain_kind == DOM_ANNULUS) {
            const float rad = sqrtf(rr);                             // 'two_spheres' (:1122-1123 / :752-753), the state before the move
            inside = rad > a->dom_a && rad < a->dom_b;
        } else if (a->domain_kind >= DOM_BOX) {                       // the boxes test the PROPOSAL (:1126-1129)
            float n_out = 0.f, n_le = 0.f;
#pragma unroll
            for (int b = 0; b < GENL_MAXDB; ++b)
                if (b < DB0) {
                    const f32x4 xi = noise_block(b);
                    const f32x4 Xp = X[b] + move_block(b, z_block(b), xi);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((16 * b + 4 * r + qv) < D) {
                            const bool lo_ok = a->domain_kind != DOM_BOX || Xp[r] >= a->dom_a, hi_ok = Xp[r] <= a->dom_b;
                            n_out += (lo_ok && hi_ok) ? 0.f : 1.f;
                            n_le += hi_ok ? 1.f : 0.f;
                        }
                }
            n_out = qsum(n_out); n_le = qsum(n_le);
            inside = a->domain_kind == DOM_BOX_UPPER_ANY ? n_le > 0.f : n_out == 0.f;
        }
        const bool in_time = inside && (t + dt) <= Tend;             // new_selection (:1119-1131), fp32
        const bool act = in_time && !stopped;
        const float actf = act ? 1.f : 0.f;
        float S = 0.f, Pz = 0.f;
        float* pblk = a->path + ((size_t)n * a->ntile16 + t16) * PBL + lane;
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
            if (b < DB0) {
                const f32x4 xi = noise_block(b);
                const f32x4 Z = z_block(b);
#pragma unroll
                for (int r = 0; r < 4; ++r) { S = fmaf(Z[r], Z[r], S); Pz = fmaf(Z[r], xi[r], Pz); }
                const f32x4 cdt = a->adaptive ? (-dt) * Z : 0.f * Z;
                f32x4 u = sqdt * xi + cdt;                           // u^ = act ((-h_z + c) dt + xi sqrt(dt)), -h_z = Z for h = -|z|^2 / 2
                if (a->h_kind == GH_QUAD) u += dt * Z;
                const f32x4 U = (actf * sig) * u;
                const f32x4 step = move_block(b, Z, xi);
                if (a->store_path && w0) {                            // the sample point is the state BEFORE the move
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pblk[(4 * b + r) * 64] = X[b][r];
                        pblk[(size_t)DB0 * 256 + (4 * b + r) * 64] = U[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool fx = (16 * b + 4 * r + qv) < D;
                    X[b][r] = (fx && act) ? X[b][r] + step[r] : X[b][r];
                }
            }
        S = qsum(S); Pz = qsum(Pz);
        float minus_h = 0.f, hy = 0.f;                               // Y update (solver.py:1141-1142): h sees V(X, t), not the running Y
        if (a->h_kind == GH_QUAD) minus_h = 0.5f * S;
        else if (a->h_kind == GH_ALLEN_CAHN) { minus_h = -(Vnow - Vnow * Vnow * Vnow); hy = 1.0f - 3.0f * Vnow * Vnow; }
        else if (a->h_kind >= GH_EXPBALL_LIN) {
            const float al = a->h_par[0];
            const float lin = 2.0f * al * (2.0f * al * rr + a->h_par[1]) + a->h_par[2];
            float nl = 0.f, nly = 0.f;
            if (a->h_kind != GH_EXPBALL_LIN) {
                const float arg = expf(2.0f * al * rr + 2.0f * a->h_par[3] * ((float)n * dt)) - Vnow * Vnow;
                if (a->h_kind == GH_EXPBALL_SQ) { nl = arg; nly = -2.0f * Vnow; }
                else { nl = sinf(arg); nly = -2.0f * Vnow * cosf(arg); }
            }
            minus_h = Vnow * lin - nl;
            hy = nly - lin;
        }
        const float zc = a->adaptive ? -S : 0.f;
        Y = Y + ((minus_h + zc) * dt + Pz * sqdt) * actf;
        if (a->store_path && w0 && q == 0) a->ahat[(size_t)n * (a->ntile16 * 16) + k] = (n == 0 ? 1.f : 0.f) - hy * dt * actf;
        t = t + dt * actf;
        put_time(t);
        if (act && q == 0) ++nact;
        msteps += act ? 1 : 0;
        stopped = stopped || !in_time;
        tile_sync<NW>();                                             // the G image is read above and rewritten by the next step
    }
    // final point: V(X_N, t_N) (solver.py:1163 / :799) as an extra value-only sample
    put_state();
    const float VN = genl_value<NW>(ga, T, A, Rr, lane, q, wave);
    if (a->store_path && w0) {
        float* pblk = a->path + ((size_t)a->N * a->ntile16 + t16) * PBL + lane;
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
            if (b < DB0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { pblk[(4 * b + r) * 64] = X[b][r]; pblk[(size_t)DB0 * 256 + (4 * b + r) * 64] = 0.f; }
            }
        if (q == 0) a->ahat[(size_t)a->N * (a->ntile16 * 16) + k] = 1.f;
    }
    if (w0 && lane == 0 && ga->nexec) ga->nexec[t16] = nex;
    // (no time input = EllipticSolver: t_N counts the active steps as m dt with one rounding, gen_kernels.h)
    if (kvalid && w0 && q == 0) { a->VN[k] = VN; a->YN[k] = Y; a->tN[k] = ga->has_time ? t : (float)msteps * dt; }
    if (kvalid && w0) {
#pragma unroll
        for (int b = 0; b < GENL_MAXDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * b + 4 * r + q;
                if (b < DB0 && f < D) a->XN[(size_t)k * D + f] = X[b][r];
            }
    }
    for (int o = 1; o < 64; o <<= 1) nact += __shfl_xor(nact, o);
    if (w0 && lane == 0 && nact) atomicAdd(a->kcount, nact);
}

// =======================================================================================
// Backward kernel: per block of 16 samples (n, tile) of the path store.
// LDS: A, Ad (a, a': TB blocks each), AB, ABd (abar, abar' of the HIDDEN segments, later zbar_i, zbar_i' in place: TB - DB0 each,
// and one more block each for the output layer seen as a layer of one unit with zbar = a, zbar' = w), then one int per
// weight-gradient tile of this launch group (its operand offsets).
// gridDim.y = launch groups of NW * MAXT weight-gradient tiles (a group recomputes the sweep and accumulates its tiles; group 0
// also carries the bias gradients): the notebooks' nets need one group.
// =======================================================================================
__host__ __device__ inline int genl_bwd_lds_bytes(int TB, int DB0, int NW) {
    return (4 * TB - 2 * DB0 + 2) * 1024 + NW * 32 * 4;
}

// MS = register slots for the hidden blocks a wave owns (hidden block hb -> wave hb mod NW, slot hb / NW): GenlGeo<NW>::MAXSLOT
// serves every net the kernels accept; the eight-wave instance with MS = 3 (sum of the hidden blocks <= 24: the Allen-Cahn notebook's
// net has 18) keeps 12 registers less per array and requests its table operands two k-step groups ahead instead of four: 37
// instead of 97 spilled dwords -- the weight-tile accumulators that lived in scratch no longer do -- and the backward of that net
// goes 3.98 -> 3.29 ms at K = 16 384 (same-box A/B).
template <int NW, bool TLDS = false, int MS = GenlGeo<NW>::MAXSLOT>
__global__ __launch_bounds__(64 * NW) void genl_bwd_kernel(const GenlArgs ga_) {
    PSP_COND_EXIT(ga_.g);
    static_assert(!TLDS || NW == 1, "LDS-resident tables: one-wave instances only");
    const KArgs ga = &ga_;
    const KGen a = &ga->g;
    const float* __restrict__ T = ga->tables;
    using Geo = GenlGeo<NW>;
    constexpr int MAXT = Geo::MAXT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int TB = ga->TB, DB0 = ga->DB0, HBS = TB - DB0;
    float* A = lds;                   // a
    float* Ad = A + TB * 256;         // a'
    float* AB = Ad + TB * 256;        // abar of hidden block hb at AB + 256 hb; zbar_i in place once layer i has been swept
    float* ABd = AB + (HBS + 1) * 256;    // abar' / zbar_i'
    int* tdesc = reinterpret_cast<int*>(ABd + (HBS + 1) * 256);
    if constexpr (TLDS) {                                            // tables behind the tile descriptors (genl_fwd_kernel)
        float* Tl = reinterpret_cast<float*>(tdesc + NW * GenlGeo<NW>::MAXT);
        for (long long i = threadIdx.x; i < ga->table_floats / 4; i += 64 * NW)
            reinterpret_cast<f32x4*>(Tl)[i] = reinterpret_cast<const f32x4*>(ga->tables)[i];
        T = Tl;
    }
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int Kpad = a->ntile16 * 16;
    const size_t PBL = (size_t)2 * DB0 * 256;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int lofsF = image_lane_offset_F(lane);
    const int tile0 = blockIdx.y * (NW * MAXT);                      // first tile of this launch group
    const bool g0 = blockIdx.y == 0;
    // tile t of layer i = (input block ib, unit block mb), mb fastest; layer L = the output layer (one unit: pseudo block HBS).
    // descriptor = A-image block | zbar-image block << 8 | layer << 16 | mb << 20
    for (int t = threadIdx.x; t < NW * MAXT; t += 64 * NW) {
        const int tt = tile0 + t;
        int dsc = -1;
        if (tt < ga->tcum[ga->L]) {
            int i = 0;
            for (int l = 1; l < GENL_MAXL; ++l) if (l < ga->L && tt >= ga->tcum[l]) i = l;
            const int loc = tt - ga->tcum[i];
            const int ib = loc / ga->HB[i], mb = loc - ib * ga->HB[i];
            dsc = ib | ((ga->off[i + 1] - DB0 + mb) << 8) | (i << 16) | (mb << 20);
        } else if (tt < ga->n_tiles) {
            dsc = (tt - ga->tcum[ga->L]) | (HBS << 8) | (ga->L << 16);
        }
        tdesc[t] = dsc;
    }
    tile_sync<NW>();
    f32x4 accW[MAXT];                 // weight tiles of this wave: slot s <-> tile tile0 + wave + NW s
    f32x4 accB[MS];         // bias gradients of the wave's hidden blocks (summed over the 16 sample lanes at the end)
    f32x4 Rr[MS], Zr[MS];
    float accb = 0.f;
#pragma unroll
    for (int s = 0; s < MAXT; ++s) accW[s] = zero4;
#pragma unroll
    for (int s = 0; s < MS; ++s) { accB[s] = zero4; Rr[s] = zero4; Zr[s] = zero4; }
    const long long nblk = (long long)(a->N + 1) * a->ntile16;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int n = (int)(blk / a->ntile16), t16 = (int)(blk % a->ntile16);
        const bool fin = (n == a->N);
        if (!fin && ga->nexec && n >= ga->nexec[t16]) continue;        // the tile had left the time loop: nothing was stored
        const int k = t16 * 16 + j;
        // per_sample (Solver's value-function ansatz): wY is (N + 1, Kpad) tangent weights and ahat holds the coefficient itself
        const float wy = a->wY[(a->per_sample ? (size_t)n * Kpad : 0) + k], wv = a->per_sample ? 0.f : a->wV[k];
        const float ah = a->ahat[(size_t)n * Kpad + k];
        const bool sval = k < a->K_local;
        const float av = sval ? (a->per_sample ? ah : (fin ? wv : wy * ah)) : 0.f;          // coefficient of grad_theta V
        const float ws = (sval && !fin) ? wy : 0.f;                  // weight of the tangent part
        const float* pb = a->path + (size_t)blk * PBL + lane;
        tile_sync<NW>();                                             // the previous block's tiles have been read
        for (int ks = wave; ks < 4 * DB0; ks += NW) { A[ks * 64 + lane] = pb[ks * 64]; Ad[ks * 64 + lane] = pb[(size_t)DB0 * 256 + ks * 64]; }
        tile_sync<NW>();
        // ---- recompute: z_i, z_i' (shared table operands), a = phi(r), a' = phi1(r) z'; r and z' of the wave's blocks stay in registers
        for (int i = 0; i < ga->L; ++i) {
            const int seg = ga->off[i + 1];
            const int KSin = 4 * seg, HBi = ga->HB[i], hoff = seg - DB0;
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                const int mb = wave + NW * s - hoff;
                if (mb >= 0 && mb < HBi) {
                    f32x4 acc = vec_get(T + ga->vB[i], mb, q), acd = zero4;
                    genl_gemm1x2<(MS < GenlGeo<NW>::MAXSLOT ? 2 : 4)>(acc, acd, T + ga->tF[i] + (size_t)mb * KSin * 64, KSin, A, Ad, lane);
                    const f32x4 r = gact_r(ga->act, acc);
                    Rr[s] = r; Zr[s] = acd;
                    img_put(A, seg + mb, gact_h(ga->act, r), lane);
                    img_put(Ad, seg + mb, gact_h1(ga->act, r) * acd, lane);
                }
            }
            tile_sync<NW>();
        }
        // ---- adjoint seeds abar = a w_out, abar' = w w_out on the hidden segments; the output layer as a one-unit layer: its
        // "zbar" block holds a in unit 0 (dw = a a_L + w a_L' then is one more row of weight tiles), db = a
        if (g0 && wave == 0 && q == 0) accb += av;
        if (wave == 0) {
            img_put(AB, HBS, f32x4{q == 0 ? av : 0.f, 0.f, 0.f, 0.f}, lane);
            img_put(ABd, HBS, f32x4{q == 0 ? ws : 0.f, 0.f, 0.f, 0.f}, lane);
        }
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int hb = wave + NW * s;
            if (hb < HBS) {
                const f32x4 w = vec_get(T + ga->vW, DB0 + hb, q);
                img_put(AB, hb, av * w, lane);
                img_put(ABd, hb, ws * w, lane);
            }
        }
        tile_sync<NW>();
        for (int i = ga->L - 1; i >= 0; --i) {
            const int seg = ga->off[i + 1];
            const int HBi = ga->HB[i], KSh = 4 * HBi, hoff = seg - DB0;
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                const int hb = wave + NW * s;
                if (hb >= hoff && hb < hoff + HBi) {
                    const f32x4 gh = img_get(AB, hb, lane), ghd = img_get(ABd, hb, lane);
                    const f32x4 p1 = gact_h1(ga->act, Rr[s]);
                    const f32x4 zbd = ghd * p1;
                    const f32x4 zb = gh * p1 + ghd * (gact_h2(ga->act, Rr[s]) * Zr[s]);
                    img_put(AB, hb, zb, lane); img_put(ABd, hb, zbd, lane);
                    if (g0) accB[s] += zb;
                }
            }
            tile_sync<NW>();
            if (i > 0) {                                             // (the input segment's adjoint is not needed: no input gradient)
                // hidden segments below layer i only; two products with the same reverse table: abar += W zbar, abar' += W zbar'
                for (int ohb = wave; ohb < hoff; ohb += NW) {
                    f32x4 acc = img_get(AB, ohb, lane), acd = img_get(ABd, ohb, lane);
                    genl_gemm1x2<(MS < GenlGeo<NW>::MAXSLOT ? 2 : 4)>(acc, acd, T + ga->tR[i] + (size_t)(DB0 + ohb) * KSh * 64, KSh, AB + hoff * 256, ABd + hoff * 256, lane);
                    img_put(AB, ohb, acc, lane); img_put(ABd, ohb, acd, lane);
                }
                tile_sync<NW>();
            }
        }
        // ---- weight-gradient tiles: dW_i[16 inputs of block ib][16 units of block mb] += sum over the 16 samples of a zbar^T + a' zbar'^T.
        // Both operands are feature-on-lane reads of the images (lane (i, q') holds feature i of samples 4 q' .. 4 q' + 3, so
        // component r is k-step r of a 16x16x4 product whose k index q' stands for sample 4 q' + r on both sides)
#pragma unroll
        for (int s = 0; s < MAXT; ++s) {
            const int dsc = __builtin_amdgcn_readfirstlane(tdesc[wave + NW * s]);
            if (dsc >= 0) {
                const int ib = dsc & 0xff, hb = (dsc >> 8) & 0xff;
                const f32x4 a4 = image_get_F(A + ib * 256, lofsF), ad4 = image_get_F(Ad + ib * 256, lofsF);
                const f32x4 z4 = image_get_F(AB + hb * 256, lofsF), zd4 = image_get_F(ABd + hb * 256, lofsF);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accW[s] = mfma16(a4[r], z4[r], accW[s]);
                    accW[s] = mfma16(ad4[r], zd4[r], accW[s]);
                }
            }
        }
    }
    // ---- partial gradients of this workgroup: row blockIdx.x of gpart; every entry of the row is written by exactly one lane
    float* gp = ga->gpart + (size_t)blockIdx.x * (size_t)ga->P;
#pragma unroll
    for (int s = 0; s < MAXT; ++s) {
        const int dsc = __builtin_amdgcn_readfirstlane(tdesc[wave + NW * s]);
        if (dsc >= 0) {
            const int ib = dsc & 0xff, i = (dsc >> 16) & 0xf, mb = (dsc >> 20) & 0xf;
            const int u = 16 * mb + j;                               // C layout: lane (column j, q) holds rows 4 q + r
            const int Hi = i < ga->L ? ga->H[i] : 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rf = genl_real_feature(ga, 16 * ib + 4 * q + r);
                if (rf >= 0 && u < Hi) gp[genl_w_index(ga, i, rf, u)] = genl_feature_scale(ga, 16 * ib + 4 * q + r) * accW[s][r];
            }
        }
    }
    if (g0) {
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int hb = wave + NW * s;
            if (hb < HBS) {
                const int i = genl_layer_of(ga, hb);
                const int mb = hb - (ga->off[i + 1] - DB0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = jsumf(accB[s][r]);
                    const int u = 16 * mb + 4 * r + q;
                    if (j == 0 && u < ga->H[i]) gp[ga->ob[i] + u] = v;
                }
            }
        }
        if (wave == 0) {
            const float v = jsumf(accb);
            if (lane == 0) gp[ga->ob[ga->L]] = v;
        }
    }
}

// host side: launches (the dynamic LDS size exceeds the 64 KiB default)
template <int NW, bool TLDS = false> inline hipError_t genl_launch_fwd(const GenlArgs& a, int ntile16, int lds_bytes, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&genl_fwd_kernel<NW, TLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((genl_fwd_kernel<NW, TLDS>), dim3(ntile16), dim3(64 * NW), lds_bytes, st, a);
    return hipGetLastError();
}
template <int NW, bool TLDS = false, int MS = GenlGeo<NW>::MAXSLOT> inline hipError_t genl_launch_bwd(const GenlArgs& a, int grid, int groups, int lds_bytes, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&genl_bwd_kernel<NW, TLDS, MS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((genl_bwd_kernel<NW, TLDS, MS>), dim3(grid, groups), dim3(64 * NW), lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace psp
