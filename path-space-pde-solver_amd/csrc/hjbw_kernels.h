// hjbw_kernels.h -- "wide" HJB rollout kernels: the same algorithm as hjb_kernels.h for state dimensions whose
// 16-trajectory panels no longer fit one wave's 256 registers next to the weight tables in LDS (d = 200, 500, ...).
//
// What changes against the narrow family:
//   * one wave per SIMD (256-thread workgroups) so a wave owns the whole 512-entry unified register file:
//     the state panel X (4 ceil(d/16) registers), the control / increment panel and the accumulators of the
//     d x d products stay in registers for any d <= 512;
//   * A-operand tables (W1, W2, W3, dt A, B; W3^T for the backward) live in GLOBAL memory (L2-resident, built per
//     call by hjbw_tables_kernel) in k-step-major order [ks][mb][64], so one k-step of all output blocks is one
//     contiguous run and the k-loop can stay ROLLED (a d = 500 product has 4000 MFMAs);
//     (each wave streams its own operands: a workgroup-shared LDS stage with one barrier per k-step was measured
//     25-40 % slower -- the L2 -> L1 stream of 32 B/clk per CU is not the limit, the barriers are);
//   * the B operand of a rolled k-loop cannot be a register array (dynamic index), so the input panel of the big
//     products is written once per step to a per-wave LDS image [ks][64] and read back one dword per k-step;
//   * the backward is streaming: a wave forms dz2 of its own sample block with a rolled k-loop over the stored
//     xi image, then owns hidden block ib for every weight-gradient tile row and walks the d/16 state blocks,
//     building the G tile of each straight from the xi image in feature-on-lane form (no G exchange at all).
// Layouts, Philox counters, loss weights, path-store format and the flat gradient layout are those of
// hjb_kernels.h; parity tests compare the two families on the same configuration.
#pragma once
#include <type_traits>

#include "hjb_kernels.h"

namespace psp {

template <int D, int H>
struct GeoW {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB, KSH = G::KSH;
    static constexpr int KP = 4 * DB;                 // k-steps of a d-dimensional contraction, padded to whole blocks
    static_assert(HB == 4, "wide kernels are built for 49..64 hidden units (4 hidden blocks)");
    static_assert(KSH == 16 || KSH == 15 || KSH == 14 || KSH == 13, "hidden k-steps");
    // k-step-major A-operand tables in global memory: element ((ks * MB + mb) * 64 + lane)
    static constexpr int tW1 = 0, tW2 = tW1 + KP * HB * 64, tW3 = tW2 + 16 * HB * 64, tA = tW3 + 16 * DB * 64,
                         tB = tA + KP * DB * 64, fwd_table_floats = tB + KP * DB * 64;
    static constexpr int bwd_table_floats = KP * HB * 64;            // W3^T
    // adjoint sweep (transposed orientations), same total size as the forward set
    static constexpr int aBT = 0, aAT = aBT + KP * DB * 64, aW3T = aAT + KP * DB * 64, aW2T = aW3T + KP * HB * 64,
                         aW1T = aW2T + 16 * HB * 64, adj_table_floats = aW1T + 16 * DB * 64;
    static_assert(adj_table_floats == fwd_table_floats, "the sweep reuses the forward kernel's table region");
    // forward LDS (floats): per-feature vectors, reduction scratch, one input image per wave
    static constexpr int vb1 = 0, vw1t = vb1 + HB * 16, vb2 = vw1t + HB * 16, vb3 = vb2 + HB * 16,
                         vdr = vb3 + DB * 16, vrun = vdr + DB * 16, vterm = vrun + DB * 16, fRed = vterm + DB * 16,
                         fImg = fRed + 64, IMG = KP * 64, fwd_lds_floats = fImg + 4 * IMG;
    // split-product forward (hjbw_fwd_kernel<.., X3>): S-step-major tables, element ((S * MB + mb) * 2 + hi/lo) * 64 + lane of
    // f16x8 (one S-step = 32 features = two 16-feature blocks; an odd last block leaves the upper half zero), and the wave's
    // input image as hi / lo packs [S][hi | lo][64] of f16x8
    static constexpr int KS8 = cdiv(DB, 2);
    static constexpr int xW1 = 0, xW2 = xW1 + KS8 * HB * 512, xW3 = xW2 + 2 * HB * 512, xA = xW3 + 2 * DB * 512,
                         xB = xA + KS8 * DB * 512, fwd_x3_table_floats = xB + KS8 * DB * 512;
    // ... and the adjoint sweep's transposed tables in split form (same total as the forward set)
    static constexpr int xaBT = 0, xaAT = xaBT + KS8 * DB * 512, xaW3T = xaAT + KS8 * DB * 512, xaW2T = xaW3T + KS8 * HB * 512,
                         xaW1T = xaW2T + 2 * HB * 512, adj_x3_table_floats = xaW1T + 2 * DB * 512;
    static_assert(adj_x3_table_floats == fwd_x3_table_floats, "the split sweep reuses the split forward's table region");
    static constexpr int IMGX = KS8 * 512, fStage = fImg + 4 * IMGX;
    // the shared table stream (gemm_img_x3s) where TWO workgroups fit a CU with the stage (d <= 256 runs two per CU and they cover
    // each other's barriers); with one wave per SIMD (d > 256) the lock-step costs more than the L2 stream saves (d = 500: 24.7 ->
    // 26.5 ms), and at d = 256 the stage would cost the second workgroup (12.6 -> 12.0 ms only)
    static constexpr bool kShare = D <= 256 && 2 * (fStage + 2 * 4 * 512) * 4 <= 160 * 1024;
    static constexpr int fwd_x3_lds_floats = fStage + (kShare ? 2 * 4 * 512 : 0);
    // (fStage: two buffers of four output blocks' hi / lo operands, 2 x 8 KiB: the table stream of the long products is fetched
    //  ONCE per workgroup and read by its four waves from LDS, gemm_img_x3s)
    // backward LDS (floats): dz2 k-step images of the four blocks of a round, double-buffered; bias staging reuses it
    static constexpr int EXB = 4 * HB * 64, bwd_lds_floats = 2 * 4 * EXB + 4 * 16 * DB;
};

// dst[((ks * MB + mb) * 64 + lane)], lane = i + 16 q  <-  src(row = 16 mb + rowmap(i), col = 4 ks + q)
template <class F>
__device__ __forceinline__ void table_fill(float* dst, int MB, int KS, long long gtid, long long gstride, F src) {
    const long long total = (long long)MB * KS * 64;
    for (long long idx = gtid; idx < total; idx += gstride) {
        const int lane = (int)(idx & 63);
        const int t = (int)(idx >> 6);
        const int mb = t % MB, ks = t / MB;
        const int i = lane & 15, q = lane >> 4;
        dst[idx] = src(16 * mb + 4 * (i & 3) + (i >> 2), 4 * ks + q);
    }
}

// split-product table (hjb_kernels.h gemm_Tx): S-step-major hi / lo f16x8 images, element ((S * MB + mb) * 2 + hi/lo) * 64 + lane;
// k = 8 g + e of S-step S <-> column 32 S + (e < 4 ? 4 e : 16 + 4 (e - 4)) + g
template <class F>
__device__ __forceinline__ void table_fill_x3(float* dstf, int MB, int NS, long long gtid, long long gstride, F src) {
    f16x8* dst = reinterpret_cast<f16x8*>(dstf);
    const long long total = (long long)NS * MB * 64;
    for (long long idx = gtid; idx < total; idx += gstride) {
        const int lane = (int)(idx & 63);
        const int t = (int)(idx >> 6);
        const int mb = t % MB, S = t / MB;
        const int i = lane & 15, g = lane >> 4;
        const int row = 16 * mb + 4 * (i & 3) + (i >> 2);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            split_f16(src(row, 32 * S + (e < 4 ? 4 * e : 16 + 4 * (e - 4)) + g), h, l);
            hi[e] = h; lo[e] = l;
        }
        dst[((long long)t * 2) * 64 + lane] = hi;
        dst[((long long)t * 2 + 1) * 64 + lane] = lo;
    }
}

template <int D, int H>
__global__ __launch_bounds__(256) void hjbw_tables_kernel(const HjbArgs a, int backward) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gs = (long long)gridDim.x * blockDim.x;
    const float* __restrict__ P = a.params;
    float* T = a.tables;
    if (backward == 1) {
        table_fill(T, W::HB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
        return;
    }
    if (backward == 3) {                               // split-product forward tables (hi / lo f16 images)
        auto fill = [&](float* dstf, int MB, int NS, auto src) { table_fill_x3(dstf, MB, NS, gtid, gs, src); };
        fill(T + W::xW1, W::HB, W::KS8, [&](int row, int col) {
            return (row < H && col < D) ? P[G::oW1 + row * (D + 1) + 1 + col] : 0.f; });
        fill(T + W::xW2, W::HB, 2, [&](int row, int col) {
            return (row < H && col < H) ? P[G::oW2 + row * H + col] : 0.f; });
        fill(T + W::xW3, W::DB, 2, [&](int row, int col) {
            return (row < D && col < H) ? P[G::oW3 + row * H + col] : 0.f; });
        if (a.drift_kind == DRIFT_DENSE) {
            const float dt = a.dt;
            const float* __restrict__ A = a.drift;
            fill(T + W::xA, W::DB, W::KS8, [&](int row, int col) {
                return (row < D && col < D) ? dt * A[row * D + col] : 0.f; });
        }
        if (a.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = a.sigma;
            fill(T + W::xB, W::DB, W::KS8, [&](int row, int col) {
                return (row < D && col < D) ? B[row * D + col] : 0.f; });
        }
        return;
    }
    if (backward == 5) {                               // hjbw_bwd_x3_kernel: W3^T as split S-step-major images
        table_fill_x3(T, W::HB, W::KS8, gtid, gs, [&](int row, int col) {
            return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
        return;
    }
    if (backward == 4) {                               // adjoint sweep, split-product tables: B^T, (dt A)^T, W3^T, W2^T, W1x^T
        auto fill = [&](float* dstf, int MB, int NS, auto src) { table_fill_x3(dstf, MB, NS, gtid, gs, src); };
        if (a.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = a.sigma;
            fill(T + W::xaBT, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? B[col * D + row] : 0.f; });
        }
        if (a.drift_kind == DRIFT_DENSE) {
            const float dt = a.dt;
            const float* __restrict__ A = a.drift;
            fill(T + W::xaAT, W::DB, W::KS8, [&](int row, int col) { return (row < D && col < D) ? dt * A[col * D + row] : 0.f; });
        }
        fill(T + W::xaW3T, W::HB, W::KS8, [&](int row, int col) { return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
        fill(T + W::xaW2T, W::HB, 2, [&](int row, int col) { return (row < H && col < H) ? P[G::oW2 + col * H + row] : 0.f; });
        fill(T + W::xaW1T, W::DB, 2, [&](int row, int col) { return (row < D && col < H) ? P[G::oW1 + col * (D + 1) + 1 + row] : 0.f; });
        return;
    }
    if (backward == 2) {                               // adjoint sweep: B^T, (dt A)^T, W3^T, W2^T, W1x^T
        if (a.sigma_kind == SIGMA_DENSE) {
            const float* __restrict__ B = a.sigma;
            table_fill(T + W::aBT, W::DB, W::KP, gtid, gs, [&](int row, int col) {
                return (row < D && col < D) ? B[col * D + row] : 0.f; });
        }
        if (a.drift_kind == DRIFT_DENSE) {
            const float dt = a.dt;
            const float* __restrict__ A = a.drift;
            table_fill(T + W::aAT, W::DB, W::KP, gtid, gs, [&](int row, int col) {
                return (row < D && col < D) ? dt * A[col * D + row] : 0.f; });
        }
        table_fill(T + W::aW3T, W::HB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
        table_fill(T + W::aW2T, W::HB, 16, gtid, gs, [&](int row, int col) {
            return (row < H && col < H) ? P[G::oW2 + col * H + row] : 0.f; });
        table_fill(T + W::aW1T, W::DB, 16, gtid, gs, [&](int row, int col) {
            return (row < D && col < H) ? P[G::oW1 + col * (D + 1) + 1 + row] : 0.f; });
        return;
    }
    table_fill(T + W::tW1, W::HB, W::KP, gtid, gs, [&](int row, int col) {
        return (row < H && col < D) ? P[G::oW1 + row * (D + 1) + 1 + col] : 0.f; });
    table_fill(T + W::tW2, W::HB, 16, gtid, gs, [&](int row, int col) {
        return (row < H && col < H) ? P[G::oW2 + row * H + col] : 0.f; });
    table_fill(T + W::tW3, W::DB, 16, gtid, gs, [&](int row, int col) {
        return (row < D && col < H) ? P[G::oW3 + row * H + col] : 0.f; });
    if (a.drift_kind == DRIFT_DENSE) {
        const float dt = a.dt;
        const float* __restrict__ A = a.drift;
        table_fill(T + W::tA, W::DB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < D && col < D) ? dt * A[row * D + col] : 0.f; });
    }
    if (a.sigma_kind == SIGMA_DENSE) {
        const float* __restrict__ B = a.sigma;
        table_fill(T + W::tB, W::DB, W::KP, gtid, gs, [&](int row, int col) {
            return (row < D && col < D) ? B[row * D + col] : 0.f; });
    }
}

// wave-uniform table pointers are forced into an SGPR pair so that every operand load is
// "SGPR base + lane offset + immediate" (no per-load 64-bit address registers for the compiler to hoist)
typedef const __attribute__((address_space(1))) float* gptr_t;
__device__ __forceinline__ gptr_t sgpr_ptr(const float* p) {
    unsigned long long addr = (unsigned long long)p;
    asm volatile("" : "+s"(addr));
    return (gptr_t)addr;
}

// The INPUT of sgpr_ptr is still ordinary arithmetic: `table + constant` is invariant over the time loop, so every distinct
// constant became a hoisted 64-bit SGPR pair -- 189 of them at d = 500, all spilled to VGPR lanes and read back with two
// v_readlane per load group.  Laundering the table pointer where the product starts makes those sums two SALU adds at the use.
__device__ __forceinline__ const float* opaque_base(const float* p) {
    unsigned long long addr = (unsigned long long)p;
    asm volatile("" : "+s"(addr));
    return (const float*)addr;
}

// acc[MB] += T . in,  T: k-step-major global table with KS k-steps, in: register panel (static indices, unrolled)
// LD = number of output blocks per k-step in the table (MB of them, starting at tbl, are used)
template <int MB, int KS, int INB, int LD = MB>
__device__ __forceinline__ void gemm_regs(f32x4 (&acc)[MB], const float* __restrict__ tbl, const f32x4 (&in)[INB], int lane) {
    static_assert(INB * 4 >= KS, "input panel too small");
    constexpr int CH = (MB >= 16) ? 1 : (MB >= 8 ? 2 : 4);
    constexpr int NCH = cdiv(KS, CH);
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    float buf[2][CH * MB];
    // one SGPR base per (k-step, 16 output blocks): every load is base + lane * 4 + immediate < 4096
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < CH; ++kk) {
            const int ks = c * CH + kk;
            if (ks < KS) {
#pragma unroll
                for (int m0 = 0; m0 < MB; m0 += 16) {
                    gptr_t cb = sgpr_ptr(tbl + (ks * LD + m0) * 64);
#pragma unroll
                    for (int mb = m0; mb < m0 + 16 && mb < MB; ++mb) buf[c & 1][kk * MB + mb] = cb[(mb - m0) * 64 + ul];
                }
            }
        }
    };
    load_chunk(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load_chunk(c + 1);
        __builtin_amdgcn_sched_barrier(0);            // the prefetch stays ahead of this chunk's MFMAs (global operands:
#pragma unroll                                        // a fence that lets VMEM cross would let the loads sink to their use)
        for (int kk = 0; kk < CH; ++kk) {
            const int ks = c * CH + kk;
            if (ks < KS) {
                const float bop = in[ks >> 2][ks & 3];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(buf[c & 1][kk * MB + mb], bop, acc[mb]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc[MB] += T . img,  T: k-step-major global table with KP k-steps (KP % 4 == 0), img: this wave's LDS image of the
// input panel (one dword per lane and k-step).  Rolled over k with a static ring of NST register stages of U k-steps
// (KP % (NST * U) == 0): the operands of a stage are requested NST - 1 stages before its MFMAs issue.
template <int MB, int KP, int LD = MB>
__device__ __forceinline__ void gemm_img(f32x4 (&acc)[MB], const float* __restrict__ tbl, const float* img, int lane) {
    constexpr int NST = 2;                            // (a 4-stage ring of single k-steps was measured 10 % slower at d = 200)
    constexpr int U = (MB >= 16) ? 1 : 2;
    static_assert(KP % (NST * U) == 0, "k padding");
    float ab[NST][U * MB], bb[NST][U];
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    auto load = [&](int st, int ks0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bb[st][u] = img[(ks0 + u) * 64 + lane];
#pragma unroll
            for (int m0 = 0; m0 < MB; m0 += 16) {
                gptr_t tp = sgpr_ptr(tbl + ((size_t)(ks0 + u) * LD + m0) * 64);       // LD: output blocks per k-step in the table
#pragma unroll
                for (int mb = m0; mb < m0 + 16 && mb < MB; ++mb) ab[st][u * MB + mb] = tp[(mb - m0) * 64 + ul];
            }
        }
    };
    auto fma_stage = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(ab[st][u * MB + mb], bb[st][u], acc[mb]);
    };
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) load(st, st * U);
#pragma unroll 1
    for (int ks = 0; ks < KP; ks += NST * U) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            // request the stage NST - 1 ahead (past the end: re-read the last k-steps, valid memory, result unused)
            const int kn = ks + (st + NST - 1) * U;
            load((st + NST - 1) % NST, kn < KP ? kn : KP - U);
            __builtin_amdgcn_sched_barrier(0);
            fma_stage(st);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- split-product versions (three v_mfma_f32_16x16x32_f16 per fp32 product; the split and its error are described at gemm_Tx
// in hjb_kernels.h).  Tables: S-step-major hi / lo images in global memory (GeoW::xW1 ...), 2 KiB per (S, output block).
typedef const __attribute__((address_space(1))) f16x8* gptr8_t;
__device__ __forceinline__ gptr8_t sgpr_ptr8(const float* p) {
    unsigned long long addr = (unsigned long long)p;
    asm volatile("" : "+s"(addr));
    return (gptr8_t)addr;
}
// (the compiler's own split code here, not hjb_kernels.h's split8: with this family's rolled loops and 512-register waves the
// inline-asm version measured 3 % slower at d = 500 -- A/B on one box, round 3)
// (round 4: split8 itself -- the classic form in the wide translation units (-DPSP_SPLIT_CLASSIC), the two-instruction pair form in
// the DenseNet-control units, which include this header and are compiled without the SLP vectoriser)
__device__ __forceinline__ void split_pack(const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) {
    split8(u0, u1, hi, lo);
}
// acc[MB] += T . img over KS8 S-steps; img: this wave's LDS image of hi / lo packs.  Rolled over S; within an S-step the output
// blocks run in chunks of CH with the operands of the next chunk (or of the next S-step's first chunk) requested one chunk ahead.
template <int MB, int KS8, int LD = MB, int GMAX = 16>
__device__ __forceinline__ void gemm_img_x3(f32x4 (&acc)[MB], const float* __restrict__ tbl, const float* img, int lane) {
    constexpr int CH = MB >= 4 ? 4 : MB;                                // (8 spills at d = 500 and under the 256-register cap of d <= 256)
    // output blocks in groups of at most 16: the correction chain of a group is 64 registers instead of 4 MB (d = 500: 128, which
    // spilled 45 dwords per step); every (S, block) operand is still read once, only the input packs are re-read from LDS per group
    constexpr int NG = cdiv(MB, GMAX), GB = cdiv(MB, NG);
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    const f16x8* imgp = reinterpret_cast<const f16x8*>(img) + lane;
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        constexpr int dummy = 0; (void)dummy;
        const int g0 = grp * GB;                                        // first output block of the group (compile-time after unrolling)
        const int gn = (g0 + GB <= MB) ? GB : MB - g0;
        const int NC = (gn + CH - 1) / CH;
        f16x8 ah[2][CH], al[2][CH], bh[2], bl[2];
        f32x4 corr[GB];
#pragma unroll
        for (int m = 0; m < GB; ++m) corr[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_a = [&](int st, int S, int c) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < CH; m += 2) {                           // fresh SGPR base every 4 KiB (two output blocks)
                if (c * CH + m < gn) {
#if defined(PSP_ABL_WIDE) && (PSP_ABL_WIDE & 1)      // timing ablation: every S-step re-reads step 0's operands (L1 hits, same instruction stream)
                    gptr8_t tp = sgpr_ptr8(tbl + ((size_t)(S & 0) * LD + g0 + c * CH + m) * 512);
#elif defined(PSP_ABL_WIDE) && (PSP_ABL_WIDE & 2)    // ... every chunk re-reads ONE 4 KiB pair of blocks
                    gptr8_t tp = sgpr_ptr8(tbl + ((size_t)(S & 0) * LD + ((g0 + c * CH + m) & 0)) * 512);
#else
                    gptr8_t tp = sgpr_ptr8(tbl + ((size_t)S * LD + g0 + c * CH + m) * 512);
#endif
#if defined(PSP_ABL_WIDE) && (PSP_ABL_WIDE & 4)      // timing ablation: half the operand bytes (lo := hi, no second load)
                    ah[st][m] = tp[ul]; al[st][m] = ah[st][m];
                    if (m + 1 < CH && c * CH + m + 1 < gn) { ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = ah[st][m + 1]; }
#elif defined(PSP_ABL_WIDE) && (PSP_ABL_WIDE & 8)    // ... no operand loads inside the loop at all
                    if (S == 0 && c == 0) { ah[st][m] = tp[ul]; al[st][m] = tp[64 + ul]; if (m + 1 < CH) { ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = tp[192 + ul]; } }
                    else { asm volatile("" : "+v"(ah[st][m]), "+v"(al[st][m])); if (m + 1 < CH) asm volatile("" : "+v"(ah[st][m + 1]), "+v"(al[st][m + 1])); }
#else
                    ah[st][m] = tp[ul]; al[st][m] = tp[64 + ul];
                    if (m + 1 < CH && c * CH + m + 1 < gn) { ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = tp[192 + ul]; }
#endif
                }
            }
        };
        auto load_b = [&](int st, int S) __attribute__((always_inline)) {
            bh[st] = imgp[(S * 2) * 64]; bl[st] = imgp[(S * 2 + 1) * 64];
        };
        load_b(0, 0);
        load_a(0, 0, 0);
#pragma unroll 1
        for (int S = 0; S < KS8; S += 2) {                              // two S-steps per trip: static ring indices
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int Sc = S + h;
                if (Sc < KS8) {
                    const int Sn = Sc + 1 < KS8 ? Sc + 1 : KS8 - 1;     // past the end: re-read the last step (unused)
                    load_b((h + 1) & 1, Sn);
#pragma unroll
                    for (int c = 0; c < cdiv(GB, CH); ++c) {
                        if (c < NC) {
                            const int cur = (h * NC + c) & 1, nxt = cur ^ 1;
                            if (c + 1 < NC) load_a(nxt, Sc, c + 1);
                            else load_a(nxt, Sn, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int m = 0; m < CH; ++m) {
                                const int mg = c * CH + m;
                                if (mg < gn) {
                                    acc[g0 + mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bh[h & 1], acc[g0 + mg], 0, 0, 0);
                                    corr[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bl[h & 1], corr[mg], 0, 0, 0);
                                    corr[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][m], bh[h & 1], corr[mg], 0, 0, 0);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < GB; ++m)
            if (m < gn) acc[g0 + m] = acc[g0 + m] + kSplitInv * corr[m];
    }
}
// gemm_img_x3 with the table stream SHARED by the four waves of the workgroup.  With split products the matrix time of a long
// product fell 4.7x while every wave still pulled its own copy of the table from L2 (d = 500: 2.2 MB per wave and step, ~26 TB/s
// over the chip: the L2 -> CU stream became the bound).  Here a chunk of four output blocks (8 KiB: hi and lo of 4 x 16 rows x
// 32 features) is fetched once per workgroup -- wave w loads block 4 c + w -- written to a double-buffered LDS stage and read by
// all four waves (8 ds_read_b128 per 12 MFMAs); one workgroup barrier per chunk.  Every wave runs every chunk (surplus waves
// of the last workgroup run along on the last tile), so the barriers are uniform.
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding vector-memory load
// (s_waitcnt vmcnt(0)), i.e. for the table fetch just issued for the NEXT chunk -- one L2 latency per chunk
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int MB, int KS8, int LD = MB>
__device__ __forceinline__ void gemm_img_x3s(f32x4 (&acc)[MB], const float* __restrict__ tbl, const float* img, float* stage,
                                             int lane, int wave) {
    constexpr int CH = 4, NC = cdiv(MB, CH);
    constexpr int NG = cdiv(MB, 16), GC = cdiv(NC, NG);                 // groups of at most 16 output blocks (correction chain 64 regs)
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    const f16x8* imgp = reinterpret_cast<const f16x8*>(img) + lane;
    f16x8* st8 = reinterpret_cast<f16x8*>(stage);
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const int c0 = grp * GC;                                        // first chunk of the group
        const int cn = (c0 + GC <= NC) ? GC : NC - c0;                  // chunks in this group
        // Pipeline of unit u = (S, chunk), all indices static by the parity of u (an S-pair holds an even number of units):
        //   top of unit u - 3   table fetch of this wave's block (global -> registers lh / ll[(u) & 1])
        //   end of unit u - 2   published to LDS stage buffer u & 1, then the unit's barrier
        //   top of unit u - 1   all four blocks read back into ah / al[u & 1] -- in flight beside the MFMAs of unit u - 1
        //   unit u              12 MFMAs
        f16x8 lh[2], ll[2], ah[2][CH], al[2][CH], bh[2], bl[2];
        f32x4 corr[GC * CH];
#pragma unroll
        for (int m = 0; m < GC * CH; ++m) corr[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int r, int S, int c) __attribute__((always_inline)) {
            while (c >= cn) { c -= cn; S += 1; }
            if (S > KS8 - 1) S = KS8 - 1;                               // past the end: re-read the last step (unused)
            const int mb = (c0 + c) * CH + wave;
            if (mb < MB) {
                gptr8_t tp = sgpr_ptr8(tbl + ((size_t)S * LD + mb) * 512);
                lh[r] = tp[ul]; ll[r] = tp[64 + ul];
            }
        };
        auto publish = [&](int buf) __attribute__((always_inline)) {
            st8[(buf * 4 + wave) * 128 + lane] = lh[buf];
            st8[(buf * 4 + wave) * 128 + 64 + lane] = ll[buf];
        };
        auto readback = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                ah[buf][m] = st8[(buf * 4 + m) * 128 + lane];
                al[buf][m] = st8[(buf * 4 + m) * 128 + 64 + lane];
            }
        };
        bh[0] = imgp[0]; bl[0] = imgp[64];
        fetch(0, 0, 0);
        fetch(1, 0, 1);
        publish(0);
        publish(1);
        lds_barrier();
        readback(0);
        lds_barrier();                                                  // (unit 0 publishes into the buffer just read)
        fetch(0, 0, 2);
#pragma unroll 1
        for (int S = 0; S < KS8; S += 2) {                              // two S-steps per trip: static ring indices
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int Sc = S + h;
                if (Sc < KS8) {
                    const int Sn = Sc + 1 < KS8 ? Sc + 1 : KS8 - 1;     // past the end: re-read the last step (unused)
                    bh[(h + 1) & 1] = imgp[(Sn * 2) * 64]; bl[(h + 1) & 1] = imgp[(Sn * 2 + 1) * 64];
#pragma unroll
                    for (int c = 0; c < GC; ++c) {
                        if (c < cn) {
                            const int cur = (h * cn + c) & 1, nxt = cur ^ 1;
                            readback(nxt);                              // unit u + 1 (published during unit u - 1)
                            fetch(nxt, Sc, c + 3);                      // unit u + 3
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int m = 0; m < CH; ++m) {
                                const int mg = c * CH + m, mb = c0 * CH + mg;
                                if (mb < MB) {
                                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bh[h & 1], acc[mb], 0, 0, 0);
                                    corr[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bl[h & 1], corr[mg], 0, 0, 0);
                                    corr[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][m], bh[h & 1], corr[mg], 0, 0, 0);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            publish(cur);                               // unit u + 2 into the buffer unit u was read from
                            lds_barrier();
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < GC * CH; ++m)
            if (c0 * CH + m < MB) acc[c0 * CH + m] = acc[c0 * CH + m] + kSplitInv * corr[m];
    }
}
// acc[MB] += T . in, in: register panel of INB <= 4 blocks (the hidden layers: two S-steps), fully unrolled
template <int MB, int INB, int LD = MB>
__device__ __forceinline__ void gemm_regs_x3(f32x4 (&acc)[MB], const float* __restrict__ tbl, const f32x4 (&in)[INB], int lane) {
    constexpr int NS = (INB + 1) / 2;
    const unsigned ul = (unsigned)lane;
    tbl = opaque_base(tbl);
    f32x4 corr[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) corr[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < NS; ++S) {
        f16x8 ah[MB], al[MB];
#pragma unroll
        for (int m = 0; m < MB; m += 2) {
            gptr8_t tp = sgpr_ptr8(tbl + ((size_t)S * LD + m) * 512);
            ah[m] = tp[ul]; al[m] = tp[64 + ul];
            if (m + 1 < MB) { ah[m + 1] = tp[128 + ul]; al[m + 1] = tp[192 + ul]; }
        }
        f16x8 bh, bl;
        split_pack(in[2 * S], (2 * S + 1 < INB) ? in[(2 * S + 1 < INB) ? 2 * S + 1 : 0] : zero4, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bh, acc[mb], 0, 0, 0);
            corr[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bl, corr[mb], 0, 0, 0);
            corr[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], bh, corr[mb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = acc[mb] + kSplitInv * corr[mb];
}

// The four waves of a workgroup stream the SAME operand tables in the same order, each for its own tile.  A workgroup
// barrier in front of every long product keeps them within a few k-steps of one another, so that a table line one wave
// misses in the CU's vector L1 is a hit (or a merged in-flight miss) for the other three: the L2 -> L1 stream, which bounds
// these kernels, is then shared instead of fetched per wave.  Every wave runs every step (surplus waves of the last
// workgroup run along on the last tile), so the barriers are uniform.  -DPSP_WIDE_NOSYNC restores free-running waves (A/B).
// Measured (round 2, d = 200 and d = 500): no gain (6.94 vs 6.9 ms, 38.9 vs 37 ms) -- the kernels are not bound by a per-wave
// L2 -> L1 stream after all; the barriers stay available as -DPSP_WIDE_SYNC_ON for A/B runs.
#if defined(PSP_WIDE_SYNC_ON) && PSP_WIDE_SYNC_ON
#define PSP_WIDE_SYNC() __syncthreads()
#else
#define PSP_WIDE_SYNC()
#endif

// =======================================================================================
// Wide forward kernel: one wave = one 16-trajectory tile for all N steps (same per-step algebra, same
// reference lines as hjb_fwd_kernel)
// =======================================================================================
// LOGU: the u_L2 log of solver.py:491-494 (psp_hjb_config.u_ref); a separate instantiation because at d = 500 the kernel
// sits on the 512-register limit and two more live accumulators cost the ordinary path 8 % (measured)
// FAST: Philox noise and no time-feature table, decided at launch (no conditional loads and joins in the time loop; training)
// X3: every product as split f16 products (psp_hjb_config.mlp_dtype = PSP_MLP_F16X3; tables built by hjbw_tables_kernel(.., 3))
// SPEC (round 4): the problem switches of the LLGC configurations -- dense drift, dense sigma, adaptive process, no running cost,
// store_path 1, not the relative-entropy loss -- as compile-time constants (hjb_kernels.h, hjb_fwd_kernel FAST_ = 2): no scalar
// branches on them inside the time loop and its rolled k-loops
template <int D, int H, bool LOGU = false, bool FAST = false, bool X3 = false, bool SPEC = false>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void hjbw_fwd_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    const int k_drift = SPEC ? (int)DRIFT_DENSE : a.drift_kind, k_sigma = SPEC ? (int)SIGMA_DENSE : a.sigma_kind;
    const int k_run = SPEC ? (int)RUN_ZERO : a.runcost_kind, k_loss = SPEC ? (int)LOSS_LOGVAR : a.loss_kind;
    const int k_store = SPEC ? 1 : a.store_path;
    const bool k_adaptive = SPEC ? true : (a.adaptive != 0);
    const uint32_t iter_now = a.iter_dev ? *a.iter_dev : a.iter;       // wave-uniform scalar load
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;
    const float* __restrict__ T = a.tables;

    stage_vec(lds + W::vb1, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob1 + f] : 0.f; });
    stage_vec(lds + W::vw1t, HB, tid, nthr, [&](int f) { return f < H ? P[G::oW1 + f * (D + 1)] : 0.f; });
    stage_vec(lds + W::vb2, HB, tid, nthr, [&](int f) { return f < H ? P[G::ob2 + f] : 0.f; });
    stage_vec(lds + W::vb3, DB, tid, nthr, [&](int f) { return f < D ? P[G::ob3 + f] : 0.f; });
    stage_vec(lds + W::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (k_drift == DRIFT_DIAG || k_drift == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + W::vrun, DB, tid, nthr, [&](int f) {
        return (f < D && k_run == RUN_DIAGQ) ? a.runcost[f] : 0.f; });
    stage_vec(lds + W::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });
    __syncthreads();

    const int t16raw = blockIdx.x * nwave + wave;     // 16-trajectory tile owned by this wave
    const bool wave_valid = t16raw < a.ntile16;       // surplus waves of the last workgroup run along on the last tile
    const int t16 = wave_valid ? t16raw : a.ntile16 - 1;   // (no divergent control flow around the rolled products) but store nothing
    const int k = t16 * 16 + j;
    const bool kvalid = wave_valid && k < a.K_local;
    const uint32_t kglob = (uint32_t)(a.k_offset + k);
    const float dt = a.dt, sqdt = a.sqdt;
    float* img = lds + W::fImg + wave * (X3 ? W::IMGX : W::IMG);       // this wave's input image [KP][64] (X3: hi / lo packs)
    [[maybe_unused]] f16x8* img8 = reinterpret_cast<f16x8*>(img) + lane;
    // table stream of the long products shared through LDS by the four waves (gemm_img_x3s; GeoW::kShare): d = 200 iteration
    // 7.08 -> 5.96 ms; the other instances keep per-wave streams (gemm_img_x3)
    constexpr bool kShare = W::kShare;
    [[maybe_unused]] const f32x4 zero4x = {0.f, 0.f, 0.f, 0.f};
    const bool store_path = k_store && wave_valid;
    const float store_cxi = (k_store == 3) ? 0.f : 1.f;
    const float store_cz = (k_store == 3) ? 1.f : (k_store == 2 ? -a.sqdt : (k_adaptive ? 0.f : a.sqdt));

    double sD = 0.0, sD2 = 0.0;
    {
        const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;         // index by block * 4
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
        [[maybe_unused]] float ULsum = 0.f;

#ifdef PSP_STAMPS
        unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#pragma unroll 1
        for (int n = 0; n < a.N; ++n) {
            PSP_STAMP(ws0);
            float tn = (float)n * dt;
            if constexpr (!FAST) { if (a.tfeat) tn = a.tfeat[n]; }
            const f32x4* vecs = opaque(vecs0);         // re-read the small vectors each step (no hoisting)
            const int qn = opaque_i(q);                // ... and rebuild the per-block Philox counters (else 3 registers
                                                       // per state block are hoisted out of the step loop)
            const f32x4* vb1 = vecs + W::vb1 / 4;
            const f32x4* vw1t = vecs + W::vw1t / 4;
            const f32x4* vb2 = vecs + W::vb2 / 4;
            const f32x4* vb3 = vecs + W::vb3 / 4;
            const f32x4* vdr = vecs + W::vdr / 4;
            const f32x4* vrun = vecs + W::vrun / 4;
            // path block of (n, tile): wave-uniform base in SGPRs, stores are "base + lane + immediate"
            typedef __attribute__((address_space(1))) float* gwptr_t;
            auto pbase = [&](int ofs) __attribute__((always_inline)) {
                return (gwptr_t)sgpr_block_addr(a.path, (unsigned long long)n * a.ntile16 + t16, (unsigned)G::PB, (unsigned)ofs);
            };
            const unsigned ul = (unsigned)lane;
            // X_n: LDS image (B operand of the W1 and drift products) and path store
            if constexpr (X3) {
#pragma unroll
                for (int S = 0; S < W::KS8; ++S) {
                    f16x8 ph, pl;
                    split_pack(X[2 * S], (2 * S + 1 < DB) ? X[(2 * S + 1 < DB) ? 2 * S + 1 : 0] : zero4x, ph, pl);
                    img8[(2 * S) * 64] = ph; img8[(2 * S + 1) * 64] = pl;
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KP; ++ks) img[ks * 64 + lane] = X[ks >> 2][ks & 3];
            }
            if (store_path) {
#pragma unroll
                for (int g = 0; g < KP / 16 + 1; ++g) {               // fresh SGPR base every 4 KiB: immediates stay < 4096
                    gwptr_t px = pbase(G::pX + g * 16 * 64);
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (g * 16 + e < KP) PSP_PATH_STORE(px + e * 64 + ul, X[(g * 16 + e) >> 2][(g * 16 + e) & 3]);
                }
            }
            PSP_STAMP(ws1);
            // ---- control net (function_space.py:190-195)
            f32x4 h1[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) h1[m] = vb1[m * 4] + tn * vw1t[m * 4];
            PSP_WIDE_SYNC();
            if constexpr (X3 && kShare) gemm_img_x3s<HB, W::KS8>(h1, T + W::xW1, img, lds + W::fStage, lane, wave);
            else if constexpr (X3) gemm_img_x3<HB, W::KS8>(h1, T + W::xW1, img, lane);
            else gemm_img<HB, KP>(h1, T + W::tW1, img, lane);
            PSP_STAMP(ws2);
            // ---- X_{n+1} = X + b(X) dt + sigma v (solver.py:471-472): the drift part now, while the image still holds X_n
            if (k_drift == DRIFT_DENSE) {
                PSP_WIDE_SYNC();
                if constexpr (X3 && kShare) gemm_img_x3s<DB, W::KS8>(X, T + W::xA, img, lds + W::fStage, lane, wave);
                else if constexpr (X3) gemm_img_x3<DB, W::KS8>(X, T + W::xA, img, lane);
                else gemm_img<DB, KP>(X, T + W::tA, img, lane);                  // X += (dt A) X_n
            } else if (k_drift == DRIFT_DIAG) {
#pragma unroll
                for (int b = 0; b < DB; ++b) X[b] += dt * (vdr[b * 4] * X[b]);
            } else if (k_drift == DRIFT_DWELL) {
#pragma unroll
                for (int b = 0; b < DB; ++b) X[b] -= dt * (4.0f * vdr[b * 4] * (X[b] * (X[b] * X[b] - 1.0f)));
            }
            PSP_STAMP(ws3);
#pragma unroll
            for (int m = 0; m < HB; ++m) h1[m] = tanh4(h1[m]);
            f32x4 h2[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) h2[m] = vb2[m * 4];
            if constexpr (X3) gemm_regs_x3<HB, HB>(h2, T + W::xW2, h1, lane);
            else gemm_regs<HB, 16, HB>(h2, T + W::tW2, h1, lane);
#pragma unroll
            for (int m = 0; m < HB; ++m) h2[m] = tanh4(h2[m]);
            if (store_path) {
                gwptr_t ph1 = pbase(G::pH1), ph2 = pbase(G::pH2);
#pragma unroll
                for (int ks = 0; ks < 4 * HB; ++ks) {
                    PSP_PATH_STORE(ph1 + ks * 64 + ul, h1[ks >> 2][ks & 3]);
                    PSP_PATH_STORE(ph2 + ks * 64 + ul, h2[ks >> 2][ks & 3]);
                }
            }
            PSP_STAMP(ws4);
            // ---- control output, Brownian increment and increment panel v, four state blocks at a time, so that Z is
            //      never live as a whole: Z_g = W3[g] h2 + b3 -> row sums |Z|^2, Z.xi (solver.py:477-478) ->
            //      v = c dt + xi sqrt(dt) (c = -Z if adaptive, solver.py:451-456) -> LDS image (dense sigma) or X
            float S = 0.f, Pz = 0.f;
            [[maybe_unused]] float UL = 0.f;
            auto z_group = [&](auto nbc, int g) __attribute__((always_inline)) {
                constexpr int NB = decltype(nbc)::value;
                f32x4 Zg[NB];
#pragma unroll
                for (int m = 0; m < NB; ++m) Zg[m] = vb3[(4 * g + m) * 4];
                if constexpr (X3) gemm_regs_x3<NB, HB, DB>(Zg, T + W::xW3 + 4 * g * 512, h2, lane);
                else gemm_regs<NB, 16, HB, DB>(Zg, T + W::tW3 + 4 * g * 64, h2, lane);
                [[maybe_unused]] f32x4 vg[4] = {zero4x, zero4x, zero4x, zero4x};
#pragma unroll
                for (int m = 0; m < NB; ++m) {
                    const int b = 4 * g + m;
                    f32x4 xi;
                    if (FAST || a.noise_mode == NOISE_PHILOX) {
                        xi = philox_block(kglob, (uint32_t)n, (uint32_t)(4 * b + qn), iter_now, a.seed_lo, a.seed_hi);
                    } else {
                        const float* xrow = a.xi + ((size_t)(n + 1) * a.K_local + (kvalid ? k : 0)) * D;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * b + 4 * r + q;
                            const float v = xrow[f < D ? f : D - 1];
                            xi[r] = (f < D && kvalid) ? v : 0.f;
                        }
                    }
                    if (16 * b + 16 > D) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (16 * b + 4 * r + q >= D) xi[r] = 0.f;
                    }
                    if (store_path) {                  // image in the xi slot: c_xi xi + c_z Z (see hjb_fwd_kernel)
                        gwptr_t pxi = pbase(G::pXi + b * 256);
                        const f32x4 wv = store_cxi * xi + store_cz * Zg[m];
#pragma unroll
                        for (int r = 0; r < 4; ++r) PSP_PATH_STORE(pxi + r * 64 + ul, wv[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S = fmaf(Zg[m][r], Zg[m][r], S);
                        Pz = fmaf(Zg[m][r], xi[r], Pz);
                    }
                    if constexpr (LOGU) {              // u_L2 logging: |-Z_n - u*(t_n)|^2 (solver.py:491-494); the table row
                        gptr_t ur = sgpr_ptr(a.uref + (size_t)n * D + 16 * b);      // base stays in SGPRs (the host pads the
#pragma unroll                                                              // table by 16 floats: no clamped index)
                        for (int r = 0; r < 4; ++r) {
                            float e = Zg[m][r] + ur[4 * r + q];
                            if (16 * b + 16 > D) e = (16 * b + 4 * r + q < D) ? e : 0.f;
                            UL = fmaf(e, e, UL);
                        }
                    }
                    const f32x4 v = k_adaptive ? (sqdt * xi - dt * Zg[m]) : (sqdt * xi);
                    if (k_sigma == SIGMA_DENSE) {
                        if constexpr (X3) vg[m] = v;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) img[(4 * b + r) * 64 + lane] = v[r];
                        }
                    } else if (k_sigma == SIGMA_SCALE) {
                        X[b] += a.sigma_scale * v;
                    } else {
                        X[b] += v;
                    }
                }
                if constexpr (X3) {                    // increment panel of this group as hi / lo packs (two S-steps per group)
                    if (k_sigma == SIGMA_DENSE) {
#pragma unroll
                        for (int s2 = 0; s2 < (NB + 1) / 2; ++s2) {
                            f16x8 ph, pl;
                            split_pack(vg[2 * s2], vg[2 * s2 + 1], ph, pl);
                            img8[(2 * (2 * g + s2)) * 64] = ph; img8[(2 * (2 * g + s2) + 1) * 64] = pl;
                        }
                    }
                }
            };
#pragma unroll
            for (int g = 0; g < DB / 4; ++g) z_group(std::integral_constant<int, 4>{}, g);
            if constexpr (DB % 4 != 0) z_group(std::integral_constant<int, DB % 4>{}, DB / 4);
            S = qsum(S);
            Pz = qsum(Pz);
            if constexpr (LOGU) ULsum = fmaf(UL, dt, ULsum);
            PSP_STAMP(ws5);
            if (k_sigma == SIGMA_DENSE) {
                PSP_WIDE_SYNC();
                if constexpr (X3 && kShare) gemm_img_x3s<DB, W::KS8>(X, T + W::xB, img, lds + W::fStage, lane, wave);
                else if constexpr (X3) gemm_img_x3<DB, W::KS8>(X, T + W::xB, img, lane);
                else gemm_img<DB, KP>(X, T + W::tB, img, lane); // X += B v
            }

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
                Y = Y - (0.5f * S + fX) * dt;           // Y carries -Zsum (hjb_fwd_kernel)
            } else {
                const float drift_y = k_adaptive ? (fX - 0.5f * S) : (fX + 0.5f * S);
                Y = Y + drift_y * dt + Pz * sqdt;
            }
            Fsum = fmaf(fX, dt, Fsum);
            PSP_STAMP(ws6);
            PSP_ACC(0, ws1, ws0);   // X image + path store
            PSP_ACC(1, ws2, ws1);   // W1 product
            PSP_ACC(2, ws3, ws2);   // drift product
            PSP_ACC(3, ws4, ws3);   // tanh, W2, tanh, h stores
            PSP_ACC(4, ws5, ws4);   // Z groups: W3 product, Philox, xi store, v
            PSP_ACC(5, ws6, ws5);   // sigma product, running cost, Y
            PSP_ACC(6, ws6, ws0);
        }
#ifdef PSP_STAMPS
        if (a.dbg && lane == 0 && wave_valid) {
            stamps[7] = (unsigned long long)a.N;
            for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
        }
#endif

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
        if constexpr (LOGU) {
            const float ULt = qsum(ULsum);
            if (kvalid && q == 0) a.ul2[k] = ULt;
        }
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
// Wide backward kernel (adaptive forward process): analytic parameter gradient, streaming over the state blocks.
// Workgroup = 4 waves (one per SIMD), persistent over rounds of 4 sample blocks:
//   phase A  wave w, block 4 round + w:  dz2 = (W3^T G)(1 - h2^2), G = w sqrt(dt) xi, as a rolled k-loop over the
//            stored xi image (one coalesced dword per lane and k-step) and the W3^T table; dz2 image -> LDS
//   phase B  wave w owns hidden block ib = w of every tile row; for each of the 4 blocks:
//            dz1 tile = (dz2^T W2[:, ib])(1 - h1^2)   (transposed product: MFMA output = operand layout)
//            dW2[:, ib] += dz2^T h1;  then for every state block ob:
//            G tile = xi tile * (w sqrt(dt)) straight from the xi image in feature-on-lane form,
//            dW3[ob, ib] += G^T h2,  dW1[ib, ob] += dz1^T X_n
//   biases: db2 from the phase-A panels, db1 / time column from the dz1 tiles, db3 from the G tiles (the ob range is
//   split over the four waves).  One barrier per round (the dz2 exchange is double-buffered).
// =======================================================================================
template <int D, int H>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void hjbw_bwd_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP, EXB = W::EXB;
    constexpr int OBW = cdiv(DB, 4);                  // state blocks per wave for the db3 sums
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;
    const float* __restrict__ T = a.tables;           // W3^T, k-step-major
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned lofsU = (unsigned)image_lane_offset_F(lane);

    const double invK = 1.0 / (double)a.K_global;
    const float meanD = (a.loss_kind == LOSS_LOGVAR) ? (float)(a.sums[0] * invK) : 0.f;
    const float coef = (float)(2.0 * invK);
    const float sqdt = a.sqdt, dt = a.dt;
    const long long nblk = (long long)a.N * a.ntile16;
    const long long nround = (nblk + 3) / 4;

    // persistent accumulators of this wave's tile row(s)
    f32x4 acc3[DB], acc1[DB], acc2[HB];
#pragma unroll
    for (int b = 0; b < DB; ++b) { acc3[b] = zero4; acc1[b] = zero4; }
#pragma unroll
    for (int m = 0; m < HB; ++m) acc2[m] = zero4;
    f32x4 bs1 = zero4, bt1 = zero4;
    float bs3[OBW], bs2 = 0.f;                        // bs2: db2 of hidden block `wave`, from the exchange tiles in phase B
#pragma unroll
    for (int i = 0; i < OBW; ++i) bs3[i] = 0.f;

    int par = 0;
#pragma unroll 1
    for (long long round = blockIdx.x; round < nround; round += gridDim.x, par ^= 1) {
        float* exch = lds + par * 4 * EXB;
        // ------------------------------------------------------------------ phase A: own block
        {
            const long long blk0 = round * 4 + wave;
            const bool bvalid = blk0 < nblk;
            const long long blk = bvalid ? blk0 : nblk - 1;
            const int t16 = (int)(blk % a.ntile16);
            const int k = t16 * 16 + j;
            const bool kvalid = bvalid && k < a.K_local;
            const float dk = a.D[kvalid ? k : 0];
            const float wk = kvalid ? (a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)) : 0.f;
            const float wks = wk * sqdt;
            const float* pb = a.path + (size_t)blk * (size_t)G::PB + lane;
            // (An L2 touch-prefetch of the images phase B streams, and of the next round's xi image, was tried in round 2 and
            //  made both kernels slower -- d=200 3.52 -> 3.73 ms, d=500 8.0 -> 8.8 ms: a whole round of d=500 images per CU is
            //  the size of the XCD's L2, the touched lines are evicted before they are used.)
            // under the 256-register cap of the d <= 256 instances (two workgroups per CU) the persistent accumulators leave
            // this phase ~90 registers: a two-stage operand ring and h2 loaded AFTER the k-loop instead of before it keep the
            // phase inside the budget (round 1: three stages + early h2 = 54 spilled registers, 1.65 GB of scratch traffic
            // per launch at d = 200)
            constexpr bool TIGHT = (D <= 256);
            constexpr int NSTG = TIGHT ? 2 : 3;
            f32x4 h2[HB];
            if constexpr (!TIGHT) {
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[m][r] = pb[G::pH2 + (4 * m + r) * 64];
            }
            f32x4 dz2[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) dz2[m] = zero4;
            // rolled k-loop, 4 k-steps per iteration, operands of the next iteration in flight
            const float* xip = pb + G::pXi;
            const unsigned ul = (unsigned)lane;
            float xb[NSTG][4], ab[NSTG][4 * HB];
            auto load = [&](int st, int ks0) __attribute__((always_inline)) {
                gptr_t tp = sgpr_ptr(T + (size_t)ks0 * (HB * 64));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xb[st][u] = xip[(size_t)(ks0 + u) * 64];
#pragma unroll
                    for (int m = 0; m < HB; ++m) ab[st][u * HB + m] = tp[(u * HB + m) * 64 + ul];
                }
            };
            auto fma_stage = [&](int st) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float g = wks * xb[st][u];
#pragma unroll
                    for (int m = 0; m < HB; ++m) dz2[m] = mfma16(ab[st][u * HB + m], g, dz2[m]);
                }
            };
            // fully unrolled, three register stages of 4 k-steps: the operands of stage g + 2 are requested before the
            // MFMAs of stage g issue (two stages = 32 MFMAs = ~1 k cycles of lead for the L2-resident table / xi image)
            constexpr int NG = KP / 4;
            load(0, 0);
            if (NSTG > 2 && NG > 1) load(1, 4);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + NSTG - 1 < NG) load((g + NSTG - 1) % NSTG, 4 * (g + NSTG - 1));
                __builtin_amdgcn_sched_barrier(0);
                fma_stage(g % NSTG);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (TIGHT) {
#pragma unroll
                for (int m = 0; m < HB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[m][r] = pb[G::pH2 + (4 * m + r) * 64];
            }
            float* ex = exch + wave * EXB + lane;
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                dz2[m] = dz2[m] * (1.0f - h2[m] * h2[m]);
#pragma unroll
                for (int r = 0; r < 4; ++r) ex[(4 * m + r) * 64] = dz2[m][r];
            }
        }
        __syncthreads();
        // ------------------------------------------------------------------ phase B: hidden block ib = wave
#pragma unroll 1
        for (int sb = 0; sb < 4; ++sb) {
            const long long c0 = round * 4 + sb;
            if (c0 >= nblk) break;                                    // wave-uniform
            const int cb = __builtin_amdgcn_readfirstlane((int)c0);
            const int n = cb / a.ntile16, t16 = cb % a.ntile16;
            const float* bp = a.path + (size_t)cb * (size_t)G::PB;   // wave-uniform block base
            const float* ex = exch + sb * EXB;
            // per-lane weights of samples 4 q .. 4 q + 3 (feature-on-lane tiles hold 4 samples per lane)
            f32x4 w4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = t16 * 16 + 4 * q + r;
                const float dk = a.D[kk < a.K_local ? kk : 0];
                w4[r] = (kk < a.K_local) ? (a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)) * sqdt : 0.f;
            }
            const f32x4 h2t = *reinterpret_cast<const f32x4*>(bp + G::pH2 + wave * 256 + lofsU);
            const f32x4 h1t = *reinterpret_cast<const f32x4*>(bp + G::pH1 + wave * 256 + lofsU);
            // dz1 tile of hidden block ib: (dz2^T W2[:, ib]) (1 - h1^2).  The B operands W2[4 ks + q][16 ib + n] are re-read per
            // sample block (16 cache-resident loads) so that they are live neither through phase A nor through the state-block
            // stream below: under the 256-register cap every persistent register is a spill somewhere else
            f32x4 dzt = zero4;
            {
                float w2b[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const int o = 4 * ks + q, i = 16 * wave + j;
                    w2b[ks] = (o < H && i < H) ? P[G::oW2 + opaque_i(0) + o * H + i] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) dzt = mfma16(ex[ks * 64 + lane], w2b[ks], dzt);
            }
            // dW2[:, ib] += dz2^T h1
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                const f32x4 a2 = tile_get(ex + m * 256, lane);
                if (m == wave) bs2 += hsum4(a2);      // db2[16 m + (lane & 15)]: this lane's four samples of the dz2 tile
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[m] = mfma16(a2[r], h1t[r], acc2[m]);
            }
            const f32x4 a1 = dzt * (1.0f - h1t * h1t);
            bs1 += a1;
            bt1 += ((float)n * dt) * a1;
            // stream over the state blocks: xi tile -> G tile -> dW3, X tile -> dW1 (tiles of block ob + 2 in flight)
            const float* xib = bp + G::pXi + lofsU;
            const float* xb = bp + G::pX + lofsU;
            constexpr int RD = (D <= 208) ? 5 : (D <= 256 ? 3 : 10);                    // ring depth: RD - 1 tiles ahead (6: ~1.3 k cycles; 4 under the register cap)
            f32x4 xit[RD], xt[RD];
#pragma unroll
            for (int i = 0; i < RD - 1; ++i) {
                if (i < DB) {
                    xit[i] = *reinterpret_cast<const f32x4*>(xib + i * 256);
                    xt[i] = *reinterpret_cast<const f32x4*>(xb + i * 256);
                }
            }
#pragma unroll
            for (int ob = 0; ob < DB; ++ob) {
                if (ob + RD - 1 < DB) {
                    xit[(ob + RD - 1) % RD] = *reinterpret_cast<const f32x4*>(xib + (ob + RD - 1) * 256);
                    xt[(ob + RD - 1) % RD] = *reinterpret_cast<const f32x4*>(xb + (ob + RD - 1) * 256);
                }
                const f32x4 g = xit[ob % RD] * w4;
                if (ob / OBW == wave) bs3[ob % OBW] += hsum4(g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc3[ob] = mfma16(g[r], h2t[r], acc3[ob]);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[ob] = mfma16(a1[r], xt[ob % RD][r], acc1[ob]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();

    // ---- write-out: tiles are disjoint between waves; D tile (row block, col block): lane (col = l & 15, qq = l >> 4),
    //      reg rr <-> [16 rowblk + 4 qq + rr][16 colblk + col]
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
    const int col = lane & 15, qq = lane >> 4, ib = wave;
#pragma unroll
    for (int ob = 0; ob < DB; ++ob)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o3 = 16 * ob + 4 * qq + rr, i3 = 16 * ib + col;
            if (o3 < D && i3 < H) gp[G::oW3 + o3 * H + i3] = acc3[ob][rr];
            const int o1 = 16 * ib + 4 * qq + rr, i1 = 16 * ob + col;
            if (o1 < H && i1 < D) gp[G::oW1 + o1 * (D + 1) + 1 + i1] = acc1[ob][rr];
        }
#pragma unroll
    for (int m = 0; m < HB; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o2 = 16 * m + 4 * qq + rr, i2 = 16 * ib + col;
            if (o2 < H && i2 < H) gp[G::oW2 + o2 * H + i2] = acc2[m][rr];
        }
    {
        const float v1 = qsum(hsum4(bs1)), vt = qsum(hsum4(bt1));
        const int f = 16 * ib + col;
        if (qq == 0 && f < H) { gp[G::ob1 + f] = v1; gp[G::oW1 + f * (D + 1)] = vt; }
    }
#pragma unroll
    for (int i = 0; i < OBW; ++i) {
        const int ob = wave * OBW + i;
        const float v = qsum(bs3[i]);
        const int f = 16 * ob + col;
        if (qq == 0 && ob < DB && f < D) gp[G::ob3 + f] = v;
    }
    // db2 of hidden block `wave`: lane (col, qq) holds the sum over its sample quarters; fixed-order sum over qq
    {
        const float v2 = qsum(bs2);
        const int f = 16 * ib + col;
        if (qq == 0 && f < H) gp[G::ob2 + f] = v2;
    }
}

// =======================================================================================
// Wide adjoint sweep (gradients through the state path; recursion and path-store protocol of hjba_kernels.h).
// One wave = one 16-trajectory tile, backwards in time: lambda' goes to the wave's LDS image, B^T lambda' and
// (dt A)^T lambda' are rolled k-loops over it (the second accumulates into lambda in place), gZ replaces the image,
// W3^T gZ is the third rolled product, W2^T and W1x^T are short unrolled ones.
// =======================================================================================
// X3: the five products as split f16 products (tables of hjbw_tables_kernel(.., 4), image as hi / lo packs); the trajectory weights
// (mu, nu, wT ~ 1 / K) are scaled per wave by a power of two and the image written back is scaled back, as in hjb_adj_kernel<.., X3>
template <int D, int H, bool X3 = false>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void hjbw_adj_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    constexpr int DB = W::DB, HB = W::HB, KP = W::KP;
    [[maybe_unused]] const f32x4 zero4x = {0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ T = a.tables;

    stage_vec(lds + W::vdr, DB, tid, nthr, [&](int f) {
        return (f < D && (a.drift_kind == DRIFT_DIAG || a.drift_kind == DRIFT_DWELL)) ? a.drift[f] : 0.f; });
    stage_vec(lds + W::vrun, DB, tid, nthr, [&](int f) {
        return (f < D && a.runcost_kind == RUN_DIAGQ) ? a.runcost[f] : 0.f; });
    stage_vec(lds + W::vterm, DB, tid, nthr, [&](int f) { return f < D ? a.term[f] : 0.f; });
    __syncthreads();

    const int t16raw = blockIdx.x * nwave + wave;
    if (t16raw >= a.ntile16) return;                   // no workgroup barriers below
    const int t16 = t16raw;
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
    const float coefW = (a.store_path == 3) ? nu * dt : mu * sqdt;
    const float wf = (mu + nu) * dt;
    const float wT = wT_in;                                                     // weight of grad g(X_N) in lambda_N
    float* img = lds + W::fImg + wave * (X3 ? W::IMGX : W::IMG);       // this wave's image (X3: hi / lo packs)
    [[maybe_unused]] f16x8* img8 = reinterpret_cast<f16x8*>(img) + lane;
    const f32x4* vecs0 = reinterpret_cast<const f32x4*>(lds) + q;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 lam[DB];                                     // lambda_N = (nu - mu) grad g(X_N)
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
        // path blocks through wave-uniform SGPR bases (loads / stores are "base + lane * 4 + immediate < 4 KiB")
        typedef __attribute__((address_space(1))) float* gwptr_t;
        auto pbase = [&](int nn, int ofs) __attribute__((always_inline)) {
            return (gwptr_t)sgpr_block_addr(a.path, (unsigned long long)nn * a.ntile16 + t16, (unsigned)G::PB, (unsigned)ofs);
        };
        const unsigned ul = (unsigned)lane;
        // lambda' = lambda_{n+1} + (mu + nu) dt grad f(X_{n+1});  X_{n+1} from the next path block (or X_N)
        if (a.runcost_kind == RUN_DIAGQ) {
            const int nx = n + 1 < a.N ? n + 1 : n;
#pragma unroll
            for (int b = 0; b < DB; ++b) {
                gwptr_t px = pbase(nx, G::pX + b * 256);
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
            if constexpr (X3) gemm_img_x3<DB, W::KS8>(qv, T + W::xaBT, img, lane);
            else gemm_img<DB, KP>(qv, T + W::aBT, img, lane);
        } else if (a.sigma_kind == SIGMA_SCALE) {
#pragma unroll
            for (int b = 0; b < DB; ++b) qv[b] = a.sigma_scale * lam[b];
        } else {
#pragma unroll
            for (int b = 0; b < DB; ++b) qv[b] = lam[b];
        }
        // lambda += dt b'(X_n)^T lambda'   (in place; the image still holds lambda')
        if (a.drift_kind == DRIFT_DENSE) {
            if constexpr (X3) gemm_img_x3<DB, W::KS8>(lam, T + W::xaAT, img, lane);
            else gemm_img<DB, KP>(lam, T + W::aAT, img, lane);
        } else if (a.drift_kind == DRIFT_DIAG) {
#pragma unroll
            for (int b = 0; b < DB; ++b) lam[b] += dt * (vdr[b * 4] * lam[b]);
        } else if (a.drift_kind == DRIFT_DWELL) {
#pragma unroll
            for (int b = 0; b < DB; ++b) {
                gwptr_t px = pbase(n, G::pX + b * 256);
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = px[r * 64 + ul];
                lam[b] -= dt * (4.0f * vdr[b * 4] * ((3.0f * x * x - 1.0f) * lam[b]));
            }
        }
        // gZ_n: back into the xi slot (as gZ / sqrt(dt)) and into the image (B operand of the W3^T product)
        [[maybe_unused]] f32x4 gzp = zero4x;           // X3: gZ of the even block of a pair, until its odd partner is formed
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            gwptr_t pw = pbase(n, G::pXi + b * 256);
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
        f32x4 dz2[HB], dz1[HB];
        {
            f32x4 h2[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[m][r] = pbase(n, G::pH2)[(4 * m + r) * 64 + ul];
#pragma unroll
            for (int m = 0; m < HB; ++m) dz2[m] = zero4;
            if constexpr (X3) gemm_img_x3<HB, W::KS8>(dz2, T + W::xaW3T, img, lane);
            else gemm_img<HB, KP>(dz2, T + W::aW3T, img, lane);
#pragma unroll
            for (int m = 0; m < HB; ++m) dz2[m] = dz2[m] * (1.0f - h2[m] * h2[m]);
        }
        {
            f32x4 h1[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[m][r] = pbase(n, G::pH1)[(4 * m + r) * 64 + ul];
#pragma unroll
            for (int m = 0; m < HB; ++m) dz1[m] = zero4;
            if constexpr (X3) gemm_regs_x3<HB, HB>(dz1, T + W::xaW2T, dz2, lane);
            else gemm_regs<HB, 16, HB>(dz1, T + W::aW2T, dz2, lane);
#pragma unroll
            for (int m = 0; m < HB; ++m) dz1[m] = dz1[m] * (1.0f - h1[m] * h1[m]);
        }
        if constexpr (X3) gemm_regs_x3<DB, HB>(lam, T + W::xaW1T, dz1, lane);
        else gemm_regs<DB, 16, HB>(lam, T + W::aW1T, dz1, lane);           // lambda_n += W1x^T dz1
    }
}

// =======================================================================================
// Role-specialised backward for the wide family, d <= 256 (the design of hjb_bwd2_kernel; round 2).
// hjbw_bwd_kernel lets every wave run every phase (two 4-wave workgroups per CU): at d = 200 it reaches 57 % of the
// matrix-pipe time of its 752 MFMAs per tile-step, as hjb_bwd_kernel did at d = 100 before it was split into roles.  Here
// one 8-wave workgroup per CU; rounds of four sample blocks; one barrier per round; double-buffered exchange:
//   * producers (waves 0-3), one block each:  G = w sqrt(dt) xi from the stored image (T layout, next round's image
//     requested a round ahead),  dz2 = (W3^T G)(1 - h2^2) as a register-chained product against the W3^T table in LDS;
//     the dz2 k-step image and the block's 16 trajectory weights go to the exchange buffer of the NEXT round; db3 / db2 are
//     element-wise running sums.  G itself is NOT exchanged (the panels of a d = 200 round would be 100 KB of LDS):
//   * consumers (waves 4-7), wave ib owns hidden block ib of every weight-gradient tile row (8 DB + 16 accumulator
//     registers): per block  dz1 = (dz2^T W2[:, ib])(1 - h1^2)  as the transposed product (W2's block in 16 registers),
//     dW3 += xi^T (w h2)  with the xi tiles read feature-on-lane straight from the path store -- the sample weight moves
//     to the h2 operand, four multiplies per block --,  dW2 += dz2^T h1,  dW1 += dz1^T X_n.  The 2 DB streamed tiles of a
//     block (xi, then X_n) run through ONE register ring that continues across block and round boundaries (the store is an
//     input: the addresses of the next round are known), requested RD - 1 tiles before their four MFMAs issue.
// Same flush layout and reduction as hjb_bwd2_kernel; parity: tests/test_gpu_wide_family.py, test_gpu_full_size.py.
// =======================================================================================
template <int D, int H>
struct GeoB2 {
    using G = Geo<D, H>;
    static constexpr int DB = G::DB, HB = G::HB, KSD = G::KSD, KSH = G::KSH;
    static constexpr int oWts = 4 * HB * 64, EXQ = oWts + 64;           // per block: dz2 k-step image, 16 trajectory weights
    static constexpr int bufs = HB * KSD * 64, lds_floats = bufs + 2 * 4 * EXQ;
    static constexpr int RD = 8;                                        // ring depth (streamed 16-feature tiles in flight); a round is
    static_assert((8 * DB) % RD == 0, "ring slots must line up across rounds");   // 8 DB tiles, so slot = tile % RD carries over
    static constexpr int RS = 16 * DB + 16 * HB;                        // per-producer bias-sum slots
    static_assert(4 * RS <= 2 * 4 * EXQ, "bias sums reuse the exchange area");
};

template <int D, int H>
__global__ __launch_bounds__(512) void hjbw_bwd2_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    using G = Geo<D, H>;
    using B2 = GeoB2<D, H>;
    constexpr int DB = B2::DB, HB = B2::HB, KSD = B2::KSD, KSH = B2::KSH, EXQ = B2::EXQ, RD = B2::RD, RS = B2::RS;
    static_assert(HB == 4, "one consumer wave per hidden block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const bool producer = wave < 4;
    const int sub = wave & 3;
    const float* __restrict__ P = a.params;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    stage_aop(lds, HB, KSD, tid, nthr, [&](int row, int col) {          // W3^T as A-operand table (producers)
        return (row < H && col < D) ? P[G::oW3 + col * H + row] : 0.f; });
    __syncthreads();
    float* bufs = lds + B2::bufs;                     // [2 buffers][4 blocks][EXQ]

    const double invK = 1.0 / (double)a.K_global;
    const float meanD = (a.loss_kind == LOSS_LOGVAR) ? (float)(a.sums[0] * invK) : 0.f;
    const float coef = (float)(2.0 * invK);
    const float sqdt = a.sqdt, dt = a.dt;
    const long long nblk = (long long)a.N * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    const int R = (int)((nround - blockIdx.x + gridDim.x - 1) / gridDim.x);   // rounds of this workgroup (>= 1)

    if (producer) {
        // ================================================================================ producers
        f32x4 sG[DB], sZ2[HB];
#pragma unroll
        for (int b = 0; b < DB; ++b) sG[b] = zero4;
#pragma unroll
        for (int m = 0; m < HB; ++m) sZ2[m] = zero4;
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
        for (int it = 0; it <= R; ++it) {
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
                const float wks = wk * sqdt;
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
                {
                    const long long n0 = own_block(it + 1);
                    const long long nblk1 = n0 >= 0 ? n0 : nblk - 1;
                    const float* pn = a.path + (size_t)nblk1 * (size_t)G::PB + lane;
                    const int k1 = (int)(nblk1 % a.ntile16) * 16 + j;
                    dkn = a.D[k1 < a.K_local ? k1 : 0];
#pragma unroll
                    for (int b = 0; b < DB; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xin[b][r] = pn[G::pXi + (4 * b + r) * 64];
                }
                if (q == 0) ex[B2::oWts + j] = wks;   // the consumers weight their h2 operand with it
                f32x4 dz2[HB];
#pragma unroll
                for (int m = 0; m < HB; ++m) dz2[m] = zero4;
                gemm_T<HB, KSD, DB>(dz2, lds, Gt, lane);
#pragma unroll
                for (int m = 0; m < HB; ++m) { dz2[m] = dz2[m] * (1.0f - h2[m] * h2[m]); sZ2[m] += dz2[m]; }
#pragma unroll
                for (int ks = 0; ks < 4 * HB; ++ks) ex[ks * 64 + lane] = dz2[ks >> 2][ks & 3];
            }
            __syncthreads();                              // swap the exchange buffers (pairs with the consumer loop)
        }
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
                const float v2 = jsumf(sZ2[m][r]);
                if (j == 0) red[16 * DB + 16 * m + 4 * r + q] = v2;
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
    f32x4 bs1 = zero4, bt1 = zero4;                       // element-wise partial sums of the dz1 tiles (db1, time column)
    float w2b[KSH];                                       // B operands of the dz1 product: W2[4 ks + q][16 ib + j]
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) {
        const int o = 4 * ks + q, i = 16 * ib + j;
        w2b[ks] = (o < H && i < H) ? P[G::oW2 + o * H + i] : 0.f;
    }
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
    constexpr int NI = 2 * DB;                            // streamed tiles of a block: xi tiles 0..DB-1, then X tiles 0..DB-1
    f32x4 st[RD];                                         // the ring
    f32x4 oh1[2], oh2[2];                                 // h1 / h2 tiles (hidden block ib) of the current and the next block
    int bb[5];                                            // blocks of the round, and the first block of the next round
    auto item_load = [&](auto gi) __attribute__((always_inline)) {     // streamed tile number gi of the round (4 NI and beyond: next round)
        constexpr int g = decltype(gi)::value, sb = g / NI, itm = g % NI;
        static_assert(sb <= 4, "ring reaches at most into the next round's first block");
        st[g % RD] = get_F(bb[sb], itm < DB ? G::pXi + itm * 256 : G::pX + (itm - DB) * 256);
    };
    const int rb0 = blockIdx.x * 4;
    bb[0] = blk_at(rb0);
    static_for<0, RD - 1>([&](auto gi) { item_load(gi); });             // (only bb[0] is needed: RD - 1 <= NI)
    static_assert(RD - 1 <= NI, "prologue stays inside the first block");
    oh1[0] = get_F(bb[0], G::pH1 + ib * 256);
    oh2[0] = get_F(bb[0], G::pH2 + ib * 256);
    __syncthreads();                                      // pairs with producer iteration 0
    for (int it = 1; it <= R; ++it) {
        const int rb = (blockIdx.x + (it - 1) * gridDim.x) * 4;
        const float* exch = bufs + ((it - 1) & 1) * 4 * EXQ;
        bb[0] = blk_at(rb); bb[1] = blk_at((long long)rb + 1); bb[2] = blk_at((long long)rb + 2); bb[3] = blk_at((long long)rb + 3);
        bb[4] = blk_at((long long)rb + 4LL * gridDim.x);
        static_for<0, 4>([&](auto sbc) {
            constexpr int sb = decltype(sbc)::value, cur = sb & 1;
            const float* ex = exch + sb * EXQ;
            // block prologue: LDS operands of this block, h1 / h2 tiles of the next one
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(ex + B2::oWts + 4 * q);    // weights of the lane's samples 4 q' .. 4 q' + 3
            f32x4 a2[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) a2[m] = tile_get(ex + m * 256, lane);
            float azk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) azk[i] = ex[i * 64 + lane];
            oh1[cur ^ 1] = get_F(bb[sb + 1], G::pH1 + ib * 256);
            oh2[cur ^ 1] = get_F(bb[sb + 1], G::pH2 + ib * 256);
            const f32x4 h2w = oh2[cur] * w4;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" ::: "memory");         // (the tied MFMAs below are inline asm: no hazard tracking for h2w)
            // ---- layer 3: dW3[:, ib] += xi^T (w h2), with the dz1 product spread over the same slots
            f32x4 dzt = zero4;
            static_for<0, DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, g = sb * NI + i;
                item_load(std::integral_constant<int, g + RD - 1>{});
#pragma unroll
                for (int r = 0; r < 4; ++r) mfma16_inplace(acc3[i], st[g % RD][r], h2w[r]);
#pragma unroll
                for (int c = 0; c < KSH; ++c) {                       // dz1 product, spread evenly over the DB slots
                    if (c * DB / KSH == i) {
                        dzt = mfma16(azk[c & 3], w2b[c], dzt);
                        if (c + 4 < KSH) azk[c & 3] = ex[(c + 4) * 64 + lane];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            const float tn = (float)(bb[sb] / a.ntile16) * dt;
            const f32x4 a1 = dzt * (1.0f - oh1[cur] * oh1[cur]);
            bs1 += a1;
            bt1 += tn * a1;
            __builtin_amdgcn_sched_barrier(0);
            // ---- layer 2: dW2[:, ib] += dz2^T h1 (the dz1 tile settles meanwhile), layer 1: dW1[ib, :] += dz1^T X_n
#pragma unroll
            for (int m = 0; m < HB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) mfma16_inplace(acc2[m], a2[m][r], oh1[cur][r]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, g = sb * NI + DB + i;
                item_load(std::integral_constant<int, g + RD - 1>{});
#pragma unroll
                for (int r = 0; r < 4; ++r) mfma16_inplace(acc1[i], a1[r], st[g % RD][r]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // the ring now holds the first RD - 1 tiles of the next round's first block; its h tiles sit in oh1[0] / oh2[0]
        __syncthreads();                                  // swap the exchange buffers (pairs with the producer loop)
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // matrix-pipe results settle before VALU / stores read them
    // ---- flush: same mapping as hjb_bwd2_kernel (tile rows = 16 ob + 4 qq + rr, columns = 16 ib + col)
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
    const int col = lane & 15, qq = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < DB; ++ob)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o3 = 16 * ob + 4 * qq + rr, i3 = 16 * ib + col;
            if (o3 < D && i3 < H) gp[G::oW3 + o3 * H + i3] = acc3[ob][rr];
            const int o1 = 16 * ib + 4 * qq + rr, i1 = 16 * ob + col;
            if (o1 < H && i1 < D) gp[G::oW1 + o1 * (D + 1) + 1 + i1] = acc1[ob][rr];
        }
#pragma unroll
    for (int ob = 0; ob < HB; ++ob)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o2 = 16 * ob + 4 * qq + rr, i2 = 16 * ib + col;
            if (o2 < H && i2 < H) gp[G::oW2 + o2 * H + i2] = acc2[ob][rr];
        }
    {
        const float v1 = qsum(hsum4(bs1)), vt = qsum(hsum4(bt1));
        const int f = 16 * ib + col;
        if (qq == 0 && f < H) {
            gp[G::ob1 + f] = v1;
            gp[G::oW1 + f * (D + 1)] = vt;
        }
    }
    __syncthreads();                                      // pairs with the producers' barrier after their LDS write
    {
        const float* red = bufs;
        const int ct = tid - 256;
        for (int f = ct; f < D; f += 256)
            gp[G::ob3 + f] = (red[f] + red[RS + f]) + (red[2 * RS + f] + red[3 * RS + f]);
        for (int f = ct; f < H; f += 256) {
            const float* r2 = red + 16 * DB + f;
            gp[G::ob2 + f] = (r2[0] + r2[RS]) + (r2[2 * RS] + r2[3 * RS]);
        }
    }
}

// LDS layout of hjbw_bwd_x3_kernel (floats)
template <int D, int H>
struct BwdX3Lds {
    using W = GeoW<D, H>;
    static constexpr int oA1 = 4 * W::EXB, oWts = oA1 + 4 * W::HB * 256, oTbl = oWts + 128;
    static constexpr int perS = W::HB * 512;                            // one S-step of the split W3^T table
    static constexpr int room = (160 * 1024 / 4 - oTbl) / perS;
    static constexpr int KSL = room < W::KS8 ? room : W::KS8;           // S-steps of the table held in LDS
    static constexpr int floats = oTbl + KSL * perS;
};

// =======================================================================================
// hjbw_bwd_x3_kernel: split-product version of hjbw_bwd_kernel for d > 256 (one wave per SIMD, 512 registers).  Same phases:
//   phase A  wave w, block 4 round + w:  dz2 = (W3^T G)(1 - h2^2) with G = w sqrt(dt) xi split on the fly per 32-feature step
//            (8 image dwords per lane) against the split W3^T table (hjbw_tables_kernel(.., 5)): 12 f16 MFMAs per step instead
//            of 32 fp32 ones; two accumulator chains (the table carries the scaled lo of table_fill_x3);
//   phase B  wave w owns hidden block ib = w; the four blocks of a round as TWO PAIRS: every weight-gradient tile contracts
//            the pair's 32 samples in three f16 MFMAs on ONE accumulator (operands split with unscaled residuals, as in
//            gen_bwd2_kernel<.., X3>): A operands G = w xi / dz2 / dz1, B operands h2 / h1 / X tiles of both blocks.
// The trajectory weights are scaled by a power of two that maps the largest |w_k| sqrt(dt) (one scan of D per workgroup) to
// [2^7, 2^8); the partial gradient is scaled back when it is written.  Gradient layout and bias sums as in hjbw_bwd_kernel.
// =======================================================================================
template <int D, int H>
__global__ __launch_bounds__(256, 1) void hjbw_bwd_x3_kernel(const HjbArgs a) {
    PSP_COND_EXIT(a);
    GradCheck<true> gchk;                                      // backward side of the range guard (hjb_kernels.h)
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    constexpr int DB = W::DB, HB = W::HB, EXB = W::EXB, KS8 = W::KS8;
    constexpr int OBW = cdiv(DB, 4);                  // state blocks per wave for the db3 sums
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const float* __restrict__ P = a.params;
    const float* __restrict__ T = a.tables;           // W3^T, split S-step-major images
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned lofsU = (unsigned)image_lane_offset_F(lane);
    const unsigned ul = (unsigned)lane;

    const double invK = 1.0 / (double)a.K_global;
    const float meanD = (a.loss_kind == LOSS_LOGVAR) ? (float)(a.sums[0] * invK) : 0.f;
    const float coef = (float)(2.0 * invK);
    const float sqdt = a.sqdt, dt = a.dt;
    const long long nblk = (long long)a.N * a.ntile16;
    const long long nround = (nblk + 3) / 4;
    auto weight_of = [&](int kk) __attribute__((always_inline)) {      // w_k sqrt(dt), unscaled
        const float dk = a.D[kk < a.K_local ? kk : 0];
        return (kk < a.K_local) ? (a.loss_kind == LOSS_WEIGHTS ? dk : coef * (dk - meanD)) * sqdt : 0.f;
    };
    // power-of-two scale: largest |w_k| sqrt(dt) -> [2^7, 2^8)
    float gs = 1.0f, ginv = 1.0f;
    {
        float am = 0.f;
        for (int k0 = tid; k0 < a.K_local; k0 += 256) am = fmaxf(am, fabsf(weight_of(k0)));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) am = fmaxf(am, __shfl_xor(am, o));
        if (lane == 0) lds[wave] = am;
        __syncthreads();
        am = fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
        const unsigned e = (__float_as_uint(am) >> 23) & 0xFFu;
        if (e >= 8u && e <= 249u) { gs = __uint_as_float((261u - e) << 23); ginv = __uint_as_float((e - 7u) << 23); }
        __syncthreads();
    }
    auto split2u = [&](const f32x4& u0, const f32x4& u1, f16x8& hi, f16x8& lo) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {                     // hi = f16(x), lo = f16(x - hi): the unscaled residual
            _Float16 h = (_Float16)u0[e];
            hi[e] = h; lo[e] = (_Float16)(u0[e] - (float)h);
            h = (_Float16)u1[e];
            hi[4 + e] = h; lo[4 + e] = (_Float16)(u1[e] - (float)h);
        }
    };
    auto fma3 = [&](f32x4& acc, const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl) __attribute__((always_inline)) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    };

    // persistent accumulators: dW3 / dW1 tiles of the state blocks ob = wave + 4 o against ALL hidden blocks; dW2[:, ib = wave]
    f32x4 acc3[OBW][HB], acc1[OBW][HB], acc2[HB];
#pragma unroll
    for (int o = 0; o < OBW; ++o)
#pragma unroll
        for (int m = 0; m < HB; ++m) { acc3[o][m] = zero4; acc1[o][m] = zero4; }
#pragma unroll
    for (int m = 0; m < HB; ++m) acc2[m] = zero4;
    f32x4 bs1 = zero4, bt1 = zero4;
    float bs3[OBW], bs2 = 0.f;
#pragma unroll
    for (int i = 0; i < OBW; ++i) bs3[i] = 0.f;

    // LDS: one exchange buffer (a wave that runs ahead into the next round's phase A writes it while the others are in phase
    // B2, which does not read it: the two barriers of a round order every other access), the dz1 tiles, the weights of two
    // rounds, then the first KSL S-steps of the split W3^T table (all of it up to d = 480; d = 500: 15 of 16).  Phase A was
    // 46 % of the round while every wave streamed the 128 KB table through its vector-memory path with two stages in flight
    // (tools/r4/wbx3_stamps.py): 5.45 -> 4.60 ms at d = 500; with the xi dwords in their own ring of seven slots and the table
    // operands one step ahead 4.47 ms (3 / 5 / 8 / 9 / 11 slots: 4.55 / 4.67 / 4.65 / 4.80 / 4.84).
    using XL = BwdX3Lds<D, H>;
    constexpr int KSL = XL::KSL;
    float* wts = lds + XL::oWts;                                        // [2 rounds][4 blocks][16] scaled trajectory weights
    float* tblL = lds + XL::oTbl;
    for (int i = tid; i < KSL * HB * 128; i += 256)                     // (the LAST KSL S-steps: see phase A)
        reinterpret_cast<f32x4*>(tblL)[i] = reinterpret_cast<const f32x4*>(T + (size_t)(KS8 - KSL) * HB * 512)[i];
    __syncthreads();
    float w2b[16];                                                      // B operands of the dz1 products: W2[4 ks + q][16 ib + j]
#pragma unroll                                                          // (512 registers here: kept for the whole kernel)
    for (int ks = 0; ks < 16; ++ks) {
        const int o = 4 * ks + q, i = 16 * wave + j;
        w2b[ks] = (o < H && i < H) ? P[G::oW2 + o * H + i] : 0.f;
    }
    int par = 0;
#ifdef PSP_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#pragma unroll 1
    for (long long round = blockIdx.x; round < nround; round += gridDim.x, par ^= 1) {
        PSP_STAMP(tx0);
        float* exch = lds;
        // ------------------------------------------------------------------ phase A: own block
        {
            const long long blk0 = round * 4 + wave;
            const bool bvalid = blk0 < nblk;
            const long long blk = bvalid ? blk0 : nblk - 1;
            const int t16 = (int)(blk % a.ntile16);
            const int k = t16 * 16 + j;
            const float wks = (bvalid ? weight_of(k) : 0.f) * gs;
            const float* pb = a.path + (size_t)blk * (size_t)G::PB + lane;
            f32x4 h2[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[m][r] = pb[G::pH2 + (4 * m + r) * 64];
            f32x4 dz2[HB], dzc[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) { dz2[m] = zero4; dzc[m] = zero4; }
            const float* xip = pb + G::pXi;
#ifndef PSP_WBX_NX
#define PSP_WBX_NX 7
#endif
            // two rings: the xi dwords of an S-step are requested NX - 1 steps ahead (global memory), its table operands one step
            // ahead (LDS; the KG <= 1 leading S-steps that did not fit are read from global memory in the prologue, where their
            // latency hides behind the first xi requests)
            constexpr int NX = PSP_WBX_NX, TS = 2, KG = KS8 - KSL;
            static_assert(KG <= 1, "at most the first S-step of the table outside the LDS");
            float xb[NX][8];
            f16x8 ah[TS][HB], al[TS][HB];
            auto load_x = [&](int st, int S) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int b = 2 * S + (e >> 2);            // block 2 S (e < 4) or 2 S + 1; past the last block: re-read it (zeroed below)
                    xb[st][e] = xip[(size_t)(4 * (b < DB ? b : DB - 1) + (e & 3)) * 64];
                }
            };
            auto load_t = [&](int st, int S) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < HB; m += 2) {
                    if (S >= KG) {                         // (S is a compile-time constant after unrolling)
                        unsigned so = (unsigned)(((S - KG) * HB + m) * 2048);   // byte offset as a scalar the compiler cannot fold:
                        asm volatile("" : "+s"(so));                       // one v_add at the use instead of 32 hoisted (and
                        const f16x8* tl = reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(tblL) + so);   // spilled) address registers
                        ah[st][m] = tl[ul]; al[st][m] = tl[64 + ul];
                        if (m + 1 < HB) { ah[st][m + 1] = tl[128 + ul]; al[st][m + 1] = tl[192 + ul]; }
                    } else {
                        gptr8_t tp = sgpr_ptr8(T + ((size_t)S * HB + m) * 512);
                        ah[st][m] = tp[ul]; al[st][m] = tp[64 + ul];
                        if (m + 1 < HB) { ah[st][m + 1] = tp[128 + ul]; al[st][m + 1] = tp[192 + ul]; }
                    }
                }
            };
            // (splitting step S + 1's increments between the MFMAs of step S -- one scheduling region -- measured slower: 4.54 -> 4.77 ms)
            auto fma_stage = [&](int sx, int st, int S) __attribute__((always_inline)) {
                f16x8 bh, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool inb = (2 * S + (e >> 2)) < DB;
                    _Float16 h, l;
                    split_f16(inb ? wks * xb[sx][e] : 0.f, h, l);
                    bh[e] = h; bl[e] = l;
                }
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    dz2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[st][m], bh, dz2[m], 0, 0, 0);
                    dzc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[st][m], bl, dzc[m], 0, 0, 0);
                    dzc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[st][m], bh, dzc[m], 0, 0, 0);
                }
            };
            load_t(0, 0);
#pragma unroll
            for (int S = 0; S < NX - 1; ++S)
                if (S < KS8) load_x(S, S);
#pragma unroll
            for (int S = 0; S < KS8; ++S) {
                if (S + NX - 1 < KS8) load_x((S + NX - 1) % NX, S + NX - 1);
                if (S + 1 < KS8) load_t((S + 1) % TS, S + 1);
                __builtin_amdgcn_sched_barrier(0);
                fma_stage(S % NX, S % TS, S);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 0) wts[par * 64 + wave * 16 + j] = wks;      // the block's 16 scaled weights, for phase B2 (no loads of D there)
            float* ex = exch + wave * EXB + lane;
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                dz2[m] = (dz2[m] + kSplitInv * dzc[m]) * (1.0f - h2[m] * h2[m]);
#pragma unroll
                for (int r = 0; r < 4; ++r) ex[(4 * m + r) * 64] = dz2[m][r];
            }
        }
        PSP_STAMP(tx1);
        __syncthreads();
        PSP_STAMP(tx2);
        // ------------------------------------------------------------------ phase B1: wave = hidden block ib -- dz1 tiles, dW2
        float* a1x = lds + XL::oA1;                                     // [4 blocks][HB] dz1 tiles (f32x4 per lane), shared below
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) {
            const long long c0 = round * 4 + 2 * pr;
            if (c0 >= nblk) break;                                    // wave-uniform
            const bool v1 = c0 + 1 < nblk;                            // second block of the pair exists (else: zero weights)
            const int cb0 = __builtin_amdgcn_readfirstlane((int)c0), cb1 = __builtin_amdgcn_readfirstlane((int)(v1 ? c0 + 1 : c0));
            const int n0 = cb0 / a.ntile16, n1 = cb1 / a.ntile16;
            const float* bp0 = a.path + (size_t)cb0 * (size_t)G::PB;
            const float* bp1 = a.path + (size_t)cb1 * (size_t)G::PB;
            const float* ex0 = exch + (2 * pr) * EXB;
            const float* ex1 = ex0 + EXB;
            const f32x4 h1t0 = *reinterpret_cast<const f32x4*>(bp0 + G::pH1 + wave * 256 + lofsU);
            const f32x4 h1t1 = *reinterpret_cast<const f32x4*>(bp1 + G::pH1 + wave * 256 + lofsU);
            // dz1 tiles of hidden block ib for both blocks: (dz2^T W2[:, ib]) (1 - h1^2)   (fp32 MFMA: 16 per block)
            f32x4 dzt0 = zero4, dzt1 = zero4;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                dzt0 = mfma16(ex0[ks * 64 + lane], w2b[ks], dzt0);
                dzt1 = mfma16(v1 ? ex1[ks * 64 + lane] : 0.f, w2b[ks], dzt1);
            }
            // dW2[:, ib] += dz2^T h1
            {
                f16x8 bh, bl;
                split2u(h1t0, h1t1, bh, bl);
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    const f32x4 a20 = tile_get(ex0 + m * 256, lane);
                    const f32x4 a21 = v1 ? tile_get(ex1 + m * 256, lane) : zero4;
                    if (m == wave) bs2 += hsum4(a20) + hsum4(a21);
                    f16x8 ahp, alp;
                    split2u(a20, a21, ahp, alp);
                    fma3(acc2[m], ahp, alp, bh, bl);
                }
            }
            const f32x4 a10 = dzt0 * (1.0f - h1t0 * h1t0);
            const f32x4 a11 = dzt1 * (1.0f - h1t1 * h1t1);
            bs1 += a10 + a11;
            bt1 += ((float)n0 * dt) * a10 + ((float)n1 * dt) * a11;
            reinterpret_cast<f32x4*>(a1x)[((2 * pr) * HB + wave) * 64 + lane] = a10;
            reinterpret_cast<f32x4*>(a1x)[((2 * pr + 1) * HB + wave) * 64 + lane] = a11;
        }
        PSP_STAMP(tx3);
        __syncthreads();
        PSP_STAMP(tx4);
        // ------------------------------------------------------------------ phase B2: wave owns the state blocks ob = wave + 4 o and ALL
        // hidden blocks: each xi / X tile of the pair is loaded and split by ONE wave (a quarter of the loads and of the VALU work
        // of a per-hidden-block stream), the h2 tiles and the dz1 tiles (LDS) of the four hidden blocks are its B / A operands
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) {
            const long long c0 = round * 4 + 2 * pr;
            if (c0 >= nblk) break;                                    // wave-uniform
            const bool v1 = c0 + 1 < nblk;
            const int cb0 = __builtin_amdgcn_readfirstlane((int)c0), cb1 = __builtin_amdgcn_readfirstlane((int)(v1 ? c0 + 1 : c0));
            const float* bp0 = a.path + (size_t)cb0 * (size_t)G::PB;
            const float* bp1 = a.path + (size_t)cb1 * (size_t)G::PB;
            // weights of the lane's samples 4 q .. 4 q + 3 of both blocks (written by phase A; an invalid block's are zero)
            const f32x4 w40 = *reinterpret_cast<const f32x4*>(wts + par * 64 + (2 * pr) * 16 + 4 * q);
            const f32x4 w41 = *reinterpret_cast<const f32x4*>(wts + par * 64 + (2 * pr + 1) * 16 + 4 * q);
            f16x8 B3h[HB], B3l[HB], A1h[HB], A1l[HB];
#pragma unroll
            for (int m = 0; m < HB; ++m) {
                const f32x4 h2t0 = *reinterpret_cast<const f32x4*>(bp0 + G::pH2 + m * 256 + lofsU);
                const f32x4 h2t1 = *reinterpret_cast<const f32x4*>(bp1 + G::pH2 + m * 256 + lofsU);
                split2u(h2t0, h2t1, B3h[m], B3l[m]);
                const f32x4 a10 = reinterpret_cast<const f32x4*>(a1x)[((2 * pr) * HB + m) * 64 + lane];
                const f32x4 a11 = reinterpret_cast<const f32x4*>(a1x)[((2 * pr + 1) * HB + m) * 64 + lane];
                split2u(a10, a11, A1h[m], A1l[m]);
            }
            const float* xib0 = bp0 + G::pXi + lofsU;
            const float* xib1 = bp1 + G::pXi + lofsU;
            const float* xb0 = bp0 + G::pX + lofsU;
            const float* xb1 = bp1 + G::pX + lofsU;
            // tile offset of this wave's o-th state block, clamped into the image (DB not a multiple of 4: the empty last slot of
            // some waves re-reads the last block; its products are discarded at the write-out)
            auto tofs = [&](int o) __attribute__((always_inline)) { const int ob = wave + 4 * o; return (ob < DB ? ob : DB - 1) * 256; };
            constexpr int RD = 3;                                     // own blocks are 4 apart: RD - 1 of them (24 MFMAs each) in flight
            f32x4 xit0[RD], xit1[RD], xt0[RD], xt1[RD];
#pragma unroll
            for (int i = 0; i < RD - 1; ++i) {
                if (i < OBW) {
                    xit0[i] = *reinterpret_cast<const f32x4*>(xib0 + tofs(i)); xit1[i] = *reinterpret_cast<const f32x4*>(xib1 + tofs(i));
                    xt0[i] = *reinterpret_cast<const f32x4*>(xb0 + tofs(i)); xt1[i] = *reinterpret_cast<const f32x4*>(xb1 + tofs(i));
                }
            }
#pragma unroll
            for (int o = 0; o < OBW; ++o) {
                if (o + RD - 1 < OBW) {
                    const int s = (o + RD - 1) % RD, of = tofs(o + RD - 1);
                    xit0[s] = *reinterpret_cast<const f32x4*>(xib0 + of); xit1[s] = *reinterpret_cast<const f32x4*>(xib1 + of);
                    xt0[s] = *reinterpret_cast<const f32x4*>(xb0 + of); xt1[s] = *reinterpret_cast<const f32x4*>(xb1 + of);
                }
                __builtin_amdgcn_sched_barrier(0);            // (a fence that lets VMEM cross lets the scheduler sink the prefetch to its use)
                const bool ovalid = wave + 4 * o < DB;            // wave-uniform (DB not a multiple of 4: the last slot of some waves is empty)
                const f32x4 g0 = ovalid ? xit0[o % RD] * w40 : zero4, g1 = ovalid ? xit1[o % RD] * w41 : zero4;
                bs3[o] += hsum4(g0) + hsum4(g1);
                f16x8 gh, gl, xh, xl;
                split2u(g0, g1, gh, gl);
                split2u(xt0[o % RD], xt1[o % RD], xh, xl);
#pragma unroll
                for (int m = 0; m < HB; ++m) {                      // (chains of different tiles interleaved: a split product is three dependent MFMAs)
                    acc3[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, B3h[m], acc3[o][m], 0, 0, 0);
                    acc1[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1h[m], xh, acc1[o][m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    acc3[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, B3l[m], acc3[o][m], 0, 0, 0);
                    acc1[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1h[m], xl, acc1[o][m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < HB; ++m) {
                    acc3[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl, B3h[m], acc3[o][m], 0, 0, 0);
                    acc1[o][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1l[m], xh, acc1[o][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PSP_STAMP(tx5);
        PSP_ACC(0, tx1, tx0); PSP_ACC(1, tx3, tx2); PSP_ACC(2, tx5, tx4); PSP_ACC(3, tx2, tx1); PSP_ACC(3, tx4, tx3); PSP_ACC(6, tx5, tx0);
#ifdef PSP_STAMPS
        stamps[7] += 1;
#endif
    }
#ifdef PSP_STAMPS
    if (a.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stamps[i];
#endif
    __syncthreads();

    // ---- write-out (layout of hjbw_bwd_kernel), scaled back by the weights' power of two
    float* gp = a.grad_partial + (size_t)blockIdx.x * G::P;
    const int col = lane & 15, qq = lane >> 4, ib = wave;
#pragma unroll
    for (int o = 0; o < OBW; ++o) {
        const int ob = wave + 4 * o;
#pragma unroll
        for (int m = 0; m < HB; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o3 = 16 * ob + 4 * qq + rr, i3 = 16 * m + col;
                if (ob < DB && o3 < D && i3 < H) { const float gv_ = ginv * acc3[o][m][rr]; gp[G::oW3 + o3 * H + i3] = gv_; gchk.see(gv_); }
                const int o1 = 16 * m + 4 * qq + rr, i1 = 16 * ob + col;
                if (ob < DB && o1 < H && i1 < D) { const float gv_ = ginv * acc1[o][m][rr]; gp[G::oW1 + o1 * (D + 1) + 1 + i1] = gv_; gchk.see(gv_); }
            }
        const float v = ginv * qsum(bs3[o]);
        const int f = 16 * ob + col;
        if (qq == 0 && ob < DB && f < D) { const float gv_ = v; gp[G::ob3 + f] = gv_; gchk.see(gv_); }
    }
#pragma unroll
    for (int m = 0; m < HB; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o2 = 16 * m + 4 * qq + rr, i2 = 16 * ib + col;
            if (o2 < H && i2 < H) { const float gv_ = ginv * acc2[m][rr]; gp[G::oW2 + o2 * H + i2] = gv_; gchk.see(gv_); }
        }
    {
        const float v1 = ginv * qsum(hsum4(bs1)), vt = ginv * qsum(hsum4(bt1));
        const int f = 16 * ib + col;
        if (qq == 0 && f < H) { { const float gv_ = v1; gp[G::ob1 + f] = gv_; gchk.see(gv_); } { const float gv_ = vt; gp[G::oW1 + f * (D + 1)] = gv_; gchk.see(gv_); } }
    }
    {
        const float v2 = ginv * qsum(bs2);
        const int f = 16 * ib + col;
        if (qq == 0 && f < H) { const float gv_ = v2; gp[G::ob2 + f] = gv_; gchk.see(gv_); }
    }
    gchk.raise(a.cond);
}

template <int D, int H>
struct HjbwLaunch {
    using G = Geo<D, H>;
    using W = GeoW<D, H>;
    static int fwd_lds(int, int) { return W::fwd_lds_floats * 4; }
    static int bwd_lds(int) { return W::bwd_lds_floats * 4; }
    static int bwd2_lds() {
        return ((D <= 256) && (GeoB2<D, H>::lds_floats * 4 <= 160 * 1024) && (G::HB == 4)) ? GeoB2<D, H>::lds_floats * 4 : W::bwd_lds_floats * 4;
    }
    static hipError_t tables(const HjbArgs& a, int backward, hipStream_t s) {
        hipLaunchKernelGGL((hjbw_tables_kernel<D, H>), dim3(256), dim3(256), 0, s, a, backward);
        return hipGetLastError();
    }
    static hipError_t fwd(const HjbArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = tables(a, 0, s);
        if (e != hipSuccess) return e;
        const int bytes = W::fwd_lds_floats * 4;
        if (a.uref) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_fwd_kernel<D, H, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_fwd_kernel<D, H, true>), dim3(grid), dim3(block), bytes, s, a);
            return hipGetLastError();
        }
        if (a.noise_mode == NOISE_PHILOX && a.tfeat == nullptr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_fwd_kernel<D, H, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_fwd_kernel<D, H, false, true>), dim3(grid), dim3(block), bytes, s, a);
            return hipGetLastError();
        }
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_fwd_kernel<D, H, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbw_fwd_kernel<D, H, false>), dim3(grid), dim3(block), bytes, s, a);
        return hipGetLastError();
    }
    // split-product forward (training launches and supplied-noise runs; the u_L2-logging instance stays fp32)
    static int fwd_x3_lds(int, int) { return W::fwd_x3_lds_floats * 4; }
    static hipError_t fwd_x3(const HjbArgs& a, int grid, int block, hipStream_t s) {
        if (a.uref) return fwd(a, grid, block, s);
        hipError_t e = tables(a, 3, s);
        if (e != hipSuccess) return e;
        const int bytes = W::fwd_x3_lds_floats * 4;
        // (a SPEC instance -- hjbw_fwd_kernel's last template argument -- measured SLOWER here: the larger scheduling regions cost the
        //  d = 500 kernel 158 spilled registers, 16.4 -> 18.6 ms, and leave d = 200 unchanged, 3.52 vs 3.56 ms; it is not instantiated)
        if (a.noise_mode == NOISE_PHILOX && a.tfeat == nullptr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_fwd_kernel<D, H, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_fwd_kernel<D, H, false, true, true>), dim3(grid), dim3(block), bytes, s, a);
            return hipGetLastError();
        }
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_fwd_kernel<D, H, false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbw_fwd_kernel<D, H, false, false, true>), dim3(grid), dim3(block), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t bwd(const HjbArgs&, int, int, hipStream_t) { return hipErrorNotSupported; }
    // d <= 256: the role-specialised kernel (8 waves, W3^T staged in LDS from the parameters: no table pass), if its LDS fits
    static constexpr bool kRoles = (D <= 256) && (GeoB2<D, H>::lds_floats * 4 <= 160 * 1024) && (G::HB == 4);
    static hipError_t bwd2(const HjbArgs& a, int grid, hipStream_t s) {
        if constexpr (kRoles) {
            const int bytes = GeoB2<D, H>::lds_floats * 4;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_bwd2_kernel<D, H>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_bwd2_kernel<D, H>), dim3(grid), dim3(512), bytes, s, a);
            return hipGetLastError();
        }
        hipError_t e = tables(a, 1, s);
        if (e != hipSuccess) return e;
        const int bytes = W::bwd_lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_bwd_kernel<D, H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbw_bwd_kernel<D, H>), dim3(grid), dim3(256), bytes, s, a);
        return hipGetLastError();
    }
    // split-product backward for the instances that run hjbw_bwd_kernel (d > 256: one wave per SIMD)
    static hipError_t bwd2_x3(const HjbArgs& a, int grid, hipStream_t s) {
        // (d <= 256: THIS entry keeps the fp32 role-specialised kernel -- hjbw_bwd_x3_kernel at d = 200 measured 3.11 against
        // 2.74 ms, round 3; hjbw_instance.hip replaces it with the split-product role-specialised kernel of hjbwx_kernels.h
        // where that one exists, round 4: 1.87 ms)
        if constexpr (kRoles || D <= 256) {
            return bwd2(a, grid, s);
        } else {
            hipError_t e = tables(a, 5, s);
            if (e != hipSuccess) return e;
            const int bytes = BwdX3Lds<D, H>::floats * 4;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_bwd_x3_kernel<D, H>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((hjbw_bwd_x3_kernel<D, H>), dim3(grid), dim3(256), bytes, s, a);
            return hipGetLastError();
        }
    }
    static hipError_t adj(const HjbArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = tables(a, 2, s);
        if (e != hipSuccess) return e;
        const int bytes = W::fwd_lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_adj_kernel<D, H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbw_adj_kernel<D, H>), dim3(grid), dim3(block), bytes, s, a);
        return hipGetLastError();
    }
    static hipError_t adj_x3(const HjbArgs& a, int grid, int block, hipStream_t s) {
        hipError_t e = tables(a, 4, s);
        if (e != hipSuccess) return e;
        const int bytes = W::fwd_x3_lds_floats * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hjbw_adj_kernel<D, H, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hjbw_adj_kernel<D, H, true>), dim3(grid), dim3(block), bytes, s, a);
        return hipGetLastError();
    }
    static HjbInstance instance() {
        HjbInstance r{D, H, G::P, &fwd_lds, &bwd_lds, &fwd, &bwd, G::PB, &bwd2_lds, &bwd2};
        r.launch_adj = &adj;
        r.launch_adj_x3 = &adj_x3;
        r.launch_bwd2_x3 = &bwd2_x3;
        r.wide = 1;
        r.bwd2_one_per_cu = kRoles ? 1 : 0;
        r.fwd_table_floats = W::fwd_table_floats > W::fwd_x3_table_floats ? W::fwd_table_floats : W::fwd_x3_table_floats;
        r.fwd_x3_lds_bytes = &fwd_x3_lds;
        r.launch_fwd_x3 = &fwd_x3;
        r.bwd_table_floats = W::bwd_table_floats > W::KS8 * W::HB * 512 ? W::bwd_table_floats : W::KS8 * W::HB * 512;
        return r;
    }
};

}  // namespace psp

#define PSP_DEFINE_WIDE_INSTANCE(D_, H_) \
    extern "C" psp::HjbInstance psp_wide_instance_##D_##_##H_() { return psp::HjbwLaunch<D_, H_>::instance(); }
#define PSP_DECLARE_WIDE_INSTANCE(D_, H_) extern "C" psp::HjbInstance psp_wide_instance_##D_##_##H_();
