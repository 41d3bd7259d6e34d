// Forward kernel for head dims 64 and 256 (every shape that is not blockwise) and for the D = 128 shapes that the
// one-wave-per-SIMD kernel does not take.  The reference fixes D = 128 (cuda_utils.cuh:11), L % 64 == 0
// (wrapper.cu:59-61) and has no mask; its README lists all three as future work (README.md:44-47).
// fa_fwd_general_kernel<D> serves every such shape too (and the blockwise entry point); this kernel is faster.
//
// What it keeps from the D = 128 kernels: transposed scores S^T = K.Q^T with v_mfma_f32_32x32x16_f16, P straight
// from the score registers into O^T += V^T.P^T, K/V tiles travelling global -> LDS by DMA (global_load_lds_dwordx4,
// swizzle applied on the source side, hidden from the compiler and awaited by hand) into a ring of stages, one
// barrier per step, the deferred rescale.  What differs:
//   * a stage is one K tile and one V tile of TK keys: 32 keys (16 KiB a tile) for D = 256, 64 keys (8 KiB) for D = 64;
//   * one workgroup per (head, query block), not persistent: Q fragments come straight from global memory, the
//     output goes straight back (these shapes are long enough for the item boundary not to matter);
//   * the two matrix products of a tile are software-pipelined INSIDE a wave: while the matrix pipe computes the
//     next tile's scores, the vector ALU turns the current tile's scores into probabilities; while it multiplies
//     those with V, the vector ALU takes the next scores' row max.  D = 256 needs this because its 128
//     accumulator registers per 32 rows leave room for one wave per SIMD only (no second wave to overlap with).
//     Both blocks of a step are written out gap by gap (one MFMA, then that gap's share of the vector work and of
//     the operand reads, sched_barrier(0)), as the phases of fa_fwd_w64_kernel are.
// Tried for D = 64, where the vector ALU binds, and measured slower: the packed forms v_pk_fma_f32 / v_pk_add_f32
// (-5 %, D = 256: -10 %), 128-key tiles (-25 %), eight waves per workgroup (-10 %), the row sums through one more
// MFMA per 16 keys (ones x P^T) instead of one addition per score (-7 % with the 48 accumulator registers it
// needs at two waves per SIMD; with the architectural registers capped for three the compiler spilled into them).
#pragma once
#include "fa_fwd_general_kernel.hpp"
#include "fa_fwd_ring_kernel.hpp"
#include "fa_fwd_w64_kernel.hpp"  // static_for

namespace fa {

template <int D>
struct HdCfg;
template <>
struct HdCfg<256> {
    static constexpr int kWaves = 4;    // 32 rows each: 128-row query blocks, one wave per SIMD
    static constexpr int kTk = 32;      // keys per tile
#ifndef FA_HD256_STAGES
#define FA_HD256_STAGES 4
#endif
    static constexpr int kStages = FA_HD256_STAGES;  // x 32 KiB: the tile sent for in step t is awaited at the end of step t + kStages - 3
    static constexpr int kOcc = 1;
#ifndef FA_HD256_AHEAD
#define FA_HD256_AHEAD 4
#endif
    static constexpr int kAhead = FA_HD256_AHEAD;  // operand fragments read this many MFMAs ahead of their use
};
// D = 128: only the masked form is used, for the shapes the one-wave-per-SIMD kernel and its masked forms do not take
// (short or mostly empty last blocks, padded batches).  32-key tiles keep a workgroup at 48 KiB of LDS and 150 + 64
// registers, so that two share a CU (with 64-key tiles and one workgroup per CU it lost to fa_fwd_general_kernel<128>).
template <>
struct HdCfg<128> {
    static constexpr int kWaves = 4;
    static constexpr int kTk = 32;
    static constexpr int kStages = 3;
    static constexpr int kOcc = 2;
    static constexpr int kAhead = 4;
};
#ifndef FA_HD64_WAVES
#define FA_HD64_WAVES 4
#define FA_HD64_TK 64
#endif
template <>
struct HdCfg<64> {
    static constexpr int kWaves = FA_HD64_WAVES;  // the vector ALU binds here (one exponential per 256 FLOP)
    static constexpr int kTk = FA_HD64_TK;
    static constexpr int kStages = 3;
    static constexpr int kOcc = FA_HD64_WAVES / 4;
#ifndef FA_HD64_AHEAD
#define FA_HD64_AHEAD 2
#endif
    static constexpr int kAhead = FA_HD64_AHEAD;
};

// All MFMAs are inline asm with explicit register files, as in fa_fwd_w64_kernel: the O^T accumulator lives in
// accumulator registers a[0 : 16 * D/32 - 1] that the asm strings name literally (the compiler never sees them, so it
// cannot re-home them into architectural registers with copies around every MFMA, which is what it does to an
// accumulator that ordinary vector code also touches, here the rescale); scores stay in architectural registers.
// tools/check_owned_regs.py audits this kernel too.
__device__ __forceinline__ void mfma_vv_first(f32x16& d, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vv(f32x16& d, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// Makes the kernel descriptor allocate a[0:31] only (D = 64: the whole accumulator).  With several waves per SIMD a
// wave's architectural and accumulator registers come out of one budget: 128 + 32 lets three workgroups share a CU.
__device__ __forceinline__ void acc_reserve32() {
    asm volatile("" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
                 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
}
__device__ __forceinline__ void acc_reserve64() {
    asm volatile("" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
                 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",
                 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47",
                 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
}
// a[LO : LO+15] = 0 through the matrix pipe; a[BASE : BASE+N-1] *= alpha through the vector ALU (rare: the rescale).
template <int LO>
__device__ __forceinline__ void acc_zero16() {
    half8 z = {};
    asm volatile("s_nop 3" : "+v"(z));  // wait states between the vector writes of z and the MFMA reading it
    asm volatile("v_mfma_f32_32x32x16_f16 a[%1:%2], %0, %0, 0" : : "v"(z), "n"(LO), "n"(LO + 15));
}
template <int BASE, int N>
__device__ __forceinline__ void acc_scale_n(float alpha) {
    acc_fence();
    static_for<N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        acc_put<BASE + i>(acc_get<BASE + i>() * alpha);
    });
}

// Scores of one tile, nothing beside them (the prologue): S^T = K.Q^T for KB key blocks, D / 16 k-steps.
template <int D, int KB>
__device__ __forceinline__ void hd_scores(const char* kb, const LaneGeomD<D>& g, const half8 (&qf)[HeadDim<D>::kSteps],
                                          f32x16 (&s)[KB]) {
    using HD = HeadDim<D>;
    static_for<HD::kSteps * KB>([&](auto ic) {
        constexpr int gi = decltype(ic)::value, b = gi % KB, ks = gi / KB;
        const half8 a = *reinterpret_cast<const half8*>(kb + g.k_rd[ks] + b * 32 * HD::kRow);
        if constexpr (ks == 0)
            mfma_vv_first(s[b], a, qf[ks]);
        else
            mfma_vv(s[b], a, qf[ks]);
    });
#pragma unroll
    for (int b = 0; b < KB; ++b) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s[b]));  // MFMA result -> vector reader
}

// Row max of raw scores (one query row per lane pair), scaled to the log2 domain.
template <int KB>
__device__ __forceinline__ float hd_row_max(const f32x16 (&s)[KB], float c) {
    float a = fmaxf(s[0][0], s[0][1]);
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int e = (b == 0 ? 2 : 0); e < 16; e += 2) a = fmaxf(fmaxf(a, s[b][e]), s[b][e + 1]);
    return half_wave_max(a) * c;
}

template <int D, int KB>
__device__ __forceinline__ half8 hd_read_v(const char* vb, const LaneGeomD<D>& g, int i) {
    using HD = HeadDim<D>;
    const int kk = i / HD::kBlocks, db = i % HD::kBlocks;
    const char* a0 = vb + g.v_rd[db] + kk * 16 * HD::kRow;
    const short4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
    const short4_t t1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0 + 8 * HD::kRow));
    return __builtin_bit_cast(half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Block 1 of a step, written out gap by gap (sched_barrier(0) after each):
//   matrix pipe   S_next = K(t+1).Q^T                     (NS = KB * D/16 MFMAs, K fragments read KA ahead)
//   vector ALU    p = exp2(S_cur c - m), row sum, P^T     (16 KB elements spread over the gaps; an element is summed
//                                                          one gap after its exponential, a pair rounded to fp16
//                                                          one gap after that; plain fp32 instructions: the packed
//                                                          forms v_pk_fma_f32 / v_pk_add_f32 measured 5-10 % slower
//                                                          beside MFMAs)
// With NEXT = false (the last tile) only the vector side runs.
//   tile transfer dma(j), j = 0 .. NDMA-1: this wave's DMA pieces of the step, one every NS / NDMA gaps (issued in one
//                 burst they stall the wave on the memory pipe's issue queue: measured 19 % of the D = 256 step)
template <int D, int KB, bool NEXT, int NDMA, typename DMA>
__device__ __forceinline__ void hd_block1(const char* kb, const LaneGeomD<D>& g, const half8 (&qf)[HeadDim<D>::kSteps],
                                          f32x16 (&s_next)[KB], f32x16 (&s_cur)[KB], float c, float m_run, float& l_run,
                                          half8 (&pf)[2 * KB], DMA&& dma) {
    using HD = HeadDim<D>;
    constexpr int NS = HD::kSteps * KB, NE = 16 * KB, KA = HdCfg<D>::kAhead;
    const float nm = -m_run;
    float sum = 0.0f;
    u32x4 pw[2 * KB] = {};
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    half8 kf[KA];
    auto k_frag = [&](auto ic) -> half8 {
        constexpr int gi = decltype(ic)::value, b = gi % KB, ks = gi / KB;
        return *reinterpret_cast<const half8*>(kb + g.k_rd[ks] + b * 32 * HD::kRow);
    };
    if constexpr (NEXT) static_for<KA>([&](auto ic) { kf[decltype(ic)::value] = k_frag(ic); });
    auto started = [](int gp) constexpr { return gp <= 0 ? 0 : (gp >= NS ? NE : gp * NE / NS); };
    auto convert_pair = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const f32x2 pr = {s_cur[(2 * q) / 16][(2 * q) % 16], s_cur[(2 * q + 1) / 16][(2 * q + 1) % 16]};
        const half2_t h = __builtin_convertvector(pr, half2_t);
        pw[q / 4][q % 4] = __builtin_bit_cast(unsigned, h);
    };
    static_for<NS>([&](auto ic) {
        constexpr int gi = decltype(ic)::value;
        if constexpr (NEXT) {
            constexpr int b = gi % KB, ks = gi / KB;
            if constexpr (ks == 0)
                mfma_vv_first(s_next[b], kf[gi % KA], qf[ks]);
            else
                mfma_vv(s_next[b], kf[gi % KA], qf[ks]);
#ifdef FA_HD_ABLATE_LDS  // (timing experiment only: every second fragment read is left out; wrong results)
            if constexpr (gi + KA < NS && (gi & 1) == 0) kf[gi % KA] = k_frag(std::integral_constant<int, gi + KA>{});
#else
            if constexpr (gi + KA < NS) kf[gi % KA] = k_frag(std::integral_constant<int, gi + KA>{});
#endif
            constexpr int every = NS / NDMA;
            if constexpr (gi % every == every - 1) dma(std::integral_constant<int, gi / every>{});
        }
        constexpr int n_done = started(gi), n_new = started(gi + 1) - n_done, n_prev = started(gi - 1),
                      n_prev2 = started(gi - 2);
#ifndef FA_HD_ABLATE_VEC
        static_for<n_new>([&](auto jc) {
            constexpr int e = n_done + decltype(jc)::value;
            s_cur[e / 16][e % 16] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[e / 16][e % 16], c, nm));
        });
#endif
        static_for<n_done - n_prev>([&](auto jc) {
            constexpr int e = n_prev + decltype(jc)::value;
            sum += s_cur[e / 16][e % 16];
        });
        static_for<n_prev / 2 - n_prev2 / 2>([&](auto jc) { convert_pair(std::integral_constant<int, n_prev2 / 2 + decltype(jc)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NE - started(NS - 1)>([&](auto jc) {
        constexpr int e = started(NS - 1) + decltype(jc)::value;
        sum += s_cur[e / 16][e % 16];
    });
    static_for<NE / 2 - started(NS - 2) / 2>([&](auto jc) { convert_pair(std::integral_constant<int, started(NS - 2) / 2 + decltype(jc)::value>{}); });
    l_run += sum;
#pragma unroll
    for (int j = 0; j < 2 * KB; ++j) pf[j] = __builtin_bit_cast(half8, pw[j]);
    // pinned here (the optimiser would sink the vector side into the consuming block), with the wait states a
    // vector write needs before an MFMA may read the register
#pragma unroll
    for (int j = 0; j < 2 * KB; ++j) asm volatile("s_nop 4" : "+v"(pf[j]));
    asm volatile("" : "+v"(l_run));
}

// Block 2 of a step:
//   matrix pipe   O^T += V(t)^T.P(t)^T                    (NP = 2 KB * D/32 MFMAs into a[0..], V^T fragments read VA ahead)
//   vector ALU    row max of S_next (from gap 3 on: the last MFMA that wrote S_next is three MFMAs back by then)
template <int D, int KB, bool NEXT>
__device__ __forceinline__ void hd_block2(const char* vb, const LaneGeomD<D>& g, const half8 (&pf)[2 * KB],
                                          const f32x16 (&s_next)[KB], float c, float& mx_next) {
    using HD = HeadDim<D>;
    constexpr int NP = 2 * KB * HD::kBlocks, VA = HdCfg<D>::kAhead;
    constexpr int NM = 8 * KB;  // links of the max3 chain
    half8 vf[VA];
    static_for<VA>([&](auto ic) { vf[decltype(ic)::value] = hd_read_v<D, KB>(vb, g, decltype(ic)::value); });
    float a = 0.0f;
    static_for<NP>([&](auto ic) {
        constexpr int gi = decltype(ic)::value, kk = gi / HD::kBlocks, db = gi % HD::kBlocks;
        mfma_acc<16 * db>(vf[gi % VA], pf[kk]);
#ifdef FA_HD_ABLATE_LDS
        if constexpr (gi + VA < NP && (gi & 1) == 0) vf[gi % VA] = hd_read_v<D, KB>(vb, g, gi + VA);
#else
        if constexpr (gi + VA < NP) vf[gi % VA] = hd_read_v<D, KB>(vb, g, gi + VA);
#endif
#ifdef FA_HD_ABLATE_MAX
        if constexpr (false) {
#else
        if constexpr (NEXT && gi >= 3) {
#endif
            // links [lo, hi) of the chain in this gap
            constexpr int lo = (gi - 3) * NM / (NP - 3), hi = (gi - 2) * NM / (NP - 3);
            static_for<hi - lo>([&](auto jc) {
                constexpr int jl = lo + decltype(jc)::value, b = (2 * jl) / 16, e = (2 * jl) % 16;
                if constexpr (jl == 0)
                    a = fmaxf(s_next[b][e], s_next[b][e + 1]);
                else
                    a = fmaxf(fmaxf(a, s_next[b][e]), s_next[b][e + 1]);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (NEXT) mx_next = half_wave_max(a) * c;
}

// Upper-bound mask on fresh scores, between the two blocks (or after the prologue's scores): element e of key block b
// in lane (r, hh) is key 32 b + 8 (e >> 2) + (e & 3) + 4 hh of the tile; it stays iff that is <= thr + 4 hh, thr per
// lane = (last key this query row may see) - (the tile's first key).  Masked scores become -inf: their exponential is 0
// and they leave the row max alone.
template <int KB>
__device__ __forceinline__ void hd_mask_scores(f32x16 (&s)[KB], int thr_minus_4hh) {
    const float ninf = -__builtin_inff();
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kc = 32 * b + 8 * (e >> 2) + (e & 3);
            s[b][e] = kc > thr_minus_4hh ? ninf : s[b][e];
        }
}

// A block of which this workgroup sees no key (blockwise mode): the carried state goes through as it is, to the
// state arrays again or, on the last block, normalised to O.  Lane (r, hh) holds columns 8c + 4hh .. + 3 of its row.
template <int D>
__device__ __forceinline__ void hd_pass_state(const FwdParams& p, size_t srow, int hh, bool row_valid, bool load,
                                              bool store) {
    if (!row_valid) return;
    if (load && store) return;  // in and out are the same arrays
    if (store) {  // a first block without keys: the empty state
        if (hh == 0) {
            p.row_max[srow] = -1e30f * 0.69314718055994530942f;
            p.row_sum[srow] = 0.0f;
        }
        float* oa = p.o_acc + srow * D;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) *reinterpret_cast<f32x4*>(oa + 8 * c + 4 * hh) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    // the last block: O = o_acc / row_sum (0 for a row that has seen no key at all)
    const float l = p.row_sum[srow];
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const float* oa = p.o_acc + srow * D;
    half_t* og = p.o + srow * D;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(oa + 8 * c + 4 * hh);
        *reinterpret_cast<half4*>(og + 8 * c + 4 * hh) = __builtin_convertvector(t * inv, half4);
    }
}

// MASKED = false: whole tiles, no mask (lq % 32 == 0, lk % TK == 0).  MASKED = true adds, outside the two blocks,
//   * any lq >= 1: query rows past the end are computed on a copy of the last row and not stored;
//   * any lk >= 0: the rows of the last tile that lie past the last key are fetched from the last key's row (finite
//     data) and their scores masked; no keys at all gives O = 0;
//   * the causal mask of attention_cpu (testing.cu:74-77: key j <= query i, both counted from the sequence's start): a
//     workgroup stops at the tile that holds its last row's diagonal, the long query blocks are handed out first;
//   * padded batches: per-sequence lengths q_len[b], k_len[b] (device arrays) in place of lq, lk, which stay the strides.
// A tile that the diagonal or the sequence's end touches gets hd_mask_scores() in a cold branch between the blocks.
// STATE (with MASKED) adds blockwise mode, as in the other kernels (kFlagLoadState / kFlagStoreState: the carried
// (o_acc, row_max, row_sum) of every row in, or out instead of O) with the mask on absolute positions q_pos0 + i,
// k_pos0 + j: a row may then see no key of the block at all, and its carried state passes through unchanged.
template <int D, bool MASKED, bool STATE = false>
__device__ __forceinline__ void fa_fwd_hd_body(const FwdParams& p) {
    static_assert(!STATE || MASKED, "blockwise mode is built on the masked form");
    using HD = HeadDim<D>;
    using CFG = HdCfg<D>;
    constexpr int NWAVES = CFG::kWaves, TK = CFG::kTk, STAGES = CFG::kStages, KB = TK / 32;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int kTile = TK * HD::kRow;             // bytes of a K (or V) tile: 16 KiB
    constexpr int kStage = 2 * kTile;                // K tile, then V tile
    constexpr int kPieces = kTile / 1024;            // 1 KiB DMA pieces per tile: 16
    constexpr int PPW = kPieces / NWAVES;            // pieces of each tile this wave moves
    constexpr int kRowsPerPiece = 1024 / HD::kRow;   // 8 (D = 64) or 2 (D = 256)
    static_assert(PPW >= 1 && PPW * NWAVES == kPieces, "tile geometry");

    __shared__ __attribute__((aligned(1024))) char ring[STAGES * kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeomD<D> g = make_lane_geom_d<D>(lane);

    int head, qb;
    decode_item(p, (int)blockIdx.x, head, qb);
    [[maybe_unused]] const bool causal = MASKED && (p.flags & kFlagCausal) != 0;
    [[maybe_unused]] const int shift = STATE ? p.q_pos0 - p.k_pos0 : 0;  // keys are counted from this block's first
    [[maybe_unused]] const bool load_state = STATE && (p.flags & kFlagLoadState) != 0;
    [[maybe_unused]] const bool store_state = STATE && (p.flags & kFlagStoreState) != 0;
    // Key ranges (STATE, p.k_splits > 1; the launcher's split for shapes with few workgroups, such as a handful of
    // query rows against a long sequence): head `head` of the p.bh is range head / bh_real of real head
    // head % bh_real; it reads that head's Q rows and keys [key_first, key_first + lk_v) and stores its state under
    // its own (virtual) head.  (A padded sequence may end before a range begins: that range stores the empty state.)
    [[maybe_unused]] int head_q = head;
    [[maybe_unused]] int key_first = 0;
    if constexpr (STATE) {
        if (p.k_splits > 1) {
            const int range = head / p.bh_real;
            head_q = head - range * p.bh_real;
            key_first = range * p.lk_split;
        }
    }
    int lq_v = p.lq, lk_v = p.lk;  // this sequence's valid rows and keys
    if constexpr (MASKED) {
        if (causal) qb = p.q_blocks - 1 - qb;  // block qb costs qb + 1 units: the long ones first
        if (p.q_len || p.k_len) {
            const int batch = (STATE ? head_q : head) / p.heads_per_batch;
            if (p.q_len) lq_v = min(max(p.q_len[batch], 0), p.lq);
            if (p.k_len) lk_v = min(max(p.k_len[batch], 0), p.lk);
        }
        if constexpr (STATE)
            if (p.k_splits > 1) lk_v = min(max(lk_v - key_first, 0), p.lk_split);
        if (qb * QB >= lq_v) return;  // (uniform over the workgroup, before its first barrier or transfer)
        // (a padded batch whose long sequences run on the fa_fwd_w64_varlen kernels, launched before this one: the list's
        // planner left this launch's thresholds behind the item count)
        if (p.long_q != 0 && varlen_on_tuned(lq_v, lk_v, p.varlen_count[1], p.varlen_count[2], causal)) return;
    }
    const int q_row0 = qb * QB + wave * kRowsPerWave;   // wave-uniform
    const bool wave_valid = q_row0 < lq_v;              // a wave past the last row computes along and stores nothing
    // this lane's query row: past the end, the last row (MASKED) or the block's first rows (whole tiles)
    const int q_abs = MASKED ? min(q_row0 + g.r, lq_v - 1) : (wave_valid ? q_row0 : 0) + g.r;
    const size_t srow = (size_t)head * p.lq + q_abs;
    const float c = p.scale_log2e;
    int n_tiles = MASKED ? (lk_v + TK - 1) / TK : p.lk / TK;
    [[maybe_unused]] const int last_key_tile = n_tiles - 1;          // the tile the sequence's end may cut
    [[maybe_unused]] const bool ragged_end = MASKED && lk_v % TK != 0;
    if constexpr (MASKED) {
        if (causal) n_tiles = max(0, min(n_tiles, (min(qb * QB + QB, lq_v) + shift + TK - 1) / TK));
        if (n_tiles == 0) {  // no keys: O = 0, or the carried state as it is
            if constexpr (STATE) {
                if (load_state || store_state) {
                    hd_pass_state<D>(p, srow, g.hh, wave_valid && q_row0 + g.r < lq_v, load_state, store_state);
                    return;
                }
            }
            if (wave_valid && q_row0 + g.r < lq_v) {
                half_t* og = p.o + srow * D;
#pragma unroll
                for (int i = 0; i < D / 4; i += 2) *reinterpret_cast<half4*>(og + 4 * (i + g.hh)) = half4{};
            }
            return;
        }
    }
    // last key this lane's row may see, and whether tile `tile` needs the mask for this wave (wave-uniform)
    [[maybe_unused]] const int key_limit = causal ? min(lk_v - 1, q_abs + shift) : lk_v - 1;
    [[maybe_unused]] auto needs_mask = [&](int tile) -> bool {
        return (ragged_end && tile == last_key_tile) || (causal && tile * TK + TK - 1 > q_row0 + shift);
    };

    // ---- DMA geometry: this wave moves pieces PPW*wave .. PPW*wave + PPW - 1 of the K tile and of the V tile, i.e.
    //      rows kRowsPerPiece * (PPW*wave + j) + rip.  Lane l fills slot l % chunks of row rip = l / chunks with the
    //      source chunk the image's swizzle maps there.  Consecutive pieces are 1 KiB apart on both sides.
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    const int d_rip = lane / HD::kChunks, d_slot = lane % HD::kChunks;
    unsigned d_koff[PPW], d_voff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = kRowsPerPiece * (PPW * wave + j) + d_rip;   // row of the tile (piece j's share added by the offset)
        const unsigned rowoff = (unsigned)((kRowsPerPiece * PPW * wave + d_rip) * HD::kRow);
        d_koff[j] = rowoff + (unsigned)((d_slot ^ (row & HD::kKMask)) << 4);
        d_voff[j] = rowoff + (unsigned)((d_slot ^ ((row & HD::kVMask) << 2)) << 4);
    }
    const unsigned d_lds = ring_lds + (unsigned)(PPW * wave) * 1024u;
    const half_t* kh = p.k + ((size_t)(STATE ? head_q : head) * p.lk + (size_t)(STATE ? key_first : 0)) * D;
    const half_t* vh = p.v + ((size_t)(STATE ? head_q : head) * p.lk + (size_t)(STATE ? key_first : 0)) * D;
    // MASKED: the same offsets for the tile the sequence's end cuts, with every row past the last key replaced by the
    // last key's row.  (The instruction's immediate offset 1024 j is added on both sides, so it is taken off here.)
    [[maybe_unused]] unsigned d_koff_end[PPW], d_voff_end[PPW];
    if constexpr (MASKED) {
        const int rows_in_last = lk_v - last_key_tile * TK;  // 1 .. TK
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = kRowsPerPiece * (PPW * wave + j) + d_rip;
            const int src = min(row, rows_in_last - 1);
            // (src - row) * kRow <= 0: folded into the 64-bit base below, the 32-bit offset stays non-negative
            d_koff_end[j] = (unsigned)(src * HD::kRow) + (unsigned)((d_slot ^ (row & HD::kKMask)) << 4);
            d_voff_end[j] = (unsigned)(src * HD::kRow) + (unsigned)((d_slot ^ ((row & HD::kVMask) << 2)) << 4);
        }
    }
    // One tile's transfer, all pieces at once (the prologue).  `cut`: the tile the sequence's end cuts.
    auto dma_tile = [&](int tile, int stage) {
        const half_t* kt = kh + (size_t)tile * TK * D;
        const half_t* vt = vh + (size_t)tile * TK * D;
        const unsigned l_ = d_lds + (unsigned)stage * (unsigned)kStage;
        const bool cut = ragged_end && tile == last_key_tile;
        static_for<PPW>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (MASKED && cut) {
                dma_set_base(l_ + 1024u * j);
                dma_piece_at<0>(kt, d_koff_end[j]);
            } else {
                dma_set_base(l_);
                dma_piece_at<1024 * j>(kt, d_koff[j]);
            }
        });
        static_for<PPW>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (MASKED && cut) {
                dma_set_base(l_ + (unsigned)kTile + 1024u * j);
                dma_piece_at<0>(vt, d_voff_end[j]);
            } else {
                dma_set_base(l_ + (unsigned)kTile);
                dma_piece_at<1024 * j>(vt, d_voff[j]);
            }
        });
    };

    // ---- prologue: the first PF = STAGES - 1 tiles on their way, Q fragments from global memory
    constexpr int PF = STAGES - 1;
    static_assert(PF >= 2 && (PF - 2) * 2 * PPW < 64, "look-ahead against the load counter's width");
    static_for<PF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i == 0 || i < n_tiles) dma_tile(i, i);
    });
    half8 qf[HD::kSteps];
    {
        const half_t* qr = p.q + (STATE ? (size_t)head_q * p.lq + q_abs : srow) * D + g.hh * 8;
#pragma unroll
        for (int ks = 0; ks < HD::kSteps; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qr + ks * 16);
    }
    float l_run = 0.0f;
    wait_all_vmem();
    __syncthreads();
    if constexpr (D == 64)
        acc_reserve32();
    else if constexpr (D == 128)
        acc_reserve64();
    else
        acc_reserve();
    static_for<HD::kBlocks>([&](auto ic) { acc_zero16<16 * decltype(ic)::value>(); });

    // scores of tile 0 and their row max: the first reference point
    f32x16 s_cur[KB], s_next[KB];
    hd_scores<D, KB>(ring, g, qf, s_cur);
    if constexpr (MASKED)
        if (needs_mask(0)) hd_mask_scores<KB>(s_cur, key_limit - 4 * g.hh);
    float m_run = hd_row_max<KB>(s_cur, c);
    [[maybe_unused]] float m_true = m_run;  // (STATE: the state goes out relative to the true running max)
    if constexpr (STATE) {
        // a row that sees nothing of tile 0 sees nothing of the block: keep its reference finite
        m_run = fmaxf(m_run, -1e30f);
        m_true = m_run;
        if (load_state) {
            const float mx0 = m_run;
            m_run = p.row_max[srow] * 1.44269504088896340736f;
            l_run = g.hh == 0 ? p.row_sum[srow] : 0.0f;
            const float* oa = p.o_acc + srow * D;
            acc_fence();
            static_for<HD::kBlocks * 4>([&](auto ic) {
                constexpr int cidx = decltype(ic)::value;  // 4 db + g4
                const f32x4 t = *reinterpret_cast<const f32x4*>(oa + 8 * cidx + 4 * g.hh);
                acc_put<4 * cidx>(t[0]);
                acc_put<4 * cidx + 1>(t[1]);
                acc_put<4 * cidx + 2>(t[2]);
                acc_put<4 * cidx + 3>(t[3]);
            });
            m_true = fmaxf(m_run, mx0);
            if (!__all(mx0 - m_run <= kRescaleSlack)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_true);
                m_run = m_true;
                l_run *= alpha;
                acc_scale_n<0, 16 * HD::kBlocks>(alpha);
            }
            wait_all_vmem();  // (the loop counts its own transfers by hand: nothing else may be in flight)
        }
    }

    // One step per tile.  The loop below is STAGES copies of it, one per ring stage, so that every stage offset is a
    // compile-time constant and the LDS reads take it as their immediate offset (with a run-time stage every read
    // paid a vector instruction to form its address: 24 of about 160 per tile and wave at D = 64).  Tile t lives in
    // stage t % STAGES.  Returns true after the last tile.
    int t = 0;
    auto step = [&](auto sc) -> bool {
        constexpr int ST = decltype(sc)::value, ST1 = (ST + 1) % STAGES, ST2 = (ST + STAGES - 1) % STAGES;
        const bool more = t + 1 < n_tiles;
        // this step's transfer: tile t+PF (past the end: the last tile once more, into a stage nobody reads any more),
        // into the stage tile t-1 occupied (its last reader has passed the barrier); issued piece by piece inside block 1
        const int t2 = t + PF < n_tiles ? t + PF : n_tiles - 1;
        const half_t* kt2 = kh + (size_t)t2 * TK * D;
        const half_t* vt2 = vh + (size_t)t2 * TK * D;
        const unsigned l2 = d_lds + (unsigned)(ST2 * kStage);
        // MASKED: the offsets of this step's pieces, those of the cut tile when it is the one sent for (chosen here,
        // outside the block; the pieces then carry their LDS address in m0 and no immediate)
        [[maybe_unused]] unsigned ko[PPW], vo[PPW];
        if constexpr (MASKED) {
            const bool cut = ragged_end && t2 == last_key_tile;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                ko[j] = cut ? d_koff_end[j] : d_koff[j] + 1024u * j;
                vo[j] = cut ? d_voff_end[j] : d_voff[j] + 1024u * j;
            }
        }
        auto dma_step = [&](auto jc) {
#ifndef FA_HD_ABLATE_DMA  // (the FA_HD_ABLATE_* builds are timing experiments: wrong results)
            constexpr int j = decltype(jc)::value;
            if constexpr (MASKED) {
                if constexpr (j < PPW) {
                    dma_set_base(l2 + 1024u * j);
                    dma_piece_at<0>(kt2, ko[j]);
                } else {
                    dma_set_base(l2 + (unsigned)kTile + 1024u * (j - PPW));
                    dma_piece_at<0>(vt2, vo[j - PPW]);
                }
            } else {
                if constexpr (j == 0) dma_set_base(l2);
                if constexpr (j == PPW) dma_set_base(l2 + (unsigned)kTile);
                if constexpr (j < PPW)
                    dma_piece_at<1024 * j>(kt2, d_koff[j]);
                else
                    dma_piece_at<1024 * (j - PPW)>(vt2, d_voff[j - PPW]);
            }
#endif
        };
        auto dma_none = [&](auto) {};
        half8 pf[2 * KB];
        float mx = m_run;
        if (__builtin_expect(more, 1)) {
            hd_block1<D, KB, true, 2 * PPW>(ring + ST1 * kStage, g, qf, s_next, s_cur, c, m_run, l_run, pf, dma_step);
            if constexpr (MASKED)
                if (__builtin_expect(needs_mask(t + 1), 0)) {
#pragma unroll
                    for (int b = 0; b < KB; ++b) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s_next[b]));  // MFMA result -> vector reader
                    hd_mask_scores<KB>(s_next, key_limit - (t + 1) * TK - 4 * g.hh);
                }
            hd_block2<D, KB, true>(ring + ST * kStage + kTile, g, pf, s_next, c, mx);
            if constexpr (STATE) m_true = fmaxf(m_true, mx);
            if (__builtin_expect(!__all(mx - m_run <= kRescaleSlack), 0)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
                acc_scale_n<0, 16 * HD::kBlocks>(alpha);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) s_cur[b] = s_next[b];
        } else {
            hd_block1<D, KB, false, 2 * PPW>(ring, g, qf, s_next, s_cur, c, m_run, l_run, pf, dma_none);
            hd_block2<D, KB, false>(ring + ST * kStage + kTile, g, pf, s_next, c, mx);
            return true;
        }
#ifndef FA_HD_ABLATE_BARRIER
        // this wave's share of tile t+2 has landed (the next step reads its K): it was sent for PF - 2 steps ago, and
        // the pieces of the steps since stay in flight (loads complete in order)
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((PF - 2) * 2 * PPW) : "memory");
        __syncthreads();
#endif
        ++t;
        return false;
    };
    for (;;) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
        if (step(std::integral_constant<int, 2>{})) break;
        if constexpr (STAGES > 3)
            if (step(std::integral_constant<int, 3>{})) break;
        if constexpr (STAGES > 4)
            if (step(std::integral_constant<int, 4>{})) break;
    }

    wait_all_vmem();  // (the last steps' redundant transfers must not outlive the workgroup's LDS)
    // O = acc / l: lane (r, hh) holds columns 32db + 8g4 + 4hh .. +3 of row r in a[16db + 4g4 ..].  Stored from there a
    // wave instruction would touch 32 rows with 16 bytes each; the rows go through LDS instead (32 rows of this wave
    // in the ring, which every wave has left: chunk c of row r at r * kRow + ((c ^ (r & kKMask)) << 4), the K tiles'
    // swizzle) and leave as whole rows, 16 bytes per lane (measured with the stores taken out: they cost 15 - 23 % of
    // the kernel at L <= 600 in the direct form).
    acc_fence();
    if constexpr (STATE) {
        if (store_state) {
            // the state relative to the true running max, whatever reference was in use
            const float beta = __builtin_amdgcn_exp2f(m_run - m_true);
            const float l_sum = half_wave_sum(l_run) * beta;
            if (wave_valid && q_row0 + g.r < lq_v) {
                if (g.hh == 0) {
                    p.row_max[srow] = m_true * 0.69314718055994530942f;
                    p.row_sum[srow] = l_sum;
                }
                float* oa = p.o_acc + srow * D;
                static_for<HD::kBlocks * 4>([&](auto ic) {
                    constexpr int cidx = decltype(ic)::value;
                    const f32x4 t = {acc_get<4 * cidx>() * beta, acc_get<4 * cidx + 1>() * beta,
                                     acc_get<4 * cidx + 2>() * beta, acc_get<4 * cidx + 3>() * beta};
                    *reinterpret_cast<f32x4*>(oa + 8 * cidx + 4 * g.hh) = t;
                });
            }
            return;
        }
    }
    const float l_tot = half_wave_sum(l_run);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
#ifdef FA_HD_DIRECT_STORE  // (comparison builds: the direct form)
    if (MASKED ? (wave_valid && q_row0 + g.r < lq_v) : wave_valid) {
        half_t* og = p.o + srow * D;
        static_for<HD::kBlocks * 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value, db = i / 4, g4 = i % 4;
            f32x4 v;
            v[0] = acc_get<16 * db + 4 * g4 + 0>() * inv;
            v[1] = acc_get<16 * db + 4 * g4 + 1>() * inv;
            v[2] = acc_get<16 * db + 4 * g4 + 2>() * inv;
            v[3] = acc_get<16 * db + 4 * g4 + 3>() * inv;
            *reinterpret_cast<half4*>(og + 32 * db + 8 * g4 + 4 * g.hh) = __builtin_convertvector(v, half4);
        });
    }
#else
    __syncthreads();  // (every wave has read its last V tile)
    char* st = ring + wave * (kRowsPerWave * HD::kRow);
    // The lane id, taken from the hardware here: anything derived from the kernel's own copy is computed in front of
    // the tile loop and kept alive through it, and the loop has no register to spare (at D = 128 the compiler then
    // parked values in accumulator registers the MFMAs own).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    {
        const int r = lane_e & 31, hh = lane_e >> 5;
        char* wr = st + r * HD::kRow + 8 * hh;
        const int sw = (r & HD::kKMask) << 4;
        static_for<HD::kBlocks * 4>([&](auto ic) {
            constexpr int cidx = decltype(ic)::value;  // 4 db + g4
            const f32x4 v = {acc_get<4 * cidx>() * inv, acc_get<4 * cidx + 1>() * inv, acc_get<4 * cidx + 2>() * inv,
                             acc_get<4 * cidx + 3>() * inv};
            *reinterpret_cast<half4*>(wr + ((cidx << 4) ^ sw)) = __builtin_convertvector(v, half4);
        });
    }
    // (the rows are written as half4 and read back as u32x4: without this the compiler, which takes accesses of
    // different types for independent, moved reads in front of writes)
    asm volatile("" ::: "memory");
    {
        constexpr int LPR = HD::kChunks, RPI = 64 / LPR, NI = kRowsPerWave / RPI;  // lanes per row, rows per instruction
        const int rr = lane_e / LPR, ch = lane_e % LPR;
        // rows of this wave that exist (whole tiles: all or none)
        const int n_rows = MASKED ? min(max(lq_v - q_row0, 0), kRowsPerWave) : (wave_valid ? kRowsPerWave : 0);
        half_t* og = p.o + ((size_t)head * p.lq + q_row0) * D + ch * 8;
        constexpr int GRP = NI < 4 ? NI : 4;  // reads in flight before their stores (registers: D = 128 has none to spare)
        static_for<NI / GRP>([&](auto gc) {
            constexpr int i0 = decltype(gc)::value * GRP;
            u32x4 v[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int row = (i0 + i) * RPI + rr;
                v[i] = *reinterpret_cast<const u32x4*>(st + row * HD::kRow + ((ch ^ (row & HD::kKMask)) << 4));
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int row = (i0 + i) * RPI + rr;
#ifdef FA_HD_ABLATE_STORE  // (timing experiments only: wrong results)
                asm volatile("" ::"v"(v[i]));
#else
                if (row < n_rows) {  // (device scope + non-temporal, as in fa::store_staged_rows)
                    const half_t* a_ = og + (size_t)row * D;
                    asm volatile("global_store_dwordx4 %0, %1, off " FA_STORE_MODS : : "v"(a_), "v"(v[i]) : "memory");
                }
#endif
            }
        });
    }
#endif
}

template <int D>
__global__ __launch_bounds__(HdCfg<D>::kWaves * 64, HdCfg<D>::kOcc) void fa_fwd_hd_kernel(const FwdParams p) {
    fa_fwd_hd_body<D, false>(p);
}
// The same with the causal mask, any lengths and padded batches.
template <int D>
__global__ __launch_bounds__(HdCfg<D>::kWaves * 64, HdCfg<D>::kOcc) void fa_fwd_hd_masked_kernel(const FwdParams p) {
    fa_fwd_hd_body<D, true>(p);
}
// The masked form in blockwise mode: carried state in and / or out, the mask on absolute positions.
// (D = 128: compiled for one workgroup per CU: with the register budget of two the carried state did not fit, and
// the compiler parked values in accumulator registers the MFMAs own; how many workgroups share a CU is decided by
// what the kernel then uses.)
template <int D>
__global__ __launch_bounds__(HdCfg<D>::kWaves * 64, D == 128 ? 1 : HdCfg<D>::kOcc) void fa_fwd_hd_state_kernel(const FwdParams p) {
    fa_fwd_hd_body<D, true, true>(p);
}

// Folds the key ranges of a split launch of fa_fwd_hd_state_kernel<D> (fa_split_merge_kernel for any head dim): row i
// of the output is  O[i] = sum_s w_s o_s[i] / sum_s w_s l_s[i],  w_s = exp(m_s[i] - max_s m_s[i])  (o_s, m_s, l_s: row
// s * rows + i of the state arrays).  D / 4 lanes per row, four columns each.  A row whose every range is empty
// (a padded sequence without keys) gets 0.
template <int D>
__global__ __launch_bounds__(256) void fa_hd_split_merge_kernel(const float* __restrict__ o_acc, const float* __restrict__ row_max,
                                                                const float* __restrict__ row_sum, half_t* __restrict__ o,
                                                                size_t rows, int n_ranges, const int* __restrict__ q_len,
                                                                int lq, int heads_per_batch) {
    constexpr int LPR = D / 4, RPB = 256 / LPR;
    const int c = threadIdx.x % LPR;
    const size_t per_pass = (size_t)gridDim.x * RPB;
    for (size_t i = (size_t)blockIdx.x * RPB + threadIdx.x / LPR; i < rows; i += per_pass) {
        if (q_len) {  // a padded batch: rows behind a sequence's length have no state and get no output
            const size_t head = i / (size_t)lq;
            if ((int)(i - head * lq) >= q_len[head / heads_per_batch]) continue;
        }
        float m = -__builtin_inff();
        for (int s = 0; s < n_ranges; ++s) m = fmaxf(m, row_max[(size_t)s * rows + i]);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        float l = 0.0f;
        for (int s = 0; s < n_ranges; ++s) {
            const size_t r = (size_t)s * rows + i;
            const float w = __expf(row_max[r] - m);
            l += w * row_sum[r];
            const f32x4 t = *(reinterpret_cast<const f32x4*>(o_acc + r * D) + c);
            acc += t * w;
        }
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        *(reinterpret_cast<half4*>(o + i * D) + c) = __builtin_convertvector(acc * inv, half4);
    }
}

}  // namespace fa
