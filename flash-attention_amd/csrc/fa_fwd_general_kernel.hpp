// Forward kernel for the shapes the tuned kernels do not take: sequence lengths that are not a
// multiple of 64, different query and key lengths, head dims 64 and 256 beside 128, and the causal
// mask of the CPU baseline (attention_cpu's `causal` argument, flash_attention_tensor_cores/
// testing.cu:74-77: key j > query i takes no part).  The reference's GPU kernel has none of these
// (wrapper.cu:53-61 rejects L % 64 != 0, cuda_utils.cuh:11 fixes D = 128, the kernel has no mask); its
// README lists them as future work (README.md:44-47).
//
// One workgroup of 4 waves per (head, 128-row query block), 32 rows per wave, in the manner of
// fa_fwd_kernel.hpp (transposed scores, P straight from the accumulator, swizzled LDS tiles, tiles
// staged global -> registers -> LDS into two buffers).  What differs:
//   * K/V rows past the end of the sequence are fetched from the last valid row instead (finite data:
//     their scores are masked, and their P is exactly 0, so what V holds there must only not be NaN);
//   * on a tile that needs it (the sequence tail, the causal diagonal) every score whose key lies
//     beyond the row's limit is replaced by -1e30 before the softmax, which is what the CPU baseline
//     does; the limit of row i is min(Lk - 1, i) under the causal mask and Lk - 1 otherwise.  Every row
//     sees key 0 in its first tile, so a row's reference maximum is a real score from then on and a
//     fully masked later tile contributes exp2(-1e30 - m) = 0;
//   * with the causal mask a workgroup stops at the tile that holds its last row's diagonal, and a wave
//     skips the tiles that lie wholly above its own rows (it still helps staging them);
//   * query rows past the end of Q are computed on a copy of the last row and not stored;
//   * blockwise mode (kFlagLoadState / kFlagStoreState, as in the tuned kernels): the carried state
//     (o_acc, row_max, row_sum) is loaded before and stored after the block instead of writing O, and
//     the causal mask compares absolute positions q_pos0 + i and k_pos0 + j, so that a sequence can be
//     folded block by block in any order.  Masked scores are -infinity here rather than the CPU
//     baseline's -1e30: a row that sees no key of a block (or of its first tile) then keeps its state
//     exactly (with -1e30 and a fresh state the baseline counts exp(0) = 1 for every masked key,
//     testing.cu:74-77 with :60,:103; it never meets that case because it only runs whole sequences);
//   * the head dim D is a template parameter: D/16 k-steps per score tile, D/32 output blocks, rows of
//     2D bytes.  The swizzles keep their form with masks that fit the row: K chunk ^ (row & min(15,
//     D/8 - 1)), V chunk ^ ((row & min(3, D/32 - 1)) << 2).  D = 256 needs 128 accumulator registers
//     and 128 KiB of LDS, i.e. one workgroup per CU.
// Measured (pattern data, H = 16): D = 128 without mask 770 / 1024 / 1098 TFLOP/s at L = 511 / 2047 / 16383
// (the tuned kernels: 1083 / 1342 / 1490 at the next multiple of 64), with the causal mask 404 / 709 /
// 1074 TFLOP/s of visible work; D = 64 920 TFLOP/s (L = 2048, 8192), D = 256 728 / 788 TFLOP/s.
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {


template <int D>
struct HeadDim {
    static_assert(D == 64 || D == 128 || D == 256, "head dim");
    static constexpr int kRow = 2 * D;              // bytes per row
    static constexpr int kTile = kKvTile * kRow;    // bytes per 64-row tile
    static constexpr int kSteps = D / 16;           // k-steps of Q.K^T
    static constexpr int kBlocks = D / 32;          // 32-column output blocks
    static constexpr int kChunks = D / 8;           // 16-byte chunks per row
    static constexpr int kKMask = (kChunks < 16 ? kChunks : 16) - 1;
    static constexpr int kVMask = (kBlocks < 4 ? kBlocks : 4) - 1;
};

// Per-lane read offsets into a K tile (operand rows of S^T = K.Q^T) and a V tile (transposed
// fragments), as LaneGeom for any head dim.
template <int D>
struct LaneGeomD {
    int r, hh;
    int k_rd[HeadDim<D>::kSteps];
    int v_rd[HeadDim<D>::kBlocks];
};
template <int D>
__device__ __forceinline__ LaneGeomD<D> make_lane_geom_d(int lane) {
    using HD = HeadDim<D>;
    LaneGeomD<D> g;
    g.r = lane & 31;
    g.hh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < HD::kSteps; ++ks) g.k_rd[ks] = g.r * HD::kRow + (((2 * ks + g.hh) ^ (g.r & HD::kKMask)) << 4);
    // lane = 16g + 4q + pp supplies row (4hh + q) [+ 16kk, + 8], 8 bytes at chunk
    // ((4db) ^ swizzle(row)) + 2(g & 1) + (pp >> 1), byte 8(pp & 1); row & 3 = q
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcg = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < HD::kBlocks; ++db)
        g.v_rd[db] = (4 * g.hh + tq) * HD::kRow +
                     ((((4 * db) ^ ((tq & HD::kVMask) << 2)) + 2 * tcg + (tp >> 1)) << 4) + 8 * (tp & 1);
    return g;
}

template <int D>
__device__ __forceinline__ void qk_tile_d(const char* kb, const LaneGeomD<D>& g, const half8 (&qf)[HeadDim<D>::kSteps],
                                          f32x16& s0, f32x16& s1) {
    using HD = HeadDim<D>;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s0[e] = 0.0f;
        s1[e] = 0.0f;
    }
#pragma unroll
    for (int ks = 0; ks < HD::kSteps; ++ks) {
        const half8 a0 = *reinterpret_cast<const half8*>(kb + g.k_rd[ks]);
        const half8 a1 = *reinterpret_cast<const half8*>(kb + g.k_rd[ks] + 32 * HD::kRow);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[ks], s1, 0, 0, 0);
    }
}

template <int D>
__device__ __forceinline__ void pv_tile_d(const char* vb, const LaneGeomD<D>& g, const half8 (&pf)[4],
                                          f32x16 (&oacc)[HeadDim<D>::kBlocks]) {
    using HD = HeadDim<D>;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int db = 0; db < HD::kBlocks; ++db) {
            const char* a0 = vb + g.v_rd[db] + kk * 16 * HD::kRow;
            const short4_t t0 =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
            const short4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(a0 + 8 * HD::kRow));
            const half8 vf = __builtin_bit_cast(half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kk], oacc[db], 0, 0, 0);
        }
    }
}

#ifndef FA_GEN64_OCC
#define FA_GEN64_OCC 2
#endif
template <int D>
__global__ __launch_bounds__(256, (D == 256 ? 1 : D == 64 ? FA_GEN64_OCC : 2)) void fa_fwd_general_kernel(const FwdParams p) {
    using HD = HeadDim<D>;
    constexpr int NWAVES = 4;
    constexpr int NT = NWAVES * 64;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int CHUNKS = (HD::kTile / 16) / NT;   // 16-byte chunks per thread per tile
    constexpr int RSTEP = NT / HD::kChunks;         // rows between a thread's consecutive chunks
    static_assert(CHUNKS >= 1 && CHUNKS * RSTEP == kKvTile, "staging geometry");

    __shared__ __attribute__((aligned(16))) char smem[4 * HD::kTile];  // [buffer][K tile | V tile]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeomD<D> g = make_lane_geom_d<D>(lane);

    int head, qb;
    decode_item(p, (int)blockIdx.x, head, qb);
    const bool causal = (p.flags & kFlagCausal) != 0;
    // Under the causal mask block qb costs qb + 1 units: hand the long blocks out first, so that the
    // workgroups still running at the end are the short ones.
    if (causal) qb = p.q_blocks - 1 - qb;
    // Padded batches: this head's sequence uses its first lq_v query rows and lk_v keys; p.lq and p.lk
    // remain the strides.  Rows from lq_v on are neither computed nor stored; a sequence without keys
    // gives O = 0.
    int lq_v = p.lq, lk_v = p.lk;
    if (p.q_len || p.k_len) {
        const int batch = head / p.heads_per_batch;
        if (p.q_len) lq_v = min(max(p.q_len[batch], 0), p.lq);
        if (p.k_len) lk_v = min(max(p.k_len[batch], 0), p.lk);
        if (qb * QB >= lq_v) return;  // (uniform over the workgroup, before its first barrier)
    }
    const int q_row0 = qb * QB + wave * kRowsPerWave;       // wave-uniform
    const bool wave_valid = q_row0 < lq_v;
    const int q_abs = min(q_row0 + g.r, lq_v - 1);           // this lane's query row (clamped)
    // key indices are relative to this block of keys; a negative limit means the row sees none of them
    const int shift = p.q_pos0 - p.k_pos0;
    const int wave_last = min(q_row0 + kRowsPerWave - 1, lq_v - 1) + shift;  // last key any row of the wave may see
    const int key_limit = causal ? min(lk_v - 1, q_abs + shift) : lk_v - 1;  // last key this row may see

    int n_tiles = (lk_v + kKvTile - 1) / kKvTile;
    if (causal) n_tiles = max(0, min(n_tiles, (min(qb * QB + QB, lq_v) + shift + kKvTile - 1) / kKvTile));

    const half_t* kh = p.k + (size_t)head * p.lk * D;
    const half_t* vh = p.v + (size_t)head * p.lk * D;

    // staging: thread t moves 16-byte chunk t % kChunks of rows t / kChunks + i * RSTEP of a tile
    const int st_row = tid / HD::kChunks, st_ch = tid % HD::kChunks;
    u32x4 st_k[CHUNKS], st_v[CHUNKS];
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int row = min(tile * kKvTile + st_row + i * RSTEP, lk_v - 1);
            const size_t off = (size_t)row * D + st_ch * 8;
            st_k[i] = *reinterpret_cast<const u32x4*>(kh + off);
            st_v[i] = *reinterpret_cast<const u32x4*>(vh + off);
        }
    };
    auto stage_write = [&](int buf) {
        char* kw = smem + buf * 2 * HD::kTile;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int row = st_row + i * RSTEP;
            *reinterpret_cast<u32x4*>(kw + row * HD::kRow + ((st_ch ^ (row & HD::kKMask)) << 4)) = st_k[i];
            *reinterpret_cast<u32x4*>(kw + HD::kTile + row * HD::kRow + ((st_ch ^ ((row & HD::kVMask) << 2)) << 4)) =
                st_v[i];
        }
    };

    // Q fragments: B operand of S^T = K.Q^T, lane (r, hh) holds Q[row][16ks + 8hh .. + 7]
    half8 qf[HD::kSteps];
    {
        const half_t* qr = p.q + ((size_t)head * p.lq + q_abs) * D + g.hh * 8;
#pragma unroll
        for (int ks = 0; ks < HD::kSteps; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qr + ks * 16);
    }

    // running state: m in the log2 domain, l a per-lane partial sum (lanes r and r + 32 are added at the end)
    f32x16 oacc[HD::kBlocks];
    float m_run = -1e30f, m_true = -1e30f, l_run = 0.0f;
    const size_t srow = (size_t)head * p.lq + q_abs;
    if (p.flags & kFlagLoadState) {
        m_run = m_true = p.row_max[srow] * 1.44269504088896340736f;
        l_run = g.hh == 0 ? p.row_sum[srow] : 0.0f;
        const float* oa = p.o_acc + srow * D;
#pragma unroll
        for (int db = 0; db < HD::kBlocks; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(oa + 32 * db + 8 * g4 + 4 * g.hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[db][4 * g4 + e] = t[e];
            }
    } else {
#pragma unroll
        for (int db = 0; db < HD::kBlocks; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] = 0.0f;
    }

    if (n_tiles > 0) {
        stage_load(0);
        stage_write(0);
    }
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) stage_load(t + 1);
        const int key0 = t * kKvTile;
        if (wave_valid && !(causal && key0 > wave_last)) {
            const char* kb = smem + buf * 2 * HD::kTile;
            f32x16 s0, s1;
            half8 pf[4];
            qk_tile_d<D>(kb, g, qf, s0, s1);
            // element e of s0 is key key0 + 8(e >> 2) + 4hh + (e & 3), of s1 the same + 32
            const int lim = key_limit - key0 - 4 * g.hh;
            if (__any(lim < 59)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = 8 * (e >> 2) + (e & 3);
                    s0[e] = (kr <= lim) ? s0[e] : -__builtin_inff();
                    s1[e] = (kr + 32 <= lim) ? s1[e] : -__builtin_inff();
                }
            }
            softmax_tile(s0, s1, p.scale_log2e, m_run, m_true, l_run, oacc, pf);
            pv_tile_d<D>(kb + HD::kTile, g, pf, oacc);
        }
        if (t + 1 < n_tiles) stage_write(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    const bool row_valid = wave_valid && q_row0 + g.r < lq_v;
    if (p.flags & kFlagStoreState) {
        // hand the state over relative to the true running max, whatever reference was in use
        const float beta = __builtin_amdgcn_exp2f(m_run - m_true);
        const float l_tot = half_wave_sum(l_run) * beta;
        if (row_valid) {
            if (g.hh == 0) {
                p.row_max[srow] = m_true * 0.69314718055994530942f;
                p.row_sum[srow] = l_tot;
            }
            float* oa = p.o_acc + srow * D;
#pragma unroll
            for (int db = 0; db < HD::kBlocks; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * g4 + e] * beta;
                    *reinterpret_cast<f32x4*>(oa + 32 * db + 8 * g4 + 4 * g.hh) = t;
                }
        }
        return;
    }
    // O = acc / l: lane (r, hh) holds columns 32db + 8g4 + 4hh .. +3 of row r
    const float l_tot = half_wave_sum(l_run);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;  // a row that has seen no key (every score masked): O = 0
    if (row_valid) {
        half_t* og = p.o + srow * D;
#pragma unroll
        for (int db = 0; db < HD::kBlocks; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[db][4 * g4 + e] * inv;
                *reinterpret_cast<half4*>(og + 32 * db + 8 * g4 + 4 * g.hh) = __builtin_convertvector(v, half4);
            }
    }
}

}  // namespace fa
