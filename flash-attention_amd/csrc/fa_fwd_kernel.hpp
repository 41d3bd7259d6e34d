// Building blocks of the forward-attention kernels for gfx950 (MI355X, CDNA4), head dim 128, fp16 in, fp32
// accumulate: what the reference kernel flash_attention_kernel computes (flash_attention_tensor_cores/
// tensor_flash_attention.cu:19-228) with a structure of its own, shared by fa_fwd_ring_kernel.hpp,
// fa_fwd_w64_kernel.hpp and (comparison builds) fa_fwd_staged_kernel.hpp:
//
//  * one wave64 owns 32 query rows (or two such sub-blocks);
//  * the wave's Q rows live in registers as MFMA B-operands for a whole work item;
//  * scores are computed TRANSPOSED, S^T = K.Q^T, with v_mfma_f32_32x32x16_f16:
//    the query index then sits on the lane, so row max / row sum are in-lane
//    reductions plus one exchange between lane l and lane l+32;
//  * the S^T accumulator registers, rounded to fp16, are directly the B operand
//    of O^T += V^T.P^T (no LDS round trip for P); V^T fragments come from the
//    row-major V tile through ds_read_b64_tr_b16;
//  * K rows are XOR-swizzled by (row & 15) 16-byte chunks so the ds_read_b128
//    operand reads are bank-conflict free; V rows are swizzled by (row & 3)
//    64-byte quarters so the transposed reads are;
//  * a per-wave LDS staging image through which Q rows come in and outputs (or the carried state of the
//    blockwise mode) go out as whole rows.
//
// The per-tile arithmetic follows the CPU baseline (testing.cu:65-123): running
// max per 64-key tile, p = exp(score - max) rounded to fp16 for P.V, row sum
// from the un-rounded p, rescale of the accumulator by exp(old_max - new_max).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace fa {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's uint4 is a struct
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), ...
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kHeadDim = 128;     // D
constexpr int kKvTile = 64;       // keys per streamed tile (the ABI's `tile`)
constexpr int kRowsPerWave = 32;  // query rows owned by one wave
constexpr int kRowBytes = kHeadDim * 2;
constexpr int kTileBytes = kKvTile * kRowBytes;  // 16 KiB
constexpr int kLdsBytes = 4 * kTileBytes;        // K and V, two buffers each (the per-wave staging images follow)
constexpr float kRescaleSlack = 8.0f;
// Cache-policy bits of the output rows' global stores (store_staged_rows; see there).
#ifndef FA_STORE_MODS
#define FA_STORE_MODS "sc1 nt"
#endif            // log2 units a row max may run ahead of its reference

enum : int {
    kFlagLoadState = 1,   // start from (o_acc, row_max, row_sum) instead of zero
    kFlagStoreState = 2,  // write the un-normalised state back instead of fp16 O
    kFlagCausal = 4,      // key k_pos0 + j > query q_pos0 + i is masked (the kernels that know the mask)
};

// Padded batches at D = 128: which sequences run on the tuned kernel's list form (fa_fwd_w64_varlen kernels) and which on the
// 128-row kernel launched beside it; both kernels and the list's planner ask this.
//   Without the mask, long_q = kVarlenByCost: by what the sequence costs on either kernel per head, in microseconds of
// one compute unit (the launcher's models of the two kernels, fitted to tools/micro/ragged_rule.py): 256-row items of
// 1.33 us per 64-key tile plus 3.6 us each, against 128-row workgroups, two to a unit, of 0.85 us per 32-key tile.
// A last block that is mostly empty weighs on the 256-row items, short key ranges on their fixed cost: 640 rows
// and keys are a tie, 700 go to the tuned kernel, 513 do not.  Any other long_q: at least long_q query rows and
// long_k keys.
//   With the mask: the same count of rows and keys (the tuned kernels' mask assumes it), and at least long_q of
// them when the last 256-row block holds more than 128 rows, long_k otherwise (a last block that is mostly idle
// weighs more on the 256-row items); with the standing thresholds also from 960 rows on when the last block
// holds at least 192 (as the launcher's rule for single sequences).
constexpr int kVarlenByCost = -1;
__host__ __device__ inline bool varlen_on_tuned(int ql, int kl, int long_q, int long_k, bool causal) {
    if (!causal) {
        if (long_q != kVarlenByCost) return ql >= long_q && kl >= long_k;
        if (ql < 64 || kl < 192) return false;
        const float tuned = (float)((ql + 255) / 256) * (1.33f * (float)((kl + 63) / 64) + 3.6f);
        const float rows128 = 0.5f * (float)((ql + 127) / 128) * (0.8535f * (float)((kl + 31) / 32) + 0.094f);
        return tuned < rows128;
    }
    if (ql != kl) return false;
    const int last_rows = ((ql - 1) & 255) + 1;
    if (long_q == 1665 && last_rows >= 192 && ql >= 960) return true;
    return ql >= (last_rows > 128 ? long_q : long_k);
}

struct FwdParams {
    const half_t* q;  // [bh][lq][128]
    const half_t* k;  // [bh][lk][128]
    const half_t* v;  // [bh][lk][128]
    half_t* o;        // [bh][lq][128]           (written unless kFlagStoreState)
    float* o_acc;     // [bh][lq][128] fp32      (blockwise mode)
    float* row_max;   // [bh][lq] running max of score/sqrt(D)
    float* row_sum;   // [bh][lq] running sum of exp(score/sqrt(D) - max)
    int bh;           // B*H independent heads
    int lq;           // query rows per head   (multiple of 64)
    int lk;           // key rows per head     (multiple of 64)
    int q_blocks;     // ceil(lq / (NWAVES*32))
    float scale_log2e;  // log2(e)/sqrt(D)
    int flags;
    unsigned long long* stamps;  // diagnostic builds only (FA_STAMPS): 32 time stamps per workgroup
    int q_pos0, k_pos0;  // absolute position of query row 0 / key row 0 (causal mask of fa_fwd_general_kernel only)
    int k_splits, lk_split, bh_real;  // fa_fwd_w64_split_kernel only: key ranges per head, keys per range, real heads
    // fa_fwd_general_kernel only (padded batches): per-batch valid lengths, device arrays of bh / heads_per_batch
    // entries or null (= lq / lk for every batch); lq and lk stay the padded lengths, i.e. the strides
    const int* q_len;
    const int* k_len;
    int heads_per_batch;
    // Padded batches at D = 128, long sequences on the tuned kernel (fa_fwd_w64_varlen_kernel): a device-built list of
    // work items, four ints each (head, 256-row block, the sequence's query rows, its keys), and their number; the
    // 128-row kernel, launched beside it for the batch's other sequences, leaves alone every sequence with at least
    // long_q query rows and long_k keys (0: none)
    const int* varlen_items;
    const int* varlen_count;
    int long_q, long_k;
};

// v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second.
// Fed the same value twice, the first result holds the low half-wave's value in every lane and the
// second the high half-wave's, so a commutative combine needs no select.
// (The results are copied to scalars before the bit casts: __builtin_bit_cast applied directly to
// an element of the returned vector reads element 0 whatever the index.)
__device__ __forceinline__ void half_wave_pair(float x, float& lo, float& hi) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned u0 = r[0], u1 = r[1];
    lo = __builtin_bit_cast(float, u0);
    hi = __builtin_bit_cast(float, u1);
}
__device__ __forceinline__ float half_wave_max(float x) {
    float lo, hi;
    half_wave_pair(x, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_wave_sum(float x) {
    float lo, hi;
    half_wave_pair(x, lo, hi);
    return lo + hi;
}

__device__ __forceinline__ half8 pack_half8(const f32x16& s, int base) {
    f32x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = s[base + i];
    return __builtin_convertvector(t, half8);
}

// ---------------------------------------------------------------------------------------------
// Per-lane constants shared by the tile routines.
// ---------------------------------------------------------------------------------------------
struct LaneGeom {
    int r;        // lane & 31: query row inside the wave / operand row
    int hh;       // lane >> 5: which half of each k-step this lane feeds
    int k_rd[8];  // K operand read: row r, chunk (2ks+hh) ^ (r&15)
    int v_rd[4];  // V transposed read, per 32-column output block
};

__device__ __forceinline__ LaneGeom make_lane_geom(int lane) {
    LaneGeom g;
    g.r = lane & 31;
    g.hh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) g.k_rd[ks] = g.r * kRowBytes + (((2 * ks + g.hh) ^ (g.r & 15)) << 4);
    // lane = 16g + 4q + pp supplies row (4hh+q) [+16kk, +8], 8 bytes at chunk
    // (4db + 2(g&1) + (pp>>1)) ^ (q<<2), byte 8(pp&1)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcg = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 4; ++db)
        g.v_rd[db] =
            (4 * g.hh + tq) * kRowBytes + ((((4 * db) ^ (tq << 2)) + 2 * tcg + (tp >> 1)) << 4) + 8 * (tp & 1);
    return g;
}

// Q fragments: B operand of S^T = K.Q^T.  Lane (r,hh) holds Q[row][16ks+8hh .. +7].
__device__ __forceinline__ void load_q_frags(half8 (&qf)[8], const half_t* q_row_ptr, int hh) {
#pragma unroll
#ifdef FA_ABLATE_CHEAP_Q
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const half8*>(q_row_ptr + hh * 8);
#else
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const half8*>(q_row_ptr + ks * 16 + hh * 8);
#endif
}

// S^T = K.Q^T : two 32-key blocks, 8 k-steps of 16 over D.  The K fragments run FA_K_AHEAD k-steps
// ahead of the MFMAs that consume them (left alone the compiler reads each pair right before its
// use and every MFMA pair waits out a full LDS round trip); the sched_group_barrier sequence pins
// that order.
#ifndef FA_K_AHEAD
#define FA_K_AHEAD 4
#endif
__device__ __forceinline__ void qk_tile(const char* kb, const LaneGeom& g, const half8 (&qf)[8], f32x16& s0,
                                        f32x16& s1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s0[e] = 0.0f;
        s1[e] = 0.0f;
    }
    half8 ka0[FA_K_AHEAD], ka1[FA_K_AHEAD];
#pragma unroll
    for (int i = 0; i < FA_K_AHEAD; ++i) {
        ka0[i] = *reinterpret_cast<const half8*>(kb + g.k_rd[i]);
        ka1[i] = *reinterpret_cast<const half8*>(kb + g.k_rd[i] + 32 * kRowBytes);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka0[ks % FA_K_AHEAD], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka1[ks % FA_K_AHEAD], qf[ks], s1, 0, 0, 0);
        if (ks + FA_K_AHEAD < 8) {
            ka0[ks % FA_K_AHEAD] = *reinterpret_cast<const half8*>(kb + g.k_rd[ks + FA_K_AHEAD]);
            ka1[ks % FA_K_AHEAD] = *reinterpret_cast<const half8*>(kb + g.k_rd[ks + FA_K_AHEAD] + 32 * kRowBytes);
        }
    }
#ifndef FA_NO_SCHED_HINTS
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * FA_K_AHEAD, 0);  // DS_READ
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
        if (ks + FA_K_AHEAD < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#endif
}

// Online softmax for query (lane & 31); this lane holds 32 of the tile's 64 keys.  The running
// reference m_run only moves when some row's tile max exceeds it by more than kRescaleSlack (log2
// units): p then stays below 2^kRescaleSlack, well inside fp16, and the 64-register rescale of the
// accumulator is skipped on most tiles.  On return s0/s1 hold p and pf the fp16 P^T fragments:
// registers 8s..8s+7 of a 32-key block are k-step s of that block, element j of lane half hh being
// key 16s + 8(j>>2) + 4hh + (j&3).
template <int DB>  // DB: 32-column blocks of the output accumulator (head dim / 32)
__device__ __forceinline__ void softmax_tile(f32x16& s0, f32x16& s1, float c, float& m_run, float& m_true,
                                             float& l_run, f32x16 (&oacc)[DB], half8 (&pf)[4]) {
#ifdef FA_ABLATE_SOFTMAX  // timing experiments only: wrong results
    pf[0] = pack_half8(s0, 0);
    pf[1] = pack_half8(s0, 8);
    pf[2] = pack_half8(s1, 0);
    pf[3] = pack_half8(s1, 8);
    return;
#endif
    // two chains of max(max(a, x), y), each link one v_max3_f32
    float mx0 = fmaxf(s0[0], s0[1]), mx1 = fmaxf(s1[0], s1[1]);
#pragma unroll
    for (int e = 2; e < 16; e += 2) {
        mx0 = fmaxf(fmaxf(mx0, s0[e]), s0[e + 1]);
        mx1 = fmaxf(fmaxf(mx1, s1[e]), s1[e + 1]);
    }
    const float mx = half_wave_max(fmaxf(mx0, mx1)) * c;
    m_true = fmaxf(m_true, mx);
    if (!__all(mx - m_run <= kRescaleSlack)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
    }
    // score*c - m and the row sum go through the packed fp32 forms (v_pk_fma_f32, v_pk_add_f32):
    // two elements per instruction
    const f32x2 c2 = {c, c}, nm2 = {-m_run, -m_run};
    f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        f32x2 a = {s0[e], s0[e + 1]}, b = {s1[e], s1[e + 1]};
        a = __builtin_elementwise_fma(a, c2, nm2);
        b = __builtin_elementwise_fma(b, c2, nm2);
        a[0] = __builtin_amdgcn_exp2f(a[0]);
        a[1] = __builtin_amdgcn_exp2f(a[1]);
        b[0] = __builtin_amdgcn_exp2f(b[0]);
        b[1] = __builtin_amdgcn_exp2f(b[1]);
        sum2 += a;
        sum2 += b;
        s0[e] = a[0];
        s0[e + 1] = a[1];
        s1[e] = b[0];
        s1[e + 1] = b[1];
    }
    l_run += sum2[0] + sum2[1];
    pf[0] = pack_half8(s0, 0);
    pf[1] = pack_half8(s0, 8);
    pf[2] = pack_half8(s1, 0);
    pf[3] = pack_half8(s1, 8);
}

// O^T += V^T.P^T : 4 k-steps of 16 keys x 4 blocks of 32 output columns
__device__ __forceinline__ void pv_tile(const char* vb, const LaneGeom& g, const half8 (&pf)[4], f32x16 (&oacc)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const char* a0 = vb + g.v_rd[db] + kk * 16 * kRowBytes;
            const short4_t t0 =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
            const short4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(a0 + 8 * kRowBytes));
            const half8 vf = __builtin_bit_cast(half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kk], oacc[db], 0, 0, 0);
        }
    }
}

// Running state.  m is kept in the log2 domain (score * log2(e)/sqrt(D)); l is a per-lane partial sum
// over this lane's 32 keys of every tile (lanes l and l+32 are added at the end).
__device__ __forceinline__ void init_state(const FwdParams& p, size_t srow, int hh, f32x16 (&oacc)[4],
                                           float& m_run, float& m_true, float& l_run) {
    if (p.flags & kFlagLoadState) {
        float mx = p.row_max[srow], sm = p.row_sum[srow];
        const float* oa = p.o_acc + srow * kHeadDim;
        f32x4 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = *reinterpret_cast<const f32x4*>(oa + 32 * (i >> 2) + 8 * (i & 3) + 4 * hh);
        // Wait for these loads HERE.  Left pending, the accumulator registers would carry "may be
        // in flight" into the tile code, and the waits the compiler then puts in front of the first
        // uses would also drain the tile and Q prefetches issued in between.
        asm volatile("" : "+v"(mx), "+v"(sm));
#pragma unroll
        for (int i = 0; i < 16; i += 4) asm volatile("" : "+v"(t[i]), "+v"(t[i + 1]), "+v"(t[i + 2]), "+v"(t[i + 3]));
        m_run = mx * 1.44269504088896340736f;
        m_true = m_run;
        l_run = hh == 0 ? sm : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[i >> 2][4 * (i & 3) + e] = t[i][e];
    } else {
        m_run = -1e30f;
        m_true = -1e30f;
        l_run = 0.0f;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] = 0.0f;
    }
}

// Lane (q = lane&31, hh) holds output row q, columns 32db + 8g4 + 4hh + (0..3) in oacc[db][4g4..4g4+3].
__device__ __forceinline__ void finish_rows(const FwdParams& p, size_t srow, int hh, f32x16 (&oacc)[4],
                                            float m_run, float m_true, float l_run) {
    if (p.flags & kFlagStoreState) {
        // hand the state over relative to the true running max, whatever reference was in use
        const float beta = __builtin_amdgcn_exp2f(m_run - m_true);
        const float l_tot = half_wave_sum(l_run) * beta;
        if (hh == 0) {
            p.row_max[srow] = m_true * 0.69314718055994530942f;
            p.row_sum[srow] = l_tot;
        }
        float* oa = p.o_acc + srow * kHeadDim;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * g4 + e] * beta;
                *reinterpret_cast<f32x4*>(oa + 32 * db + 8 * g4 + 4 * hh) = t;
            }
    } else {
        const float l_tot = half_wave_sum(l_run);
        const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;  // a row that has seen no key: O = 0, not 0/0
        half_t* og = p.o + srow * kHeadDim;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * g4 + e] * inv;
#ifdef FA_ABLATE_NO_STORE
                asm volatile("" ::"v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
#else
                *reinterpret_cast<half4*>(og + 32 * db + 8 * g4 + 4 * hh) = __builtin_convertvector(t, half4);
#endif
            }
    }
}

// Work item -> (head, query block).  Workgroups are dealt round-robin over the 8 XCDs, so ids
// congruent mod 8 share an L2: all query blocks of one head get one residue class, and the head's
// K/V are re-read from that L2.
__device__ __forceinline__ void decode_item(const FwdParams& p, int item, int& head, int& qb) {
    if ((p.bh & 7) == 0) {
        const int j = item >> 3;
        head = (j / p.q_blocks) * 8 + (item & 7);
        qb = j % p.q_blocks;
    } else {
        head = item / p.q_blocks;
        qb = item % p.q_blocks;
    }
}

// Per-wave LDS staging image (32 rows x 256 B, chunk c of row q at q*256 + ((c ^ (q&15)) << 4)):
// the same image serves the coalesced write-out of an output tile and the coalesced read-in of the
// next item's Q rows.  In both directions one side touches it fragment-wise (lane = row) and the
// other row-wise (16 lanes per row, 4 whole rows per wave instruction); the XOR keeps both free of
// bank conflicts.  Only the owning wave touches its image, and a wave's LDS operations execute in
// order, so no barrier is involved.
__device__ __forceinline__ int stage_img_off(int row, int chunk) { return row * kRowBytes + ((chunk ^ (row & 15)) << 4); }

// The item-boundary code runs once per item but sits inside the loop over items, so the optimiser
// would hoist its two dozen lane-constant addresses out and keep them in registers across the tile
// loop, which has none to spare.  Passing the lane id through an empty asm makes them recomputed
// where they are used.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// How the row-wise helpers below reach the 64 elements of a sub-block's O^T (element 16 db + e): as a value
// array here; the one-wave-per-SIMD kernel passes an accessor over the registers its inline asm owns, so that
// no 64-register copy of the accumulator exists while the whole tile loop's state is live.
struct ArrayAcc {
    f32x16 (&o)[4];
    template <int I>
    __device__ __forceinline__ float get() const { return o[I >> 4][I & 15]; }
    template <int I>
    __device__ __forceinline__ void put(float x) const { o[I >> 4][I & 15] = x; }
};
__device__ __forceinline__ void store_staged_rows(half_t* o_rows, const char* img, int lane, int first_row);
// O = acc / l as fp16, written through the staging image so that global stores are whole rows.
// All eight row-wise reads are issued before the first store (the accumulators are dead by then, so
// registers are plentiful); addresses are one base plus immediates: (4i+row0)&15 = 4(i&3) + row0, so
// the swizzled chunk of read i is (ch ^ row0) ^ 4(i&3).
// Rows below first_row (of the 32) are not stored.
template <typename ACC>
__device__ __forceinline__ void store_output_rows_acc(half_t* o_rows, char* img, int lane, const ACC& acc, float l_run,
                                                      int first_row = 0) {
    const float l_tot = half_wave_sum(l_run);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;  // a row that has seen no key: O = 0, not 0/0
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
    static_for<16>([&](auto ic) {
        constexpr int cidx = decltype(ic)::value;  // 4 db + g4
        const f32x4 t = {acc.template get<4 * cidx>() * inv, acc.template get<4 * cidx + 1>() * inv,
                         acc.template get<4 * cidx + 2>() * inv, acc.template get<4 * cidx + 3>() * inv};
        // columns 32db + 8g4 + 4hh .. +3  =  chunk 4db+g4, second half when hh
        *reinterpret_cast<half4*>(img + stage_img_off(r, cidx) + 8 * hh) = __builtin_convertvector(t, half4);
    });
    // (written as half4, read back as u32x4: accesses of different types, which the compiler may reorder)
    asm volatile("" ::: "memory");
    store_staged_rows(o_rows, img, lane, first_row);
}
__device__ __forceinline__ void store_output_rows(half_t* o_rows, char* img, int lane, f32x16 (&oacc)[4],
                                                  float l_run, int first_row = 0) {
    store_output_rows_acc(o_rows, img, lane, ArrayAcc{oacc}, l_run, first_row);
}
// The second half of the above: 32 fp16 rows that sit in the staging image go out as whole rows.
__device__ __forceinline__ void store_staged_rows(half_t* o_rows, const char* img, int lane, int first_row) {
    lane = opaque(lane);
    const int row0 = lane >> 4, ch = lane & 15;
    const char* rd = img + row0 * kRowBytes;
    int sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sw[j] = ((ch ^ row0) ^ (4 * j)) << 4;
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(rd + i * 4 * kRowBytes + sw[i & 3]);
    half_t* wr = o_rows + row0 * kHeadDim + ch * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#ifdef FA_ABLATE_NO_STORE
        asm volatile("" ::"v"(v[i]));
#else
        // Device-scope, non-temporal store ("sc1 nt"): the rows go through the XCD's L2 to memory as they are
        // written.  A plain store leaves the output dirty in the L2 (32 MiB across the chip, half of a 64 MiB
        // output), and the write-back at the end of the kernel then delays the next launch: measured at L = 512,
        // 6.7 us between consecutive launches, 4.6 us with the non-temporal hint alone (lines marked for early
        // eviction, still written back), and a little less again with the device scope on top (tools/ab.py, one
        // process, four boxes: 65.3 -> 63.4, 63.2 -> 62.0, 62.9 -> 62.5 and 62.2 -> 61.9 us per launch, i.e. 0.5 - 3 %; on one
        // time axis the time between two launches falls from 4.5 to 4.1 us and the rest is inside the spread;
        // "sc1", "sc0 sc1" and "sc0 sc1 nt" within 0.5 us of it, plain stores 67.2; nothing measurable from
        // L = 1024 on or on N(0,1) data; same bits).  -DFA_STORE_MODS="..." builds another policy for comparison.
        if (first_row <= 0 || 4 * i + row0 >= first_row) {
            const half_t* base_ = wr + (i >> 2) * 16 * kHeadDim;
            if ((i & 3) == 0) asm volatile("global_store_dwordx4 %0, %1, off " FA_STORE_MODS : : "v"(base_), "v"(v[i]) : "memory");
            if ((i & 3) == 1) asm volatile("global_store_dwordx4 %0, %1, off offset:1024 " FA_STORE_MODS : : "v"(base_), "v"(v[i]) : "memory");
            if ((i & 3) == 2) asm volatile("global_store_dwordx4 %0, %1, off offset:2048 " FA_STORE_MODS : : "v"(base_), "v"(v[i]) : "memory");
            if ((i & 3) == 3) asm volatile("global_store_dwordx4 %0, %1, off offset:3072 " FA_STORE_MODS : : "v"(base_), "v"(v[i]) : "memory");
        }
#endif
    }
}

// The carried state of 32 rows (blockwise mode, key-split partials) through the same per-wave image, so that
// global memory sees 256-byte pieces of whole rows instead of the 16-byte pieces a lane holds (lane = row:
// 32 rows x 32 bytes per instruction).  An fp32 row is 512 bytes, twice the image's row, so the accumulator
// goes in two column halves of 64; in each the chunk index 8 (db & 1) + 2 g4 + hh takes the place of the
// fp16 output's 4 db + g4.  `row0` is the first of the 32 rows; rows below first_row are not stored.
template <typename ACC>
__device__ __forceinline__ void store_state_rows(const FwdParams& p, size_t row0, char* img, int lane,
                                                 const ACC& acc, float m_run, float m_true, float l_run,
                                                 int first_row = 0) {
    // hand the state over relative to the true running max, whatever reference was in use
    const float beta = __builtin_amdgcn_exp2f(m_run - m_true);
    const float l_tot = half_wave_sum(l_run) * beta;
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
    if (hh == 0 && r >= first_row) {
        p.row_max[row0 + r] = m_true * 0.69314718055994530942f;
        p.row_sum[row0 + r] = l_tot;
    }
    const int rrow = lane >> 4, ch = lane & 15;
    const char* rd = img + rrow * kRowBytes;
    int sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sw[j] = ((ch ^ rrow) ^ (4 * j)) << 4;
    float* wr = p.o_acc + (row0 + rrow) * kHeadDim + ch * 4;
    static_for<2>([&](auto hc) {
        constexpr int half = decltype(hc)::value;
        static_for<8>([&](auto ic) {
            constexpr int dbl = decltype(ic)::value >> 2, g4 = decltype(ic)::value & 3;
            constexpr int i0 = 16 * (2 * half + dbl) + 4 * g4;
            const f32x4 t = {acc.template get<i0>() * beta, acc.template get<i0 + 1>() * beta,
                             acc.template get<i0 + 2>() * beta, acc.template get<i0 + 3>() * beta};
            *reinterpret_cast<f32x4*>(img + stage_img_off(r, 8 * dbl + 2 * g4 + hh)) = t;
        });
        asm volatile("" ::: "memory");  // (f32x4 writes, u32x4 reads: keep them in this order)
        // (four rows-of-four at a time: this runs with the whole tile loop's state live)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(rd + (4 * grp + j) * 4 * kRowBytes + sw[j]);
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (first_row <= 0 || 4 * (4 * grp + j) + rrow >= first_row)
                    *reinterpret_cast<u32x4*>(wr + (4 * grp + j) * 4 * kHeadDim + half * 64) = v[j];
        }
    });
}
// The way back: (o_acc, row_max, row_sum) of 32 rows into a sub-block's registers (what init_state does lane by
// lane), row pieces of 256 bytes -> image -> fragments.
template <typename ACC>
__device__ __forceinline__ void load_state_rows(const FwdParams& p, size_t row0, char* img, int lane,
                                                const ACC& acc, float& m_run, float& m_true, float& l_run) {
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
    const float mx = p.row_max[row0 + r], sm = p.row_sum[row0 + r];
    const int rrow = lane >> 4, ch = lane & 15;
    char* wr = img + rrow * kRowBytes;
    int sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sw[j] = ((ch ^ rrow) ^ (4 * j)) << 4;
    const float* rd = p.o_acc + (row0 + rrow) * kHeadDim + ch * 4;
    static_for<2>([&](auto hc) {
        constexpr int half = decltype(hc)::value;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(rd + (4 * grp + j) * 4 * kHeadDim + half * 64);
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(wr + (4 * grp + j) * 4 * kRowBytes + sw[j]) = v[j];
        }
        asm volatile("" ::: "memory");  // (u32x4 writes, f32x4 reads: keep them in this order)
        static_for<8>([&](auto ic) {
            constexpr int dbl = decltype(ic)::value >> 2, g4 = decltype(ic)::value & 3;
            constexpr int i0 = 16 * (2 * half + dbl) + 4 * g4;
            const f32x4 t = *reinterpret_cast<const f32x4*>(img + stage_img_off(r, 8 * dbl + 2 * g4 + hh));
            acc.template put<i0>(t[0]);
            acc.template put<i0 + 1>(t[1]);
            acc.template put<i0 + 2>(t[2]);
            acc.template put<i0 + 3>(t[3]);
        });
    });
    m_run = mx * 1.44269504088896340736f;
    m_true = m_run;
    l_run = hh == 0 ? sm : 0.0f;
}

// Next item's Q rows travel global -> LDS directly (global_load_lds_dwordx4: no registers, 4 whole
// rows per wave instruction).  The LDS side of such a load is linear (piece base + 16*lane), so the
// image's XOR swizzle is applied to the SOURCE address: lane l of piece i fills slot l&15 of row
// 4i + (l>>4) with chunk (l&15) ^ (row&15).
__device__ __forceinline__ void dma_q_rows(const half_t* q_rows, char* img, int lane) {
    lane = opaque(lane);
    const int row0 = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + row0;
        const half_t* src = q_rows + row * kHeadDim + ((slot ^ (row & 15)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(img + i * 1024), 16, 0, 0);
    }
}
// They are picked up as MFMA B-operand fragments: lane (r,hh) takes chunk 2ks+hh of row r.
__device__ __forceinline__ void fetch_q_frags(half8 (&qf)[8], const char* img, int lane) {
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const half8*>(img + stage_img_off(r, 2 * ks + hh));
}

}  // namespace fa
