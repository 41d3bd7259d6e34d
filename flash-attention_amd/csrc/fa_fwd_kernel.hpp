// FlashAttention-2 forward for gfx950 (MI355X, CDNA4), head dim 128, fp16 in,
// fp32 accumulate, non-causal.  Computes what the reference kernel
// flash_attention_kernel computes (flash_attention_tensor_cores/
// tensor_flash_attention.cu:19-228) with a structure of its own:
//
//  * one wave64 owns 32 query rows; a workgroup is NWAVES such waves;
//  * the wave's Q rows live in registers as MFMA B-operands for the whole kernel;
//  * K/V tiles of 64 keys are staged global -> registers -> LDS (128-bit loads,
//    two LDS buffers, one barrier per tile); the next tile's loads are issued
//    before the current tile's math so their latency hides under the MFMAs;
//  * scores are computed TRANSPOSED, S^T = K.Q^T, with v_mfma_f32_32x32x16_f16:
//    the query index then sits on the lane, so row max / row sum are in-lane
//    reductions plus one exchange between lane l and lane l+32;
//  * the S^T accumulator registers, rounded to fp16, are directly the B operand
//    of O^T += V^T.P^T (no LDS round trip for P); V^T fragments come from the
//    row-major V tile through ds_read_b64_tr_b16;
//  * K rows are XOR-swizzled by (row & 15) 16-byte chunks so the ds_read_b128
//    operand reads are bank-conflict free; V rows are swizzled by (row & 3)
//    64-byte quarters so the transposed reads are.
//
// The per-tile arithmetic follows the CPU baseline (testing.cu:65-123): running
// max per 64-key tile, p = exp(score - max) rounded to fp16 for P.V, row sum
// from the un-rounded p, rescale of the accumulator by exp(old_max - new_max).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's uint4 is a struct

constexpr int kHeadDim = 128;     // D
constexpr int kKvTile = 64;       // keys per streamed tile (the ABI's `tile`)
constexpr int kRowsPerWave = 32;  // query rows owned by one wave
constexpr int kRowBytes = kHeadDim * 2;
constexpr int kTileBytes = kKvTile * kRowBytes;  // 16 KiB
constexpr int kLdsBytes = 4 * kTileBytes;        // K and V, two buffers each
constexpr float kRescaleSlack = 8.0f;            // log2 units a row max may run ahead of its reference

enum : int {
    kFlagLoadState = 1,   // start from (o_acc, row_max, row_sum) instead of zero
    kFlagStoreState = 2,  // write the un-normalised state back instead of fp16 O
};

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

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void fa_fwd_kernel(const FwdParams p) {
    constexpr int NT = NWAVES * 64;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int CHUNKS = (kTileBytes / 16) / NT;  // 16-byte chunks per thread per tile
    static_assert(CHUNKS >= 1, "too many threads for one tile");

    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;   // query row inside the wave / operand row
    const int hh = lane >> 5;  // which half of each k-step this lane feeds

    // ---- workgroup -> (head, query block).  Workgroups are dealt round-robin over the 8 XCDs,
    //      so ids congruent mod 8 share an L2: give all query blocks of one head one residue.
    int head, qb;
    {
        const int bid = blockIdx.x;
        if ((p.bh & 7) == 0) {
            const int j = bid >> 3;
            head = (j / p.q_blocks) * 8 + (bid & 7);
            qb = j % p.q_blocks;
        } else {
            head = bid / p.q_blocks;
            qb = bid % p.q_blocks;
        }
    }
    const size_t q_head_off = (size_t)head * p.lq * kHeadDim;
    const size_t k_head_off = (size_t)head * p.lk * kHeadDim;
    const half_t* __restrict__ qg = p.q + q_head_off;
    const half_t* __restrict__ kg = p.k + k_head_off;
    const half_t* __restrict__ vg = p.v + k_head_off;

    const int q_row0 = qb * QB + wave * kRowsPerWave;
    const bool wave_valid = q_row0 < p.lq;  // wave-uniform; invalid waves still stage K/V
    const int q_row = (wave_valid ? q_row0 : 0) + r;

    // ---- Q fragments: B operand of S^T = K.Q^T.  Lane (r,hh) holds Q[r][16ks+8hh .. +7].
    half8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        qf[ks] = *reinterpret_cast<const half8*>(qg + (size_t)q_row * kHeadDim + ks * 16 + hh * 8);

    // ---- staging geometry: thread t moves 16-byte chunk (t + i*NT) of a tile, i.e. row
    //      (t>>4) + i*RSTEP, chunk t&15.  RSTEP is a multiple of 16, so the swizzle term is the
    //      same for every i and the per-i offsets are compile-time constants.
    constexpr int RSTEP = NT / 16;
    const int st_row = tid >> 4, st_ch = tid & 15;
    const int st_goff = st_row * kHeadDim + st_ch * 8;                          // elements
    const int st_koff = st_row * kRowBytes + ((st_ch ^ (st_row & 15)) << 4);    // bytes in a K buffer
    const int st_voff = st_row * kRowBytes + ((st_ch ^ ((st_row & 3) << 2)) << 4);  // bytes in a V buffer
    u32x4 st_k[CHUNKS], st_v[CHUNKS];

#define FA_STAGE_LOAD(tile)                                                                     \
    do {                                                                                        \
        const half_t* kt = kg + (size_t)(tile) * (kKvTile * kHeadDim) + st_goff;                \
        const half_t* vt = vg + (size_t)(tile) * (kKvTile * kHeadDim) + st_goff;                \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i) {                                    \
            st_k[i] = *reinterpret_cast<const u32x4*>(kt + i * RSTEP * kHeadDim);               \
            st_v[i] = *reinterpret_cast<const u32x4*>(vt + i * RSTEP * kHeadDim);               \
        }                                                                                       \
    } while (0)
#define FA_STAGE_WRITE(buf)                                                                     \
    do {                                                                                        \
        char* kw = smem + (buf) * 2 * kTileBytes;                                               \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i) {                                    \
            *reinterpret_cast<u32x4*>(kw + st_koff + i * RSTEP * kRowBytes) = st_k[i];          \
            *reinterpret_cast<u32x4*>(kw + kTileBytes + st_voff + i * RSTEP * kRowBytes) = st_v[i]; \
        }                                                                                       \
    } while (0)

    // ---- operand read addresses (per lane, tile- and buffer-independent parts)
    // K: row r (and r+32), chunk (2ks+hh) ^ (r&15)
    int k_rd[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r * kRowBytes + (((2 * ks + hh) ^ (r & 15)) << 4);
    // V (transposed read): lane = 16g + 4q + pp supplies row (4hh+q) [+16kk, +8], 8 bytes at
    // chunk (4db + 2(g&1) + (pp>>1)) ^ (q<<2), byte 8(pp&1)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcg = (lane >> 4) & 1;
    int v_rd[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
        v_rd[db] = (4 * hh + tq) * kRowBytes + ((((4 * db) ^ (tq << 2)) + 2 * tcg + (tp >> 1)) << 4) + 8 * (tp & 1);

    // ---- running state.  m is kept in the log2 domain (score * log2(e)/sqrt(D)); l is a per-lane
    //      partial sum over this lane's 32 keys of every tile (lanes l and l+32 are added at the end).
    f32x16 oacc[4];
    float m_run, l_run, m_true;
    if (p.flags & kFlagLoadState) {
        const size_t srow = (size_t)head * p.lq + q_row;
        m_run = p.row_max[srow] * 1.44269504088896340736f;
        m_true = m_run;
        l_run = hh == 0 ? p.row_sum[srow] : 0.0f;
        const float* oa = p.o_acc + srow * kHeadDim;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(oa + 32 * db + 8 * g4 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[db][4 * g4 + e] = t[e];
            }
    } else {
        m_run = -1e30f;
        m_true = -1e30f;
        l_run = 0.0f;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] = 0.0f;
    }

    const int n_tiles = p.lk / kKvTile;
    const float c = p.scale_log2e;

    FA_STAGE_LOAD(0);
    FA_STAGE_WRITE(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) FA_STAGE_LOAD(t + 1);

        const char* kb = smem + buf * 2 * kTileBytes;
        const char* vb = kb + kTileBytes;

        // ---- S^T = K.Q^T : two 32-key blocks, 8 k-steps of 16 over D
        f32x16 s0, s1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = 0.0f;
            s1[e] = 0.0f;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const half8 ka0 = *reinterpret_cast<const half8*>(kb + k_rd[ks]);
            const half8 ka1 = *reinterpret_cast<const half8*>(kb + k_rd[ks] + 32 * kRowBytes);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka1, qf[ks], s1, 0, 0, 0);
        }

        // ---- online softmax for query (lane & 31); this lane holds 32 of the tile's 64 keys.
        //      The running reference m_run only moves when some row's tile max exceeds it by more
        //      than kRescaleSlack (log2 units): p then stays below 2^kRescaleSlack, well inside
        //      fp16, and the 64-register rescale of the accumulator is skipped on most tiles.
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(s0[e], s1[e]));
        mx = half_wave_max(mx) * c;
        m_true = fmaxf(m_true, mx);
        if (!__all(mx - m_run <= kRescaleSlack)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
        }
        float rowsum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], c, -m_run));
            s1[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], c, -m_run));
            rowsum += s0[e] + s1[e];
        }
        l_run += rowsum;

        // P^T fragments: registers 8s..8s+7 of a 32-key block are k-step s of that block, with
        // element j of lane half hh being key 16s + 8(j>>2) + 4hh + (j&3)
        half8 pf[4];
        pf[0] = pack_half8(s0, 0);
        pf[1] = pack_half8(s0, 8);
        pf[2] = pack_half8(s1, 0);
        pf[3] = pack_half8(s1, 8);

        // ---- O^T += V^T.P^T : 4 k-steps of 16 keys x 4 blocks of 32 output columns
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* a0 = vb + v_rd[db] + kk * 16 * kRowBytes;
                const short4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4_t*)(a0));
                const short4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4_t*)(a0 + 8 * kRowBytes));
                const half8 vf = __builtin_bit_cast(
                    half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kk], oacc[db], 0, 0, 0);
            }
        }

        if (t + 1 < n_tiles) FA_STAGE_WRITE(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    if (!wave_valid) return;
    const size_t srow = (size_t)head * p.lq + q_row;
    if (p.flags & kFlagStoreState) {
        // hand the state over relative to the true running max, whatever reference was in use
        const float beta = __builtin_amdgcn_exp2f(m_run - m_true);
        const float l_tot = half_wave_sum(l_run) * beta;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] *= beta;
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
                for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * g4 + e];
                *reinterpret_cast<f32x4*>(oa + 32 * db + 8 * g4 + 4 * hh) = t;
            }
    } else {
        const float inv = 1.0f / half_wave_sum(l_run);
        half_t* og = p.o + srow * kHeadDim;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = oacc[db][4 * g4 + e] * inv;
                *reinterpret_cast<half4*>(og + 32 * db + 8 * g4 + 4 * hh) = __builtin_convertvector(t, half4);
            }
    }
}

#undef FA_STAGE_LOAD
#undef FA_STAGE_WRITE

}  // namespace fa
