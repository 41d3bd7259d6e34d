// Forward kernel for 8-wave workgroups whose K/V tiles reach LDS by DMA into a three-stage ring.
//
// fa_fwd_kernel moves every tile global -> registers -> LDS: 16 staging registers per lane, eight
// ds_write_b128 per wave and tile on the LDS store path (which is much slower than the read path),
// and the stores, the barrier and the first operand reads of the next tile all sit between the last
// MFMA of one tile and the first MFMA of the next.  Here
//   * a tile travels global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, the
//     image's XOR swizzle applied on the source side); no registers, no LDS store instructions;
//   * the ring has three stages (K tile + V tile, 32 KiB each).  During step g the waves read tile g,
//     tile g+1 is already complete, and tile g+2 is in flight into the stage tile g-1 occupied.
//     One workgroup barrier per step, at its end, serves both directions: every wave has finished
//     tile g (so its stage may be refilled in step g+1) and has waited for its own share of tile g+2
//     (so every wave may read that tile from step g+1 on);
//   * because tile g+1 was published by the PREVIOUS barrier, a wave reads its first K fragments of
//     tile g+1 before it enters the barrier: they land while it waits, and the first MFMA of the next
//     step issues as soon as the barrier opens.
//
// The DMA instructions are issued through inline asm.  A DMA the compiler knows of is drained
// (s_waitcnt vmcnt(0)) in front of every barrier and every LDS read that might alias it; these have
// to stay in flight across a whole step of LDS reads.  Completion is awaited by hand at the end of
// the step (s_waitcnt vmcnt(0): nothing younger than the step's DMA needs to stay in flight).
// Vector-memory operations complete in order, so the compiler's own counted waits, which do not
// count the hidden instructions, can only wait longer than needed, never shorter.
//
// The stream of work items, the Q staging image, the folded write-out and the deferred rescale are
// those of fa_fwd_kernel.
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {

constexpr int kStageBytes = 2 * kTileBytes;  // K tile, then V tile
#ifndef FA_RING_STAGES
#define FA_RING_STAGES 3
#endif
constexpr int kRingStages = FA_RING_STAGES;

// LDS-DMA of one 1 KiB piece (4 rows of 256 B): the LDS base goes to m0 (dma_set_base), then lane l
// fills bytes 16l..16l+15 at m0 + OFS with the 16 bytes at base + off + OFS; the compile-time offset
// is applied by the instruction to BOTH sides.  m0 is left changed: the kernels that use these have
// no other use for it.
__device__ __forceinline__ void dma_set_base(unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds_addr) : "memory");
}
template <int OFS>
__device__ __forceinline__ void dma_piece_at(const half_t* base, unsigned off_bytes) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(off_bytes), "s"(base), "n"(OFS) : "memory");
}
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Q rows of the next item -> this wave's staging image (as dma_q_rows, hidden from the compiler).
__device__ __forceinline__ void dma_q_rows_asm(const half_t* q_rows, char* img, int lane) {
    lane = opaque(lane);
    const int row0 = lane >> 4, slot = lane & 15;
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)img);
    // piece i holds rows 4i + row0: row & 15 = 4(i&3) + row0, so the swizzled chunk is
    // (slot ^ row0) ^ 4(i&3); pieces i and i+4 share it and lie 4 KiB apart on both sides
    const unsigned off0 = (unsigned)(row0 * kRowBytes);
    const int sw = slot ^ row0;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) off[i] = off0 + (unsigned)((sw ^ (4 * i)) << 4);
    dma_set_base(lds_base);
    dma_piece_at<0>(q_rows, off[0]);
    dma_piece_at<1024>(q_rows, off[1]);
    dma_piece_at<2048>(q_rows, off[2]);
    dma_piece_at<3072>(q_rows, off[3]);
    dma_set_base(lds_base + 4096u);
    dma_piece_at<0>(q_rows + 16 * kHeadDim, off[0]);
    dma_piece_at<1024>(q_rows + 16 * kHeadDim, off[1]);
    dma_piece_at<2048>(q_rows + 16 * kHeadDim, off[2]);
    dma_piece_at<3072>(q_rows + 16 * kHeadDim, off[3]);
}

// First FA_K_AHEAD k-steps of a tile's K operand fragments.
__device__ __forceinline__ void k_prefetch(const char* kb, const LaneGeom& g, half8 (&ka0)[FA_K_AHEAD],
                                           half8 (&ka1)[FA_K_AHEAD]) {
#pragma unroll
    for (int i = 0; i < FA_K_AHEAD; ++i) {
        ka0[i] = *reinterpret_cast<const half8*>(kb + g.k_rd[i]);
        ka1[i] = *reinterpret_cast<const half8*>(kb + g.k_rd[i] + 32 * kRowBytes);
    }
}
// S^T = K.Q^T with those fragments already on their way; the rest are read FA_K_AHEAD k-steps ahead.
__device__ __forceinline__ void qk_tile_pre(const char* kb, const LaneGeom& g, const half8 (&qf)[8],
                                            half8 (&ka0)[FA_K_AHEAD], half8 (&ka1)[FA_K_AHEAD], f32x16& s0,
                                            f32x16& s1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s0[e] = 0.0f;
        s1[e] = 0.0f;
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
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
        if (ks + FA_K_AHEAD < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS_READ
    }
}

#ifndef FA_V_AHEAD
#define FA_V_AHEAD 4
#endif
__device__ __forceinline__ half8 read_v_frag(const char* vb, const LaneGeom& g, int kk, int db) {
    const char* a0 = vb + g.v_rd[db] + kk * 16 * kRowBytes;
    const short4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
    const short4_t t1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0 + 8 * kRowBytes));
    return __builtin_bit_cast(half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}
// O^T += V^T.P^T with the V fragments read FA_V_AHEAD MFMAs ahead of their use.
__device__ __forceinline__ void pv_tile_ahead(const char* vb, const LaneGeom& g, const half8 (&pf)[4],
                                              f32x16 (&oacc)[4]) {
    static_assert(FA_V_AHEAD >= 1 && FA_V_AHEAD <= 16, "ring depth");
    half8 ring[FA_V_AHEAD];
#pragma unroll
    for (int i = 0; i < FA_V_AHEAD; ++i) ring[i] = read_v_frag(vb, g, i >> 2, i & 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[i % FA_V_AHEAD], pf[i >> 2], oacc[i & 3], 0, 0, 0);
        if (i + FA_V_AHEAD < 16) ring[i % FA_V_AHEAD] = read_v_frag(vb, g, (i + FA_V_AHEAD) >> 2, (i + FA_V_AHEAD) & 3);
    }
#pragma unroll
    for (int i = 0; i < FA_V_AHEAD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int i = 0; i + FA_V_AHEAD < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#pragma unroll
    for (int i = 16 - FA_V_AHEAD; i < 16; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
}

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void fa_fwd_ring_kernel(const FwdParams p) {
    static_assert(NWAVES == 8, "a tile is 16 pieces per matrix: two of each per wave");
    static_assert(kRingStages == 3, "the step's waits and barrier are laid out for three stages");
    constexpr int QB = NWAVES * kRowsPerWave;

    __shared__ __attribute__((aligned(1024))) char ring[kRingStages * kStageBytes];
    __shared__ __attribute__((aligned(1024))) char img_all[NWAVES * kRowsPerWave * kRowBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeom g = make_lane_geom(lane);
    char* img = img_all + wave * (kRowsPerWave * kRowBytes);

    // ---- DMA geometry.  A tile is 16 pieces of 4 rows (1 KiB); this wave moves pieces 2*wave and
    //      2*wave + 1 of the K tile and of the V tile, i.e. rows 8*wave + 4j + (lane >> 4).  The two
    //      pieces are 1 KiB apart on both sides, so one LDS base and the instruction offset address
    //      them.  Lane l fills slot l & 15 of its row with the source chunk the image's swizzle maps
    //      there: for K that depends on row & 15 (one offset register per piece), for V on row & 3.
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    const int d_r4 = lane >> 4, d_slot = lane & 15;
    const unsigned d_rowoff = (unsigned)((8 * wave + d_r4) * kRowBytes);
    const unsigned d_koff0 = d_rowoff + (unsigned)((d_slot ^ ((8 * wave + d_r4) & 15)) << 4);
    const unsigned d_koff1 = d_rowoff + (unsigned)((d_slot ^ ((8 * wave + 4 + d_r4) & 15)) << 4);
    const unsigned d_voff = d_rowoff + (unsigned)((d_slot ^ (d_r4 << 2)) << 4);
    const unsigned d_lds = ring_lds + (unsigned)wave * 2048u;

    const int n_items = p.bh * p.q_blocks;
    const int n_tiles = p.lk / kKvTile;
    const int stride = (int)gridDim.x;
    const float c = p.scale_log2e;
    constexpr size_t kTileElems = (size_t)kKvTile * kHeadDim;

    // ---- compute stream: first item
    int item = blockIdx.x;
    int head, qb;
    decode_item(p, item, head, qb);
    int q_row0 = qb * QB + wave * kRowsPerWave;
    bool wave_valid = q_row0 < p.lq;
    size_t wrow = (size_t)head * p.lq + (wave_valid ? q_row0 : 0);
    size_t srow = wrow + g.r;

    // ---- load cursor: the tile the next DMA fetches, as an element offset common to K and V.  It
    //      runs two tiles ahead of the compute stream, across item boundaries; past the end of the
    //      stream it stays on the last tile (a redundant transfer into a stage nobody reads).
    int c_item = item, c_tile = 0;
    size_t c_off = (size_t)head * p.lk * kHeadDim;
#define RING_DMA_TILE(stage_)                                                                  \
    do {                                                                                       \
        const half_t* kt_ = p.k + c_off;                                                       \
        const half_t* vt_ = p.v + c_off;                                                       \
        const unsigned l_ = d_lds + (unsigned)(stage_) * (unsigned)kStageBytes;                \
        dma_set_base(l_);                                                                      \
        dma_piece_at<0>(kt_, d_koff0);                                                         \
        dma_piece_at<1024>(kt_, d_koff1);                                                      \
        dma_set_base(l_ + (unsigned)kTileBytes);                                               \
        dma_piece_at<0>(vt_, d_voff);                                                          \
        dma_piece_at<1024>(vt_, d_voff);                                                       \
        if (c_tile + 1 < n_tiles) {                                                            \
            ++c_tile;                                                                          \
            c_off += kTileElems;                                                               \
        } else if (c_item + stride < n_items) {                                                \
            c_item += stride;                                                                  \
            c_tile = 0;                                                                        \
            int h_, q_;                                                                        \
            decode_item(p, c_item, h_, q_);                                                    \
            c_off = (size_t)h_ * p.lk * kHeadDim;                                              \
        }                                                                                      \
    } while (0)

#ifdef FA_STAMPS  // diagnostic build: 100 MHz time stamps of wave 0 at entry, after the prologue, after every step
    int n_stamp = 0;
#define RING_STAMP()                                                                           \
    do {                                                                                       \
        if (p.stamps && tid == 0 && n_stamp < 60)                                              \
            p.stamps[(size_t)blockIdx.x * 64 + n_stamp++] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#else
#define RING_STAMP() do {} while (0)
#endif
    RING_STAMP();
    // ---- prologue: tiles 0 and 1 of the stream and the first item's Q rows
    half8 qf[8];
    half8 ka0[FA_K_AHEAD], ka1[FA_K_AHEAD];
    RING_DMA_TILE(0);
    RING_DMA_TILE(1);
    dma_q_rows_asm(p.q + wrow * kHeadDim, img, lane);
    wait_all_vmem();
    __syncthreads();
    RING_STAMP();
    fetch_q_frags(qf, img, lane);
    k_prefetch(ring, g, ka0, ka1);

    f32x16 oacc[4];
    float m_run = 0.f, l_run = 1.f, m_true = 0.f;  // set per item by init_state
    int st = 0, st_fill = 2;                       // stage of the current tile / stage the step's DMA fills
    bool out_pending = false, out_valid = false;
    size_t out_wrow = 0;

#define RING_WRITE_OUT()                                                                       \
    do {                                                                                       \
        if (p.flags & kFlagStoreState)                                                         \
            finish_rows(p, out_wrow + g.r, g.hh, oacc, m_run, m_true, l_run);                  \
        else                                                                                   \
            store_output_rows(p.o + out_wrow * kHeadDim, img, lane, oacc, l_run);              \
    } while (0)

    // One loop, one step body; an item's first and last tile are told apart by wave-uniform flags.
    int t = 0;
    for (;;) {
        const bool first = (t == 0), last = (t + 1 == n_tiles);
        RING_DMA_TILE(st_fill);  // tile g+2
        const char* kb = ring + st * kStageBytes;
        f32x16 s0, s1;
        qk_tile_pre(kb, g, qf, ka0, ka1, s0, s1);
        if (first) {
            // the finished item's output leaves behind the new item's first Q.K^T
            if (out_pending && out_valid) RING_WRITE_OUT();
            init_state(p, srow, g.hh, oacc, m_run, m_true, l_run);
        }
        const int next_item = item + stride;
        const bool more_items = next_item < n_items;
        int nhead = head, nqb = qb;
        bool nvalid = wave_valid;
        size_t nwrow = wrow;
        if (last && more_items) {
            // the staging image is idle from here to the end of the item (any write-out through it
            // was issued above and its reads have been consumed by its stores)
            decode_item(p, next_item, nhead, nqb);
            const int nq_row0 = nqb * QB + wave * kRowsPerWave;
            nvalid = nq_row0 < p.lq;
            nwrow = (size_t)nhead * p.lq + (nvalid ? nq_row0 : 0);
            dma_q_rows_asm(p.q + nwrow * kHeadDim, img, lane);
        }
        half8 pf[4];
        softmax_tile(s0, s1, c, m_run, m_true, l_run, oacc, pf);
        pv_tile_ahead(kb + kTileBytes, g, pf, oacc);
        if (last) {
            out_pending = true;
            out_valid = wave_valid;
            out_wrow = wrow;
            if (!more_items) {
                if (out_valid) RING_WRITE_OUT();
                wait_all_vmem();  // the last step's (redundant) transfer must not outlive the workgroup's LDS
                RING_STAMP();
                break;
            }
        }
        wait_all_vmem();  // this wave's share of tile g+2 (and the Q rows, if any) has landed
        if (last) {
            fetch_q_frags(qf, img, lane);
            item = next_item;
            head = nhead;
            qb = nqb;
            wave_valid = nvalid;
            wrow = nwrow;
            srow = nwrow + g.r;
            t = -1;
        }
        st_fill = st;
        st = (st == kRingStages - 1) ? 0 : st + 1;
        k_prefetch(ring + st * kStageBytes, g, ka0, ka1);  // tile g+1: complete since the last barrier
        __syncthreads();
        RING_STAMP();
        ++t;
    }
}

#undef RING_STAMP
#undef RING_DMA_TILE
#undef RING_WRITE_OUT

}  // namespace fa
