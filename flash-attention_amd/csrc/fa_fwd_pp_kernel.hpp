// Two-phase ("ping-pong") schedule of the forward kernel for 8-wave workgroups.
//
// A CU's SIMD hosts two of the workgroup's waves.  In fa_fwd_kernel both run the same tile at the
// same time, so they want the matrix pipe together (Q.K^T, P.V) and the vector ALU together
// (softmax) and each unit idles while the other is contended.  Here every tile step g of a wave is
// cut into
//     Y_g  vector phase : softmax of S(g)  ->  P(g)            (plus item set-up / write-out)
//     X_g  matrix phase : O += V(g)^T.P(g)^T , then S(g+1) = K(g+1).Q^T
// separated by workgroup barriers, and waves 4-7 run one phase behind waves 0-3.  At any time one
// wave of a SIMD is in a matrix phase and its partner in a vector phase.
//
// K therefore runs one tile ahead of V.  Ring use at step g (par = g & 1):
//     X_g reads   V(g)   in V[par]      and  K(g+1) in K[par^1]
//     X_g ends by writing K(g+2) -> K[par]  and  V(g+1) -> V[par^1]
// (global loads for those two tiles are issued at the start of Y_g).  With the late half one phase
// behind, a buffer's last read and its next write are always separated by a barrier: K(g) in K[par]
// is last read in X_(g-1) (global phases 2g-1 / 2g for the two halves) and rewritten at the ends of
// X_g (phases 2g+1 / 2g+2); the same holds for V.
//
// Work items form one stream of tile steps per workgroup exactly as in fa_fwd_kernel: the load
// cursors simply run ahead across item boundaries, the next item's Q rows are fetched by LDS-DMA
// during the item's last vector phase and picked up between the last P.V and the next Q.K^T, and the
// finished item's output is written out in the first vector phase of the next item.
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {

// LDS-DMA of the next item's Q rows, issued through inline asm so that the compiler does not know
// of it: a DMA it knows of is drained (s_waitcnt vmcnt(0)) in front of every workgroup barrier, and
// this one has to stay in flight across one.  It is always issued AFTER the step's K/V loads, so the
// compiler's own counted waits on those can only over-wait; completion is awaited by hand
// (wait_dma) before the image is read.  m0 carries the LDS base of each 1 KiB piece.
__device__ __forceinline__ void dma_q_rows_hidden(const half_t* q_rows, char* img, int lane) {
    lane = opaque(lane);
    const int row0 = lane >> 4, slot = lane & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)img;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + row0;
        const half_t* src = q_rows + row * kHeadDim + ((slot ^ (row & 15)) << 3);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + i * 1024);
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
    }
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Advance a load cursor by one tile along the workgroup's stream of tiles; past the end of the
// stream it stays on the last tile (the redundant loads keep every wait a counted one).
__device__ __forceinline__ void advance_cursor(const FwdParams& p, int n_tiles, int n_items, int stride, int& item,
                                               int& tile, const half_t*& ptr, const half_t* base) {
    if (tile + 1 < n_tiles) {
        ++tile;
        ptr += (size_t)kKvTile * kHeadDim;
    } else if (item + stride < n_items) {
        item += stride;
        tile = 0;
        int h, qb;
        decode_item(p, item, h, qb);
        ptr = base + (size_t)h * p.lk * kHeadDim;
    }
}

// O^T += V^T.P^T with the V fragments read PP_V_AHEAD MFMAs ahead of their use.  A wave in its
// matrix phase is alone on the SIMD's matrix pipe (its partner is in a vector phase), so every LDS
// round trip it does not cover with its own MFMAs is pipe idle time; left alone the compiler keeps
// two or three fragments in flight.  sched_group_barrier pins the interleave.
#ifndef PP_X_PRIO
#define PP_X_PRIO 1
#endif
#ifndef PP_V_AHEAD
#define PP_V_AHEAD 4
#endif
__device__ __forceinline__ half8 read_v_frag(const char* vb, const LaneGeom& g, int kk, int db) {
    const char* a0 = vb + g.v_rd[db] + kk * 16 * kRowBytes;
    const short4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0));
    const short4_t t1 =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(a0 + 8 * kRowBytes));
    return __builtin_bit_cast(half8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void pv_tile_ahead(const char* vb, const LaneGeom& g, const half8 (&pf)[4],
                                              f32x16 (&oacc)[4]) {
    static_assert(PP_V_AHEAD >= 1 && PP_V_AHEAD <= 16, "ring depth");
    half8 ring[PP_V_AHEAD];
#pragma unroll
    for (int i = 0; i < PP_V_AHEAD; ++i) ring[i] = read_v_frag(vb, g, i >> 2, i & 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        oacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[i % PP_V_AHEAD], pf[i >> 2], oacc[i & 3], 0, 0, 0);
        if (i + PP_V_AHEAD < 16) ring[i % PP_V_AHEAD] = read_v_frag(vb, g, (i + PP_V_AHEAD) >> 2, (i + PP_V_AHEAD) & 3);
    }
#pragma unroll
    for (int i = 0; i < PP_V_AHEAD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int i = 0; i + PP_V_AHEAD < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#pragma unroll
    for (int i = 16 - PP_V_AHEAD; i < 16; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
}

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void fa_fwd_pp_kernel(const FwdParams p) {
    static_assert(NWAVES == 8, "two waves per SIMD are the point of this schedule");
    constexpr int NT = NWAVES * 64;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int CHUNKS = (kTileBytes / 16) / NT;
    constexpr int RSTEP = NT / 16;

    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];  // K[0] V[0] K[1] V[1], 16 KiB each
    __shared__ __attribute__((aligned(16))) char img_all[NWAVES * kRowsPerWave * kRowBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= NWAVES / 2;  // this half runs one phase behind the other
    const LaneGeom g = make_lane_geom(lane);
    char* img = img_all + wave * (kRowsPerWave * kRowBytes);

    const int st_row = tid >> 4, st_ch = tid & 15;
    const int st_goff = st_row * kHeadDim + st_ch * 8;
    const int st_koff = st_row * kRowBytes + ((st_ch ^ (st_row & 15)) << 4);
    const int st_voff = st_row * kRowBytes + ((st_ch ^ ((st_row & 3) << 2)) << 4);

#define PP_LOAD(dst_, src_)                                                                      \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i)                                       \
            dst_[i] = *reinterpret_cast<const u32x4*>((src_) + st_goff + i * RSTEP * kHeadDim);  \
    } while (0)
#define PP_WRITE_K(buf_, src_)                                                                   \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i)                                       \
            *reinterpret_cast<u32x4*>(smem + (buf_) * 2 * kTileBytes + st_koff + i * RSTEP * kRowBytes) = src_[i]; \
    } while (0)
#define PP_WRITE_V(buf_, src_)                                                                   \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i)                                       \
            *reinterpret_cast<u32x4*>(smem + (buf_) * 2 * kTileBytes + kTileBytes + st_voff + i * RSTEP * kRowBytes) = src_[i]; \
    } while (0)

    const int n_items = p.bh * p.q_blocks;
    const int n_tiles = p.lk / kKvTile;
    const int stride = (int)gridDim.x;
    const float c = p.scale_log2e;

    // ---- compute stream: first item
    int item = blockIdx.x;
    int head, qb;
    decode_item(p, item, head, qb);
    int q_row0 = qb * QB + wave * kRowsPerWave;
    bool wave_valid = q_row0 < p.lq;
    size_t wrow = (size_t)head * p.lq + (wave_valid ? q_row0 : 0);
    size_t srow = wrow + g.r;

    // ---- load cursors
    int k_item = item, k_tile = 0, v_item = item, v_tile = 0;
    const half_t* kp = p.k + (size_t)head * p.lk * kHeadDim;
    const half_t* vp = p.v + (size_t)head * p.lk * kHeadDim;

    // ---- prologue, both halves in step: Q rows -> image; K(0), K(1), V(0) -> ring
    half8 qf[8];
    {
        u32x4 a[CHUNKS], b[CHUNKS], d[CHUNKS];
        PP_LOAD(a, kp);
        advance_cursor(p, n_tiles, n_items, stride, k_item, k_tile, kp, p.k);
        PP_LOAD(b, kp);
        advance_cursor(p, n_tiles, n_items, stride, k_item, k_tile, kp, p.k);
        PP_LOAD(d, vp);
        advance_cursor(p, n_tiles, n_items, stride, v_item, v_tile, vp, p.v);
        dma_q_rows_hidden(p.q + wrow * kHeadDim, img, lane);
        PP_WRITE_K(0, a);
        PP_WRITE_K(1, b);
        PP_WRITE_V(0, d);
        wait_dma();
    }
    __syncthreads();
    fetch_q_frags(qf, img, lane);
    if (late) __syncthreads();

    f32x16 oacc[4];
    float m_run = 0.f, l_run = 1.f, m_true = 0.f;  // set per item by init_state
    f32x16 s0, s1;                                 // S of the step whose vector phase comes next
    qk_tile(smem, g, qf, s0, s1);                  // X_(-1): S(0) from K[0]
    __syncthreads();

    int par = 0;
    bool out_pending = false, out_valid = false;
    size_t out_wrow = 0;

#define PP_WRITE_OUT()                                                                         \
    do {                                                                                       \
        if (p.flags & kFlagStoreState)                                                         \
            finish_rows(p, out_wrow + g.r, g.hh, oacc, m_run, m_true, l_run);                  \
        else                                                                                   \
            store_output_rows(p.o + out_wrow * kHeadDim, img, lane, oacc, l_run);              \
    } while (0)

    // One loop, one step body; the item's first and last tile are told apart by wave-uniform flags
    // (separate code copies per kind of step made the register allocator shuffle and spill the
    // ~130 loop-carried registers at every join).
#ifdef FA_STAMPS
    int n_stamp = 0;
#define PP_STAMP()                                                                                       \
    do {                                                                                                 \
        if (p.stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && n_stamp < 60)        \
            p.stamps[(wave ? 64 : 0) + n_stamp++] = __builtin_amdgcn_s_memtime();                        \
    } while (0)
#else
#define PP_STAMP() do {} while (0)
#endif
    int t = 0;
    for (;;) {
        const bool first = (t == 0), last = (t + 1 == n_tiles);
        PP_STAMP();
        /* ---- Y: vector phase ---- */
        if (first) {  // before the tile loads: their registers are not live during the write-out
            if (out_pending && out_valid) PP_WRITE_OUT();
            init_state(p, srow, g.hh, oacc, m_run, m_true, l_run);
        }
        u32x4 st_k[CHUNKS], st_v[CHUNKS];
        PP_LOAD(st_k, kp);  // K(g+2)
        PP_LOAD(st_v, vp);  // V(g+1)
        advance_cursor(p, n_tiles, n_items, stride, k_item, k_tile, kp, p.k);
        advance_cursor(p, n_tiles, n_items, stride, v_item, v_tile, vp, p.v);
        const int next_item = item + stride;
        const bool more_items = next_item < n_items;
        int nhead = head, nqb = qb;
        bool nvalid = wave_valid;
        size_t nwrow = wrow;
        if (last && more_items) {
            decode_item(p, next_item, nhead, nqb);
            const int nq_row0 = nqb * QB + wave * kRowsPerWave;
            nvalid = nq_row0 < p.lq;
            nwrow = (size_t)nhead * p.lq + (nvalid ? nq_row0 : 0);
            dma_q_rows_hidden(p.q + nwrow * kHeadDim, img, lane);
        }
        half8 pf[4];
        softmax_tile(s0, s1, c, m_run, m_true, l_run, oacc, pf);
        PP_STAMP();
        __syncthreads();
        PP_STAMP();
        /* ---- X: matrix phase ---- */
        __builtin_amdgcn_s_setprio(PP_X_PRIO);
        pv_tile_ahead(smem + par * 2 * kTileBytes + kTileBytes, g, pf, oacc);
        if (last) {
            out_pending = true;
            out_valid = wave_valid;
            out_wrow = wrow;
            if (!more_items) {
                __builtin_amdgcn_s_setprio(0);
                if (out_valid) PP_WRITE_OUT();
                break;
            }
            wait_dma();
            fetch_q_frags(qf, img, lane);
            item = next_item;
            head = nhead;
            qb = nqb;
            wave_valid = nvalid;
            wrow = nwrow;
            srow = nwrow + g.r;
            t = -1;
        }
        qk_tile(smem + (par ^ 1) * 2 * kTileBytes, g, qf, s0, s1);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP();
        PP_WRITE_K(par, st_k);
        PP_WRITE_V(par ^ 1, st_v);
        PP_STAMP();
        __syncthreads();
        par ^= 1;
        ++t;
    }
    if (!late) __syncthreads();  // the early half owes the barrier the late half spent at the start
}

#undef PP_WRITE_OUT
#undef PP_STAMP
#undef PP_LOAD
#undef PP_WRITE_K
#undef PP_WRITE_V

}  // namespace fa
