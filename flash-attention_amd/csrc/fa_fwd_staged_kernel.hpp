// The register-staged forward kernel (global -> registers -> ds_write tiles, two LDS buffers, one barrier per
// tile), 8 waves x 32 rows or 4 waves x 32 rows: the first of this library's kernels, kept for comparisons only.
// It is compiled into the comparison build alone (-DFA_AB_KERNELS: libflash_attention_ab.so, FA_RING=0 /
// FA_NWAVES=4 select it); the product library does not contain it.  Building blocks: fa_fwd_kernel.hpp.
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {

// The kernel.  gridDim.x workgroups walk the work items (head, query block) item = blockIdx.x,
// blockIdx.x + gridDim.x, ...  as ONE stream of K/V tiles: the loads for the next tile are issued
// before the current tile's math whether that tile belongs to this item or to the next one, the
// next item's Q rows are fetched right after the current item's last Q.K^T, and an item's output
// stores drain behind the next item's first tiles.  With gridDim.x == number of items it
// degenerates to one item per workgroup.
//
// The last tile of an item is a separate copy of the tile body: the steady-state loop then carries
// nothing but the K/V staging registers across its back edge, and every wait on a global load is a
// counted one (a conditional Q reload inside the loop made the compiler drain all loads, the
// prefetched tile included, in front of every Q.K^T).
template <int NWAVES>
// (second launch bound: waves per SIMD.  A 4-wave workgroup with its 96 KiB of LDS is alone on its CU,
// one wave per SIMD, and may use the whole register file; two 4-wave halves of an 8-wave one share a SIMD.)
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void fa_fwd_kernel(const FwdParams p) {
    constexpr int NT = NWAVES * 64;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int CHUNKS = (kTileBytes / 16) / NT;  // 16-byte chunks per thread per tile
    static_assert(CHUNKS >= 1, "too many threads for one tile");

    // Two separate LDS objects: the K/V ring and the per-wave staging images.  With one array the
    // compiler must assume that a tile read may alias an in-flight Q transfer into an image and
    // drains every outstanding load in front of the first read after the transfer was issued.
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
    __shared__ __attribute__((aligned(16))) char img_all[NWAVES * kRowsPerWave * kRowBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeom g = make_lane_geom(lane);
    char* img = img_all + wave * (kRowsPerWave * kRowBytes);

    // ---- staging geometry: thread t moves 16-byte chunk (t + i*NT) of a tile, i.e. row
    //      (t>>4) + i*RSTEP, chunk t&15.  RSTEP is a multiple of 16, so the swizzle term is the
    //      same for every i and the per-i offsets are compile-time constants.
    constexpr int RSTEP = NT / 16;
    const int st_row = tid >> 4, st_ch = tid & 15;
    const int st_goff = st_row * kHeadDim + st_ch * 8;                              // elements
    const int st_koff = st_row * kRowBytes + ((st_ch ^ (st_row & 15)) << 4);        // bytes in a K buffer
    const int st_voff = st_row * kRowBytes + ((st_ch ^ ((st_row & 3) << 2)) << 4);  // bytes in a V buffer
    u32x4 st_k[CHUNKS], st_v[CHUNKS];

#define FA_STAGE_LOAD(kt_, vt_)                                                                 \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < CHUNKS; ++i) {                                    \
            st_k[i] = *reinterpret_cast<const u32x4*>((kt_) + st_goff + i * RSTEP * kHeadDim);  \
            st_v[i] = *reinterpret_cast<const u32x4*>((vt_) + st_goff + i * RSTEP * kHeadDim);  \
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

    const int n_items = p.bh * p.q_blocks;
    const int n_tiles = p.lk / kKvTile;
    const float c = p.scale_log2e;
    constexpr size_t kTileElems = (size_t)kKvTile * kHeadDim;

    // ---- first item
    int item = blockIdx.x;
    int head, qb;
    decode_item(p, item, head, qb);
#ifdef FA_ABLATE_SAME_KV
#define FA_KV_HEAD(h) ((h) & 7)
#else
#define FA_KV_HEAD(h) (h)
#endif
    const half_t* kt = p.k + (size_t)FA_KV_HEAD(head) * p.lk * kHeadDim;  // tile being staged
    const half_t* vt = p.v + (size_t)FA_KV_HEAD(head) * p.lk * kHeadDim;
    int q_row0 = qb * QB + wave * kRowsPerWave;
    bool wave_valid = q_row0 < p.lq;                                   // wave-uniform; waves past the last
    size_t wrow = (size_t)head * p.lq + (wave_valid ? q_row0 : 0);     // row still stage K/V
    size_t srow = wrow + g.r;

    // Q takes the same route for the first item as for every later one (rows -> staging image ->
    // fragments), so the fragments are never the destination of a global load and no wait on them
    // can reach into the tile loop.
    half8 qf[8];
    dma_q_rows(p.q + wrow * kHeadDim, img, lane);
    FA_STAGE_LOAD(kt, vt);
    FA_STAGE_WRITE(0);
    f32x16 oacc[4];
    float m_run = 0.f, l_run = 1.f, m_true = 0.f;  // set per item by init_state
    __syncthreads();
    fetch_q_frags(qf, img, lane);

#ifdef FA_STAMPS
    int n_stamp = 0;
#define FA_STAMP()                                                                                  \
    do {                                                                                            \
        if (p.stamps && tid == 0 && n_stamp < 60)                                                   \
            p.stamps[(size_t)blockIdx.x * 64 + n_stamp++] = __builtin_amdgcn_s_memrealtime();       \
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memtime(); \
        if (p.stamps && tid == 0 && n_stamp == 1) p.stamps[(size_t)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FA_STAMP() do {} while (0)
#endif
    FA_STAMP();
#ifdef FA_PRIO_SKEW
    // waves w and w + NWAVES/2 share a SIMD: a fixed priority difference lets one run ahead, so that
    // its softmax falls under the other's MFMAs instead of both doing the same kind of work at once
    if (wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(FA_PRIO_SKEW);
#endif
    int buf = 0;
    // Output of the item just finished, still in the accumulators: it is written out under the
    // first Q.K^T of the next item (vector/LDS/store work beside matrix work) rather than between
    // the two items.
    bool out_pending = false, out_valid = false;
    size_t out_wrow = 0;

#define FA_WRITE_OUT()                                                                         \
    do {                                                                                       \
        if (p.flags & kFlagStoreState)                                                         \
            finish_rows(p, out_wrow + g.r, g.hh, oacc, m_run, m_true, l_run);                  \
        else                                                                                   \
            store_output_rows(p.o + out_wrow * kHeadDim, img, lane, oacc, l_run);              \
    } while (0)
#define FA_TILE(kb_)                                                                           \
    do {                                                                                       \
        f32x16 s0, s1;                                                                         \
        half8 pf[4];                                                                           \
        qk_tile((kb_), g, qf, s0, s1);                                                         \
        softmax_tile(s0, s1, c, m_run, m_true, l_run, oacc, pf);                               \
        FA_MID_BARRIER();                                                                      \
        pv_tile((kb_) + kTileBytes, g, pf, oacc);                                              \
    } while (0)

#ifdef FA_HACK_STAGGER
#define FA_MID_BARRIER() __syncthreads()
    if (wave >= NWAVES / 2) __syncthreads();
#else
#define FA_MID_BARRIER() do {} while (0)
#endif
    for (;;) {
        if (n_tiles > 1) {
            // ---- tile 0 of the item, with the previous item's write-out folded in
            kt += kTileElems;
            vt += kTileElems;
            FA_STAGE_LOAD(kt, vt);
            {
                const char* kb = smem + buf * 2 * kTileBytes;
                f32x16 s0, s1;
                half8 pf[4];
                // (the tile's Q.K^T is spelled out in both branches so that the write-out shares a
                // basic block with its MFMAs and can be scheduled between them)
                if (out_pending && out_valid) {
                    qk_tile(kb, g, qf, s0, s1);
                    FA_WRITE_OUT();
                } else {
                    qk_tile(kb, g, qf, s0, s1);
                }
                init_state(p, srow, g.hh, oacc, m_run, m_true, l_run);
                softmax_tile(s0, s1, c, m_run, m_true, l_run, oacc, pf);
                FA_MID_BARRIER();
                pv_tile(kb + kTileBytes, g, pf, oacc);
            }
            FA_STAGE_WRITE(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            FA_STAMP();

            // ---- tiles 1 .. n_tiles-2: the next tile is the next 64 keys of the same head
            for (int t = 1; t + 1 < n_tiles; ++t) {
                kt += kTileElems;
                vt += kTileElems;
                FA_STAGE_LOAD(kt, vt);
                FA_TILE(smem + buf * 2 * kTileBytes);
                FA_STAGE_WRITE(buf ^ 1);
                __syncthreads();
                buf ^= 1;
                FA_STAMP();
            }
        } else {
            if (out_pending && out_valid) FA_WRITE_OUT();
            init_state(p, srow, g.hh, oacc, m_run, m_true, l_run);
        }

        FA_STAMP();
        // ---- last tile of the item: the next tile, if any, is tile 0 of the next item's head
        const int next_item = item + (int)gridDim.x;
        const bool more_items = next_item < n_items;
        int nhead = head, nqb = qb;
        bool nvalid = wave_valid;
        size_t nwrow = wrow;
        if (more_items) {
            decode_item(p, next_item, nhead, nqb);
            const int nq_row0 = nqb * QB + wave * kRowsPerWave;
            nvalid = nq_row0 < p.lq;
            nwrow = (size_t)nhead * p.lq + (nvalid ? nq_row0 : 0);
            // the staging image is idle during the tile (the Q fragments are in registers, the output
            // goes through it only afterwards): start the next item's Q rows towards it now
            dma_q_rows(p.q + nwrow * kHeadDim, img, lane);
            kt = p.k + (size_t)FA_KV_HEAD(nhead) * p.lk * kHeadDim;
            vt = p.v + (size_t)FA_KV_HEAD(nhead) * p.lk * kHeadDim;
            FA_STAGE_LOAD(kt, vt);
        }
        FA_TILE(smem + buf * 2 * kTileBytes);
        FA_STAMP();

        out_pending = true;
        out_valid = wave_valid;
        out_wrow = wrow;
        if (!more_items) {
            if (out_valid) FA_WRITE_OUT();
#ifdef FA_HACK_STAGGER
            if (wave < NWAVES / 2) __syncthreads();
#endif
            break;
        }
        // waits for the tile loads and, with them, for the older Q transfer; the fragments leave the
        // image before the write-out (in the next tile) reuses it
        FA_STAGE_WRITE(buf ^ 1);
        fetch_q_frags(qf, img, lane);
        FA_STAMP();

        item = next_item;
        head = nhead;
        qb = nqb;
        wave_valid = nvalid;
        wrow = nwrow;
        srow = nwrow + g.r;
        __syncthreads();
        buf ^= 1;
        FA_STAMP();
    }
}

#undef FA_WRITE_OUT
#undef FA_MID_BARRIER
#undef FA_TILE
#undef FA_STAMP
#undef FA_KV_HEAD
#undef FA_STAGE_LOAD
#undef FA_STAGE_WRITE

}  // namespace fa
