// Forward kernel for the shapes the tuned kernels do not take: sequence lengths that are not a
// multiple of 64, different query and key lengths, and the causal mask of the CPU baseline
// (attention_cpu's `causal` argument, flash_attention_tensor_cores/testing.cu:74-77: key j > query i
// takes no part).  The reference's GPU kernel has neither (wrapper.cu:53-61 rejects L % 64 != 0, the
// kernel has no mask); its README lists both as future work (README.md:44-47).
//
// One workgroup of 4 waves per (head, 128-row query block), 32 rows per wave, built from the tile
// routines of fa_fwd_kernel.hpp (transposed scores, P straight from the accumulator, swizzled LDS
// tiles).  What differs:
//   * K/V rows past the end of the sequence are fetched from the last valid row instead (finite data:
//     their scores are masked, and their P is exactly 0, so what V holds there must only not be NaN);
//   * on a tile that needs it (the sequence tail, the causal diagonal) every score whose key lies
//     beyond the row's limit is replaced by -1e30 before the softmax, which is what the CPU baseline
//     does; the limit of row i is min(Lk - 1, i) under the causal mask and Lk - 1 otherwise.  Every row
//     sees key 0 in its first tile, so a row's reference maximum is a real score from then on and a
//     fully masked later tile contributes exp2(-1e30 - m) = 0;
//   * with the causal mask a workgroup stops at the tile that holds its last row's diagonal, and a wave
//     skips the tiles that lie wholly above its own rows (it still helps staging them);
//   * query rows past the end of Q are computed on a copy of the last row and not stored.
// Measured (pattern data, B*L = 16384, H = 16): 770 / 1043 / 1136 TFLOP/s without mask at L = 511 / 2047 /
// 16383 (the tuned kernels: 1083 / 1405 / 1522 at the next multiple of 64).
#pragma once
#include "fa_fwd_kernel.hpp"

namespace fa {

constexpr int kFlagCausal = 4;  // FwdParams::flags: key j > query i is masked (needs lq == lk)

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void fa_fwd_general_kernel(const FwdParams p) {
    constexpr int NT = NWAVES * 64;
    constexpr int QB = NWAVES * kRowsPerWave;
    constexpr int CHUNKS = (kTileBytes / 16) / NT;
    constexpr int RSTEP = NT / 16;
    static_assert(CHUNKS >= 1 && RSTEP % 16 == 0, "staging geometry");

    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];  // [buffer][K tile | V tile]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeom g = make_lane_geom(lane);

    int head, qb;
    decode_item(p, (int)blockIdx.x, head, qb);
    const bool causal = (p.flags & kFlagCausal) != 0;
    // Under the causal mask block qb costs qb + 1 units: hand the long blocks out first, so that the
    // workgroups still running at the end are the short ones.
    if (causal) qb = p.q_blocks - 1 - qb;
    const int q_row0 = qb * QB + wave * kRowsPerWave;       // wave-uniform
    const bool wave_valid = q_row0 < p.lq;
    const int q_abs = min(q_row0 + g.r, p.lq - 1);           // this lane's query row (clamped)
    const int wave_last = min(q_row0 + kRowsPerWave - 1, p.lq - 1);
    const int key_limit = causal ? min(p.lk - 1, q_abs) : p.lk - 1;  // last key this row may see

    int n_tiles = (p.lk + kKvTile - 1) / kKvTile;
    if (causal) n_tiles = min(n_tiles, (min(qb * QB + QB, p.lq) + kKvTile - 1) / kKvTile);

    const half_t* kh = p.k + (size_t)head * p.lk * kHeadDim;
    const half_t* vh = p.v + (size_t)head * p.lk * kHeadDim;

    // staging: thread t moves 16-byte chunk t & 15 of rows (t >> 4) + i * RSTEP of a tile
    const int st_row = tid >> 4, st_ch = tid & 15;
    const int st_koff = st_row * kRowBytes + ((st_ch ^ (st_row & 15)) << 4);
    const int st_voff = st_row * kRowBytes + ((st_ch ^ ((st_row & 3) << 2)) << 4);
    u32x4 st_k[CHUNKS], st_v[CHUNKS];
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int row = min(tile * kKvTile + st_row + i * RSTEP, p.lk - 1);
            const size_t off = (size_t)row * kHeadDim + st_ch * 8;
            st_k[i] = *reinterpret_cast<const u32x4*>(kh + off);
            st_v[i] = *reinterpret_cast<const u32x4*>(vh + off);
        }
    };
    auto stage_write = [&](int buf) {
        char* kw = smem + buf * 2 * kTileBytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            *reinterpret_cast<u32x4*>(kw + st_koff + i * RSTEP * kRowBytes) = st_k[i];
            *reinterpret_cast<u32x4*>(kw + kTileBytes + st_voff + i * RSTEP * kRowBytes) = st_v[i];
        }
    };

    half8 qf[8];
    load_q_frags(qf, p.q + ((size_t)head * p.lq + q_abs) * kHeadDim, g.hh);

    f32x16 oacc[4];
    float m_run = -1e30f, m_true = -1e30f, l_run = 0.0f;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[db][e] = 0.0f;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) stage_load(t + 1);
        const int key0 = t * kKvTile;
        if (wave_valid && !(causal && key0 > wave_last)) {
            const char* kb = smem + buf * 2 * kTileBytes;
            f32x16 s0, s1;
            half8 pf[4];
            qk_tile(kb, g, qf, s0, s1);
            // element e of s0 is key key0 + 8(e >> 2) + 4hh + (e & 3), of s1 the same + 32
            const int lim = key_limit - key0 - 4 * g.hh;
            if (__any(lim < 59)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = 8 * (e >> 2) + (e & 3);
                    s0[e] = (kr <= lim) ? s0[e] : -1e30f;
                    s1[e] = (kr + 32 <= lim) ? s1[e] : -1e30f;
                }
            }
            softmax_tile(s0, s1, p.scale_log2e, m_run, m_true, l_run, oacc, pf);
            pv_tile(kb + kTileBytes, g, pf, oacc);
        }
        if (t + 1 < n_tiles) stage_write(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // O = acc / l: lane (r, hh) holds columns 32db + 8g4 + 4hh .. +3 of row r
    const float inv = 1.0f / half_wave_sum(l_run);
    if (wave_valid && q_row0 + g.r < p.lq) {
        half_t* og = p.o + ((size_t)head * p.lq + q_row0 + g.r) * kHeadDim;
#pragma unroll
        for (int db = 0; db < 4; ++db)
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
