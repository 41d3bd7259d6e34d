// Forward kernel with ONE wave per SIMD: 4 waves per workgroup, 64 query rows per wave, the whole
// 512-entry register file per lane.
//
// With two waves on a SIMD (fa_fwd_kernel, fa_fwd_ring_kernel) the matrix pipe idles while both do
// softmax and the vector ALU idles while both issue MFMAs; alternating their phases with barriers
// buys little, because every phase change costs a barrier, an LDS prologue and an arbitration
// fight.  Here the two things that should overlap live in ONE instruction stream.  A wave's 64 rows
// are two 32-row sub-blocks A and B, each with its own Q fragments, S^T tile, P^T fragments, O^T
// accumulator and running state, and B runs half a tile behind A:
//
//     phase 1 of step t :  MFMA  S_B(t)   = K(t).Q_B^T ,  O_B += V(t-1)^T.P_B(t-1)^T
//                          VALU  P_A(t)   = exp2(S_A(t).c - m_A),  l_A += rowsum
//                          then  mx_B = rowmax S_B(t), rescale decision for B
//     phase 2 of step t :  MFMA  S_A(t+1) = K(t+1).Q_A^T , O_A += V(t)^T.P_A(t)^T
//                          VALU  P_B(t)   = exp2(S_B(t).c - m_B),  l_B += rowsum
//                          then  mx_A = rowmax S_A(t+1), rescale decision for A
//
// Each phase is one basic block of 32 MFMAs with the other sub-block's exponentials, the row-max
// chain of the scores just produced and the LDS operand reads placed in the gaps between them
// (written out gap by gap with a sched_barrier(0) after each; an MFMA holds the SIMD's vector issue
// for only 8 of its 32 cycles).  The rescale decision is the only branch and sits between the
// blocks, where no MFMA is writing the accumulator.
//
// K therefore runs one tile ahead of V, and the tile one sub-block multiplies in a phase 2 is the one
// the other multiplies in the following phase 1.  The K fragments of a tile (64 registers) and its
// V^T fragments (64 accumulator registers) therefore stay in registers for two phases and every
// fragment is read from LDS once; phase 1 re-fills each slot right after its last use, phase 2
// reads nothing.  LDS: a K ring and a V ring of three 16 KiB slots each, both filled by DMA
// (global_load_lds, hidden from the compiler as in fa_fwd_ring_kernel) two steps ahead of their
// use.  One barrier per step, at its end, after each wave has waited for everything but the step's
// own transfers.
//
// Items (head, 256-row query block) form one stream of tiles per workgroup, as in the other
// kernels.  The next item's Q rows are DMAed into the wave's 16 KiB staging image during the item's
// last-but-one step; the fragment reads into the Q registers are requested one phase before the
// registers are needed and complete behind that phase (A's during phase 1 of the last step, for
// phase 2, which already computes S_A of the next item; B's during that phase 2, for phase 1 of the
// next step).  A sub-block's output is written out (through its half of the image, as whole rows)
// right after the phase that finished it.  The kernel needs lq % 64 == 0 (in a last block of fewer than
// 256 rows some waves have no rows of their own: they compute along on a stand-in and store nothing)
// and at least 3 tiles per item; the launcher sends other shapes to the other kernels.
//
// The body is a template: fa_fwd_w64_kernel is its plain form; the causal mask and lengths that are not
// whole tiles are two template parameters that add per-item tile counts, a shifted last tile and a mask
// applied between the phases (see fa_fwd_w64_body), and leave the phases as they are.
#pragma once
#include <type_traits>
#include <utility>

#include "fa_fwd_ring_kernel.hpp"

namespace fa {

struct RowState {
    float m_run, m_true, l_run;
};

// ---------------------------------------------------------------------------------------------
// The O^T accumulators are OWNED by the inline asm below: sub-block A in a[0:63], sub-block B in
// a[64:127] (block db of 32 output columns in 16 consecutive registers); so are the Q fragments
// (a[128:191]), which an MFMA can take as its B operand from there.  They are named literally
// in the instruction strings and are not C++ values, so the compiler cannot move, copy or re-home
// them (as a value, an accumulator that is also read by ordinary vector code somewhere in the loop
// is kept in architectural registers and copied in and out around every MFMA).  The kernel keeps
// its own register demand under 256, so the compiler has no use for accumulator registers; one
// statement with the whole range in its clobber list makes the kernel descriptor allocate them.
// Every statement that touches them is volatile: program order is the only ordering there is.
//
// The compiler does not know these statements are MFMAs and inserts no wait states for them:
//   * a register an MFMA wrote is read by other instructions only several MFMAs later (the phase
//     below is ordered that way) or behind acc_fence();
//   * their operands are LDS read results (the compiler's s_waitcnt covers those) or registers
//     written by vector instructions at least a few instructions earlier.
// ---------------------------------------------------------------------------------------------
constexpr int kAccA = 0, kAccB = 64;     // O^T of sub-blocks A and B
constexpr int kQregA = 128, kQregB = 160;  // their Q fragments: k-step ks in a[base + 4ks : base + 4ks + 3]
constexpr int kVbuf = 192;                 // V^T fragments of one tile: fragment i in a[192 + 4i : 192 + 4i + 3]

__device__ __forceinline__ void acc_reserve() {
    asm volatile(""
                 :
                 :
                 :
                   "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14",
                   "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28",
                   "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42",
                   "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56",
                   "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70",
                   "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84",
                   "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98",
                   "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110",
                   "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122",
                   "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134",
                   "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146",
                   "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158",
                   "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170",
                   "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182",
                   "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194",
                   "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206",
                   "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218",
                   "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230",
                   "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242",
                   "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254",
                   "a255");
}
// S (architectural registers): d = a.q  /  d += a.q, with q = a[QLO:QLO+3]
template <int QLO>
__device__ __forceinline__ void mfma_s_first(f32x16& d, const half8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(a), "n"(QLO), "n"(QLO + 3));
}
template <int QLO>
__device__ __forceinline__ void mfma_s(f32x16& d, const half8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "n"(QLO), "n"(QLO + 3));
}
// O: a[LO:LO+15] += a.b
template <int LO>
__device__ __forceinline__ void mfma_acc(const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]"
                 :
                 : "v"(a), "v"(b), "n"(LO), "n"(LO + 15));
}
// O: a[LO:LO+15] += v.b with the V^T fragment v taken from a[VLO:VLO+3]
template <int LO, int VLO>
__device__ __forceinline__ void mfma_acc_av(const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%1:%2], a[%3:%4], %0, a[%1:%2]"
                 :
                 : "v"(b), "n"(LO), "n"(LO + 15), "n"(VLO), "n"(VLO + 3));
}
// One V^T fragment (two transposed 8-byte reads) straight into a[VLO:VLO+3].  The compiler does
// not see these reads: their completion is awaited by hand (lds_wait_all) before the MFMAs that
// use the registers.  Its own counted waits for the reads it does see can only become longer.
template <int VLO, int OFS>
__device__ __forceinline__ void vbuf_read(unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%3:%4], %0 offset:%6"
                 :
                 : "v"(addr), "n"(VLO), "n"(VLO + 1), "n"(VLO + 2), "n"(VLO + 3), "n"(OFS), "n"(OFS + 8 * kRowBytes)
                 : "memory");
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Wait states between the last MFMA on an accumulator and a v_accvgpr_read of it.
__device__ __forceinline__ void acc_fence() { asm volatile("s_nop 15\n\ts_nop 3"); }
template <int R>
__device__ __forceinline__ float acc_get() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
    return x;
}
template <int R>
__device__ __forceinline__ void acc_put(float x) {
    asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(x), "n"(R));
}
template <int R>
__device__ __forceinline__ void acc_put_u32(unsigned x) {
    asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(x), "n"(R));
}
// Q fragments of one sub-block: staging image -> a[QBASE : QBASE+31].  The reads are hidden from the
// compiler like vbuf_read: lds_wait_all() stands between them and the first MFMA that takes the
// registers.  q_install_issue alone lets a whole phase pass in between (a sub-block's Q registers
// are idle while the other sub-block is on the matrix side), which hides the reads' latency.
template <int QBASE>
__device__ __forceinline__ void q_install_issue(const char* img_half, int lane) {
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
    // lane (r, hh) takes chunk 2ks + hh of row r; the image's slot of chunk c in row r is c ^ (r & 15),
    // i.e. ((2ks) ^ (r & 14)) plus a lane-constant low bit: k-step ks sits at base ^ (ks << 5)
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)img_half +
                          (unsigned)(r * kRowBytes + ((hh ^ (r & 15)) << 4));
    static_for<8>([&](auto ic) {
        constexpr int ks = decltype(ic)::value;
        const unsigned addr = base ^ (unsigned)(ks << 5);
        asm volatile("ds_read_b128 a[%1:%2], %0" : : "v"(addr), "n"(QBASE + 4 * ks), "n"(QBASE + 4 * ks + 3) : "memory");
    });
}
template <int QBASE>
__device__ __forceinline__ void q_install(const char* img_half, int lane) {
    q_install_issue<QBASE>(img_half, lane);
    lds_wait_all();
}
// A sub-block's accumulator for the row-wise helpers (ArrayAcc's interface): no copy in architectural registers
// (64 fewer live at the item boundaries: the kernel's register count falls from 252 to 216; bit-identical
// results).  Whole tiles only: in the RAGGED forms the same change made the compiler keep scalars in vector
// lanes inside the tile loop (1 - 3 % slower at L = 16383), so those still copy the accumulator out.
template <int BASE>
struct OwnedAcc {
    template <int I>
    __device__ __forceinline__ float get() const { return acc_get<BASE + I>(); }
    template <int I>
    __device__ __forceinline__ void put(float x) const { acc_put<BASE + I>(x); }
};
// whole sub-block accumulator <-> values
template <int BASE>
__device__ __forceinline__ void acc_load(f32x16 (&o)[4]) {
    acc_fence();
    static_for<64>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        o[i >> 4][i & 15] = acc_get<BASE + i>();
    });
}
template <int BASE>
__device__ __forceinline__ void acc_store(const f32x16 (&o)[4]) {
    static_for<64>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        acc_put<BASE + i>(o[i >> 4][i & 15]);
    });
}
// a[BASE : BASE+63] = 0 through the matrix pipe (four MFMAs of zero operands with C = 0)
template <int BASE>
__device__ __forceinline__ void acc_zero() {
    half8 z = {};
    asm volatile("s_nop 3" : "+v"(z));  // wait states between the vector writes of z and the MFMAs reading it
    static_for<4>([&](auto ic) {
        constexpr int LO = BASE + 16 * decltype(ic)::value;
        const half8 zz = z;
        asm volatile("v_mfma_f32_32x32x16_f16 a[%1:%2], %0, %0, 0" : : "v"(zz), "n"(LO), "n"(LO + 15));
    });
}
template <int BASE>
__device__ __forceinline__ void acc_scale(float alpha) {
    acc_fence();
    static_for<64>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        acc_put<BASE + i>(acc_get<BASE + i>() * alpha);
    });
}

// Building blocks of the write-out below.  volatile: it relies on the order they are written in.
__device__ __forceinline__ f32x2 pk_mul(f32x2 x, f32x2 y) {
    f32x2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ unsigned cvt_pk_f16(f32x2 x) {
    unsigned w;
    const float lo = x[0], hi = x[1];
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
    return w;
}
__device__ __forceinline__ unsigned pinned(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
// O = acc / l of one sub-block as fp16, through its half of the staging image, straight from the registers the asm
// owns (what store_output_rows_acc does with an accessor; same arithmetic, same bits).  This runs on the single
// wave's critical path at every item boundary, and with one wave per SIMD nothing hides a vector instruction's
// latency: measured (tools/micro/issue_cost.hip) an independent vector instruction issues every 4.7 cycles, one that
// depends on its predecessor every 8.3, v_pk_mul_f32 every 6.7 (3.4 per element), v_fma_mix only every 8.7, and a
// ds_write_b64 every 24.5 while all four waves write.  So the pieces go four at a time, in a fixed order with
// every result used at least four instructions after it was produced:
//   16 reads of the next four pieces | 8 v_pk_mul_f32 | 8 v_cvt_pk_f16_f32 | 4 address XORs | 4 ds_write_b64.
// A piece is 8 bytes: columns 32 db + 8 g4 + 4 hh .. +3 of the lane's row, chunk 4 db + g4; its address is one XOR
// away from chunk 0's (the image's swizzle only touches address bits 4-7, and the lane's slot of chunk 0 has the
// row's low bits there).  With zero_after the accumulator is zeroed through the matrix pipe as soon as its last
// element has been read, behind the row-wise reads and the stores instead of after them.
#ifndef W64_LEAN_OUT
#define W64_LEAN_OUT 1
#endif
template <int BASE, typename STAMP>
__device__ __forceinline__ void store_output_rows_owned(half_t* o_rows, char* img_half, int lane, float l_run,
                                                        bool zero_after, STAMP&& stamp) {
    stamp(1);
    const float l_tot = half_wave_sum(l_run);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;  // a row that has seen no key: O = 0, not 0/0
    f32x2 inv2 = {inv, inv};
    lane = opaque(lane);
    const int r = lane & 31, hh = lane >> 5;
    unsigned wr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)img_half +
                  (unsigned)(r * kRowBytes + ((r & 15) << 4) + 8 * hh);
    asm volatile("" : "+v"(inv2), "+v"(wr));
    stamp(2);
    f32x2 x[2][8];
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        x[0][i >> 1][i & 1] = acc_get<BASE + i>();
    });
    static_for<4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        f32x2(&cur)[8] = x[g & 1];
        if constexpr (g < 3) {
            static_for<16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                x[(g + 1) & 1][i >> 1][i & 1] = acc_get<BASE + 16 * (g + 1) + i>();
            });
        }
        static_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            cur[i] = pk_mul(cur[i], inv2);
        });
        unsigned w[8];
        static_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            w[i] = cvt_pk_f16(cur[i]);
        });
        unsigned ad[4];
        static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ad[i] = pinned(wr ^ (unsigned)((4 * g + i) << 4));
        });
        static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const u32x2 both = {w[2 * i], w[2 * i + 1]};
            *(__attribute__((address_space(3))) u32x2*)(size_t)ad[i] = both;
        });
    });
    if (zero_after) acc_zero<BASE>();
    // (written as 8-byte pieces, read back as u32x4: accesses of different types, which the compiler may reorder)
    asm volatile("" ::: "memory");
    stamp(3);
    store_staged_rows(o_rows, img_half, lane, 0);
    stamp(4);
}
template <int BASE>
__device__ __forceinline__ void store_output_rows_owned(half_t* o_rows, char* img_half, int lane, float l_run,
                                                        bool zero_after) {
    store_output_rows_owned<BASE>(o_rows, img_half, lane, l_run, zero_after, [](int) {});
}
// The rescale decision of fa::softmax_tile, for scores whose scaled row max mx is already known.
// Runs between two phases: no MFMA is in flight on the accumulator.
// need: some row's max runs ahead of its reference by more than the slack.
// (a lane mask: the rows concerned)
typedef unsigned long long lane_mask_t;
__device__ __forceinline__ lane_mask_t rescale_needed(float mx, float m_run) {
    return __builtin_amdgcn_ballot_w64(!(mx - m_run <= kRescaleSlack));
}
template <int BASE>
__device__ __forceinline__ void rescale_if_needed(lane_mask_t need, float mx, RowState& st) {
    st.m_true = fmaxf(st.m_true, mx);
    if (__builtin_expect(need != 0, 0)) {
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);
        st.m_run = m_new;
        st.l_run *= alpha;
        acc_scale<BASE>(alpha);
    }
}

// Operand reads with the ring slot folded into the instruction's immediate offset: the per-lane
// parts of the addresses (W64Geom, absolute LDS byte addresses including the ring's base) never
// change, and the tile loop is unrolled over the three slots so that every slot offset is a
// compile-time constant.  No address arithmetic is left in the loop.
struct W64Geom {
    unsigned k_ad[8];  // K ring base + lane part of the fragment of k-step ks, key block 0 (block 1: + 32 rows)
    unsigned v_ad[4];  // V ring base + lane part of the transposed fragment of output block db
};
typedef const __attribute__((address_space(3))) half8* lds_half8_ptr;
typedef __attribute__((address_space(3))) short4_t* lds_short4_ptr;
// Passing the addresses through an empty asm at the top of a phase keeps "address + slot offset"
// from being hoisted out of the tile loop as three dozen separate registers: the sum is then
// formed where it is used, i.e. folded into the read's immediate offset.
__device__ __forceinline__ void launder(W64Geom& r) {
    asm volatile(""
                 : "+v"(r.k_ad[0]), "+v"(r.k_ad[1]), "+v"(r.k_ad[2]), "+v"(r.k_ad[3]), "+v"(r.k_ad[4]), "+v"(r.k_ad[5]),
                   "+v"(r.k_ad[6]), "+v"(r.k_ad[7]), "+v"(r.v_ad[0]), "+v"(r.v_ad[1]), "+v"(r.v_ad[2]), "+v"(r.v_ad[3]));
}
template <int OFS>
__device__ __forceinline__ half8 read_k_at(const W64Geom& w, int ks, int blk) {
    return *(lds_half8_ptr)(size_t)(w.k_ad[ks] + (unsigned)(OFS + blk * 32 * kRowBytes));
}

// Schedule of the vector slice: how many of Y's 32 score elements have had their exponential started
// before gap g.  An element costs about 18 issue cycles (fma 4, exp 8, add 4, half a conversion) of the 24
// an MFMA leaves free, so a gap holds one; W64_SLICE picks where the surplus goes.
//   0: two per gap in gaps 0-3, one per gap in gaps 4-27 (everything finished inside the phase)
//   1: one per gap in all 32 gaps (the last sum and the last conversion fall behind the phase);
//      measured 0.5-2 % faster than 0
//   2: one per gap in gaps 0-29, a second one in gaps 16 and 17 (the lightest ones); as fast as 1
#ifndef W64_SLICE
#define W64_SLICE 1
#endif
constexpr int slice_started(int g) {
    if (g <= 0) return 0;
#if W64_SLICE == 0
    return g < 4 ? 2 * g : (g < 28 ? g + 4 : 32);
#elif W64_SLICE == 1
    return g < 32 ? g : 32;
#else
    return g >= 30 ? 32 : g + (g > 16 ? 1 : 0) + (g > 17 ? 1 : 0);
#endif
}

// Where the step's barrier sits.  The tiles a step reads from LDS were sent for two steps earlier; every wave
// waits for its own pieces at the end of the step before (s_waitcnt) and the workgroup's barrier publishes
// them.  Nothing needs that barrier before the step's first LDS read, and the first MFMAs of phase 1 take
// operands that are already in registers: with W64_BAR_GAP = g >= 0 the barrier moves from the end of a step
// into gap g of the next step's phase 1, behind g + 1 issued MFMAs that run while the slowest wave arrives;
// the K fragment reads of gaps 0 .. g follow it.  The step's transfers start in gap 2 and overwrite what
// other waves read in the step before, so g <= 1.  -1: barrier at the end of the step.
#ifndef W64_BAR_GAP
#define W64_BAR_GAP -1
#endif
static_assert(W64_BAR_GAP >= -1 && W64_BAR_GAP <= 1, "the step's transfers start in gap 2");
__device__ __forceinline__ void step_barrier() { __builtin_amdgcn_s_barrier(); }

// One phase (see the file comment), written out gap by gap: 32 MFMAs, and after each of them the
// LDS reads and the slice of vector work that share its 32 cycles; sched_barrier(0) after every gap
// keeps the compiler from re-ordering across gaps.  X is the sub-block on the matrix side
// (accumulator base XACC, Q fragments at QBASE), Y the one on the vector side.
//
// Operands.  The K fragments of one tile (kb0/kb1: key blocks 0 and 1, k-steps 0-7, 64 registers) and
// the V^T fragments of one tile (a[kVbuf..], 64 registers) stay in registers for TWO consecutive
// phases, so that both sub-blocks use them and every fragment is read from LDS once: sub-block A is
// half a tile ahead, so the tile A multiplies in a phase 2 is the one B multiplies in the following
// phase 1.  Phase 1 (REFILL) therefore re-fills each slot right after its MFMA has taken it: k-step ks
// of the next K tile after gaps 2ks / 2ks+1, fragment i of the next V tile after gap 16+i.  Phase 2 reads
// nothing from LDS.  Data returning from LDS competes with vector instructions for the register
// file's write port (measured: 48 reads + the vector slice cost 13 cycles per MFMA more than either
// alone), which is why halving the reads matters.
//   KB_NEXT  K slot (byte offset in the K ring) the refill reads     VB_NEXT  V slot the refill reads
//   sx0/sx1  out: S_X                          mx_x     out: scaled row max of S_X
//   m_x      X's running reference max         need_x   out: rescale_needed(mx_x, m_x)
//   sy0/sy1  in: S_Y (overwritten by p)        py       out: P_Y fragments;   l_y += row sum
//   dma(j)   issues piece j (0..7) of the step's tile transfers (j is an integral_constant); called in
//            gaps 2, 6, 10, ... 30
// Gaps 0-15 are S_X (k-step g/2, key block g&1), gaps 16-31 O_X (V fragment g-16).  The row max of
// S_X starts in gap 19, three MFMAs after the last one that wrote S_X.
//   xdma(g)  a caller's own transfers in gap g (g is an integral_constant; called in every gap): the next item's Q
//            rows trickling in beside the tiles (QSTREAM below); nothing otherwise
template <bool REFILL, int XACC, int QBASE, int KB_NEXT, int VB_NEXT, typename DMA, typename XDMA>
__device__ __forceinline__ void fused_phase(W64Geom& w, half8 (&kb0)[8], half8 (&kb1)[8], f32x16& sx0, f32x16& sx1,
                                            const half8 (&px)[4], float& mx_x, float m_x, lane_mask_t& need_x,
                                            f32x16& sy0, f32x16& sy1, float m_y, float& l_y, half8 (&py)[4], float c,
                                            DMA&& dma, XDMA&& xdma) {
    launder(w);
    const float nm = -m_y;
    float sum = 0.0f;
    float a0 = 0.0f, a1 = 0.0f;
    u32x4 pw[4] = {};
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#define W64_Y(e_) (((e_) < 16) ? sy0[(e_) & 15] : sy1[(e_) & 15])
    static_for<32>([&](auto ic) {
        constexpr int gi = decltype(ic)::value;
        // ---- the MFMA
        if constexpr (gi < 16) {
            constexpr int ks = gi >> 1;
            if constexpr (gi == 0)
                mfma_s_first<QBASE>(sx0, kb0[0]);
            else if constexpr (gi == 1)
                mfma_s_first<QBASE>(sx1, kb1[0]);
            else if constexpr (gi & 1)
                mfma_s<QBASE + 4 * ks>(sx1, kb1[ks]);
            else
                mfma_s<QBASE + 4 * ks>(sx0, kb0[ks]);
        } else {
            constexpr int i = gi - 16;
            // all V fragments were requested at least a phase ago; the one wait sits where it cannot stall
            if constexpr (i == 0 && !REFILL) lds_wait_all();
            mfma_acc_av<XACC + 16 * (i & 3), kVbuf + 4 * i>(px[i >> 2]);
        }
        // ---- the workgroup's barrier of the step before (W64_BAR_GAP, see there)
        if constexpr (REFILL && gi == W64_BAR_GAP) step_barrier();
        // ---- this gap's share of the tile transfer
        if constexpr ((gi & 3) == 2) dma(std::integral_constant<int, (gi >> 2)>{});
        xdma(std::integral_constant<int, gi>{});
        // ---- LDS reads: re-fill the slot the MFMA above has just taken
        if constexpr (REFILL) {
#ifndef W64_ABLATE_KREAD
            if constexpr (gi < 16 && gi >= W64_BAR_GAP) {
                // (the reads of the gaps in front of the barrier follow it)
                static_for<(gi == W64_BAR_GAP ? gi : 0) + 1>([&](auto rc) {
                    constexpr int gr = gi - decltype(rc)::value;
                    constexpr int ks = gr >> 1;
                    if constexpr (gr & 1)
                        kb1[ks] = read_k_at<KB_NEXT>(w, ks, 1);
                    else
                        kb0[ks] = read_k_at<KB_NEXT>(w, ks, 0);
                });
            }
#endif
#ifndef W64_ABLATE_VREAD
            if constexpr (gi >= 16) {
                constexpr int i = gi - 16;
                vbuf_read<kVbuf + 4 * i, VB_NEXT + (i >> 2) * 16 * kRowBytes>(w.v_ad[i & 3]);
            }
#endif
        }
        // ---- vector slice: Y's 32 scores, spread over the gaps as slice_started() says; each element is
        //      summed one gap after its exponential, each pair rounded to fp16 one gap after that
        {
            constexpr int n_done = slice_started(gi);                 // elements started before this gap
            constexpr int n_new = slice_started(gi + 1) - n_done;     // started in this gap
            constexpr int n_prev = slice_started(gi - 1);
#ifndef W64_ABLATE_VEC  // (timing experiments only: wrong results)
            static_for<n_new>([&](auto jc) {
                constexpr int e = n_done + decltype(jc)::value;
#ifdef W64_ABLATE_FMA  // (timing experiments only: wrong results)
                if constexpr (e < 16)
                    sy0[e] = __builtin_amdgcn_exp2f(sy0[e]);
                else
                    sy1[e - 16] = __builtin_amdgcn_exp2f(sy1[e - 16]);
#else
                if constexpr (e < 16)
                    sy0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sy0[e], c, nm));
                else
                    sy1[e - 16] = __builtin_amdgcn_exp2f(__builtin_fmaf(sy1[e - 16], c, nm));
#endif
            });
            // elements [n_prev, n_done) got their exponential in the previous gap
            static_for<n_done - n_prev>([&](auto jc) {
                constexpr int e = n_prev + decltype(jc)::value;
                sum += W64_Y(e);
            });
#endif
            // pairs completed (both summed) before the previous gap's additions: [n_prev2/2, n_prev/2)
            constexpr int n_prev2 = slice_started(gi - 2);
            static_for<n_prev / 2 - n_prev2 / 2>([&](auto jc) {
                constexpr int q = n_prev2 / 2 + decltype(jc)::value;
                const f32x2 pr = {W64_Y(2 * q), W64_Y(2 * q + 1)};
                const half2_t h = __builtin_convertvector(pr, half2_t);
                pw[q >> 2][q & 3] = __builtin_bit_cast(unsigned, h);
            });
        }
        // ---- vector slice: row max of S_X
#ifndef W64_ABLATE_MAX
        if constexpr (gi == 19) {
            a0 = fmaxf(sx0[0], sx0[1]);
            a1 = fmaxf(sx1[0], sx1[1]);
        }
        if constexpr (gi >= 20 && gi <= 26) {
            constexpr int k2 = 2 * (gi - 19);
            a0 = fmaxf(fmaxf(a0, sx0[k2]), sx0[k2 + 1]);
            a1 = fmaxf(fmaxf(a1, sx1[k2]), sx1[k2 + 1]);
        }
        if constexpr (gi == 28) mx_x = half_wave_max(fmaxf(a0, a1)) * c;
        if constexpr (gi == 30) {
            // the rescale decision, taken here so that the branch after the phase finds its condition ready
            need_x = rescale_needed(mx_x, m_x);
            asm volatile("" : "+s"(need_x));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    });
    // what the schedule leaves behind the last gap: sums of the elements started in it, the last conversions
    static_for<32 - slice_started(31)>([&](auto jc) {
        constexpr int e = slice_started(31) + decltype(jc)::value;
        sum += W64_Y(e);
    });
    static_for<16 - slice_started(30) / 2>([&](auto jc) {
        constexpr int q = slice_started(30) / 2 + decltype(jc)::value;
        const f32x2 pr = {W64_Y(2 * q), W64_Y(2 * q + 1)};
        const half2_t h = __builtin_convertvector(pr, half2_t);
        pw[q >> 2][q & 3] = __builtin_bit_cast(unsigned, h);
    });
    l_y += sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) py[j] = __builtin_bit_cast(half8, pw[j]);
    // the results are pinned here: the optimiser would otherwise sink the whole vector side into the
    // block that consumes it
    asm volatile("" : "+v"(py[0]), "+v"(py[1]), "+v"(py[2]), "+v"(py[3]), "+v"(l_y), "+v"(mx_x));
#undef W64_Y
}

template <bool REFILL, int XACC, int QBASE, int KB_NEXT, int VB_NEXT, typename DMA>
__device__ __forceinline__ void fused_phase(W64Geom& w, half8 (&kb0)[8], half8 (&kb1)[8], f32x16& sx0, f32x16& sx1,
                                            const half8 (&px)[4], float& mx_x, float m_x, lane_mask_t& need_x,
                                            f32x16& sy0, f32x16& sy1, float m_y, float& l_y, half8 (&py)[4], float c,
                                            DMA&& dma) {
    fused_phase<REFILL, XACC, QBASE, KB_NEXT, VB_NEXT>(w, kb0, kb1, sx0, sx1, px, mx_x, m_x, need_x, sy0, sy1, m_y, l_y,
                                                       py, c, dma, [](auto) {});
}

// O_X += V^T.P_X^T alone (the stream's last half tile), from the fragments already in registers.
template <int XACC>
__device__ __forceinline__ void pv_only(const half8 (&px)[4]) {
    lds_wait_all();
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        mfma_acc_av<XACC + 16 * (i & 3), kVbuf + 4 * i>(px[i >> 2]);
    });
}

// The stream's very first tile: its K fragments into the buffer (they stay for the first phase 1),
// S = K.Q^T and its row max (nothing to overlap with yet).
template <int QBASE>
__device__ __forceinline__ void first_scores(W64Geom& w, half8 (&kb0)[8], half8 (&kb1)[8], f32x16& sx0, f32x16& sx1,
                                             float& mx_x, float c) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        kb0[ks] = read_k_at<0>(w, ks, 0);
        kb1[ks] = read_k_at<0>(w, ks, 1);
    }
    static_for<8>([&](auto ic) {
        constexpr int ks = decltype(ic)::value;
        if constexpr (ks == 0) {
            mfma_s_first<QBASE>(sx0, kb0[0]);
            mfma_s_first<QBASE>(sx1, kb1[0]);
        } else {
            mfma_s<QBASE + 4 * ks>(sx0, kb0[ks]);
            mfma_s<QBASE + 4 * ks>(sx1, kb1[ks]);
        }
    });
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sx0), "+v"(sx1));  // MFMA result -> vector reader
    float a0 = fmaxf(sx0[0], sx0[1]), a1 = fmaxf(sx1[0], sx1[1]);
#pragma unroll
    for (int e = 2; e < 16; e += 2) {
        a0 = fmaxf(fmaxf(a0, sx0[e]), sx0[e + 1]);
        a1 = fmaxf(fmaxf(a1, sx1[e]), sx1[e + 1]);
    }
    mx_x = half_wave_max(fmaxf(a0, a1)) * c;
}

#ifdef W64_FINE_STAMPS
// Diagnostic build (tools/stamps_w64.py): the low 32 bits of the shader clock, kept in lane `idx` of one of two
// vector registers per wave (no memory traffic inside the loop) and dumped at the kernel's end.  m0 is the lane
// select: stamps are taken between phases only, where no transfer relies on it.
__device__ __forceinline__ void fine_stamp(unsigned& sv0, unsigned& sv1, int idx) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned lo = (unsigned)t;
    if (idx < 64)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(sv0) : "s"(lo), "s"(idx) : "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(sv1) : "s"(lo), "s"(idx - 64) : "m0");
}
// The same inside a phase (-DW64_GAP_STAMP=g -DW64_GAP_STAMP_PHASE=1|2: one stamp in gap g of that phase, kept in the
// slot of the phase's end stamp, which such a build leaves out): m0, the lane select, belongs to the tile transfers
// there and is put back.  The counter is sampled when the instruction issues; the wait for its
// result follows and is not part of the value.
__device__ __forceinline__ void gap_stamp(unsigned& sv0, unsigned& sv1, int idx) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned lo = (unsigned)t;
    unsigned keep;
    if (idx < 64)
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                     : "+v"(sv0), "=&s"(keep) : "s"(lo), "s"(idx));
    else
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                     : "+v"(sv1), "=&s"(keep) : "s"(lo), "s"(idx - 64));
}
#endif

// Causal item order.  With the mask of attention_cpu (testing.cu:74-77: key <= query row) query block qb
// needs only the tiles up to its diagonal, 4 (qb + 1) of them, so items differ in length.  Items are
// numbered long ones first (n = 0 is the last query block of head 0), and the workgroups take them in
// rounds of `grid`, every second round in reverse workgroup order, so that each workgroup's items add up
// to the same length within one block.  With B*H and the grid multiples of 8 the reversal leaves the low
// three bits alone: workgroup ids congruent mod 8 share an XCD and keep serving the same heads.
__device__ __forceinline__ void decode_item_causal(const FwdParams& p, int item, int grid, int& head, int& qb) {
    const int n_items = p.bh * p.q_blocks;
    const int k = item / grid, s = item - k * grid;
    const bool reversed = (k & 1) && (k + 1) * grid <= n_items;  // a partial last round is dealt in plain order
    int n;
    if (((p.bh | grid) & 7) == 0) {
        const int wi = s >> 3;
        n = k * grid + (reversed ? (grid >> 3) - 1 - wi : wi) * 8 + (s & 7);
    } else {
        n = k * grid + (reversed ? grid - 1 - s : s);
    }
    const int q = n / p.bh;
    head = n - q * p.bh;
    qb = p.q_blocks - 1 - q;
}

// Mask on one sub-block's fresh scores, between two phases.  Register e of key block kb in lane (r, hh)
// holds row kk + 4 hh of the tile, kk = 32 kb + 8 (e >> 2) + (e & 3), against query row r of the sub-block.
// The element stays iff lo <= kk <= thr, both bounds per lane:
//   thr = (the query row's position) - (position of the tile's first row) - 4 hh    (causal mask; else large)
//   lo  = (rows at the head of the tile that repeat the tile before) - 4 hh            (ragged last tile; else 0)
// Masked scores become -inf (their exponential is 0 and they leave the row max alone); mx is recomputed
// from what stays (-inf for a row that sees nothing of the tile).
// (LO = false: no lower bound, one comparison per score: the diagonal tiles of a causal sequence in whole tiles,
// several per item, sit on the critical path of short sequences.)
template <bool LO>
__device__ __forceinline__ void mask_scores(f32x16& s0, f32x16& s1, int thr, int lo, float& mx, float c) {
    const float ninf = -__builtin_inff();
    float a = ninf;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int kc = 8 * (e >> 2) + (e & 3);
        s0[e] = (kc > thr || (LO && kc < lo)) ? ninf : s0[e];
        s1[e] = (kc + 32 > thr || (LO && kc + 32 < lo)) ? ninf : s1[e];
        a = fmaxf(fmaxf(a, s0[e]), s1[e]);
    }
    mx = half_wave_max(a) * c;
}

// <false, false>: whole tiles, no mask (lq % 64 == 0, lk % 64 == 0): the default kernel.  Without touching the
// phases,
//   CAUSAL adds the causal mask (lq == lk, positions from 0): per-item tile counts, the item order above;
//   RAGGED adds any lk >= 192: the last tile's source is shifted down to end at the last key, so it repeats
//     dup = 64 n_tiles - lk rows of the tile before, which are masked; and any lq >= 64: a wave whose 64 rows
//     would pass the end takes the last 64 rows instead and stores only the rows no other wave owns (the
//     others are computed twice and stored once).
// A tile the diagonal touches or the ragged last tile gets mask_scores() in a cold branch between two phases.
// (Four instantiations rather than run-time flags: the kernel has no scalar registers to spare; with both
// features behind run-time flags the compiler kept scalars in vector lanes inside the tile loop, 7 % slower.)
//   SPLIT (whole tiles, no mask) divides every head's keys into p.k_splits equal ranges of p.lk_split keys and
//     makes each (range, head) pair a head of its own: p.bh counts those, virtual head vh = range * p.bh_real +
//     head reads Q of `head`, keys [range * lk_split, (range + 1) * lk_split) of `head`, and hands its
//     un-normalised state to row vh * lq + i of (o_acc, row_max, row_sum) (the launcher sets kFlagStoreState and
//     points them at a workspace; fa_split_merge_kernel folds the ranges).  For launches with fewer items than
//     compute units.
//   QSTREAM (whole tiles, no mask, items of 7 tiles or more) brings the next item's Q rows in four pieces per wave
//     and step during the item's steps 1 to 4, one piece per MFMA gap beside the tile transfers, instead of
//     sixteen at once in the last step but one: a workgroup's 64 pieces of 1 KiB take the unit's address path
//     about a thousand cycles, and sent in one go every wave sits out the queue (measured with cycle stamps at
//     L = 512: that step ran 2 000 - 2 900 cycles longer than its neighbours).  A phase must stay one basic
//     block, so the four extra transfers are issued in EVERY step; outside steps 1 to 4 they repeat four of
//     the step's own K pieces (same source, same destination).  The per-lane offsets of a K piece fit the Q
//     rows as well (same rows-of-four, same swizzle; the wave's row block moves into the scalar base).
//     Measured (bit-identical results): the step that used to send the rows runs like its neighbours, but the
//     four steps that carry them run 150 - 250 cycles longer and every other step 80: +1 % at L = 448 and 512,
//     nothing at 640 - 1024, -1 ... 2 % from 1280 on.  Not instantiated in the library; -DW64_QSTREAM_DEFAULT=true
//     builds the plain kernel in this form (tools/build_w64_only.sh).
//   VARLEN (with RAGGED, with or without CAUSAL) takes its items from a list in device memory instead of numbering them: entry i =
//     (head, 256-row block, query rows, keys) of a padded batch's sequence, so that every item has its own lengths
//     (tile count, shifted last tile, last query rows) while the strides stay the padded ones; *p.varlen_count
//     items.  For the long sequences of fa_forward_varlen's batches; the list is built on the device by
//     fa_varlen_plan_kernel, because the lengths live there.
template <bool CAUSAL, bool RAGGED, bool SPLIT = false, bool QSTREAM = false, bool VARLEN = false>
__device__ __forceinline__ void fa_fwd_w64_body(const FwdParams& p) {
    static_assert(!SPLIT || (!CAUSAL && !RAGGED), "the key split is built for whole tiles without a mask");
    static_assert(!QSTREAM || (!CAUSAL && !RAGGED && !SPLIT), "Q streaming is built for the plain form");
    static_assert(!VARLEN || (RAGGED && !SPLIT && !QSTREAM), "per-item lengths are built on the ragged forms");
    constexpr bool MASKED = CAUSAL || RAGGED;
    constexpr int NWAVES = 4;
    constexpr int QB = 256;                        // query rows per item
    constexpr int kImgBytes = 64 * kRowBytes;      // staging image per wave: 64 rows
    constexpr int kSlots = 3;

    __shared__ __attribute__((aligned(1024))) char k_ring[kSlots * kTileBytes];
    __shared__ __attribute__((aligned(1024))) char v_ring[kSlots * kTileBytes];
    __shared__ __attribute__((aligned(1024))) char img_all[NWAVES * kImgBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LaneGeom g = make_lane_geom(lane);
    char* img = img_all + wave * kImgBytes;  // rows 0-31: sub-block A, rows 32-63: sub-block B

    // ---- DMA geometry: a tile is 16 pieces of 4 rows (1 KiB); this wave moves pieces 4*wave .. 4*wave+3,
    //      i.e. rows 16*wave + 4j + (lane >> 4).  Consecutive pieces are 1 KiB apart on both sides, so
    //      one LDS base per tile and the instruction offset address them.  Lane l fills slot l & 15
    //      of its row with the source chunk the image's swizzle maps there: for K that depends on
    //      row & 15 = 4j + (lane >> 4), one offset register per piece; for V only on row & 3.
    const unsigned k_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)k_ring);
    const unsigned v_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)v_ring);
    const int d_r4 = lane >> 4, d_slot = lane & 15;
    const unsigned d_rowoff = (unsigned)((16 * wave + d_r4) * kRowBytes);
    unsigned d_koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d_koff[j] = d_rowoff + (unsigned)((d_slot ^ (4 * j + d_r4)) << 4);
    const unsigned d_voff = d_rowoff + (unsigned)((d_slot ^ (d_r4 << 2)) << 4);
    const unsigned d_piece = (unsigned)wave * 4096u;
    W64Geom w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w.k_ad[i] = k_lds + (unsigned)g.k_rd[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) w.v_ad[i] = v_lds + (unsigned)g.v_rd[i];

    const int n_items = VARLEN ? *p.varlen_count : p.bh * p.q_blocks;
    if constexpr (VARLEN)
        if ((int)blockIdx.x >= n_items) return;  // (uniform, before the first transfer or barrier)
    // (VARLEN: tile counts, the last tile's shift and the lengths are per item: *_c of the item being computed, kc.* of
    // the load cursor's, n* of the next one; everywhere else they are the launch's)
    const int n_tiles_all = SPLIT ? p.lk_split / kKvTile : RAGGED ? (p.lk + kKvTile - 1) / kKvTile : p.lk / kKvTile;  // >= 3
    constexpr bool causal = CAUSAL;
    [[maybe_unused]] std::conditional_t<VARLEN, int, const int> dup = RAGGED ? n_tiles_all * kKvTile - p.lk : 0;
    [[maybe_unused]] int lq_c = p.lq, lk_c = p.lk;  // VARLEN: of the item being computed
    [[maybe_unused]] int nlq = p.lq, nlk = p.lk;    // VARLEN: of the next item
    // one list entry: head, block, query rows, keys
    [[maybe_unused]] auto list_item = [&](int it, int& hd, int& blk, int& lq_, int& lk_) {
        const int* e = p.varlen_items + 4 * (size_t)it;
        hd = e[0];
        blk = e[1];
        lq_ = e[2];
        lk_ = e[3];
    };
    // tiles of a list item and the rows its last tile repeats: all of the sequence's tiles, the last one shifted to
    // end at the last key; with the mask only those up to the block's diagonal, and no shift unless they reach the end
    [[maybe_unused]] auto item_shape = [&](int blk, int lk_, int& nt_, int& dup_) {
        const int full = (lk_ + kKvTile - 1) / kKvTile;
        nt_ = CAUSAL && 4 * (blk + 1) < full ? 4 * (blk + 1) : full;
        dup_ = nt_ == full ? full * kKvTile - lk_ : 0;
    };
#define W64_LQ_C (VARLEN ? lq_c : p.lq)
#define W64_LQ_N (VARLEN ? nlq : p.lq)
#define W64_LK_C (VARLEN ? lk_c : p.lk)
    const int stride = (int)gridDim.x;
    // tiles of an item: all of them, or with the mask those up to the query block's diagonal (>= 3 as well)
    auto item_tiles = [&](int blk) -> int {
        if constexpr (CAUSAL) {
            const int n = 4 * (blk + 1);
            return n < n_tiles_all ? n : n_tiles_all;
        } else {
            return n_tiles_all;
        }
    };
    auto decode = [&](int it, int& hd, int& blk) {
        if constexpr (CAUSAL)
            decode_item_causal(p, it, stride, hd, blk);
        else
            decode_item(p, it, hd, blk);
    };
    const float c = p.scale_log2e;
    constexpr size_t kTileElems = (size_t)kKvTile * kHeadDim;
    // element offset of a head's first key row (SPLIT: of the virtual head's key range)
    auto kv_base = [&](int hd) -> size_t {
        if constexpr (SPLIT) {
            const int range = hd / p.bh_real;
            return ((size_t)(hd - range * p.bh_real) * p.lk + (size_t)range * p.lk_split) * kHeadDim;
        } else {
            return (size_t)hd * p.lk * kHeadDim;
        }
    };

    // ---- load cursor (an element offset common to K and V): the tile the next K transfer fetches.  It
    //      runs three tiles ahead of the compute stream; the V transfer of a step fetches the tile the
    //      K transfer of the step before fetched.  Past the end of the stream the cursor stays on the
    //      last tile (redundant transfers into slots nobody reads any more).
    struct Cursor {
        int item, tile, nt;  // nt: tiles of the cursor's item
        size_t off;
        int head, qb;  // of the cursor's item
        int dup;       // VARLEN: rows its last tile repeats
    };
    // Items advance by `stride`, so (head, block) of the next item follow from the current ones without a
    // division (an emulated one costs about 220 cycles at every item boundary).  With the XCD-aware numbering
    // (decode_item: item = 8 j + x, head = 8 (j / q_blocks) + x, block = j % q_blocks) j advances by stride / 8 and
    // x stays, provided stride is a multiple of 8; with the plain numbering item itself plays j's part.  Either way
    //   block += inc_qb, head += inc_head, and on block >= q_blocks: block -= q_blocks, head += inc_wrap.
    // inc_wrap = 0 marks the one case this does not cover (XCD-aware numbering on a grid that is no multiple of 8);
    // the causal order has its own numbering and keeps the division.
    [[maybe_unused]] int inc_qb = 0, inc_head = 0, inc_wrap = 0;
    if constexpr (!CAUSAL) {
        const bool xcd = (p.bh & 7) == 0;
        if (!xcd || (stride & 7) == 0) {
            const int u = xcd ? stride >> 3 : stride;
            inc_wrap = xcd ? 8 : 1;
            inc_head = (u / p.q_blocks) * inc_wrap;
            inc_qb = u % p.q_blocks;
        }
    }
#define W64_ADVANCE(cur_)                                                                      \
    do {                                                                                       \
        if (__builtin_expect((cur_).tile + 1 < ((CAUSAL || VARLEN) ? (cur_).nt : n_tiles_all), 1)) { \
            ++(cur_).tile;                                                                     \
            (cur_).off += kTileElems;                                                          \
            if (VARLEN) {                                                                      \
                if ((cur_).tile == (cur_).nt - 1) (cur_).off -= (size_t)(cur_).dup * kHeadDim; \
            } else if (RAGGED && (cur_).tile == n_tiles_all - 1) {                             \
                (cur_).off -= (size_t)dup * kHeadDim;                                          \
            }                                                                                  \
        } else if ((cur_).item + stride < n_items) {                                           \
            (cur_).item += stride;                                                             \
            (cur_).tile = 0;                                                                   \
            if (VARLEN) {                                                                      \
                int lq_, lk_;                                                                  \
                list_item((cur_).item, (cur_).head, (cur_).qb, lq_, lk_);                      \
                item_shape((cur_).qb, lk_, (cur_).nt, (cur_).dup);                             \
                (cur_).off = kv_base((cur_).head);                                             \
            } else if (!CAUSAL && inc_wrap != 0) {                                             \
                (cur_).qb += inc_qb;                                                           \
                (cur_).head += inc_head;                                                       \
                if ((cur_).qb >= p.q_blocks) {                                                 \
                    (cur_).qb -= p.q_blocks;                                                   \
                    (cur_).head += inc_wrap;                                                   \
                }                                                                              \
            } else {                                                                           \
                decode((cur_).item, (cur_).head, (cur_).qb);                                   \
            }                                                                                  \
            if (!VARLEN) {                                                                     \
                (cur_).nt = item_tiles((cur_).qb);                                             \
                (cur_).off = kv_base((cur_).head);                                             \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#define W64_DMA_K(slot_)                                                                       \
    do {                                                                                       \
        const half_t* t_ = p.k + kc.off;                                                       \
        dma_set_base(k_lds + (unsigned)(slot_) * (unsigned)kTileBytes + d_piece);              \
        dma_piece_at<0>(t_, d_koff[0]);                                                        \
        dma_piece_at<1024>(t_, d_koff[1]);                                                     \
        dma_piece_at<2048>(t_, d_koff[2]);                                                     \
        dma_piece_at<3072>(t_, d_koff[3]);                                                     \
    } while (0)
#define W64_DMA_V(slot_)                                                                       \
    do {                                                                                       \
        const half_t* t_ = p.v + kc.off;                                                       \
        dma_set_base(v_lds + (unsigned)(slot_) * (unsigned)kTileBytes + d_piece);              \
        dma_piece_at<0>(t_, d_voff);                                                           \
        dma_piece_at<1024>(t_, d_voff);                                                        \
        dma_piece_at<2048>(t_, d_voff);                                                        \
        dma_piece_at<3072>(t_, d_voff);                                                        \
    } while (0)

    // ---- compute stream: first item
    int item = blockIdx.x;
    int head, qb;
    if constexpr (VARLEN) {
        list_item(item, head, qb, lq_c, lk_c);
        nlq = lq_c;
        nlk = lk_c;
    } else {
        decode(item, head, qb);
    }
    std::conditional_t<CAUSAL || VARLEN, int, const int> n_tiles = item_tiles(qb);  // of the item being computed
    if constexpr (VARLEN) item_shape(qb, lk_c, n_tiles, dup);
    // A wave's 64 rows of a query block, or, when the block is the sequence's last and ends before them
    // (lq is a multiple of 64, not necessarily of 256), the block's first 64 rows as a stand-in source:
    // such a wave computes along (it moves its share of every tile and meets every barrier) and stores
    // nothing.  kNoRows marks "no rows to store".
    constexpr size_t kNoRows = ~(size_t)0;
    // position (row within the head) of the first of the 64 rows this wave computes for block blk
    // (lq_: the rows of the item's sequence; p.lq is the stride between heads either way)
    auto row_pos = [&](int blk, int lq_) -> int {
        int r0 = blk * QB + wave * 64;
        if (r0 >= lq_) r0 = blk * QB;
        if constexpr (RAGGED)
            if (r0 > lq_ - 64) r0 = lq_ - 64;
        return r0;
    };
    auto source_row = [&](int hd, int blk, int lq_) -> size_t { return (size_t)hd * p.lq + (size_t)row_pos(blk, lq_); };
    // where those rows' Q comes from: the same place, except that a virtual head (SPLIT) reads its real head's
    auto q_row = [&](int hd, int blk, int lq_) -> size_t {
        if constexpr (SPLIT)
            return (size_t)(hd % p.bh_real) * p.lq + (size_t)row_pos(blk, lq_);
        else
            return source_row(hd, blk, lq_);
    };
    auto owns_rows = [&](int blk, int lq_) -> bool { return blk * QB + wave * 64 < lq_; };
    // rows at the head of the wave's 64 that another wave owns (RAGGED, last rows of the sequence)
    [[maybe_unused]] auto rows_skipped = [&](int blk, int lq_) -> int { return blk * QB + wave * 64 - row_pos(blk, lq_); };
    size_t wrow = source_row(head, qb, W64_LQ_C);  // this wave's first query row (where its output or state goes)
    Cursor kc = {item, 0, n_tiles, kv_base(head), head, qb, dup};
    // Causal mask: `lim` is the position of the wave's first query row (in keys); a tile whose last key
    // lies beyond a sub-block's first row takes the mask (mask_scores, between the phases).
    [[maybe_unused]] int lim = row_pos(qb, W64_LQ_C);
    // needs_mask(tile, pos): does tile `tile` need a mask for the sub-block whose first row has position pos
    // (wave-uniform; the ragged last tile, or a tile the diagonal touches)?  apply_mask() then masks the
    // scores and recomputes their row max.  The call sites mark the branch unlikely, which keeps the mask
    // code out of the steady-state loop's layout.
    [[maybe_unused]] auto needs_mask = [&](int tile, int pos) -> bool {
        const bool tail = RAGGED && dup != 0 && tile == (VARLEN ? n_tiles : n_tiles_all) - 1;
        const int key0 = tail ? W64_LK_C - kKvTile : kKvTile * tile;
        return tail || (causal && key0 + 63 > pos);
    };
    [[maybe_unused]] auto apply_mask = [&](f32x16& s0, f32x16& s1, float& mx, int tile, int pos) {
        const bool tail = RAGGED && dup != 0 && tile == (VARLEN ? n_tiles : n_tiles_all) - 1;
        const int key0 = tail ? W64_LK_C - kKvTile : kKvTile * tile;
        const int l_ = opaque(lane);
        const int r_ = l_ & 31, h4_ = 4 * (l_ >> 5);
        if (tail)
            mask_scores<true>(s0, s1, causal ? pos + r_ - key0 - h4_ : 1000, dup - h4_, mx, c);
        else
            mask_scores<false>(s0, s1, pos + r_ - key0 - h4_, 0, mx, c);
    };
    // ---- prologue, in three groups so that the first scores start on as little data as possible:
    //        Q_A rows, K(0)            | waited for first: S_A(0) needs nothing else
    //        Q_B rows, K(1), V(0)      | waited for before step 0 (behind S_A(0) and the state set-up)
    //        V(1), K(2)                | stay in flight: nothing reads them before step 1, and the wait at
    //                                    the end of step 0 covers them
    {
        const size_t qsrc = q_row(head, qb, W64_LQ_C);
        dma_q_rows_asm(p.q + qsrc * kHeadDim, img, lane);
        W64_DMA_K(0);
        dma_q_rows_asm(p.q + (qsrc + 32) * kHeadDim, img + 32 * kRowBytes, lane);
    }
    const size_t off0 = kc.off;
    W64_ADVANCE(kc);
    W64_DMA_K(1);
    {
        const size_t off1 = kc.off;
        kc.off = off0;
        W64_DMA_V(0);
        kc.off = off1;
    }
    W64_DMA_V(1);
    W64_ADVANCE(kc);
    W64_DMA_K(2);
    size_t v_next_off = kc.off;  // V(2): what step 0 transfers
    W64_ADVANCE(kc);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // 8 + 4 + 4 + 8 loads behind the first group
    __syncthreads();

#ifdef FA_STAMPS  // diagnostic build: shader-clock and 100 MHz counters at entry and exit
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    acc_reserve();
    q_install<kQregA>(img, lane);

    f32x16 sa0, sa1, sb0, sb1;
    half8 pa[4], pb[4];
    RowState ra, rb;
    half8 kb0[8], kb1[8];  // K fragments of the tile both sub-blocks multiply next (key blocks 0 / 1)
    float mx_a, mx_b;
    lane_mask_t need_a = 0, need_b = 0;

    // S_A(0) and its state; P_B = 0 makes the first phase's O_B update a no-op on values that
    // init_state replaces anyway, and S_B needs no value before phase 1 writes it.
    first_scores<kQregA>(w, kb0, kb1, sa0, sa1, mx_a, c);
    if constexpr (MASKED)
        if (needs_mask(0, lim)) apply_mask(sa0, sa1, mx_a, 0, lim);
    if constexpr (RAGGED) {
        f32x16 o[4];
        if (p.flags & kFlagLoadState)  // (sub-block A's half of the image is free: its Q fragments are installed)
            load_state_rows(p, wrow, img, lane, ArrayAcc{o}, ra.m_run, ra.m_true, ra.l_run);
        else
            init_state(p, 0, g.hh, o, ra.m_run, ra.m_true, ra.l_run);
        acc_store<kAccA>(o);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] = 0.0f;
        acc_store<kAccB>(o);
        need_a = rescale_needed(mx_a, ra.m_run);
    } else {
        if (p.flags & kFlagLoadState) {
            load_state_rows(p, wrow, img, lane, OwnedAcc<kAccA>{}, ra.m_run, ra.m_true, ra.l_run);
            need_a = rescale_needed(mx_a, ra.m_run);
        } else {  // a fresh item takes its first tile's row max as reference at once
            acc_zero<kAccA>();
            ra.m_run = ra.m_true = mx_a;
            ra.l_run = 0.0f;
        }
        acc_zero<kAccB>();
    }
    rescale_if_needed<kAccA>(need_a, mx_a, ra);
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = half8{};
    rb.m_run = rb.m_true = -1e30f;
    rb.l_run = 0.0f;

    // the prologue's second group has landed behind the work above
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    q_install<kQregB>(img + 32 * kRowBytes, lane);

    // first row of the item whose O_B is being accumulated (none before the stream's first phase 1; during an
    // item's first phase 1 still the previous item).  O_A always belongs to the current item when it is written.
    size_t out_b = kNoRows;
    [[maybe_unused]] int skip_b = 0;  // rows_skipped() of the item out_b names

#ifdef FA_ABLATE_NO_WRITEOUT  // (timing experiments only: wrong results)
#define W64_WRITE_OUT(base_, st_, row_, off_, img_, skip_, zero_, zeroed_) do { zeroed_ = false; } while (0)
#else
#define W64_WRITE_OUT(base_, st_, row_, off_, img_, skip_, zero_, zeroed_)                     \
    do {                                                                                       \
        zeroed_ = false;                                                                       \
        if ((row_) != kNoRows) {                                                               \
            const size_t r_ = (row_) + (off_);                                                 \
            if constexpr (RAGGED) {                                                            \
                f32x16 o_[4];                                                                  \
                acc_load<base_>(o_);                                                           \
                if (p.flags & kFlagStoreState)                                                 \
                    store_state_rows(p, r_, (img_), lane, ArrayAcc{o_}, (st_).m_run, (st_).m_true, \
                                     (st_).l_run, (skip_) - (int)(off_));                      \
                else                                                                           \
                    store_output_rows(p.o + r_ * kHeadDim, (img_), lane, o_, (st_).l_run,      \
                                      (skip_) - (int)(off_));                                  \
            } else {                                                                           \
                acc_fence();                                                                   \
                if (p.flags & kFlagStoreState)                                                 \
                    store_state_rows(p, r_, (img_), lane, OwnedAcc<base_>{}, (st_).m_run, (st_).m_true, \
                                     (st_).l_run, 0);                                          \
                else if (W64_LEAN_OUT) {                                                       \
                    store_output_rows_owned<base_>(p.o + r_ * kHeadDim, (img_), lane, (st_).l_run, (zero_), \
                                                   wo_stamp);                                  \
                    zeroed_ = (zero_);                                                         \
                }                                                                              \
                else                                                                           \
                    store_output_rows_acc(p.o + r_ * kHeadDim, (img_), lane, OwnedAcc<base_>{}, \
                                          (st_).l_run, 0);                                     \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#endif
    // Start of an item for one sub-block, given the scaled row max mx_ of its first tile.  A fresh item
    // takes that max as its reference at once (nothing to rescale: O = 0, l = 0); a continued one
    // (blockwise mode) loads its state and decides as for any other tile.  Sets need_ accordingly.
#define W64_INIT(base_, st_, row_, mx_, need_, img_, zeroed_)                                  \
    do {                                                                                       \
        if (p.flags & kFlagLoadState) {                                                        \
            if constexpr (RAGGED) {                                                            \
                f32x16 o_[4];                                                                  \
                load_state_rows(p, (row_), (img_), lane, ArrayAcc{o_}, (st_).m_run, (st_).m_true, \
                                (st_).l_run);                                                  \
                acc_store<base_>(o_);                                                          \
            } else {                                                                           \
                load_state_rows(p, (row_), (img_), lane, OwnedAcc<base_>{}, (st_).m_run, (st_).m_true, \
                                (st_).l_run);                                                  \
            }                                                                                  \
            need_ = rescale_needed(mx_, (st_).m_run);                                          \
        } else {                                                                               \
            if (!(zeroed_)) acc_zero<base_>();                                                 \
            (st_).m_run = (st_).m_true = (mx_);                                                \
            (st_).l_run = 0.0f;                                                                \
            need_ = 0;                                                                         \
        }                                                                                      \
    } while (0)

    // One step.  Tile j of either ring lives in slot j % 3; SC = t % 3.  Phase 1 re-fills the fragment
    // buffers from K(t+1) (slot SC1) and V(t) (slot SC) while this step's transfers, K(t+3) and V(t+2),
    // go to slots SC and SC2, whose tiles K(t) and V(t-1) were last read in step t-1.  The transfers
    // have two steps to land: the wait at the end of a step leaves the step's own eight pieces in
    // flight.  Returns true at the end of the stream.
    int t = 0;
    // (head, block) of item `it` = the item after (hd, blk) in this workgroup's sequence
    [[maybe_unused]] auto next_of = [&](int hd, int blk, int it, int& nh, int& nb) {
        if constexpr (VARLEN) {
            list_item(it, nh, nb, nlq, nlk);
        } else if (!CAUSAL && inc_wrap != 0) {
            nb = blk + inc_qb;
            nh = hd + inc_head;
            if (nb >= p.q_blocks) {
                nb -= p.q_blocks;
                nh += inc_wrap;
            }
        } else {
            decode(it, nh, nb);
        }
    };
    // The step in which a workgroup sends for the next item's Q rows.  The staging image is free from the item's
    // second step on and the rows are needed in its last one, so any step from 1 to n_tiles - 2 will do; which
    // one depends on the workgroup: with equal items all workgroups of a launch reach their item boundaries
    // together, and sent in the same step everywhere the rows are one burst of 64 KB per compute unit (17 MB
    // across the chip at L = 512) that HBM serves in more than a microsecond while every wave waits (cycle stamps:
    // that step took 2 000 - 2 900 cycles longer than its neighbours, the same time at either clock).  Spread
    // over up to six steps by workgroup (bits 3-5 of its id: the XCDs stay mixed) the bursts are a sixth each.
#ifndef W64_Q_SPREAD
#define W64_Q_SPREAD 6
#endif
    const int q_class = (int)((blockIdx.x >> 3) % W64_Q_SPREAD);
    auto q_step = [&]() -> int {
        const int room = n_tiles - 3;  // steps before the last but one that may take it
        return n_tiles - 2 - (q_class < room ? q_class : room);
    };
    [[maybe_unused]] const unsigned img_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)img);
    int nhead = head, nqb = qb;  // the next item, known from its last-but-one step on
    size_t nwrow = wrow;
#ifdef W64_FINE_STAMPS
    // seven stamps per step (those of W64_FINE_STAMPS_MASK: a stamp costs about a hundred cycles and drains the
    // wave's LDS reads) for 18 steps, from the workgroup's W64_FINE_STAMPS-th step on
#ifndef W64_FINE_STAMPS_MASK
#define W64_FINE_STAMPS_MASK 0x7f
#endif
    unsigned sv0 = 0, sv1 = 0;
    int gs = 0;  // steps this workgroup has run
    auto dump_stamps = [&]() {
        if (p.stamps) {
            unsigned* o = reinterpret_cast<unsigned*>(p.stamps + 4 * 256) + ((size_t)blockIdx.x * 4 + wave) * 128;
            o[lane] = sv0;
            o[64 + lane] = sv1;
        }
    };
#define W64_STAMP(k_)                                                                          \
    do {                                                                                       \
        if ((W64_FINE_STAMPS_MASK >> (k_) & 1) && gs >= W64_FINE_STAMPS && gs < W64_FINE_STAMPS + 18) \
            fine_stamp(sv0, sv1, 7 * (gs - W64_FINE_STAMPS) + (k_));                           \
    } while (0)
#else
#define W64_STAMP(k_) do { } while (0)
#endif
#if defined(W64_FINE_STAMPS) && defined(W64_STAMP_WO)  // (points 1-4 inside sub-block A's write-out instead)
    auto wo_stamp = [&](int k_) {
        if (t + 1 == n_tiles) W64_STAMP(k_);
    };
#else
    auto wo_stamp = [](int) {};
#endif
    auto step = [&](auto sc) -> bool {
        constexpr int SC = decltype(sc)::value, SC1 = (SC + 1) % kSlots, SC2 = (SC + 2) % kSlots;
        W64_STAMP(0);
        const bool first = (t == 0), last = (t + 1 == n_tiles);
        const int next_item = item + stride;
        const bool more_items = next_item < n_items;

        // ---- this step's transfers, issued piece by piece inside phase 1: K(t+3), V(t+2)
        const half_t* k_src = p.k + kc.off;
        const half_t* v_src = p.v + v_next_off;
        v_next_off = kc.off;
        const unsigned k_dst = k_lds + (unsigned)(SC * kTileBytes) + d_piece;
        const unsigned v_dst = v_lds + (unsigned)(SC2 * kTileBytes) + d_piece;
        W64_ADVANCE(kc);
        // pieces 0-3: K, 4-7: V
        auto dma_step = [&](auto jc) {
#ifndef W64_ABLATE_DMA  // (timing experiments only: wrong results)
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) dma_set_base(k_dst);
            if constexpr (j == 4) dma_set_base(v_dst);
            if constexpr (j < 4)
                dma_piece_at<1024 * j>(k_src, d_koff[j]);
            else
                dma_piece_at<1024 * (j - 4)>(v_src, d_voff);
#endif
        };
        auto dma_none = [&](auto) {};
        auto xdma_phase2 = [&](auto gc) {
#if defined(W64_FINE_STAMPS) && defined(W64_GAP_STAMP)
            constexpr int gx = decltype(gc)::value;
            if constexpr (W64_GAP_STAMP_PHASE == 2 && gx == W64_GAP_STAMP)
                if (gs >= W64_FINE_STAMPS && gs < W64_FINE_STAMPS + 18) gap_stamp(sv0, sv1, 7 * (gs - W64_FINE_STAMPS) + 4);
#endif
        };
        // ---- QSTREAM: this step's four extra pieces (gaps 0, 1, 15, 16: one per gap, the LDS base set in the first
        //      of each pair; the tile transfers set their own bases in gaps 2 and 18)
        [[maybe_unused]] unsigned x_base = k_dst;
        [[maybe_unused]] const half_t* x_src = k_src;
        if constexpr (QSTREAM) {
            if (__builtin_expect(t == 1 && more_items, 0)) {
                next_of(head, qb, next_item, nhead, nqb);
                nwrow = source_row(nhead, nqb, W64_LQ_N);
            }
            if (more_items && t >= 1 && t <= 4) {
                // group t - 1 of the next item's 64 rows: 16 rows, 4 KiB on both sides
                x_base = img_lds + 4096u * (unsigned)(t - 1);
                x_src = p.q + (q_row(nhead, nqb, W64_LQ_N) + (size_t)(16 * (t - 1))) * kHeadDim - (size_t)wave * 2048;
            }
        }
        auto xdma_step = [&](auto gc) {
            constexpr int gx = decltype(gc)::value;
#if defined(W64_FINE_STAMPS) && defined(W64_GAP_STAMP)
            if constexpr (W64_GAP_STAMP_PHASE == 1 && gx == W64_GAP_STAMP)
                if (gs >= W64_FINE_STAMPS && gs < W64_FINE_STAMPS + 18) gap_stamp(sv0, sv1, 7 * (gs - W64_FINE_STAMPS) + 2);
#endif
            if constexpr (QSTREAM && (gx == 0 || gx == 1 || gx == 15 || gx == 16)) {
                constexpr int e = gx < 2 ? gx : gx - 13;
                if constexpr (gx == 0 || gx == 15) dma_set_base(x_base);
                dma_piece_at<1024 * e>(x_src, d_koff[e]);
            }
        };
        // ---- the next item's Q rows, once per item, in step q_step() (see there); the item's head and block are
        //      kept for the item's last step, which installs the rows.
        if (!QSTREAM && __builtin_expect(more_items && (t == q_step()), 0)) {
            next_of(head, qb, next_item, nhead, nqb);
            nwrow = source_row(nhead, nqb, W64_LQ_N);
#ifndef FA_ABLATE_NO_QDMA  // (timing experiments only: wrong results)
            const size_t qsrc = q_row(nhead, nqb, W64_LQ_N);
            dma_q_rows_asm(p.q + qsrc * kHeadDim, img, lane);
            dma_q_rows_asm(p.q + (qsrc + 32) * kHeadDim, img + 32 * kRowBytes, lane);
#endif
        }

        // ---- phase 1: matrix side B (S_B(t), O_B += V(t-1).P_B(t-1)), vector side A
        // In an item's last step A's next Q fragments are requested here: Q_A is idle while B is on the
        // matrix side, and the rows arrived with the previous step's end-of-step wait.
        if (__builtin_expect(last && more_items, 0)) q_install_issue<kQregA>(img, lane);
#ifndef W64_STAMP_WO
        W64_STAMP(1);
#endif
        fused_phase<true, kAccB, kQregB, SC1 * kTileBytes, SC * kTileBytes>(
            w, kb0, kb1, sb0, sb1, pb, mx_b, rb.m_run, need_b, sa0, sa1, ra.m_run, ra.l_run, pa, c, dma_step, xdma_step);
#ifndef W64_STAMP_WO
        W64_STAMP(2);
#endif
        if constexpr (MASKED) {
            // S_B(t): rows lim + 32 .. lim + 63 against tile t
            if (__builtin_expect(needs_mask(t, lim + 32), 0)) {
                apply_mask(sb0, sb1, mx_b, t, lim + 32);
                need_b = rescale_needed(mx_b, rb.m_run);
            }
        }
        if (__builtin_expect(first, 0)) {
            // O_B of the previous item is complete: write it out, then start the new item's state
            bool zeroed;
            W64_WRITE_OUT(kAccB, rb, out_b, 32, img + 32 * kRowBytes, skip_b, true, zeroed);
            out_b = owns_rows(qb, W64_LQ_C) ? wrow : kNoRows;
            if constexpr (RAGGED) skip_b = rows_skipped(qb, W64_LQ_C);
            W64_INIT(kAccB, rb, wrow + 32, mx_b, need_b, img + 32 * kRowBytes, zeroed);
        }
        rescale_if_needed<kAccB>(need_b, mx_b, rb);

        // ---- phase 2: matrix side A (S_A(t+1), O_A += V(t).P_A(t)), vector side B
        // In the last step S_A(t+1) belongs to the next item: Q_A must be complete.  B's next fragments are
        // requested now (Q_B is idle in phase 2, whose own wait in gap 16 covers them).
        if (__builtin_expect(last && more_items, 0)) {
            lds_wait_all();
            q_install_issue<kQregB>(img + 32 * kRowBytes, lane);
        }
#ifndef W64_STAMP_WO
        W64_STAMP(3);
#endif
        fused_phase<false, kAccA, kQregA, 0, 0>(
            w, kb0, kb1, sa0, sa1, pa, mx_a, ra.m_run, need_a, sb0, sb1, rb.m_run, rb.l_run, pb, c, dma_none, xdma_phase2);
#ifndef W64_STAMP_WO
        W64_STAMP(4);
#endif
        if constexpr (MASKED) {
            // S_A(t+1), or in an item's last step S_A(0) of the next item: rows lim .. lim + 31
            const int tile_a = last ? 0 : t + 1, pos_a = last ? row_pos(nqb, W64_LQ_N) : lim;
            if (__builtin_expect(needs_mask(tile_a, pos_a), 0)) {
                apply_mask(sa0, sa1, mx_a, tile_a, pos_a);
                need_a = rescale_needed(mx_a, ra.m_run);
            }
        }
        if (__builtin_expect(last, 0)) {
            bool zeroed;
            W64_WRITE_OUT(kAccA, ra, (owns_rows(qb, W64_LQ_C) ? wrow : kNoRows), 0, img, rows_skipped(qb, W64_LQ_C),
                          more_items, zeroed);
            if (!more_items) {
                // B is half a tile behind: O_B += V(T-1).P_B(T-1), then its rows
                pv_only<kAccB>(pb);
                W64_WRITE_OUT(kAccB, rb, out_b, 32, img + 32 * kRowBytes, skip_b, false, zeroed);
                wait_all_vmem();  // the last step's (redundant) transfers must not outlive the workgroup's LDS
#ifdef W64_FINE_STAMPS
                dump_stamps();
#endif
#ifdef FA_STAMPS
                if (p.stamps && tid == 0) {
                    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
                    o[0] = st_t0;
                    o[1] = st_r0;
                    o[2] = __builtin_amdgcn_s_memtime();
                    o[3] = __builtin_amdgcn_s_memrealtime();
                }
#endif
                return true;
            }
            W64_INIT(kAccA, ra, nwrow, mx_a, need_a, img, zeroed);
        }
        rescale_if_needed<kAccA>(need_a, mx_a, ra);

        // ---- end of step
        W64_STAMP(5);
#ifdef W64_PAD_SCALAR  // (timing experiments only: extra scalar instructions where no MFMA is in flight)
        {
            int pad_ = t;
#define W64_STR2(x_) #x_
#define W64_STR(x_) W64_STR2(x_)
            asm volatile(".rept " W64_STR(W64_PAD_SCALAR) "\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(pad_) : : "scc");
            asm volatile("" : : "s"(pad_));
        }
#endif
#ifndef W64_ABLATE_BARRIER  // (timing experiments only: wrong results)
        // Everything older than this step's eight pieces has landed: K(t+2), V(t+1), any Q rows.  (Loads
        // complete in order among themselves; stores may overtake them, which only makes this wait longer.)
#ifndef W64_ABLATE_WAIT  // (timing experiments only: wrong results)
        if constexpr (QSTREAM)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // (twelve pieces per step)
        else
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
        W64_STAMP(6);
#ifndef W64_ABLATE_SYNC  // (timing experiments only: wrong results)
        if constexpr (W64_BAR_GAP < 0) __syncthreads();  // (else inside the next step's phase 1)
#endif
#endif
#ifdef W64_FINE_STAMPS
        ++gs;
#endif
        if (__builtin_expect(last, 0)) {
            item = next_item;
            head = nhead;
            qb = nqb;
            wrow = nwrow;
            if constexpr (VARLEN) {
                lq_c = nlq;
                lk_c = nlk;
                item_shape(qb, lk_c, n_tiles, dup);
            } else if constexpr (CAUSAL) {
                n_tiles = item_tiles(qb);
            }
            if constexpr (MASKED) lim = row_pos(qb, W64_LQ_C);
            t = -1;
        }
        ++t;
        return false;
    };
    for (;;) {
        if (step(std::integral_constant<int, 0>{})) break;
        if (step(std::integral_constant<int, 1>{})) break;
        if (step(std::integral_constant<int, 2>{})) break;
    }
}

#ifndef W64_QSTREAM_DEFAULT  // (experiment builds: the plain kernel in the Q-streaming form, see QSTREAM)
#define W64_QSTREAM_DEFAULT false
#endif
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_kernel(const FwdParams p) {
    fa_fwd_w64_body<false, false, false, W64_QSTREAM_DEFAULT>(p);
}

#ifndef FA_W64_ONLY  // (experiment builds of the default kernel alone, tools/build_w64_only.sh, leave the other forms out)
// The same with the causal mask, lengths in whole tiles.
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_causal_kernel(const FwdParams p) { fa_fwd_w64_body<true, false>(p); }
// The same for lengths that are not whole tiles, without and with the causal mask.
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_ragged_kernel(const FwdParams p) { fa_fwd_w64_body<false, true>(p); }
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_ragged_causal_kernel(const FwdParams p) {
    fa_fwd_w64_body<true, true>(p);
}
// The ragged forms over a device-built list of items with their own lengths (see VARLEN above).
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_varlen_kernel(const FwdParams p) {
    fa_fwd_w64_body<false, true, false, false, true>(p);
}
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_varlen_causal_kernel(const FwdParams p) {
    fa_fwd_w64_body<true, true, false, false, true>(p);
}
// The item list of the fa_fwd_w64_varlen kernels, built where the lengths live.  One workgroup: per sequence the
// valid lengths (clamped to the padded ones; a null array means the padded length) and whether the tuned kernel
// takes the sequence (varlen_on_tuned); the sequences in descending order of keys (batches of up to kVarlenSortMax
// sequences, larger ones keep their order); a scan over groups of entries; then every thread writes its share of
// the entries (head, block, query rows, keys).
//   Without the mask an item costs its sequence's keys, and a group is a sequence: H * ceil(q / 256) entries, the
// heads in groups of eight with the group's blocks interleaved (block 0 of heads 0-7, block 1 of heads 0-7, ...):
// with H a multiple of 8 an entry's index is then congruent to its head mod 8, workgroup ids congruent mod 8 share
// an XCD, and a head's K/V stay in one L2, as in decode_item.
//   With the mask an item costs its block's diagonal, 4 (block + 1) tiles, and a group is a block index, the
// highest first (decode_item_causal's order): the sequences that have that block are a prefix of the sorted order,
// H entries each.  (This order needs the sort: the launcher keeps larger masked batches off this kernel.)
//   A batch whose tuned sequences hold less than min_work rows x keys (summed over the heads), or fewer items than
// `grid` workgroups, is left to the 128-row kernel altogether (count 0, thresholds out of reach): the launcher's rule
// on the padded lengths, applied to the lengths themselves.
//   Workgroup w of `grid` takes entries w, w + grid, ...: the entries are laid down in rounds of `grid`, every other
// round back to front (the last one forwards, so that it has no holes), which hands the workgroup with the
// costliest item of one round the cheapest of the next.  With a few items per workgroup whose costs differ by a
// factor of three, the busiest workgroup then carries 7 - 10 % more than the mean instead of 15 - 18 %.
constexpr int kVarlenMaxBatch = 4096;
constexpr int kVarlenSortMax = 512;
constexpr int kVarlenMaxBlocks = 4096;  // (with the mask: block indices are groups)
__global__ __launch_bounds__(256) void fa_varlen_plan_kernel(const int* __restrict__ q_len, const int* __restrict__ k_len,
                                                             int B, int H, int lq_pad, int lk_pad, int long_q, int long_k,
                                                             int causal, int grid, float min_work,
                                                             int* __restrict__ items, int* __restrict__ count) {
    __shared__ int first[kVarlenMaxBatch + 1];  // first[g]: index of group g's first entry
    __shared__ int len_q[kVarlenMaxBatch], len_k[kVarlenMaxBatch];  // valid lengths; len_k = 0: not the tuned kernel's
    __shared__ unsigned short order[kVarlenMaxBatch];               // order[r]: which sequence comes r-th
    __shared__ int scan[256];
    __shared__ float wsum[2][256];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) {
        carry = 0;
        first[0] = 0;
    }
    // The thresholds of this launch.  The two kernels run one after the other, and a handful of sequences below the
    // thresholds makes the second one a launch of a few workgroups whose longest sets its time; when the
    // sequences below them that the list form can take at all (64 rows and 192 keys; with the mask as many rows as
    // keys) hold at most a quarter of the batch's work, they join the list instead and the thresholds drop to
    // that minimum.  count[1], count[2] carry them to the 128-row kernel.
    float w_long = 0.0f, w_mid = 0.0f;
    const int min_q = causal ? 3 * kKvTile : kKvTile, min_k = 3 * kKvTile;
    for (int b = tid; b < B; b += 256) {
        const int ql = q_len ? min(max(q_len[b], 0), lq_pad) : lq_pad;
        const int kl = k_len ? min(max(k_len[b], 0), lk_pad) : lk_pad;
        len_q[b] = ql;
        len_k[b] = kl;
        const float w = (float)ql * (float)kl;
        if (varlen_on_tuned(ql, kl, long_q, long_k, causal != 0))
            w_long += w;
        else if (varlen_on_tuned(ql, kl, min_q, min_k, causal != 0))
            w_mid += w;
    }
    wsum[0][tid] = w_long;
    wsum[1][tid] = w_mid;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) {
            wsum[0][tid] += wsum[0][tid + d];
            wsum[1][tid] += wsum[1][tid + d];
        }
        __syncthreads();
    }
    float w_tuned = wsum[0][0];
    if (wsum[0][0] > 0.0f && wsum[1][0] <= 0.25f * (wsum[0][0] + wsum[1][0])) {
        long_q = min_q;
        long_k = min_k;
        w_tuned += wsum[1][0];
    }
    // Too little for the list form to pay (min_work: rows x keys summed over heads and the sequences it would take;
    // twice that with the mask, half of which is above the diagonal): it takes nothing, the 128-row kernel
    // everything.
    if ((float)H * w_tuned < (causal ? 2.0f : 1.0f) * min_work) long_q = long_k = 0x7fffffff;
    if (tid == 0) {
        count[1] = long_q;
        count[2] = long_k;
    }
    for (int b = tid; b < B; b += 256)
        if (!varlen_on_tuned(len_q[b], len_k[b], long_q, long_k, causal != 0)) len_k[b] = 0;
    __syncthreads();
    if (B <= kVarlenSortMax) {
        for (int b = tid; b < B; b += 256) {  // rank by counting: costlier sequences first, equal ones by index
            const int c = len_k[b];
            int rank = 0;
            for (int j = 0; j < B; ++j) rank += len_k[j] > c || (len_k[j] == c && j < b);
            order[rank] = (unsigned short)b;
        }
    } else {
        for (int b = tid; b < B; b += 256) order[b] = (unsigned short)b;
    }
    __syncthreads();
    // blocks of the r-th sequence (0: not the tuned kernel's)
    auto blocks_of = [&](int r) { return len_k[order[r]] > 0 ? (len_q[order[r]] + 255) / 256 : 0; };
    const int top = causal ? blocks_of(0) : 0;  // masked: groups are block indices top - 1 ... 0
    const int groups = causal ? top : B;
    for (int g0 = 0; g0 < groups; g0 += 256) {  // inclusive scan of the groups' entry counts, 256 at a time
        const int g = g0 + tid;
        int n = 0;
        if (g < groups) {
            if (causal) {  // sequences with more than `blk` blocks: a prefix of the (descending) order
                const int blk = top - 1 - g;
                int lo = 0, hi = B;  // first r with blocks_of(r) <= blk
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (blocks_of(mid) > blk) lo = mid + 1; else hi = mid;
                }
                n = H * lo;
            } else {
                n = H * blocks_of(g);
            }
        }
        scan[tid] = n;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? scan[tid - d] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        if (g < groups) first[g + 1] = carry + scan[tid];
        __syncthreads();
        if (tid == 255) carry += scan[255];
        __syncthreads();
    }
    if (min_work > 0.0f && carry < grid) {  // (uniform: carry is shared)
        if (tid == 0) {
            count[0] = 0;
            count[1] = count[2] = 0x7fffffff;
        }
        return;
    }
    if (tid == 0) *count = carry;
    const int total = carry;
    const int last_round = (total - 1) / grid;
    for (int i = tid; i < total; i += 256) {
        int lo = 0, hi = groups - 1;  // the group g with first[g] <= i < first[g + 1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (first[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int j = i - first[lo];
        int b, blk, hd;
        if (causal) {
            const int r = j / H;
            b = order[r];
            hd = j - r * H;
            blk = top - 1 - lo;
        } else {
            b = order[lo];
            const int nblk = (len_q[b] + 255) / 256;
            // groups of eight heads (the last one smaller), a group's blocks interleaved
            const int per_group = 8 * nblk, grp = j / per_group;
            const int gh = min(8, H - 8 * grp), r = j - grp * per_group;
            blk = r / gh;
            hd = 8 * grp + (r - blk * gh);
        }
        const int round = i / grid, pos = i - round * grid;
        const int dest = (last_round - round) & 1 ? round * grid + (grid - 1 - pos) : i;
        int* e = items + 4 * (size_t)dest;
        e[0] = b * H + hd;
        e[1] = blk;
        e[2] = len_q[b];
        e[3] = len_k[b];
    }
}

// Whole tiles, no mask, every head's keys split into ranges (see SPLIT above).
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_split_kernel(const FwdParams p) {
    fa_fwd_w64_body<false, false, true>(p);
}
#endif

// Folds the key ranges of fa_fwd_w64_split_kernel: row i of the output is
//   O[i] = sum_s w_s o_s[i] / sum_s w_s l_s[i],   w_s = exp(m_s[i] - max_s m_s[i]),
// the online-softmax merge of testing.cu:91-96 applied across ranges (o_s, m_s, l_s: row s * rows + i of the
// state arrays, m in natural-log units as the blockwise entry points keep it).  Half a wave per row: lane c holds
// columns 4c .. 4c+3.  A range that saw no key (l = 0) contributes nothing.
__global__ __launch_bounds__(256) void fa_split_merge_kernel(const float* __restrict__ o_acc, const float* __restrict__ row_max,
                                                             const float* __restrict__ row_sum, half_t* __restrict__ o,
                                                             size_t rows, int n_ranges) {
    const int c = threadIdx.x & 31;
    const size_t per_pass = (size_t)gridDim.x * 8;
    for (size_t i = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5); i < rows; i += per_pass) {
        float m = -__builtin_inff();
        for (int s = 0; s < n_ranges; ++s) m = fmaxf(m, row_max[(size_t)s * rows + i]);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        float l = 0.0f;
        for (int s = 0; s < n_ranges; ++s) {
            const size_t r = (size_t)s * rows + i;
            const float w = __expf(row_max[r] - m);
            l += w * row_sum[r];
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(o_acc + r * kHeadDim) + c);
            acc += t * w;
        }
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        const half4 h = __builtin_convertvector(acc * inv, half4);
        __builtin_nontemporal_store(h, reinterpret_cast<half4*>(o + i * kHeadDim) + c);
    }
}

#undef W64_STAMP
#undef W64_LQ_C
#undef W64_LQ_N
#undef W64_LK_C
#undef W64_ADVANCE
#undef W64_DMA_K
#undef W64_DMA_V
#undef W64_WRITE_OUT
#undef W64_INIT

}  // namespace fa
