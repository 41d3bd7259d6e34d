/*
 * C ABI of libflash_attention.so (MI355X / gfx950 build).
 *
 * The first two functions are the drop-in boundary: same names, parameter order
 * and widths as the reference's
 *   flash_attention_tensor_cores/wrapper.cuh:8-19   run_tensor_flash_attention
 *   flash_attention_tensor_cores/wrapper.cuh:21-32  run_tensor_flash_attention_host_half
 * (`__half*` is a 16-bit pointer, `cudaStream_t` a pointer-sized handle; here
 * they are spelled uint16_t* and void* so that the header needs no GPU
 * toolkit).  The ctypes view used by benchmarks/compare_impls.py:34-40,
 * [POINTER(c_uint16)]*4 + [c_int]*5 + [c_void_p, POINTER(c_float)], restype
 * None, binds to them unchanged.
 *
 * Tensors are contiguous row-major [B][H][L][D] IEEE binary16
 * (element (((b*H+h)*L+i)*D+d), testing.cu:49-51).  D must be 128,
 * `tile` must be 64 (kept for compatibility: it is the key-tile cadence of the
 * online softmax, wrapper.cu:53-57), L must be a multiple of 64
 * (wrapper.cu:59-61).  Unlike the reference (wrapper.cu:89-185), every B, H and
 * every such L is accepted and honoured at run time.
 *
 * The fa_* functions are additions of this library (status-returning twins,
 * the blockwise entry point used by the sequence-parallel ring, the benchmark
 * driver's input generator, device facts for the roofline).
 */
#ifndef FLASH_ATTENTION_H
#define FLASH_ATTENTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define FA_DEFAULT(x) = x
extern "C" {
#else
#define FA_DEFAULT(x)
#endif

/* ---------------------------------------------------------------------------
 * Drop-in entry points
 * ------------------------------------------------------------------------- */

/* Replaces wrapper.cu:39-203.  Device pointers.  Asynchronous on `stream` when
 * elapsed_ms is NULL; when non-NULL the call brackets the kernel with events,
 * waits for it and stores the kernel-only milliseconds (wrapper.cu:79-85,
 * 188-196).  Errors (bad tile / L / D, launch failure) are raised the way the
 * reference raises them: a std::runtime_error thrown through the C boundary
 * (wrapper.cu:56,60,201), i.e. std::terminate under a C or ctypes caller. */
void run_tensor_flash_attention(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO,
                                int B, int H, int L, int D, int tile,
                                void* stream FA_DEFAULT(0), float* elapsed_ms FA_DEFAULT(0));

/* Replaces wrapper.cu:9-35.  Host pointers: allocates four device buffers,
 * copies Q,K,V host->device on `stream`, runs the kernel, copies O back, waits,
 * frees.  *elapsed_ms is kernel-only time, copies excluded. */
void run_tensor_flash_attention_host_half(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV,
                                          uint16_t* hO, int B, int H, int L, int D, int tile,
                                          void* stream FA_DEFAULT(0), float* elapsed_ms FA_DEFAULT(0));

/* ---------------------------------------------------------------------------
 * Additions (no counterpart in the reference unless cited)
 * ------------------------------------------------------------------------- */

enum {
    FA_OK = 0,
    FA_ERR_TILE = 1,      /* tile != 64            (wrapper.cu:53-57) */
    FA_ERR_SEQLEN = 2,    /* L % tile != 0 or <= 0 (wrapper.cu:59-61) */
    FA_ERR_HEAD_DIM = 3,  /* D != 128 (the reference hard-codes it: cuda_utils.cuh:11) */
    FA_ERR_SHAPE = 4,     /* B or H <= 0, or sizes overflow */
    FA_ERR_POINTER = 5,   /* NULL tensor pointer */
    FA_ERR_RUNTIME = 6,   /* HIP runtime / launch failure; text in fa_last_error() */
    FA_ERR_WORKSPACE = 7  /* fa_forward_noalloc: the call needs more scratch memory than the static buffer holds */
};

/* Same work and arguments as run_tensor_flash_attention, but returns a status
 * instead of throwing; the two drop-in functions are thin shells over these.
 *
 * Launches with fewer (head, 256-row query block) items than TWICE the resident workgroups (one per compute unit,
 * less fa_set_reserved_cus) may divide every head's keys into ranges when a cost model says it pays
 * (fa_split_ways() is the authority; typically few heads and a long sequence, e.g. one rank's two heads of B=1,
 * H=16, L=16384 sharded over eight GPUs: 128 items on 256 units): they run the ranges as heads of their own and fold them with a
 * second, small kernel.  The reference never meets this: its grid is (L/64, B, H) workgroups of 64 rows
 * (wrapper.cu:74-76).  The workspace (520 bytes per query row and range) is taken from the stream's memory pool
 * (hipMallocAsync / hipFreeAsync: stream-ordered, no device synchronisation) and returned to it by the same call;
 * other launches allocate nothing.  On a stream under capture nothing is allocated: such a shape is launched
 * undivided there (fa_forward_noalloc launches the divided form in a capture as well). */
int fa_forward(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, int B, int H, int L,
               int D, int tile, void* stream, float* elapsed_ms);
int fa_forward_host_half(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV, uint16_t* hO, int B,
                         int H, int L, int D, int tile, void* stream, float* elapsed_ms);

/* The host-pointer entry point over several GPUs of one node, from ONE process.  No counterpart in the reference
 * (README.md:47 lists multi-GPU as future work); the partition follows from its kernel's grid, whose (batch, head)
 * axes are independent (tensor_flash_attention.cu:30-31): the flattened B*H axis is cut into n_devices contiguous
 * ranges (the first B*H % n_devices of them one head longer), range i runs on GPU devices[i] (devices == NULL:
 * GPU i; n_devices <= 0 with devices == NULL: every visible GPU).  Each range is one contiguous piece of the four
 * host buffers and is handled by a host thread of its own exactly as fa_forward_host_half handles the whole
 * tensor (device buffers, copy in, kernel, copy out, free, on a stream of its own), so the copies of different
 * GPUs run side by side on their own links and nothing is exchanged between GPUs.  A GPU may be named more than
 * once (its ranges then share the device); ranges without a head (n_devices > B*H) do nothing.
 * elapsed_ms_per_device: NULL, or n_devices floats that receive each range's kernel-only milliseconds (0 for an
 * empty range); total FLOPs over their maximum is the aggregate rate.  Returns after every range has finished;
 * the first failing range's status is returned and its message (with the device) is in fa_last_error().  The
 * caller's current device is left as it was. */
int fa_forward_host_half_sharded(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV, uint16_t* hO, int B,
                                 int H, int L, int D, int tile, const int* devices, int n_devices,
                                 float* elapsed_ms_per_device);

/* fa_forward for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid it.  No counterpart in the reference.  Same arguments, result and kernels as fa_forward
 * minus elapsed_ms; it calls no hipMalloc / hipFree / hipMallocAsync or any other allocating
 * routine, creates no events and never waits.  A launch that divides its keys (see fa_forward) takes its workspace
 * from a buffer of 34 078 720 bytes that is part of the library's device image (65 536 query rows x ranges: B*H*L*S
 * <= 65536, e.g. B=1, H=2, L=16384 in two ranges); every GPU the library runs on holds that buffer until the
 * process ends.  The cost model is bounded by that buffer: the entry point takes the best split that fits, or launches
 * undivided (which needs no scratch at all; most shapes: a launch is only ever divided when it has fewer
 * (head, 256-row block) items than twice the resident workgroups, see fa_split_ways), so its output has fa_forward's
 * bits whenever both choose the same split and passes the same parity gate otherwise.  Only a split FORCED with
 * FA_SPLIT=n that does not fit returns FA_ERR_WORKSPACE (sizes in fa_last_error()) and launches nothing.  All calls on
 * one GPU share the buffer: two calls that divide their keys and overlap in time on one GPU (different streams
 * or host threads) race on it; calls on one stream are ordered and safe.  The first call on a device looks
 * the buffer's address up and must not be made under stream capture. */
int fa_forward_noalloc(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, int B, int H, int L,
                       int D, int tile, void* stream);

/* Blockwise forward: folds one K/V block of Lk keys into the carried online-
 * softmax state of Lq query rows, the rule of testing.cu:91-96 /
 * tensor_flash_attention.cu:158-170 applied across blocks instead of across
 * 64-key tiles.  This is the compute step of the sequence-parallel ring.
 *   dQ            [BH][Lq][128] fp16       dK, dV  [BH][Lk][128] fp16
 *   o_acc         [BH][Lq][128] fp32 un-normalised accumulator
 *   row_max       [BH][Lq] fp32 running max of score/sqrt(D)
 *   row_sum       [BH][Lq] fp32 running sum of exp(score/sqrt(D)-row_max)
 *   first != 0    ignore the incoming state (start from zero)
 *   last  != 0    write dO = o_acc/row_sum as fp16 and leave the state untouched;
 *                 otherwise update o_acc,row_max,row_sum and leave dO untouched.
 *                 A row that has seen no key at all by then (row_sum == 0: possible only through
 *                 fa_forward_block_general's mask) gets O = 0.
 * Lq and Lk must be multiples of 64.  Asynchronous on `stream`. */
int fa_forward_block(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, float* o_acc,
                     float* row_max, float* row_sum, int BH, int Lq, int Lk, int D, int first, int last,
                     void* stream);

/* General forward: any Lq >= 1 and Lk >= 1 (no multiple-of-64 rule, query and key lengths may
 * differ), head dim D of 64, 128 or 256 and, with FA_FLAG_CAUSAL, the causal mask of the
 * reference's CPU baseline (attention_cpu's `causal` argument, testing.cu:41-48,74-77: key
 * j > query i takes no part; needs Lq == Lk).  The reference's GPU path has none of these: its
 * wrapper rejects L % 64 != 0 (wrapper.cu:59-61), D is fixed at 128 (cuda_utils.cuh:11) and its
 * kernel has no mask; README.md:44-47 lists them as future work.
 *   dQ, dO  [B][H][Lq][D] fp16        dK, dV  [B][H][Lk][D] fp16        scale 1/sqrt(D)
 * Shapes the tuned kernels take (D == 128, Lq == Lk, a multiple of 64, no flags) are passed on to
 * fa_forward; other D == 128 shapes (the causal mask, lengths that are not multiples of 64, Lq != Lk)
 * run on masked forms of the same kernel where those are faster than the general one
 * (fa_kernel_name_general tells).  Asynchronous on `stream` unless elapsed_ms is non-NULL (as fa_forward).
 * A launch without the mask that would leave most of the chip idle (a few query rows against a long sequence,
 * or few heads) divides every head's keys over workgroups (fa_split_ways_general tells): its workspace,
 * (D + 2) floats per query row and range, comes from the stream's memory pool and goes back to it in the same
 * call, as in fa_forward; a stream under capture gets the undivided launch.  fa_forward_varlen does the same. */
enum { FA_FLAG_CAUSAL = 1 };
int fa_forward_general(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, int B, int H,
                       int Lq, int Lk, int D, int flags, void* stream, float* elapsed_ms);

/* Padded batches (the padding mask): sequence b of the batch uses only its first q_len[b] query rows and its first
 * k_len[b] keys of the padded [B][H][Lq][D] / [B][H][Lk][D] tensors; what lies behind them is never read (keys,
 * values) or written (rows of dO from q_len[b] on keep their bytes).  Per sequence the result is what
 * attention_cpu gives on the truncated tensors: its 64-key tiles end at the sequence's last key
 * (testing.cu:65-66).  d_q_len and d_k_len are DEVICE arrays of B ints, read by the kernel (no host
 * synchronisation); either may be NULL (= Lq, Lk for every sequence); values are clamped to [0, Lq] and [0, Lk]; a
 * sequence without keys gets O = 0.  With FA_FLAG_CAUSAL key j is visible to query i iff j <= i (both counted from
 * the sequence's start; Lq and Lk may differ).  D of 64, 128 or 256.
 *   At D = 128 the long sequences of a batch (without the mask those that cost less there by a model of both
 * kernels, from about 640 rows and keys on; with it as many rows as keys and at least 1665, or 2305 when the last
 * 256-row block is at most half full) run on the default
 * kernel's ragged forms from a device-built item list (16 bytes per head and 256-row block of the padded batch, from
 * the stream's memory pool, hipMallocAsync; not on a stream under capture), the other sequences after them on the
 * 128-row kernel, unless they hold so little of the work that they join the list; batches too small to fill the
 * chip that way stay on the 128-row kernel.  FA_VARLEN_W64=0 in the environment keeps everything on the 128-row
 * kernel, 1 takes the list form wherever it can run.
 *   The reference has no mask of this kind: its
 * README lists "causal and padding masks" as future work (README.md:44), its naive kernel knows the causal one
 * only (normal_attention.cu:50,64,81). */
int fa_forward_varlen(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, int B, int H, int Lq,
                      int Lk, int D, int flags, const int* d_q_len, const int* d_k_len, void* stream,
                      float* elapsed_ms);

/* Blockwise form of fa_forward_general (the state and the first/last convention of fa_forward_block,
 * any Lq, Lk >= 1, D of 64, 128 or 256).  With FA_FLAG_CAUSAL key j of this block is visible to query
 * i of this block iff k_pos0 + j <= q_pos0 + i, the positions being those of the rows in the whole
 * sequence: the blocks of a causal sequence can then be folded in any order (a row that sees no key
 * of a block keeps its state), which the zig-zag sequence-parallel ring needs.  attention_cpu has
 * the mask for whole sequences only (testing.cu:74-77).  Setting `last` while some row has seen no key
 * of any block (a block wholly above the diagonal folded alone, q_pos0 < k_pos0 on a first block) is
 * accepted: those rows of dO are written as 0, never NaN. */
int fa_forward_block_general(const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV, uint16_t* dO, float* o_acc,
                             float* row_max, float* row_sum, int BH, int Lq, int Lk, int D, int flags, int q_pos0,
                             int k_pos0, int first, int last, void* stream);

/* fa_forward_general on host pointers: device buffers allocated per call, copies in and out on
 * `stream`, waits (the shell of wrapper.cu:21-34). */
int fa_forward_general_host_half(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV, uint16_t* hO, int B,
                                 int H, int Lq, int Lk, int D, int flags, void* stream, float* elapsed_ms);

/* Entry points of the reference's two sibling libraries, so that its three-way caller
 * (benchmarks/compare_impls.py:17-28,42-63) finds all three symbols in this library (point
 * libnaive_attention.so and libflash_attention_cuda_cores.so at it).  Same kernels as above; host
 * pointers; errors thrown as the reference throws them.
 *   run_naive_attention_host_half             replaces naive_gpu_attention/wrapper.cu:7-39;
 *                                             `causal` != 0 applies the mask (normal_attention.cu:50,64,81)
 *   run_cuda_cores_flash_attention_host_half  replaces flash_attention_cuda_cores/wrapper.cu:7-40; its two
 *                                             launch-shape arguments are checked for being positive and
 *                                             otherwise ignored */
void run_naive_attention_host_half(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV, uint16_t* hO, int B,
                                   int H, int L, int D, int causal, void* stream, float* elapsed_ms);
void run_cuda_cores_flash_attention_host_half(const uint16_t* hQ, const uint16_t* hK, const uint16_t* hV,
                                              uint16_t* hO, int B, int H, int L, int D, int KQVO_block_y,
                                              int number_of_warps_per_block, void* stream, float* elapsed_ms);

/* Device-side input generator of the reference's throughput driver
 * (benchmark_tensor_flash_attention.cu:34-41,68-72): element i of the buffer
 * becomes half(fmodf((float)((i*16807+seed)&0xFFFF),1024)/1024-0.5). */
int fa_init_half(uint16_t* dptr, size_t n, unsigned long long seed, void* stream);

/* Facts about the current device for the roofline denominator:
 * compute units, max engine clock in kHz, and the dense fp16 MFMA peak in
 * FLOP/s computed as CUs * 4 SIMDs * 1024 FLOP/clk/SIMD * clock
 * (v_mfma_f32_32x32x16_f16: 32768 FLOP per 32 cycles per SIMD). */
int fa_device_info(int* compute_units, int* clock_khz, double* fp16_dense_peak_flops);

/* Persistent launches (the D = 128 kernels: one resident workgroup per compute unit, each holding the unit's
 * whole LDS) leave k compute units unused from the next call on, so that kernels of OTHER streams can become
 * resident while an attention kernel runs: the transport kernels of a ring step's send/recv next to its
 * compute.  Process-wide, 0 by default (initial value: environment variable FA_RESERVED_CUS).  No counterpart
 * in the reference, which has no multi-device mode. */
int fa_set_reserved_cus(int k);
int fa_get_reserved_cus(void);

/* The library reads its FA_* environment switches (kernel selection for A/B comparisons and tests) once per
 * process; this takes a new snapshot.  For tests; not for concurrent use with launches. */
void fa_debug_reload_env(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char* fa_last_error(void);

/* Library version string. */
const char* fa_version(void);

/* Name of the device kernel that serves Lq query rows against Lk keys per head (as rocprofv3
 * reports it, without template arguments' spelling differences); for benchmark records.  No
 * counterpart in the reference, whose wrapper picks among seven instantiations of one kernel
 * (flash_attention_tensor_cores/wrapper.cu:89-185). */
const char* fa_kernel_name(int lq, int lk);
/* Into how many key ranges a call of fa_forward on B*H = bh heads of lq query rows and lk keys divides each head
 * on the current device (1: no split; then fa_kernel_name names the only kernel, otherwise
 * fa::fa_fwd_w64_split_kernel runs, followed by fa::fa_split_merge_kernel). */
int fa_split_ways(int bh, int lq, int lk);
/* The same for a call of fa_forward_general (flags as there).  For unmasked lengths that are not whole tiles at
 * D = 128 the choice between the ragged form of the default kernel and the 128-row kernel also depends on the batch
 * (whole rounds of 256-row items or not): fa_kernel_name_general_bh takes B*H as well and names the kernel the call
 * launches; without it (bh = 0) the answer is the one for batches that do not fill the chip. */
const char* fa_kernel_name_general(int lq, int lk, int D, int flags);
const char* fa_kernel_name_general_bh(int bh, int lq, int lk, int D, int flags);
/* Into how many key ranges a call of fa_forward_general divides each head when fa_kernel_name_general names one of the
 * fa::fa_fwd_hd_* kernels (a handful of query rows against a long sequence, or few heads: too few workgroups to fill
 * the chip); 1: no split; otherwise fa::fa_fwd_hd_state_kernel<D> runs on the ranges, followed by
 * fa::fa_hd_split_merge_kernel<D>.  No counterpart in the reference, whose grid is (L/64, B, H)
 * (flash_attention_tensor_cores/wrapper.cu:74-76). */
int fa_split_ways_general(int bh, int lq, int lk, int D, int flags);
/* The same for a call of fa_forward_block_general. */
const char* fa_kernel_name_block_general(int lq, int lk, int D, int flags, int q_pos0, int k_pos0);

#ifdef __cplusplus
}
#endif
#undef FA_DEFAULT
#endif /* FLASH_ATTENTION_H */
