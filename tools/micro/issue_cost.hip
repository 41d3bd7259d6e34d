// Microbenchmark: cycles per instruction for ONE wave per SIMD (256 threads per workgroup, one workgroup per CU),
// the situation of the one-wave-per-SIMD attention kernel's item-boundary code: straight-line runs of one
// instruction kind, independent of each other or each depending on the one before.
//   hipcc -O3 --offload-arch=gfx950 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
    __shared__ float lds[64 * 1024 / 4];
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, y = 0.5f;
    unsigned u0 = threadIdx.x * 8, u1 = 1;
    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
    h8_t h8 = {1, 2, 3, 4, 5, 6, 7, 8};
    asm volatile("" : "+v"(h8));
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(y), "+v"(u0), "+v"(u1));
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0" ::"v"(y) : "a0", "a1", "a2", "a3");
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if constexpr (KIND == 0) {  // v_mov independent (8 different registers)
        REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                           "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
    } else if constexpr (KIND == 1) {  // v_mul dependent chain
        REP256(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(y));)
    } else if constexpr (KIND == 2) {  // v_mul independent
        REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                           "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
    } else if constexpr (KIND == 3) {  // v_accvgpr_read independent
        REP16(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a0\n v_accvgpr_read_b32 %5, a1\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n"
                           "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a0\n v_accvgpr_read_b32 %5, a1\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3"
                           : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7));)
    } else if constexpr (KIND == 4) {  // v_fma_mixlo independent
        REP16(asm volatile("v_fma_mixlo_f16 %0, %0, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %1, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %2, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %3, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %4, %4, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %5, %5, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %6, %6, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %7, %7, %8, 0 op_sel_hi:[0,0,0]\n"
                           "v_fma_mixlo_f16 %0, %0, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %1, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %2, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %3, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %4, %4, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %5, %5, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %6, %6, %8, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %7, %7, %8, 0 op_sel_hi:[0,0,0]"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
    } else if constexpr (KIND == 5) {  // v_cvt_pk_f16_f32 independent
        REP16(asm volatile("v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8\n"
                           "v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
    } else if constexpr (KIND == 6) {  // ds_write_b64, all four waves at once
        REP16(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n"
                           "ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:4608\n ds_write_b64 %0, %1 offset:5120\n ds_write_b64 %0, %1 offset:5632\n ds_write_b64 %0, %1 offset:6144\n ds_write_b64 %0, %1 offset:6656\n ds_write_b64 %0, %1 offset:7168\n ds_write_b64 %0, %1 offset:7680"
                           : : "v"(u0), "v"(*(double*)&a0) : "memory");)
        asm volatile("s_waitcnt lgkmcnt(0)");
    } else if constexpr (KIND == 7) {  // exp independent
        REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                           "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 8) {  // s_add independent-ish
        int s0 = 1;
        asm volatile("" : "+s"(s0));
        REP256(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");)
        asm volatile("" : : "s"(s0));
    } else if constexpr (KIND == 9) {  // s_cmp + not-taken branch
        int s0 = 0;
        asm volatile("" : "+s"(s0));
        REP256(asm volatile("s_bitcmp1_b32 %0, 0\n s_cbranch_scc1 9f\n9:" : : "s"(s0) : "scc");)
    } else if constexpr (KIND == 10) {  // v_pk_mul_f32 independent
        REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                           "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                           : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&a0));)
    }
    else if constexpr (KIND == 11) {  // v_exp_f32 and v_mul_f32 alternating, independent
        REP16(asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %4, %4, %8\n v_exp_f32 %1, %1\n v_mul_f32 %5, %5, %8\n v_exp_f32 %2, %2\n v_mul_f32 %6, %6, %8\n v_exp_f32 %3, %3\n v_mul_f32 %7, %7, %8\n"
                           "v_exp_f32 %0, %0\n v_mul_f32 %4, %4, %8\n v_exp_f32 %1, %1\n v_mul_f32 %5, %5, %8\n v_exp_f32 %2, %2\n v_mul_f32 %6, %6, %8\n v_exp_f32 %3, %3\n v_mul_f32 %7, %7, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
    } else if constexpr (KIND == 12) {  // one MFMA 32x32x16 per 7 independent v_mul (8 instructions per group)
        REP16(asm volatile("v_mfma_f32_32x32x16_f16 a[16:31], %8, %8, a[16:31]\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n"
                           "v_mfma_f32_32x32x16_f16 a[32:47], %8, %8, a[32:47]\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h8), "v"(y)
                           : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");)
    } else if constexpr (KIND == 13) {  // MFMA 32x32x16 alone, two accumulators alternating
        REP16(asm volatile(".rept 8\n v_mfma_f32_32x32x16_f16 a[16:31], %0, %0, a[16:31]\n v_mfma_f32_32x32x16_f16 a[32:47], %0, %0, a[32:47]\n .endr"
                           : : "v"(h8)
                           : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");)
    } else if constexpr (KIND == 14) {  // one MFMA per 3 v_mul + 1 v_exp + 1 v_add-like (the phase's mix: 6 per group)
        REP16(asm volatile(".rept 4\n v_mfma_f32_32x32x16_f16 a[16:31], %8, %8, a[16:31]\n v_fma_f32 %0, %0, %9, %9\n v_exp_f32 %1, %1\n v_add_f32 %2, %2, %9\n v_cvt_pk_f16_f32 %3, %4, %5\n"
                           "v_mfma_f32_32x32x16_f16 a[32:47], %8, %8, a[32:47]\n v_fma_f32 %4, %4, %9, %9\n v_exp_f32 %5, %5\n v_add_f32 %6, %6, %9\n .endr"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h8), "v"(y)
                           : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x] + (float)u1;
}

template <int KIND>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
    std::vector<unsigned long long> h(256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d_out, d_sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, sum = 0;
    for (auto v : h) { mn = v < mn ? v : mn; sum += v; }
    printf("%-34s 256 instructions: min %5llu mean %6.0f cycles  -> %.1f per instruction\n", name, mn, sum / 256.0, mn / 256.0);
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 256 * 8); hipMalloc(&d_sink, 256 * 256 * 4);
    run<0>("v_mov_b32 independent", d_out, d_sink);
    run<1>("v_mul_f32 dependent chain", d_out, d_sink);
    run<2>("v_mul_f32 independent", d_out, d_sink);
    run<3>("v_accvgpr_read_b32", d_out, d_sink);
    run<4>("v_fma_mixlo_f16", d_out, d_sink);
    run<5>("v_cvt_pk_f16_f32", d_out, d_sink);
    run<6>("ds_write_b64 (4 waves)", d_out, d_sink);
    run<7>("v_exp_f32", d_out, d_sink);
    run<8>("s_add_u32 dependent", d_out, d_sink);
    run<9>("s_bitcmp1 + s_cbranch (not taken) x256 pairs", d_out, d_sink);
    run<10>("v_pk_mul_f32", d_out, d_sink);
    run<11>("v_exp_f32 / v_mul_f32 alternating", d_out, d_sink);
    run<12>("1 MFMA + 7 v_mul per group (x32)", d_out, d_sink);
    run<13>("MFMA 32x32x16 alone (x256)", d_out, d_sink);
    run<14>("MFMA + fma, exp, add, (cvt) mix (x72 groups of ~4.5)", d_out, d_sink);
    return 0;
}
