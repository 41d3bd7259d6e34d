// Row sums of P on the matrix pipe.  P^T of one 16-key slab sits in the B-operand layout of
// v_mfma_f32_32x32x16_f16 (lane l: query l % 32, keys 8 (l / 32) .. + 7).  The same four registers fed to
// v_mfma_f32_16x16x32_f16 as B are read as (column l % 16, key group l / 16); with an A operand that is 1.0 in
// lanes {0, 32} (row 0, key groups 0 and 2) and {20, 52} (row 4, key groups 1 and 3) and 0 elsewhere, register 0
// of the 16 x 16 result holds, in lane l < 32, the slab's sum for query l, and 0 in lanes 32-63.
// This program checks that claim against sums computed lane by lane.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/rowsum_mfma tools/micro/rowsum_mfma.hip && tools/micro/rowsum_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const _Float16* p /* [32 queries][16 keys] */, float* out /* [64 lanes][4] */) {
    const int l = threadIdx.x;
    half8 b;
    for (int j = 0; j < 8; ++j) b[j] = p[(l % 32) * 16 + 8 * (l / 32) + j];
    const _Float16 one = (l == 0 || l == 32 || l == 20 || l == 52) ? (_Float16)1.0f : (_Float16)0.0f;
    half8 a;
    for (int j = 0; j < 8; ++j) a[j] = one;
    f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(d) : "v"(a), "v"(b));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = d[i];
}

int main() {
    std::vector<_Float16> p(32 * 16);
    srand(1);
    for (auto& x : p) x = (_Float16)((rand() % 2048) / 2048.0f);
    _Float16* dp;
    float* dout;
    hipMalloc(&dp, p.size() * 2);
    hipMalloc(&dout, 64 * 4 * 4);
    hipMemcpy(dp, p.data(), p.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dp, dout);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float want = 0.0f;
        if (l < 32)
            for (int key = 0; key < 16; ++key) want += (float)p[l * 16 + key];
        if (out[l * 4] != want) { ++bad; printf("lane %d: got %g want %g\n", l, out[l * 4], want); }
        for (int i = 1; i < 4; ++i)
            if (out[l * 4 + i] != 0.0f) { ++bad; printf("lane %d reg %d: %g (want 0)\n", l, i, out[l * 4 + i]); }
    }
    printf("rowsum_mfma: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad != 0;
}
