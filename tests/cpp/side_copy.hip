// Test helper (not part of the product): a kernel shaped like a collective library's transport kernel, for
// checking that another stream's kernel can become resident while a persistent attention kernel runs.  A few
// workgroups of 256-512 threads, each holding a block of LDS, copy a buffer device-to-device in a grid-stride
// loop of 16-byte pieces; `repeat` passes make it last as long as a transfer over a slower link would.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void side_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int repeat,
                                 int lds_words) {
    extern __shared__ unsigned lds[];
    // touch the LDS block so that it is really part of the kernel's footprint
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = (unsigned)i;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned keep = lds[threadIdx.x % (lds_words > 0 ? lds_words : 1)];
    for (int r = 0; r < repeat; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            u32x4 v = __builtin_nontemporal_load(src + i);
            __builtin_nontemporal_store(v, dst + i);
        }
    if (keep == 0xffffffffu) dst[0].x = keep;  // (never true: keeps the LDS reads alive)
}

extern "C" int side_copy_launch(const void* src, void* dst, size_t bytes, int blocks, int threads, int lds_bytes,
                                int repeat, void* stream) {
    if (!src || !dst || bytes % 16 != 0 || blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 ||
        lds_bytes > 65536 || repeat <= 0)
        return 1;
    hipLaunchKernelGGL(side_copy_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes,
                       static_cast<hipStream_t>(stream), static_cast<const u32x4*>(src), static_cast<u32x4*>(dst),
                       bytes / 16, repeat, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
