// Does data written by one workgroup reach another workgroup of the SAME XCD through that XCD's L2 without a
// device-scope release (which writes the whole L2 back)?  Workgroup w (producer, w < n/2 ... see below) writes a block
// with plain stores, waits for them (s_waitcnt vmcnt(0)), then raises a flag with an atomic; its partner w + 8 * k
// (same id mod 8 = same XCD, by the dispatcher's round-robin) spins on the flag with an atomic read and then reads
// the block with loads that miss the unit's L1 (sc0 sc1).  Printed: XCC_ID per workgroup id mod 8 (the assumption),
// mismatching words over all rounds for same-XCD partners and, for comparison, for partners on different XCDs.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/xcd_handoff tools/micro/xcd_handoff.hip && tools/micro/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kWords = 32768;  // 128 KiB per pair, about a key range's state

__global__ __launch_bounds__(256) void handoff(unsigned* data, unsigned* flags, unsigned* xcc, unsigned* bad, int pairs,
                                               int partner_shift, unsigned round) {
    const int w = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[w] = id & 0xf;
    }
    const bool producer = w < pairs;
    const int pair = producer ? w : w - pairs - partner_shift;
    if (pair < 0 || pair >= pairs) return;
    unsigned* blk = data + (size_t)pair * kWords;
    if (producer) {
        for (int i = threadIdx.x; i < kWords; i += 256) blk[i] = round * 0x9e3779b9u + (unsigned)i + (unsigned)pair * 7u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[pair], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_fetch_add(&flags[pair], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round && spins < (1L << 18)) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
            }
        }
        __syncthreads();
        unsigned wrong = 0;
        for (int i = threadIdx.x; i < kWords; i += 256) {
            unsigned v;
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(blk + i) : "memory");
            wrong += v != round * 0x9e3779b9u + (unsigned)i + (unsigned)pair * 7u;
        }
        if (wrong) atomicAdd(&bad[0], wrong);
    }
}

int main() {
    const int pairs = 128;  // producers 0..127, consumers 128..255 (+ shift): w and w + 128 share w mod 8
    unsigned *data, *flags, *xcc, *bad;
    hipMalloc(&data, (size_t)pairs * kWords * 4);
    hipMalloc(&flags, pairs * 4);
    hipMalloc(&xcc, 512 * 4);
    hipMalloc(&bad, 4);
    for (int shift : {0, 1, 4}) {  // 0: same XCD; 1, 4: the partner sits on another XCD
        hipMemset(flags, 0, pairs * 4);
        hipMemset(bad, 0, 4);
        hipMemset(data, 0, (size_t)pairs * kWords * 4);
        for (unsigned round = 1; round <= 200; ++round)
            hipLaunchKernelGGL(handoff, dim3(2 * pairs + shift), dim3(256), 0, 0, data, flags, xcc, bad, pairs, shift, round);
        hipDeviceSynchronize();
        unsigned h_bad = 0;
        hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> h_xcc(512);
        hipMemcpy(h_xcc.data(), xcc, 512 * 4, hipMemcpyDeviceToHost);
        int off = 0;
        for (int w = 0; w < 2 * pairs; ++w) off += h_xcc[w] != h_xcc[w % 8];
        printf("partner shift %d: %u stale words in 200 rounds x %d pairs x %d words; workgroups whose XCC_ID differs from id mod 8's: %d (XCC of ids 0-7:",
               shift, h_bad, pairs, kWords, off);
        for (int w = 0; w < 8; ++w) printf(" %u", h_xcc[w]);
        printf(")\n");
    }
    return 0;
}
