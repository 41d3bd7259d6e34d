// Throughput driver: counterpart of the reference's
// flash_attention_tensor_cores/benchmark_tensor_flash_attention.cu:43-230.
// Same six shapes (H=16, D=128, (B,L) = (32,512) ... (1,16384)), same input
// pattern and seeds, warm-up 3 / reps 5 with one event pair per rep, same FLOP
// formula and the same result line.  It calls the exported device-pointer entry
// point instead of launching a kernel template directly, allocates per shape
// (every shape touches 2^25 elements per tensor) and adds the fraction of the
// device's dense fp16 MFMA peak to each line.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

#include "../../include/flash_attention.h"

#define CHECK_HIP(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            std::fprintf(stderr, "HIP error %s at %s:%d -> %s\n", #call, __FILE__, __LINE__,         \
                         hipGetErrorString(e_));                                                     \
            std::exit(EXIT_FAILURE);                                                                 \
        }                                                                                            \
    } while (0)

#define CHECK_FA(call)                                                                               \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != FA_OK) {                                                                          \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fa_last_error());               \
            std::exit(EXIT_FAILURE);                                                                 \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    const int D = 128, tile = 64, warmup = 3, reps = 5, H = 16;
    // optional argument: milliseconds of preheat for the steady-state line of each shape (0 = reference figures only)
    const double steady_ms = argc > 1 ? std::atof(argv[1]) : 1000.0;
    const std::vector<int> seqs = {512, 1024, 2048, 4096, 8192, 16384};
    const std::vector<int> batch = {32, 16, 8, 4, 2, 1};

    int cus = 0, khz = 0;
    double peak = 0.0;
    CHECK_FA(fa_device_info(&cus, &khz, &peak));
    std::printf("Benchmark Tensor Flash Attention Kernel\n");
    std::printf("D=%d tile=%d warmup=%d reps=%d  device: %d CUs @ %.0f MHz, dense fp16 MFMA peak %.1f TFLOP/s\n", D,
                tile, warmup, reps, cus, khz / 1e3, peak / 1e12);

    size_t maxElems = 0;
    for (size_t i = 0; i < seqs.size(); ++i) {
        const size_t e = (size_t)batch[i] * H * seqs[i] * D;
        if (e > maxElems) maxElems = e;
    }
    uint16_t *dQ, *dK, *dV, *dO;
    CHECK_HIP(hipMalloc((void**)&dQ, maxElems * 2));
    CHECK_HIP(hipMalloc((void**)&dK, maxElems * 2));
    CHECK_HIP(hipMalloc((void**)&dV, maxElems * 2));
    CHECK_HIP(hipMalloc((void**)&dO, maxElems * 2));
    // the pattern depends only on the flat index, so this equals the prefix of the
    // reference's max-shape buffer
    CHECK_FA(fa_init_half(dQ, maxElems, 1234ULL, nullptr));
    CHECK_FA(fa_init_half(dK, maxElems, 2345ULL, nullptr));
    CHECK_FA(fa_init_half(dV, maxElems, 3456ULL, nullptr));
    CHECK_HIP(hipDeviceSynchronize());

    for (size_t i = 0; i < seqs.size(); ++i) {
        const int B = batch[i], L = seqs[i];
        if (L % tile != 0) {
            std::printf("Skipping B=%d H=%d L=%d (L not multiple of tile)\n", B, H, L);
            continue;
        }
        for (int w = 0; w < warmup; ++w) CHECK_FA(fa_forward(dQ, dK, dV, dO, B, H, L, D, tile, nullptr, nullptr));
        CHECK_HIP(hipDeviceSynchronize());

        hipEvent_t start, stop;
        CHECK_HIP(hipEventCreate(&start));
        CHECK_HIP(hipEventCreate(&stop));
        float total_ms = 0.0f;
        for (int r = 0; r < reps; ++r) {
            CHECK_HIP(hipEventRecord(start));
            CHECK_FA(fa_forward(dQ, dK, dV, dO, B, H, L, D, tile, nullptr, nullptr));
            CHECK_HIP(hipEventRecord(stop));
            CHECK_HIP(hipEventSynchronize(stop));
            float ms;
            CHECK_HIP(hipEventElapsedTime(&ms, start, stop));
            total_ms += ms;
        }
        CHECK_HIP(hipEventDestroy(start));
        CHECK_HIP(hipEventDestroy(stop));
        const float avg_ms = total_ms / reps;
        const double flops = 4.0 * (double)L * (double)L * (double)D * (double)H * (double)B;
        const double tflops = (flops / (avg_ms / 1000.0)) / 1e12;
        std::printf("B=%d H=%d L=%d: avg_time=%.3f ms, est_TFLOPs=%.3f (%.1f%% of dense fp16 peak)\n", B, H, L, avg_ms,
                    tflops, 100.0 * tflops * 1e12 / peak);

        // The figure above is the reference driver's: one launch per event pair on a chip that idles in
        // between, so it carries the launch latency and a clock that has not come up.  Steady state:
        // about a second of back-to-back launches, then `steady_reps` launches between one event pair.
        if (steady_ms > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < steady_ms) {
                for (int w = 0; w < 50; ++w) CHECK_FA(fa_forward(dQ, dK, dV, dO, B, H, L, D, tile, nullptr, nullptr));
                CHECK_HIP(hipDeviceSynchronize());
            }
            const int steady_reps = L <= 2048 ? 100 : 20;
            CHECK_HIP(hipEventCreate(&start));
            CHECK_HIP(hipEventCreate(&stop));
            CHECK_HIP(hipEventRecord(start));
            for (int r = 0; r < steady_reps; ++r) CHECK_FA(fa_forward(dQ, dK, dV, dO, B, H, L, D, tile, nullptr, nullptr));
            CHECK_HIP(hipEventRecord(stop));
            CHECK_HIP(hipEventSynchronize(stop));
            float ms;
            CHECK_HIP(hipEventElapsedTime(&ms, start, stop));
            CHECK_HIP(hipEventDestroy(start));
            CHECK_HIP(hipEventDestroy(stop));
            const double st = (flops / (ms / steady_reps / 1000.0)) / 1e12;
            std::printf("    steady state (%d back-to-back launches): %.3f ms, %.3f TFLOP/s (%.1f%%)\n", steady_reps,
                        ms / steady_reps, st, 100.0 * st * 1e12 / peak);
        }
    }
    CHECK_HIP(hipFree(dQ));
    CHECK_HIP(hipFree(dK));
    CHECK_HIP(hipFree(dV));
    CHECK_HIP(hipFree(dO));
    return 0;
}
