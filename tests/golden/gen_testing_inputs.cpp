// Regenerates the input recipe of the reference's correctness driver
// (flash_attention_tensor_cores/testing.cu:158-165): std::mt19937 seeded with
// 42, std::normal_distribution<float>(0,1) plus a 0.5 bias, drawn per element
// in the order Q[i], K[i], V[i].  Writes the fp32 values (Q then K then V) to
// stdout as raw little-endian floats; the caller rounds them to fp16.
// The sequence depends on libstdc++'s normal_distribution, which is why the
// rounded result is committed as a fixture instead of being regenerated.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 64 * 128;
    std::vector<float> q(n), k(n), v(n);
    std::mt19937 gen(42);
    std::normal_distribution<float> d(0.0f, 1.0f);
    const float bias = 0.5f;
    for (size_t i = 0; i < n; ++i) {
        q[i] = d(gen) + bias;
        k[i] = d(gen) + bias;
        v[i] = d(gen) + bias;
    }
    std::fwrite(q.data(), sizeof(float), n, stdout);
    std::fwrite(k.data(), sizeof(float), n, stdout);
    std::fwrite(v.data(), sizeof(float), n, stdout);
    return 0;
}
