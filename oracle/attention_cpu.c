/*
 * CPU oracle for the forward-attention hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference's CPU baseline
 * `attention_cpu` (flash_attention_tensor_cores/testing.cu:41-132) plus the
 * small helpers the reference's drivers use around it.  Only tests, the smoke
 * check and the `cpu_baseline` leg of the benchmark may call into it, and only
 * as the checker or the timed CPU baseline; the GPU library never links it.
 *
 * Pinning: the reference's attention_cpu lives in a .cu file that includes
 * CUDA headers, so it cannot be compiled in this environment.  This restatement
 * is pinned by (1) golden vectors produced by running the reference notebook's
 * `flash_attention_numpy` (benchmarks/proof_correctness.ipynb, cell 2100755c)
 * on the reference's two input recipes, (2) a dense fp64 softmax, and (3)
 * bit-equality between the two loop orders implemented below.
 *
 * Arithmetic recipe (all citations are testing.cu):
 *   scale = 1/sqrtf(D)                                      :52
 *   keys are visited in tiles of 64                          :53,65
 *   dot   = sum_d float(q)*float(k), sequential in d, fp32   :78-84
 *   score = dot*scale ; local_max over the tile              :85-86
 *   old_max = max ; max = max(max, local_max)                :91-92
 *   acc[d] *= expf(old_max-max) ; sum *= expf(old_max-max)   :94-96
 *   ex = expf(score-max) ; P = half(ex) ; tile_sum += ex     :103-109
 *   sum += tile_sum                                          :110
 *   acc[d] += sum_j float(P_j)*float(V_jd), sequential in j  :114-122
 *   O = acc/sum  (fp32, NOT rounded to fp16)                 :126-128
 *   max starts at -1e30f, sum at 0                           :60-61
 *   causal: key j > query i gets score -1e30f and does not
 *           take part in local_max                           :74-77
 *
 * Build: gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off keeps every multiply and add separately rounded, as the
 * reference's x86-64 host build does.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_TILE 64

/* ---- IEEE binary16 <-> binary32, portable, round-to-nearest-even ---------- */

float oracle_half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t oracle_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        uint32_t man = ax & 0x7FFFFFu;
        return (uint16_t)(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0u));
    }
    if (ax >= 0x477FF000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x33000001u) { /* <= 2^-25 rounds to zero (tie goes to even = 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;
    uint32_t base;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        base = 0;
    } else {
        shift = 13;
        base = ((uint32_t)(e + 15) << 10) - 0x400u; /* implicit bit adds 0x400 back */
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | (base + q));
}

void oracle_halfs_to_floats(const uint16_t* src, float* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = oracle_half_to_float(src[i]);
}

void oracle_floats_to_halfs(const float* src, uint16_t* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = oracle_float_to_half(src[i]);
}

/* ---- literal loop order of testing.cu:55-131 (slow; used to pin the fast one) */

void oracle_attention_cpu_literal(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O,
                                  int B, int H, int L, int D, int causal) {
    const float scale = 1.0f / sqrtf((float)D);
    float* acc = (float*)malloc(sizeof(float) * (size_t)D);
    float tile_scores[ORACLE_TILE];
    uint16_t p_tile[ORACLE_TILE];
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < L; ++i) {
                const size_t base = ((size_t)b * H + h) * (size_t)L * D;
                float max_val = -1e30f, sum_val = 0.0f;
                for (int d = 0; d < D; ++d) acc[d] = 0.0f;
                for (int t = 0; t < L; t += ORACLE_TILE) {
                    const int t_end = (t + ORACLE_TILE < L) ? t + ORACLE_TILE : L;
                    float local_max = -1e30f;
                    for (int j = t; j < t_end; ++j) {
                        if (causal && j > i) {
                            tile_scores[j - t] = -1e30f;
                            continue;
                        }
                        float dot = 0.0f;
                        for (int d = 0; d < D; ++d) {
                            float qv = oracle_half_to_float(Q[base + (size_t)i * D + d]);
                            float kv = oracle_half_to_float(K[base + (size_t)j * D + d]);
                            dot += qv * kv;
                        }
                        tile_scores[j - t] = dot * scale;
                        if (tile_scores[j - t] > local_max) local_max = tile_scores[j - t];
                    }
                    const float old_max = max_val;
                    if (local_max > max_val) max_val = local_max;
                    const float rescale = expf(old_max - max_val);
                    for (int d = 0; d < D; ++d) acc[d] *= rescale;
                    sum_val *= rescale;
                    float tile_sum = 0.0f;
                    for (int j = t; j < t_end; ++j) {
                        float ex = expf(tile_scores[j - t] - max_val);
                        p_tile[j - t] = oracle_float_to_half(ex);
                        tile_sum += ex;
                    }
                    sum_val += tile_sum;
                    for (int d = 0; d < D; ++d) {
                        float dot_pv = 0.0f;
                        for (int j = t; j < t_end; ++j)
                            dot_pv += oracle_half_to_float(p_tile[j - t]) *
                                      oracle_half_to_float(V[base + (size_t)j * D + d]);
                        acc[d] += dot_pv;
                    }
                }
                for (int d = 0; d < D; ++d) O[base + (size_t)i * D + d] = acc[d] / sum_val;
            }
    free(acc);
}

/* ---- fast form: identical arithmetic per output element, loops reordered so
 *      that the compiler can vectorise ACROSS keys (scores) and ACROSS d (PV).
 *      Every individual sum keeps the reference's order (d ascending for the
 *      dot product, j ascending for P.V), so results are bit-identical. -------- */

typedef struct {
    float max_val;
    float sum_val;
} row_state;

/* One query row against keys [k0, k0+nk) of one head.  kT is the head's keys
 * transposed to [D][Lk] floats, vf the values as [Lk][D] floats.  `qpos` is the
 * query's absolute position and `kpos0` the absolute position of key 0 of this
 * block (both only matter for the causal mask).  nk must be a multiple of 64
 * unless it is the sequence's tail. */
__attribute__((target_clones("avx2", "default")))
static void row_update(const float* q, const float* kT, const float* vf, int Lk, int D, int nk,
                       int causal, long qpos, long kpos0, float scale, float* acc, row_state* st) {
    float scores[ORACLE_TILE];
    float pq[ORACLE_TILE];
    float dot_pv[512];
    for (int t = 0; t < nk; t += ORACLE_TILE) {
        const int n = (t + ORACLE_TILE < nk) ? ORACLE_TILE : nk - t;
        for (int j = 0; j < n; ++j) scores[j] = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float qd = q[d];
            const float* kr = kT + (size_t)d * Lk + t;
            for (int j = 0; j < n; ++j) scores[j] += qd * kr[j];
        }
        float local_max = -1e30f;
        for (int j = 0; j < n; ++j) {
            if (causal && (kpos0 + t + j) > qpos) {
                scores[j] = -1e30f;
            } else {
                scores[j] = scores[j] * scale;
                if (scores[j] > local_max) local_max = scores[j];
            }
        }
        const float old_max = st->max_val;
        if (local_max > st->max_val) st->max_val = local_max;
        const float rescale = expf(old_max - st->max_val);
        for (int d = 0; d < D; ++d) acc[d] *= rescale;
        st->sum_val *= rescale;
        float tile_sum = 0.0f;
        for (int j = 0; j < n; ++j) {
            float ex = expf(scores[j] - st->max_val);
            pq[j] = oracle_half_to_float(oracle_float_to_half(ex));
            tile_sum += ex;
        }
        st->sum_val += tile_sum;
        for (int d = 0; d < D; ++d) dot_pv[d] = 0.0f;
        for (int j = 0; j < n; ++j) {
            const float pj = pq[j];
            const float* vr = vf + (size_t)(t + j) * D;
            for (int d = 0; d < D; ++d) dot_pv[d] += pj * vr[d];
        }
        for (int d = 0; d < D; ++d) acc[d] += dot_pv[d];
    }
}

static void head_to_float(const uint16_t* K, const uint16_t* V, int Lk, int D, float* kT, float* vf) {
    for (int j = 0; j < Lk; ++j)
        for (int d = 0; d < D; ++d) {
            kT[(size_t)d * Lk + j] = oracle_half_to_float(K[(size_t)j * D + d]);
            vf[(size_t)j * D + d] = oracle_half_to_float(V[(size_t)j * D + d]);
        }
}

/* Full attention, optionally restricted to a subset of query rows.
 * rows == NULL  : every row; O is [B][H][L][D].
 * rows != NULL  : the nrows listed rows of every (b,h); O is [B][H][nrows][D].
 * nthreads <= 0 : all OpenMP threads.  Returns 0, or -1 on a bad argument. */
int oracle_attention_cpu_rows(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O, int B,
                              int H, int L, int D, int causal, const int* rows, int nrows,
                              int nthreads) {
    if (B <= 0 || H <= 0 || L <= 0 || D <= 0 || D > 512) return -1;
    const float scale = 1.0f / sqrtf((float)D);
    const int nr = rows ? nrows : L;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    for (int bh = 0; bh < B * H; ++bh) {
        const size_t base = (size_t)bh * L * D;
        float* kT = (float*)malloc(sizeof(float) * (size_t)L * D);
        float* vf = (float*)malloc(sizeof(float) * (size_t)L * D);
        if (!kT || !vf) {
            free(kT);
            free(vf);
            return -1;
        }
        head_to_float(K + base, V + base, L, D, kT, vf);
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
        for (int r = 0; r < nr; ++r) {
            const int i = rows ? rows[r] : r;
            float q[512], acc[512];
            for (int d = 0; d < D; ++d) {
                q[d] = oracle_half_to_float(Q[base + (size_t)i * D + d]);
                acc[d] = 0.0f;
            }
            row_state st = {-1e30f, 0.0f};
            row_update(q, kT, vf, L, D, L, causal, i, 0, scale, acc, &st);
            float* o = O + ((size_t)bh * nr + r) * D;
            for (int d = 0; d < D; ++d) o[d] = acc[d] / st.sum_val;
        }
        free(kT);
        free(vf);
    }
    return 0;
}

/* Same signature as the reference's attention_cpu (testing.cu:41-48), plus a
 * thread count.  nthreads = 1 is the faithful single-threaded figure. */
int oracle_attention_cpu(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O, int B, int H,
                         int L, int D, int causal, int nthreads) {
    return oracle_attention_cpu_rows(Q, K, V, O, B, H, L, D, causal, NULL, 0, nthreads);
}

/* ---- blockwise form: one K/V block folded into carried (acc, max, sum) ------
 * The carried state is exactly the reference's per-row state (testing.cu:60-62)
 * between two key tiles, so folding the blocks of a sequence in order gives the
 * same bits as one full pass.  Q is [B][H][Lq][D], K/V are [B][H][Lk][D],
 * acc is [B][H][Lq][D] fp32, mx and sm are [B][H][Lq] fp32.  first != 0 resets
 * the state before folding.  q_pos0 / k_pos0 are the absolute positions of the
 * first query / key row (causal mask only). */
int oracle_attention_block_update(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* acc,
                                  float* mx, float* sm, int B, int H, int Lq, int Lk, int D, int causal,
                                  long q_pos0, long k_pos0, int first, int nthreads) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || D <= 0 || D > 512) return -1;
    const float scale = 1.0f / sqrtf((float)D);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    for (int bh = 0; bh < B * H; ++bh) {
        const size_t qbase = (size_t)bh * Lq * D, kbase = (size_t)bh * Lk * D;
        float* kT = (float*)malloc(sizeof(float) * (size_t)Lk * D);
        float* vf = (float*)malloc(sizeof(float) * (size_t)Lk * D);
        if (!kT || !vf) {
            free(kT);
            free(vf);
            return -1;
        }
        head_to_float(K + kbase, V + kbase, Lk, D, kT, vf);
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
        for (int i = 0; i < Lq; ++i) {
            float q[512];
            float* a = acc + qbase + (size_t)i * D;
            row_state st;
            for (int d = 0; d < D; ++d) q[d] = oracle_half_to_float(Q[qbase + (size_t)i * D + d]);
            if (first) {
                for (int d = 0; d < D; ++d) a[d] = 0.0f;
                st.max_val = -1e30f;
                st.sum_val = 0.0f;
            } else {
                st.max_val = mx[(size_t)bh * Lq + i];
                st.sum_val = sm[(size_t)bh * Lq + i];
            }
            row_update(q, kT, vf, Lk, D, Lk, causal, q_pos0 + i, k_pos0, scale, a, &st);
            mx[(size_t)bh * Lq + i] = st.max_val;
            sm[(size_t)bh * Lq + i] = st.sum_val;
        }
        free(kT);
        free(vf);
    }
    return 0;
}

/* O = acc / sum (testing.cu:126-128). */
void oracle_attention_block_finalize(const float* acc, const float* sm, float* O, size_t nrows, int D) {
    for (size_t r = 0; r < nrows; ++r)
        for (int d = 0; d < D; ++d) O[r * D + d] = acc[r * D + d] / sm[r];
}

/* ---- the benchmark driver's input pattern
 *      (benchmark_tensor_flash_attention.cu:34-41): element idx of a buffer
 *      seeded with `seed` is half(fmodf((idx*16807+seed)&0xFFFF, 1024)/1024-0.5).
 *      idx0 lets a caller generate a window of a larger buffer. -------------- */
void oracle_init_half(uint16_t* dst, size_t idx0, size_t n, unsigned long long seed) {
    for (size_t k = 0; k < n; ++k) {
        unsigned long long idx = (unsigned long long)(idx0 + k);
        float v = fmodf((float)((idx * 16807ULL + seed) & 0xFFFFULL), 1024.f) / 1024.f - 0.5f;
        dst[k] = oracle_float_to_half(v);
    }
}

/* ---- error metrics of testing.cu:214-236.  out[0..4] = mean abs, max abs,
 *      mean rel, max rel (denominator |ref|+1e-6), global relative L2. -------- */
void oracle_error_metrics(const float* ref, const float* test, size_t n, double* out) {
    float max_abs = 0.f, mean_abs = 0.f, max_rel = 0.f, mean_rel = 0.f;
    double ssd = 0.0, ssr = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float diff = fabsf(ref[i] - test[i]);
        if (diff > max_abs) max_abs = diff;
        mean_abs += diff;
        float rel = diff / (fabsf(ref[i]) + 1e-6f);
        if (rel > max_rel) max_rel = rel;
        mean_rel += rel;
        ssd += (double)diff * diff;
        ssr += (double)ref[i] * ref[i];
    }
    out[0] = n ? mean_abs / (float)n : 0.0;
    out[1] = max_abs;
    out[2] = n ? mean_rel / (float)n : 0.0;
    out[3] = max_rel;
    out[4] = (float)(sqrt(ssd) / (sqrt(ssr) + 1e-6f));
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
