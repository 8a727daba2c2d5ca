/*
 * knn_oracle.c — CPU restatement of the cosine-kNN search of nhahn/cslam.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (cslam_amd/, libcslam_hip.so)
 * never does.
 *
 * Follows   cslam/nns_matching.py:44-63  (search):
 *     similarity = cosine_similarity(data[:n], q)          (eps = 1e-8, nns_matching.py:61)
 *     ns = argsort(similarity, descending=True)[:k]        (nns_matching.py:62)
 *     return items[ns], similarity[ns]                     (nns_matching.py:63)
 * and       cslam/nns_matching.py:65-78  (search_best = search(q, 1)).
 *
 * Numeric contract (what "bit-exact" means for the HIP path):
 *   the reference leaves the fp32 summation order to torch; this restatement pins ONE order
 *   so that indices and similarities are reproducible bit for bit:
 *     dot(x,q)  : k is cut into chunks of 512 floats; inside a chunk ONE fmaf chain, k visited
 *                 in 16-blocks, inside a block k = kb + 4*g + j for j=0..3 (outer), g=0..3
 *                 (inner) — the order in which v_mfma_f32_16x16x4_f32 consumes a lane's
 *                 float4 operand; chunk partials are added to the total in chunk order
 *                 (plain fp32 adds).  The chunking keeps |sim - fp64 cosine| < 4e-7 at
 *                 D = 4096 (a single 4096-long chain drifts to 2.4e-6, beyond the 1e-6 tie
 *                 tolerance of the reference's own test)
 *     ||x||^2   : 64 interleaved partial fmaf chains (lane l takes floats 256*t + 4*l + c),
 *                 combined by an xor-butterfly (32,16,8,4,2,1)           — a wave64 reduction
 *     rn(x)     : 1 / max(sqrt(||x||^2), 1e-8)     (IEEE sqrt and divide)
 *     sim       : (dot * rn(x)) * rn(q),  -0 canonicalised to +0
 *   ordering    : similarity descending, ties → lower row index first (torch.argsort is not
 *                 stable, so any tie order is a valid reference output).
 *   Against the imported reference (torch CPU) the indices agree wherever the reference's
 *   neighbouring similarities differ by more than 1e-6 — the tolerance the reference's own
 *   test grants (tests/test_sparse_matching.py:75-78); tests/test_oracle_knn.py checks that
 *   on the committed golden vectors.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline int64_t pad16(int64_t d) { return (d + 15) / 16 * 16; }

/* ||x||^2 in wave64 order over the zero-padded row of length dp (multiple of 16). */
static float sumsq_wave_order(const float* x, int64_t d, int64_t dp) {
    float acc[64];
    for (int l = 0; l < 64; ++l) acc[l] = 0.0f;
    for (int64_t t = 0; t * 256 < dp; ++t) {
        for (int l = 0; l < 64; ++l) {
            int64_t base = t * 256 + 4 * (int64_t)l;
            for (int c = 0; c < 4; ++c) {
                int64_t k = base + c;
                if (k < dp) {
                    float v = (k < d) ? x[k] : 0.0f;
                    acc[l] = fmaf(v, v, acc[l]);
                }
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = acc[l] + acc[l ^ off];
        memcpy(acc, nxt, sizeof(acc));
    }
    return acc[0];
}

static float inv_norm(const float* x, int64_t d) {
    float ss = sumsq_wave_order(x, d, pad16(d));
    float nrm = sqrtf(ss);
    if (!(nrm > 1e-8f)) nrm = 1e-8f; /* max(norm, eps); NaN → eps like fmaxf */
    return 1.0f / nrm;
}

/* dot(x,q): fmaf chains in MFMA-16x16x4 operand order over chunks of 512 floats (32 k-blocks),
 * chunk partials folded into the total by plain fp32 adds in chunk order. */
#define ORACLE_CHUNK_KB 32
static float dot_chain(const float* x, const float* q, int64_t d) {
    float total = 0.0f, part = 0.0f;
    int64_t dp = pad16(d);
    int64_t nkb = dp / 16;
    for (int64_t kb = 0; kb < nkb; ++kb) {
        for (int j = 0; j < 4; ++j)
            for (int g = 0; g < 4; ++g) {
                int64_t k = kb * 16 + 4 * g + j;
                float a = (k < d) ? x[k] : 0.0f;
                float b = (k < d) ? q[k] : 0.0f;
                part = fmaf(a, b, part);
            }
        if ((kb + 1) % ORACLE_CHUNK_KB == 0 || kb + 1 == nkb) {
            total = total + part;
            part = 0.0f;
        }
    }
    return total;
}

/* 1/max(||row||,1e-8) for n rows of length d (dense, stride d). */
void oracle_inv_norms(const float* rows, int64_t n, int64_t d, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = inv_norm(rows + i * d, d);
}

/* sims[q*n + i] for every query q and row i. */
void oracle_similarities(const float* rows, int64_t n, int64_t d, const float* queries,
                         int64_t m, float* sims) {
    float* rnx = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    oracle_inv_norms(rows, n, d, rnx);
    for (int64_t q = 0; q < m; ++q) {
        const float* qv = queries + q * d;
        float rnq = inv_norm(qv, d);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            float s = (dot_chain(rows + i * d, qv, d) * rnx[i]) * rnq;
            if (s == 0.0f) s = 0.0f; /* -0 → +0 */
            sims[q * n + i] = s;
        }
    }
    free(rnx);
}

typedef struct {
    float s;
    int32_t i;
} cand_t;

static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* search(): top-k per query.  out_idx/out_sim are [m][k]; unused slots are -1 / -inf.
 * out_count[q] = min(k, n). */
void oracle_search(const float* rows, int64_t n, int64_t d, const float* queries, int64_t m,
                   int64_t k, int32_t* out_idx, float* out_sim, int32_t* out_count) {
    int64_t kk = k < n ? k : n;
    for (int64_t q = 0; q < m; ++q) {
        for (int64_t j = 0; j < k; ++j) {
            out_idx[q * k + j] = -1;
            out_sim[q * k + j] = -INFINITY;
        }
        out_count[q] = (int32_t)kk;
    }
    if (n == 0 || k == 0) return;
    float* sims = (float*)malloc(sizeof(float) * (size_t)n);
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)n);
    for (int64_t q = 0; q < m; ++q) {
        oracle_similarities(rows, n, d, queries + q * d, 1, sims);
        for (int64_t i = 0; i < n; ++i) {
            c[i].s = sims[i];
            c[i].i = (int32_t)i;
        }
        qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);
        for (int64_t j = 0; j < kk; ++j) {
            out_idx[q * k + j] = c[j].i;
            out_sim[q * k + j] = c[j].s;
        }
    }
    free(c);
    free(sims);
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
