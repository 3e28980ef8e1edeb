/*
 * ucfp_oracle_index.c -- CPU restatement of the kNN hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * COSINE: PINNED.  Restates, line for line in meaning, the in-tree reference code
 *   dot_product  src/index/embedded/mod.rs:454-472  (8 independent f32 accumulators over
 *                chunks_exact(8), sequential sum of the 8, then the sequential remainder)
 *   l2_norm      :475-477   insert_topk :484-495   knn phase 2 :324-356
 * Scores are bit-identical to the reference's arithmetic.  TIES: the reference has no total
 * order -- insert_topk puts a later equal score IN FRONT while the buffer is filling
 * (partition_point(|s| s > score)), rejects it once full, and rayon's fold/reduce split decides
 * which rows meet in which buffer -- so equal scores come out in an unspecified order.  Two
 * entry points therefore:
 *   ucfp_oracle_cosine_knn           the specification the HIP path implements:
 *                                    (score desc, record_id asc), a total order;
 *   ucfp_oracle_cosine_knn_ref_fold  the reference's insert_topk run as ONE sequential fold in
 *                                    ascending record_id (redb range-scan order, :300-302),
 *                                    kept to show both agree whenever scores are distinct.
 * Pinned by the reference's own tests (src/index/embedded/mod.rs:522-589,
 * src/server/tests.rs:53-113), replayed in tests/test_reference_pins.py (oracle) and tests/test_index_gpu.py (HIP path).
 *
 * HAMMING: the reference has no Hamming search -> nothing to pin; this is the
 * specification itself: d = popcount(q ^ x), order (d asc, id asc).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float dot_product(const float* a, const float* b, size_t len) {
    float accs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t chunks = len / 8;
    for (size_t c = 0; c < chunks; c++)
        for (int j = 0; j < 8; j++) accs[j] += a[c * 8 + j] * b[c * 8 + j];
    float sum = 0.0f; /* accs.iter().sum(): f32 Sum folds from 0.0 */
    for (int j = 0; j < 8; j++) sum += accs[j];
    for (size_t i = chunks * 8; i < len; i++) sum += a[i] * b[i];
    return sum;
}

static float l2_norm(const float* v, size_t len) { return sqrtf(dot_product(v, v, len)); }

typedef struct {
    uint64_t id;
    float score;
} hit_t;

/* insert_topk: sorted-descending vec, partition_point(|s| s > score) */
static void insert_topk(hit_t* local, size_t* len, uint64_t rid, float score, size_t k) {
    if (*len < k) {
        size_t pos = 0;
        while (pos < *len && local[pos].score > score) pos++;
        memmove(local + pos + 1, local + pos, (*len - pos) * sizeof(hit_t));
        local[pos].id = rid;
        local[pos].score = score;
        (*len)++;
    } else if (*len > 0 && score > local[*len - 1].score) {
        size_t pos = 0;
        while (pos < *len && local[pos].score > score) pos++;
        memmove(local + pos + 1, local + pos, (*len - 1 - pos) * sizeof(hit_t));
        local[pos].id = rid;
        local[pos].score = score;
    }
}

typedef struct {
    uint64_t id;
    size_t row;
} idrow_t;
static int cmp_idrow(const void* a, const void* b) {
    uint64_t x = ((const idrow_t*)a)->id, y = ((const idrow_t*)b)->id;
    return (x > y) - (x < y);
}

/* returns number of hits (<= k). rows: n x dim. Faithful sequential fold (see header). */
size_t ucfp_oracle_cosine_knn_ref_fold(const uint64_t* ids, const float* rows, size_t n, size_t dim,
                              const float* query, size_t k, uint64_t* out_ids, float* out_scores) {
    if (dim == 0 || k == 0) return 0; /* :275-277 */
    float q_norm = l2_norm(query, dim);
    if (q_norm == 0.0f) return 0; /* :283-286 */
    idrow_t* order = (idrow_t*)malloc((n ? n : 1) * sizeof(idrow_t));
    for (size_t i = 0; i < n; i++) {
        order[i].id = ids[i];
        order[i].row = i;
    }
    qsort(order, n, sizeof(idrow_t), cmp_idrow);
    hit_t* local = (hit_t*)malloc((k + 1) * sizeof(hit_t));
    size_t len = 0;
    for (size_t i = 0; i < n; i++) {
        const float* v = rows + order[i].row * dim;
        float v_norm = l2_norm(v, dim);
        if (v_norm == 0.0f) continue; /* :328-330 */
        float score = dot_product(query, v, dim) / (q_norm * v_norm);
        if (score != score) continue; /* NaN: dropped (reference: unspecified) */
        insert_topk(local, &len, order[i].id, score, k);
    }
    for (size_t i = 0; i < len; i++) {
        out_ids[i] = local[i].id;
        out_scores[i] = local[i].score;
    }
    free(local);
    free(order);
    return len;
}

static int cmp_hit_total(const void* a, const void* b) {
    const hit_t* x = (const hit_t*)a;
    const hit_t* y = (const hit_t*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* The specification: reference arithmetic, total order (score desc, id asc). */
size_t ucfp_oracle_cosine_knn(const uint64_t* ids, const float* rows, size_t n, size_t dim,
                              const float* query, size_t k, uint64_t* out_ids, float* out_scores) {
    if (dim == 0 || k == 0) return 0;
    float q_norm = l2_norm(query, dim);
    if (q_norm == 0.0f) return 0;
    hit_t* all = (hit_t*)malloc((n ? n : 1) * sizeof(hit_t));
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const float* v = rows + i * dim;
        float v_norm = l2_norm(v, dim);
        if (v_norm == 0.0f) continue;
        float score = dot_product(query, v, dim) / (q_norm * v_norm);
        if (score != score) continue;
        all[m].id = ids[i];
        all[m].score = score;
        m++;
    }
    qsort(all, m, sizeof(hit_t), cmp_hit_total);
    size_t len = m < k ? m : k;
    for (size_t i = 0; i < len; i++) {
        out_ids[i] = all[i].id;
        out_scores[i] = all[i].score;
    }
    free(all);
    return len;
}

typedef struct {
    uint32_t d;
    uint64_t id;
} hd_t;
static int cmp_hd(const void* a, const void* b) {
    const hd_t* x = (const hd_t*)a;
    const hd_t* y = (const hd_t*)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* Hamming top-k for nq queries; outputs nq x k (padded with id = ~0, d = ~0). */
void ucfp_oracle_hamming_topk(const uint64_t* ids, const uint64_t* codes, size_t n,
                              const uint64_t* queries, size_t nq, size_t k, uint64_t* out_ids,
                              uint32_t* out_dist, uint32_t* out_counts) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t q = 0; q < nq; q++) {
        /* bounded selection: keep the k best in a small sorted buffer */
        hd_t* best = (hd_t*)malloc((k + 1) * sizeof(hd_t));
        size_t len = 0;
        for (size_t i = 0; i < n; i++) {
            hd_t c;
            c.d = (uint32_t)__builtin_popcountll(queries[q] ^ codes[i]);
            c.id = ids[i];
            if (len == k && cmp_hd(&c, &best[len - 1]) >= 0) continue;
            size_t pos = len;
            while (pos > 0 && cmp_hd(&c, &best[pos - 1]) < 0) pos--;
            if (len < k) len++;
            memmove(best + pos + 1, best + pos, (len - 1 - pos) * sizeof(hd_t));
            best[pos] = c;
        }
        for (size_t i = 0; i < k; i++) {
            out_ids[q * k + i] = i < len ? best[i].id : ~0ull;
            out_dist[q * k + i] = i < len ? best[i].d : 0xffffffffu;
        }
        out_counts[q] = (uint32_t)len;
        free(best);
    }
}


/* ---- timed CPU baselines (bench.py cpu_baseline legs) ---------------------------------------------------
 * The reference's knn phase 2 is `par_iter().fold(local top-k via insert_topk).reduce(merge)` over the candidate
 * vectors (src/index/embedded/mod.rs:324-340); rayon becomes OpenMP here: row chunks are folded in parallel with the
 * reference's own dot_product / l2_norm / insert_topk, the per-chunk lists of a query are merged in the total order
 * (score desc, id asc).  Scores are bit-identical to ucfp_oracle_cosine_knn. */
void ucfp_oracle_cosine_knn_batch_omp(const uint64_t* ids, const float* rows, size_t n, size_t dim, const float* queries,
                                      size_t nq, size_t k, uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
    for (size_t q = 0; q < nq; q++) out_counts[q] = 0;
    if (dim == 0 || k == 0 || nq == 0) return;
    const size_t chunk = nq >= 64 ? 4096 : 512;     /* enough (chunk, query) tiles for every core at nq = 1 too */
    const size_t nchunks = (n + chunk - 1) / chunk;
    float* vnorm = (float*)malloc((n ? n : 1) * sizeof(float));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) vnorm[i] = l2_norm(rows + i * dim, dim);
    hit_t* part = (hit_t*)malloc((nchunks ? nchunks : 1) * nq * (k + 1) * sizeof(hit_t));
    size_t* plen = (size_t*)calloc((nchunks ? nchunks : 1) * nq, sizeof(size_t));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (size_t c = 0; c < nchunks; c++)
        for (size_t q = 0; q < nq; q++) {
            const float* qv = queries + q * dim;
            const float q_norm = l2_norm(qv, dim);
            hit_t* local = part + (c * nq + q) * (k + 1);
            size_t len = 0;
            if (q_norm != 0.0f) {
                const size_t i1 = (c + 1) * chunk < n ? (c + 1) * chunk : n;
                for (size_t i = c * chunk; i < i1; i++) {
                    if (vnorm[i] == 0.0f) continue;
                    const float score = dot_product(qv, rows + i * dim, dim) / (q_norm * vnorm[i]);
                    if (score != score) continue;
                    insert_topk(local, &len, ids[i], score, k);
                }
            }
            plen[c * nq + q] = len;
        }
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < nq; q++) {
        size_t m = 0;
        for (size_t c = 0; c < nchunks; c++) m += plen[c * nq + q];
        hit_t* all = (hit_t*)malloc((m ? m : 1) * sizeof(hit_t));
        size_t w = 0;
        for (size_t c = 0; c < nchunks; c++)
            for (size_t e = 0; e < plen[c * nq + q]; e++) all[w++] = part[(c * nq + q) * (k + 1) + e];
        qsort(all, m, sizeof(hit_t), cmp_hit_total);
        const size_t len = m < k ? m : k;
        for (size_t e = 0; e < k; e++) {
            out_ids[q * k + e] = e < len ? all[e].id : ~0ull;
            out_scores[q * k + e] = e < len ? all[e].score : 0.0f;
        }
        out_counts[q] = (uint32_t)len;
        free(all);
    }
    free(part);
    free(plen);
    free(vnorm);
}

/* One (corpus slice, query) tile of the timed scan: the k best of codes[i0, i1) by (d, id), sorted, into best[]; returns
 * their number.  Scalar form. */
static size_t hamming_tile_scalar(const uint64_t* ids, const uint64_t* codes, size_t i0, size_t i1, uint64_t qv, hd_t* best,
                                  size_t k) {
    size_t len = 0;
    for (size_t i = i0; i < i1; i++) {
        hd_t c;
        c.d = (uint32_t)__builtin_popcountll(qv ^ codes[i]);
        if (len == k && c.d > best[len - 1].d) continue;      /* the common case: one compare */
        c.id = ids[i];
        if (len == k && cmp_hd(&c, &best[len - 1]) >= 0) continue;
        size_t pos = len;
        while (pos > 0 && cmp_hd(&c, &best[pos - 1]) < 0) pos--;
        if (len < k) len++;
        memmove(best + pos + 1, best + pos, (len - 1 - pos) * sizeof(hd_t));
        best[pos] = c;
    }
    return len;
}

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define UCFP_HAVE_AVX512_TILE 1
/* The same tile with AVX-512 VPOPCNTDQ (a scalar loop with a data-dependent branch understates the CPU by an
 * order of magnitude): 32 codes per trip -- four 512-bit xor + vpopcntq + compare against the current k-th distance --
 * and the scalar insert only for the (rare) codes that pass.  Chosen at run time (__builtin_cpu_supports), so the
 * portable build the tests use runs it too wherever the CPU has it; same answer as the scalar form by construction. */
__attribute__((target("avx512f,avx512vpopcntdq"))) static size_t hamming_tile_avx512(const uint64_t* ids,
                                                                                      const uint64_t* codes, size_t i0,
                                                                                      size_t i1, uint64_t qv, hd_t* best,
                                                                                      size_t k) {
    size_t len = 0, i = i0;
    const __m512i q = _mm512_set1_epi64((long long)qv);
    __m512i thr = _mm512_set1_epi64(64);          /* accept everything while the list is filling */
    for (; i + 32 <= i1; i += 32) {
        __m512i d[4];
        __mmask8 m[4];
        for (int u = 0; u < 4; u++) {
            d[u] = _mm512_popcnt_epi64(_mm512_xor_si512(_mm512_loadu_si512((const void*)(codes + i + 8 * u)), q));
            m[u] = _mm512_cmple_epu64_mask(d[u], thr);
        }
        if (!(m[0] | m[1] | m[2] | m[3])) continue;
        for (int u = 0; u < 4; u++) {
            if (!m[u]) continue;
            uint64_t dv[8];
            _mm512_storeu_si512((void*)dv, d[u]);
            for (unsigned mm = m[u]; mm; mm &= mm - 1) {
                const int b = __builtin_ctz(mm);
                hd_t c;
                c.d = (uint32_t)dv[b];
                if (len == k && c.d > best[len - 1].d) continue;
                c.id = ids[i + 8 * (size_t)u + (size_t)b];
                if (len == k && cmp_hd(&c, &best[len - 1]) >= 0) continue;
                size_t pos = len;
                while (pos > 0 && cmp_hd(&c, &best[pos - 1]) < 0) pos--;
                if (len < k) len++;
                memmove(best + pos + 1, best + pos, (len - 1 - pos) * sizeof(hd_t));
                best[pos] = c;
            }
        }
        if (len == k) thr = _mm512_set1_epi64((long long)best[k - 1].d);
    }
    /* tail (< 32 codes): merge the scalar scan of it into the list */
    for (; i < i1; i++) {
        hd_t c;
        c.d = (uint32_t)__builtin_popcountll(qv ^ codes[i]);
        if (len == k && c.d > best[len - 1].d) continue;
        c.id = ids[i];
        if (len == k && cmp_hd(&c, &best[len - 1]) >= 0) continue;
        size_t pos = len;
        while (pos > 0 && cmp_hd(&c, &best[pos - 1]) < 0) pos--;
        if (len < k) len++;
        memmove(best + pos + 1, best + pos, (len - 1 - pos) * sizeof(hd_t));
        best[pos] = c;
    }
    return len;
}
#endif

/* 1 when the timed Hamming scan runs its AVX-512 VPOPCNTDQ tile on this CPU (reported next to the baseline) */
int ucfp_oracle_hamming_simd(void) {
#ifdef UCFP_HAVE_AVX512_TILE
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vpopcntdq");
#else
    return 0;
#endif
}

/* Hamming top-k parallel over corpus slices as well (a single query must still use every core): slices x queries
 * tiles keep k-best lists, merged per query in (d asc, id asc).  Same answer as ucfp_oracle_hamming_topk.
 * force_scalar != 0 keeps the scalar tile (the tests compare the two). */
void ucfp_oracle_hamming_topk_omp2(const uint64_t* ids, const uint64_t* codes, size_t n, const uint64_t* queries, size_t nq,
                                   size_t k, uint64_t* out_ids, uint32_t* out_dist, uint32_t* out_counts, int force_scalar) {
    if (k == 0 || nq == 0) return;
    const int simd = !force_scalar && ucfp_oracle_hamming_simd();
    const size_t slice = (size_t)1 << 16;       /* 512 KiB of codes: stays in a core's L2 while the queries walk over it */
    const size_t ns = n ? (n + slice - 1) / slice : 1;
    hd_t* part = (hd_t*)malloc(ns * nq * (k + 1) * sizeof(hd_t));
    uint32_t* plen = (uint32_t*)calloc(ns * nq, sizeof(uint32_t));
    /* a thread takes a slice and walks a run of queries over it (the slice is read from memory once per run) */
    const size_t qrun = nq < 64 ? 1 : 64;
    const size_t nruns = (nq + qrun - 1) / qrun;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (size_t sl = 0; sl < ns; sl++)
        for (size_t r = 0; r < nruns; r++) {
            const size_t i0 = sl * slice, i1 = (sl + 1) * slice < n ? (sl + 1) * slice : n;
            const size_t q1 = (r + 1) * qrun < nq ? (r + 1) * qrun : nq;
            for (size_t q = r * qrun; q < q1; q++) {
                hd_t* best = part + (sl * nq + q) * (k + 1);
                size_t len;
#ifdef UCFP_HAVE_AVX512_TILE
                if (simd) len = hamming_tile_avx512(ids, codes, i0, i1, queries[q], best, k);
                else
#endif
                    len = hamming_tile_scalar(ids, codes, i0, i1, queries[q], best, k);
                plen[sl * nq + q] = (uint32_t)len;
            }
        }
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < nq; q++) {
        size_t m = 0;
        for (size_t sl = 0; sl < ns; sl++) m += plen[sl * nq + q];
        hd_t* all = (hd_t*)malloc((m ? m : 1) * sizeof(hd_t));
        size_t w = 0;
        for (size_t sl = 0; sl < ns; sl++)
            for (size_t e = 0; e < plen[sl * nq + q]; e++) all[w++] = part[(sl * nq + q) * (k + 1) + e];
        qsort(all, m, sizeof(hd_t), cmp_hd);
        const size_t len = m < k ? m : k;
        for (size_t e = 0; e < k; e++) {
            out_ids[q * k + e] = e < len ? all[e].id : ~0ull;
            out_dist[q * k + e] = e < len ? all[e].d : 0xffffffffu;
        }
        out_counts[q] = (uint32_t)len;
        free(all);
    }
    free(part);
    free(plen);
}

void ucfp_oracle_hamming_topk_omp(const uint64_t* ids, const uint64_t* codes, size_t n, const uint64_t* queries, size_t nq,
                                  size_t k, uint64_t* out_ids, uint32_t* out_dist, uint32_t* out_counts) {
    ucfp_oracle_hamming_topk_omp2(ids, codes, n, queries, nq, k, out_ids, out_dist, out_counts, 0);
}

