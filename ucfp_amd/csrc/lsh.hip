// lsh.hip -- banded MinHash LSH index over 1032-byte MinHash-128 records.
//
// The reference only re-tags a MinHash record as `minhash-lsh-h128` (src/modality/text.rs:437-446);
// no band index exists in it, and its docs disagree on (bands, rows).  This is the
// structure BASELINE configs[3] asks for next to the signatures: candidate lookup by band-key
// equality, verification by slot agreement.  Spec (ours, DESIGN.md "LSH"):
//   key_b   = fold of slots [b*rows, (b+1)*rows):  h = 0xcbf29ce484222325; h = (h ^ slot) * 0x100000001b3
//             per slot, then the splitmix64 finaliser
//   build   per band, (key, row) pairs sorted by key (stable radix sort: rows ascending inside a key)
//   query   per band the first `cand_per_band` rows whose key equals the query's; union over bands;
//           score = (#equal slots) / 128 (the MinHash Jaccard estimate); best k by (score desc, id asc)
// One wave per query: binary search is wave-uniform, candidates are gathered 64 at a time, the
// agreement count is two 64-bit compares per lane and a ballot popcount.  The sort is rocPRIM's
// device radix sort (a plain library primitive); everything else is hand-written.

#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM's texture iterator calls the host memset without including it

#include <rocprim/rocprim.hpp>

#include <mutex>
#include <new>

#include "../../include/ucfp_hip.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
int ctx_device(const ucfp_ctx* ctx);
}  // namespace ucfp
using ucfp::capi_fail;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return capi_fail(UCFP_E_INDEX, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

namespace {

constexpr int kMaxCand = 1024;  // unique candidates examined per query

// records: n x 1032 bytes (8-byte header + 128 u64, only 4-byte alignment guaranteed)
__device__ __forceinline__ uint64_t load_slot(const uint8_t* rec, uint32_t i) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(rec + 8 + 8 * (size_t)i);
    return (uint64_t)p[0] | ((uint64_t)p[1] << 32);
}

// keys: band-major [bands][n]; rows_out (optional): [bands][n] = row index; sigs (optional): [n][128]
__global__ void lsh_keys_kernel(const uint8_t* __restrict__ records, size_t n, uint32_t bands, uint32_t rows,
                                uint64_t* __restrict__ keys, uint32_t* __restrict__ rows_out,
                                uint64_t* __restrict__ sigs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * bands) return;
    const size_t doc = t / bands;
    const uint32_t b = (uint32_t)(t - doc * bands);
    const uint8_t* rec = records + doc * 1032;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t r = 0; r < rows; r++) {
        const uint64_t v = load_slot(rec, b * rows + r);
        if (sigs) sigs[doc * 128 + b * rows + r] = v;
        h = (h ^ v) * 0x100000001b3ull;
    }
    h ^= h >> 30;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 27;
    h *= 0x94D049BB133111EBull;
    h ^= h >> 31;
    keys[(size_t)b * n + doc] = h;
    if (rows_out) rows_out[(size_t)b * n + doc] = (uint32_t)doc;
    // slots not covered by any band (bands * rows < 128) are copied by band 0's thread
    if (sigs && b == 0)
        for (uint32_t i = bands * rows; i < 128; i++) sigs[doc * 128 + i] = load_slot(rec, i);
}

// one wave per query
__global__ __launch_bounds__(64) void lsh_query_kernel(const uint8_t* __restrict__ qrecords, uint32_t nq,
                                                       const uint64_t* __restrict__ skeys,
                                                       const uint32_t* __restrict__ srows, size_t n, uint32_t bands,
                                                       uint32_t rows, uint32_t cand_per_band,
                                                       const uint64_t* __restrict__ sigs,
                                                       const uint64_t* __restrict__ ids, uint32_t k,
                                                       uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                       uint32_t* __restrict__ out_counts) {
    __shared__ uint32_t cand[kMaxCand];
    __shared__ uint32_t top_agree[UCFP_INDEX_MAX_K];
    __shared__ uint64_t top_id[UCFP_INDEX_MAX_K];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* qrec = qrecords + (size_t)q * 1032;
    const uint64_t qa = load_slot(qrec, lane), qb = load_slot(qrec, lane + 64);
    uint32_t ncand = 0;  // wave-uniform
    for (uint32_t b = 0; b < bands; b++) {
        // band key of the query: lanes [b*rows, (b+1)*rows) hold the slots; fold sequentially
        uint64_t h = 0xcbf29ce484222325ull;
        for (uint32_t r = 0; r < rows; r++) {
            const uint32_t i = b * rows + r;
            const uint64_t v = i < 64 ? __shfl(qa, (int)i, 64) : __shfl(qb, (int)(i - 64), 64);
            h = (h ^ v) * 0x100000001b3ull;
        }
        h ^= h >> 30;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 27;
        h *= 0x94D049BB133111EBull;
        h ^= h >> 31;
        // lower bound of h in skeys[b] (wave-uniform binary search)
        const uint64_t* kb = skeys + (size_t)b * n;
        size_t lo = 0, hi = n;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (kb[mid] < h) lo = mid + 1;
            else hi = mid;
        }
        // gather the run of equal keys, 64 at a time, up to cand_per_band
        for (uint32_t taken = 0; taken < cand_per_band && ncand < (uint32_t)kMaxCand; taken += 64) {
            const size_t i = lo + taken + lane;
            const bool ok = i < n && taken + lane < cand_per_band && kb[i] == h;
            const uint64_t m = __ballot(ok);
            if (!m) break;
            const uint32_t pos = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (ok && pos < (uint32_t)kMaxCand) cand[pos] = srows[(size_t)b * n + i];
            ncand += (uint32_t)__popcll(m);
            if (ncand > (uint32_t)kMaxCand) ncand = kMaxCand;
            if (__popcll(m) < 64) break;   // the run ended inside this group
        }
    }
    ucfp::wave_lds_sync();
    // verify candidates (duplicates across bands are skipped), keep the best k
    uint32_t kept = 0;
    for (uint32_t c = 0; c < ncand; c++) {
        const uint32_t row = cand[c];
        // seen before in the list?
        bool dup = false;
        for (uint32_t j = lane; j < c; j += 64) dup |= cand[j] == row;
        if (__any(dup)) continue;
        const uint64_t* sg = sigs + (size_t)row * 128;
        const uint32_t agree = (uint32_t)__popcll(__ballot(sg[lane] == qa)) + (uint32_t)__popcll(__ballot(sg[lane + 64] == qb));
        const uint64_t id = ids[row];
        // insertion into the sorted (agree desc, id asc) list: k <= 128, so each lane owns entries
        // `lane` and `lane + 64`; the insert position is the number of entries the newcomer does NOT beat
        const uint32_t j1 = lane, j2 = lane + 64;
        uint32_t a1 = 0, a2 = 0;
        uint64_t i1 = 0, i2 = 0;
        if (j1 < kept) {
            a1 = top_agree[j1];
            i1 = top_id[j1];
        }
        if (j2 < kept) {
            a2 = top_agree[j2];
            i2 = top_id[j2];
        }
        const bool keep1 = j1 < kept && !(agree > a1 || (agree == a1 && id < i1));
        const bool keep2 = j2 < kept && !(agree > a2 || (agree == a2 && id < i2));
        const uint32_t pos = (uint32_t)__popcll(__ballot(keep1)) + (uint32_t)__popcll(__ballot(keep2));
        if (pos >= k) continue;
        ucfp::wave_lds_sync();   // every lane has read its entries before any lane overwrites one
        const uint32_t last = kept < k ? kept : k - 1;   // entries [pos, last) move one place down
        if (j1 >= pos && j1 < last) {
            top_agree[j1 + 1] = a1;
            top_id[j1 + 1] = i1;
        }
        if (j2 >= pos && j2 < last) {
            top_agree[j2 + 1] = a2;
            top_id[j2 + 1] = i2;
        }
        if (lane == 0) {
            top_agree[pos] = agree;
            top_id[pos] = id;
        }
        if (kept < k) kept++;
        ucfp::wave_lds_sync();
    }
    ucfp::wave_lds_sync();
    for (uint32_t j = lane; j < k; j += 64) {
        const bool v = j < kept;
        out_ids[(size_t)q * k + j] = v ? top_id[j] : ~0ull;
        out_scores[(size_t)q * k + j] = v ? (float)top_agree[j] * (1.0f / 128.0f) : -1.0f;
    }
    if (lane == 0) out_counts[q] = kept;
}

struct DevArr {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (cap >= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(&p, bytes + 256));
        cap = bytes + 256;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct ucfp_lsh {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    uint32_t bands = 0, rows = 0, cand_per_band = 64;
    size_t n = 0;
    std::mutex mu;
    DevArr keys_a, keys_b, rows_a, rows_b, sigs, ids, tmp;
    uint64_t* skeys = nullptr;   // sorted keys, band-major
    uint32_t* srows = nullptr;   // rows in sorted order
};

extern "C" {

int ucfp_text_lsh_band_keys_dev(ucfp_ctx* ctx, const uint8_t* d_records, size_t n, uint32_t bands, uint32_t rows,
                                uint64_t* d_keys, void* stream) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (bands == 0 || rows == 0 || rows > 64 || bands * rows > 128)
        return capi_fail(UCFP_E_INVALID, "need 1 <= rows <= 64 and bands * rows <= 128 (got %u x %u)", bands, rows);
    if (n == 0) return UCFP_OK;
    if (!d_records || !d_keys) return capi_fail(UCFP_E_INVALID, "records/keys is NULL");
    const size_t t = n * bands;
    hipLaunchKernelGGL(lsh_keys_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_records,
                       n, bands, rows, d_keys, (uint32_t*)nullptr, (uint64_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

int ucfp_lsh_create(ucfp_ctx* ctx, uint32_t bands, uint32_t rows, uint32_t cand_per_band, ucfp_lsh** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (bands == 0 || rows == 0 || rows > 64 || bands * rows > 128)
        return capi_fail(UCFP_E_INVALID, "need 1 <= rows <= 64 and bands * rows <= 128 (got %u x %u)", bands, rows);
    ucfp_lsh* l = new (std::nothrow) ucfp_lsh();
    if (!l) return capi_fail(UCFP_E_INDEX, "out of host memory");
    l->ctx = ctx;
    l->device = ucfp::ctx_device(ctx);
    l->bands = bands;
    l->rows = rows;
    l->cand_per_band = cand_per_band ? cand_per_band : 64;
    *out = l;
    return UCFP_OK;
}

void ucfp_lsh_destroy(ucfp_lsh* l) {
    if (!l) return;
    (void)hipSetDevice(l->device);
    (void)hipDeviceSynchronize();
    l->keys_a.release();
    l->keys_b.release();
    l->rows_a.release();
    l->rows_b.release();
    l->sigs.release();
    l->ids.release();
    l->tmp.release();
    delete l;
}

int ucfp_lsh_build_dev(ucfp_lsh* l, const uint64_t* d_ids, const uint8_t* d_records, size_t n, void* stream) {
    if (!l) return capi_fail(UCFP_E_INVALID, "lsh is NULL");
    if (n && (!d_ids || !d_records)) return capi_fail(UCFP_E_INVALID, "ids/records is NULL");
    if (n > 0xfffffff0u) return capi_fail(UCFP_E_INVALID, "at most 2^32 - 16 rows per LSH shard");
    std::lock_guard<std::mutex> lk(l->mu);
    HIP_TRY(hipSetDevice(l->device));
    hipStream_t st = (hipStream_t)stream;
    l->n = n;
    if (n == 0) return UCFP_OK;
    const size_t tot = n * l->bands;
    int rc;
    if ((rc = l->keys_a.ensure(tot * 8)) || (rc = l->keys_b.ensure(tot * 8)) || (rc = l->rows_a.ensure(tot * 4)) ||
        (rc = l->rows_b.ensure(tot * 4)) || (rc = l->sigs.ensure(n * 128 * 8)) || (rc = l->ids.ensure(n * 8)))
        return rc;
    HIP_TRY(hipMemcpyAsync(l->ids.p, d_ids, n * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lsh_keys_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_records, n, l->bands,
                       l->rows, (uint64_t*)l->keys_a.p, (uint32_t*)l->rows_a.p, (uint64_t*)l->sigs.p);
    HIP_TRY(hipGetLastError());
    // per band: stable radix sort of (key, row)
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)l->keys_a.p, (uint64_t*)l->keys_b.p,
                                      (uint32_t*)l->rows_a.p, (uint32_t*)l->rows_b.p, n, 0, 64, st));
    if ((rc = l->tmp.ensure(tmp_bytes))) return rc;
    for (uint32_t b = 0; b < l->bands; b++) {
        const size_t o = (size_t)b * n;
        HIP_TRY(rocprim::radix_sort_pairs(l->tmp.p, tmp_bytes, (uint64_t*)l->keys_a.p + o, (uint64_t*)l->keys_b.p + o,
                                          (uint32_t*)l->rows_a.p + o, (uint32_t*)l->rows_b.p + o, n, 0, 64, st));
    }
    l->skeys = (uint64_t*)l->keys_b.p;
    l->srows = (uint32_t*)l->rows_b.p;
    return UCFP_OK;
}

int ucfp_lsh_query_dev(ucfp_lsh* l, const uint8_t* d_query_records, size_t nq, uint32_t k, uint64_t* d_out_ids,
                       float* d_out_scores, uint32_t* d_out_counts, void* stream) {
    if (!l) return capi_fail(UCFP_E_INVALID, "lsh is NULL");
    if (k == 0 || k > UCFP_INDEX_MAX_K) return capi_fail(UCFP_E_INVALID, "k must be in [1, %u]", UCFP_INDEX_MAX_K);
    if (nq == 0) return UCFP_OK;
    if (!d_query_records || !d_out_ids || !d_out_scores || !d_out_counts)
        return capi_fail(UCFP_E_INVALID, "query/output buffer is NULL");
    if (nq > 0x7fffffffu) return capi_fail(UCFP_E_INVALID, "too many queries");
    std::lock_guard<std::mutex> lk(l->mu);
    HIP_TRY(hipSetDevice(l->device));
    hipStream_t st = (hipStream_t)stream;
    // an empty index runs the same kernel: every binary search ends at 0 and no candidate is gathered
    hipLaunchKernelGGL(lsh_query_kernel, dim3((unsigned)nq), dim3(64), 0, st, d_query_records, (uint32_t)nq, l->skeys,
                       l->srows, l->n, l->bands, l->rows, l->cand_per_band, (const uint64_t*)l->sigs.p,
                       (const uint64_t*)l->ids.p, k, d_out_ids, d_out_scores, d_out_counts);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

}  // extern "C"
