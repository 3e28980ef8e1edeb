// common.h -- internal declarations shared by the HIP translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ucfp {

// Intra-wave LDS hand-off: the DS unit executes one wave's LDS instructions in issue order, so a
// later read sees an earlier write of another lane; what is needed is that the COMPILER keeps the
// order (and, defensively, that outstanding LDS ops have landed).  Unlike a workgroup-scope fence
// this does not wait for global loads/stores in flight (vmcnt), so prefetches stay overlapped.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Same-wave LDS hand-off without the drain: the LDS executes one wave's instructions in order, so a ds_read issued
// after a ds_write of the same wave sees the data; only the compiler has to keep the order.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// image.hip
int launch_image_hash(uint32_t algo, const uint8_t* frames, size_t n, uint32_t w, uint32_t h,
                      size_t row_stride, size_t frame_stride, int pixfmt, uint32_t min_dim,
                      uint32_t max_dim, const uint8_t* exact, uint8_t* out, int32_t* status,
                      uint8_t* norm_ws, size_t norm_ws_frames, hipStream_t stream);
bool image_hash_needs_ws(const uint8_t* frames, uint32_t w, uint32_t h, size_t row_stride, size_t frame_stride,
                         int pixfmt, uint32_t min_dim, uint32_t max_dim);
// the fused any-geometry kernel (ragged batches; uniform ones of a geometry the square kernels do not take)
bool image_any_plan(const uint8_t* base, uint64_t src, uint32_t w, uint32_t h, size_t row_stride, int pixfmt, uint32_t* cls,
                    uint32_t* magic, uint32_t* shift);
int image_any_group(uint32_t cls);           // which of the three kernels (launches) a frame of this class belongs to: 0, 1, 2
size_t image_any_item_bytes();
void image_any_item_write(void* dst, size_t i, uint64_t src, uint32_t w, uint32_t h, uint32_t row_stride, uint32_t slot, uint32_t cls,
                          uint32_t magic, uint32_t shift);
int launch_image_hash_any(uint32_t algo, const uint8_t* base, const void* d_items, size_t n, int group, uint32_t proto_w,
                          uint32_t proto_h, uint32_t proto_row_stride, uint32_t proto_cls, uint32_t proto_magic, uint32_t proto_shift,
                          size_t frame_stride, const uint8_t* lo, const uint8_t* hi, const uint8_t* exact, uint8_t* out,
                          int32_t* status, const uint32_t* d_geo, hipStream_t stream);
// d_geo: the context's table area, image_any_geometry_bytes() per width, indexed by width (0 .. 2048); a width's table is
// written once (image_any_geometry_table on the host, copied up) before the first launch that has a frame of that width
size_t image_any_geometry_bytes();
void image_any_geometry_table(uint32_t w, uint32_t* tab);
// (slot, status) pairs: zero record (out may be NULL) + the status for frames that are not hashed
int launch_image_preset_list(const uint32_t* d_entries, size_t n, uint8_t* out, uint32_t rec, int32_t* status, hipStream_t stream);
int launch_image_record_codes(const uint8_t* records, size_t n, uint32_t rec_bytes, uint32_t offset, uint64_t* codes,
                              hipStream_t stream);
int launch_image_synth(uint8_t* frames, size_t n, uint32_t w, uint32_t h, size_t first,
                       hipStream_t stream);

// blake3.hip
size_t blake3_ws_bytes(size_t n, size_t blob_bytes);
int launch_blake3_batch(const uint8_t* blob, const uint64_t* offsets, size_t n, uint8_t* ws, uint8_t* out, hipStream_t stream);

// ---- encoded uploads (png.hip, jpeg.hip): one file of a decode batch as the kernels see it ----
// A RAGGED batch carries a device table of these, built by the host from the per-file probe results (every upload its own
// geometry: the reference's route takes any image, src/server/handlers.rs:232-302); a UNIFORM batch -- ONE announced
// geometry, the round-2/3 entry points -- passes no table and the kernels derive entry k from the launch's parameters.
struct UpItem {
    uint32_t file;        // index into the caller's offsets / status / records
    uint32_t w, h;        // announced geometry: the file's own header must agree
    int32_t pixfmt;       // announced decoded format (PNG; a JPEG's luma plane is GRAY8)
    uint32_t row_stride;  // rows of the decoded frame, bytes
    uint32_t bxp, byp;    // JPEG: the coefficient plane in blocks
    uint32_t max_seg;     // JPEG: restart segments the file's table holds
    uint64_t frame_off;   // decoded frame: bytes from the launch's frame base
    uint64_t aux_off;     // PNG: filtered scanlines, bytes into the raw area; JPEG: coefficient plane, int16 units into the coefficient area
    uint64_t seg_off;     // JPEG: segment table, words into the segment area
};
struct UpUniform {        // the same for entry k of a uniform batch: file = k, offsets k x stride
    uint32_t w = 0, h = 0;
    int32_t pixfmt = 0;
    uint32_t row_stride = 0, bxp = 0, byp = 0, max_seg = 0;
    uint64_t frame_stride = 0, aux_stride = 0;
};
__device__ __forceinline__ UpItem up_item(const UpItem* __restrict__ items, const UpUniform& u, size_t k) {
    if (items) return items[k];
    return UpItem{(uint32_t)k, u.w, u.h, u.pixfmt, u.row_stride, u.bxp, u.byp, u.max_seg, k * u.frame_stride, k * u.aux_stride,
                  k * (uint64_t)(u.max_seg + 2)};
}

// png.hip
struct PngWs {
    size_t zbuf = 0, info = 0, raw = 0, raw_stride = 0, raw_n = 0, total = 0;
    size_t tok = 0, tinfo = 0;      // two-pass inflate: 16-bit token words (2 bytes per filtered byte at most), per-file counts
};
size_t png_ws_bytes(size_t n, size_t png_bytes, uint32_t w, uint32_t h, int pixfmt, PngWs* ws);
// side / fork / join (optional): a second stream and two events of the caller's -- a batch of >= 1600 files is then decoded as
// two halves side by side (the halves' stages fill each other's tails: 2000 files 240 k -> 250 k images/s, 8000 files 303 k ->
// 317 k; smaller batches lose a few per cent that way), joined into `stream` before the call returns
int launch_png_decode(const uint8_t* png, const uint64_t* offsets, size_t n, uint32_t w, uint32_t h, int pixfmt, uint8_t* ws,
                      const PngWs& l, uint8_t* frames, size_t row_stride, size_t frame_stride, int32_t* status,
                      hipStream_t stream, hipStream_t side = nullptr, hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
int launch_png_merge_status(const uint8_t* ws, const PngWs& l, size_t n, uint8_t* out, uint32_t rec, int32_t* status,
                            hipStream_t stream, const UpItem* d_items = nullptr);
// ragged: n files listed in d_items (device); raw_total = bytes of the raw area (sum of the files' aligned scanline sizes),
// max_row[f] = widest decoded row in bytes among the files announced as pixel format f (0: none; sizes the unfilter's LDS)
size_t png_ragged_ws_bytes(size_t n, size_t png_bytes, size_t raw_total, PngWs* ws);
size_t png_raw_bytes(uint32_t w, uint32_t h, int pixfmt);      // aligned bytes of one file's filtered scanlines in the raw area
int launch_png_decode_ragged(const uint8_t* png, const uint64_t* offsets, const UpItem* d_items, size_t n, const uint32_t max_w[3],
                             uint8_t* ws, const PngWs& l, uint8_t* frames, int32_t* status, hipStream_t stream);

// jpeg.hip
struct JpegWs {
    size_t clean = 0, info = 0, seg = 0, qtab = 0, coef = 0, coef_stride = 0, total = 0;
    size_t jpg_bytes = 0;   // the batch's encoded bytes (picks the decoder's waves per file)
    uint32_t bxp = 0, byp = 0, max_seg = 0;
};
size_t jpeg_ws_bytes(size_t n, size_t jpg_bytes, uint32_t w, uint32_t h, JpegWs* ws);
int launch_jpeg_decode(const uint8_t* jpg, const uint64_t* offsets, size_t n, uint32_t w, uint32_t h, uint8_t* ws,
                       const JpegWs& l, uint8_t* frames, size_t row_stride, size_t frame_stride, int32_t* status,
                       hipStream_t stream);
int launch_jpeg_merge_status(const uint8_t* ws, const JpegWs& l, size_t n, uint8_t* out, uint32_t rec, int32_t* status,
                             hipStream_t stream, const UpItem* d_items = nullptr);
// ragged: n files listed in d_items (device); seg_words / coef_words = the batch's totals of segment-table words and
// coefficient-plane int16 (per file: jpeg_plane_geometry); d_first: n + 1 device words, the first inverse-DCT workgroup of
// every entry (ceil(bxp byp / 256) each), d_first[n] = idct_groups
void jpeg_plane_geometry(uint32_t w, uint32_t h, uint32_t* bxp, uint32_t* byp, uint32_t* max_seg);
size_t jpeg_ragged_ws_bytes(size_t n, size_t jpg_bytes, size_t seg_words, size_t coef_words, JpegWs* ws);
int launch_jpeg_decode_ragged(const uint8_t* jpg, const uint64_t* offsets, const UpItem* d_items, size_t n, const uint32_t* d_first,
                              size_t idct_groups, uint8_t* ws, const JpegWs& l, uint8_t* frames, int32_t* status, hipStream_t stream);

// hamming.hip
struct HammingPlan {
    uint32_t qgroups = 0;       // ceil(nq / 64)
    int cap = 24;               // lane-private candidate list capacity (robust tier)
    size_t sample_n = 0;        // codes in the tau0 sample pre-pass
    uint32_t sample_parts = 0;
    size_t per_part = 0;
    size_t robust_n = 0;        // robust tier covers [0, robust_n): everything when `fast` is off, nothing when on
    uint32_t slices = 0;        // robust tier: one wave per (slice, qgroup)
    size_t per_slice = 0;
    bool bound = false;         // first bound from the matrix-core bound pass over [0, bound_n) instead of the sample histogram
    size_t bound_n = 0;
    bool fast = false;          // matrix-core filter in stages over [0, stage_end[0]), [stage_end[0], stage_end[1]), ...
    uint32_t nstages = 0;
    size_t stage_end[12] = {0};
    uint32_t cand_cap = 0;      // candidate slots per query in global memory
    uint32_t log_cap = 0;       // suspect-block records per wave and stage
    uint32_t fb_slices = 0;     // fallback robust scan over [0, n), device-gated on overflow
    size_t fb_per_slice = 0;
};
constexpr uint32_t kHammingMaxBatch = 4096;   // queries per launch_hamming_search call (workspace and log sizing)
HammingPlan hamming_plan(size_t n, uint32_t nq, uint32_t k);
size_t hamming_workspace_bytes(const HammingPlan& p, uint32_t nq, uint32_t k);
int launch_hamming_search(const uint64_t* codes, const uint64_t* ids, size_t n,
                          const uint64_t* queries, uint32_t nq, uint32_t k, uint8_t* ws,
                          const HammingPlan& p, uint64_t* out_ids, uint32_t* out_dist,
                          float* out_scores, uint32_t* out_cnt, hipStream_t stream,
                          const uint32_t* ids_ascending = nullptr);
// ids_ascending: device word, non-zero while the shard's record ids ascend with the row number (kept by
// launch_ids_order_update for append-only shards): later stages may then use strict thresholds (hamming_list_tau)
int launch_ids_order_update(const uint64_t* ids, size_t n, bool first, uint32_t* state, hipStream_t stream);
int launch_hamming_scores(const uint32_t* dist, size_t total, float* scores, hipStream_t stream);

// hamming_direct.hip: 1..8 queries, k <= 32, ONE launch (the request shape of /v1/query).  `state` =
// hamming_direct_state_bytes() of device memory whose first kHammingDirectZeroBytes were zeroed once (the kernel leaves
// them zero).
constexpr uint32_t kHammingDirectMaxQ = 8, kHammingDirectMaxK = 32, kHammingDirectZeroBytes = 12288;
bool hamming_direct_ok(size_t n, uint32_t nq, uint32_t k);
size_t hamming_direct_state_bytes();
int launch_hamming_direct(const uint64_t* codes, const uint64_t* ids, size_t n, const uint64_t* queries, uint32_t nq,
                          uint32_t k, uint8_t* state, uint64_t* out_ids, uint32_t* out_dist, float* out_scores,
                          uint32_t* out_cnt, hipStream_t stream, const uint32_t* ids_ascending = nullptr);

// topk.hip
struct SelectPlan {
    uint32_t slices = 0;
    size_t per_slice = 0;
};
SelectPlan select_plan(size_t n, uint32_t nq);
// few queries (<= 16): chunk minima -> threshold -> gather (topk.hip); `mins` = select_pruned_ws_bytes(n, nq) bytes
bool select_pruned_ok(size_t n, uint32_t nq, uint32_t k);
size_t select_pruned_ws_bytes(size_t n, uint32_t nq);
int launch_select_pruned_u32(const uint32_t* keys, const uint64_t* ids, size_t n, uint32_t nq, uint32_t k, uint32_t* mins,
                             uint64_t* out_ids, uint32_t* out_key, uint32_t* out_cnt, hipStream_t stream,
                             const uint32_t* run_flag);
int launch_select_topk_u32(const uint32_t* keys, const uint64_t* ids, size_t n, const SelectPlan& p,
                           uint32_t nq, uint32_t k, uint64_t* part_ids, uint32_t* part_key,
                           uint32_t* part_cnt, hipStream_t stream, const uint32_t* run_flag = nullptr);
// best k of (a base list of k entries + a list of candidate row numbers) per query; see topk.hip
int launch_topk_select_lists_u32(const uint64_t* base_ids, const uint32_t* base_key, const uint32_t* ckey,
                                 const uint32_t* crow, const uint32_t* ccnt, uint32_t cap, const uint64_t* ids,
                                 uint32_t nq, uint32_t k, uint64_t* out_ids, uint32_t* out_key, uint32_t* out_cnt,
                                 uint32_t* overflow, hipStream_t stream);
// 5 .. 48 cosine queries without a key matrix (cosine.hip CosinePrune, topk.hip prune_*): chunk minima -> threshold and
// ~k listed chunks per query -> their keys recomputed -> answer; *flag != 0 afterwards = take the dense path instead
struct CosinePrunePlan {
    uint32_t cs_shift = 5;   // rows per chunk = 1 << cs_shift
    uint32_t qpad = 16;      // minima per chunk (queries padded to the keys kernel's tile): mins[chunk][qpad]
    uint32_t nchunks = 0;
    uint32_t waves = 0;      // waves of the minima launch: wmin[qpad][waves]
    uint32_t capq = 32;      // listed chunks per query at most
};
constexpr uint32_t kPruneCand = 2048;   // chunks per query at or below the waves' bound, ranked exactly
bool cosine_prune_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n, uint32_t k);
CosinePrunePlan cosine_prune_plan(size_t n, uint32_t nq_pass, uint32_t k, bool approx = false);
// the minima through the f16 matrix pipe (cosine.hip cosine_mins_f16): approximate within cosine_mins_eps(dim) of the exact
// score, 5 .. 64 queries per pass; the prune kernels take the same eps; *flag raised when a norm or a score leaves the range
// the bound holds in.  image: nq x dim halves, written with the norms by launch_cosine_norms_image
float cosine_mins_eps(uint32_t dim);
uint32_t cosine_list_queries(uint32_t dim);   // queries one exact list pass holds
bool cosine_mins_f16_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n, uint32_t k);
int launch_cosine_norms_image(const float* queries, size_t nq, uint32_t dim, float* norms, void* image, uint32_t* zero2,
                              hipStream_t stream);   // zero2: two words the kernel zeroes (the pass's flag and list counter)
int launch_cosine_mins_f16(const float* rows, const float* norms, size_t n, uint32_t dim, const void* image, const float* qnorm,
                           uint32_t nq_pass, const CosinePrunePlan& p, uint32_t* mins, uint32_t* wmin, uint32_t* flag,
                           hipStream_t stream);
int launch_cosine_keys_dense_mfma(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                                  const float* qnorm, uint32_t nq_pass, uint32_t* keys, const uint32_t* run_flag,
                                  hipStream_t stream);
int launch_cosine_keys_mins(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                            const float* qnorm, uint32_t nq_pass, const CosinePrunePlan& p, uint32_t* mins, uint32_t* wmin,
                            hipStream_t stream);
int launch_cosine_keys_list(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                            const float* qnorm, uint32_t nq_pass, const CosinePrunePlan& p, const void* list,
                            const uint32_t* nlist, uint32_t* ckeys, const uint32_t* fallback_flag, hipStream_t stream);
// bound[q] = k-th smallest wave minimum -> cand[q] = (minimum, chunk) of the chunks at or below it -> tau[q] = k-th smallest chunk
// minimum; list = (query, chunk) of every chunk whose minimum is <= tau[q]; qrange[q] = (first entry, entries); *nlist = entries
// in all; *flag raised when a query lists more than capq chunks or has no threshold.  ws: bound[nq] ccnt[nq] cand[nq][kPruneCand] x 8 B
size_t prune_tau_ws_bytes(uint32_t nq);
int launch_prune_tau(const uint32_t* mins, const uint32_t* wmin, const CosinePrunePlan& p, uint32_t nq, uint32_t k, uint8_t* ws,
                     uint32_t* tau, void* list, uint32_t* nlist, void* qrange, uint32_t* flag, hipStream_t stream,
                     float eps = 0.f);   // eps: the minima are within eps of the exact scores (topk.hip relax_key)
// best k by (key, id) of the listed chunks' keys <= tau[q]; does nothing once *flag is raised (and raises it if its own list overflows)
int launch_prune_final(const uint32_t* ckeys, const CosinePrunePlan& p, const void* list, const void* qrange,
                       const uint32_t* tau, const uint64_t* ids, size_t n, uint32_t nq, uint32_t k, uint64_t* out_ids,
                       uint32_t* out_key, uint32_t* out_cnt, uint32_t* flag, hipStream_t stream);
constexpr uint32_t kCosineListCap = 1024;   // candidates kept per query by the filtered cosine pass
// run_flag: optional device word; when non-null the merge only runs if it is non-zero
int launch_topk_merge_u32(const uint64_t* part_ids, const uint32_t* part_key, uint32_t parts,
                          uint32_t nq, uint32_t k, uint64_t* out_ids, uint32_t* out_key,
                          uint32_t* out_cnt, const uint32_t* run_flag, hipStream_t stream,
                          float* hamming_scores = nullptr);   // optional: 1 - key / 64 of the final keys, written even when run_flag is 0
// wire format of the sharded search: 16-byte entries {id u64, key u32, pad u32}
int launch_topk_pack_entries(const uint64_t* ids, const uint32_t* keys, size_t total, void* entries, hipStream_t stream);
// missing (optional, device): bit p set = part p's first entry carries the 0xffffffff pad word of a shard that could not scan
int launch_topk_merge_packed(const void* entries, uint32_t parts, uint32_t nq, uint32_t k, uint64_t* out_ids,
                             uint32_t* out_key, uint32_t* out_cnt, hipStream_t stream, uint64_t* missing = nullptr);
// tree merge (fan-in 64 per level); tmp_* hold 2 x topk_merge_tmp_entries(parts, nq, k) entries
size_t topk_merge_tmp_entries(uint32_t parts, uint32_t nq, uint32_t k);
int launch_topk_merge_tree_u32(const uint64_t* part_ids, const uint32_t* part_key, uint32_t parts, uint32_t nq,
                               uint32_t k, uint64_t* tmp_ids, uint32_t* tmp_key, uint64_t* out_ids, uint32_t* out_key,
                               uint32_t* out_cnt, hipStream_t stream, const uint32_t* run_flag = nullptr);

// cosine.hip
int launch_cosine_norms(const float* rows, size_t n, uint32_t dim, float* norms, hipStream_t stream);
int cosine_queries_per_pass(uint32_t dim, size_t nq);
int launch_cosine_keys(const float* rows, const float* norms, size_t n, uint32_t dim, const float* queries,
                       const float* qnorm, uint32_t nq_pass, uint32_t* keys, hipStream_t stream,
                       const uint32_t* run_flag = nullptr);
// the batch (GEMM) path with a per-query key threshold: rows whose key is <= tau[q] are appended to q's candidate
// list instead of writing the nq x n key matrix; false if this (rows, dim, batch) does not take the GEMM path
bool cosine_filter_ok(const float* rows, uint32_t dim, const float* queries, uint32_t nq_pass, size_t n);
int launch_cosine_tau(const uint32_t* base_key, uint32_t k, const float* qnorm, uint32_t nq, uint32_t* tau, float* uq,
                      uint32_t* ccnt, hipStream_t stream);
int launch_cosine_keys_filtered(const float* rows, const float* norms, size_t n, size_t row_base, uint32_t dim,
                                const float* queries, const float* qnorm, uint32_t nq_pass, const uint32_t* tau,
                                const float* uq, uint32_t* ccnt, uint32_t* ckey, uint32_t* crow, uint32_t cap,
                                hipStream_t stream);
int launch_cosine_scores_from_keys(const uint32_t* keys, size_t total, float* scores, hipStream_t stream);

// text.hip
int launch_text_minhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint32_t k,
                        uint8_t* out, int32_t* status, hipStream_t stream);
int launch_text_simhash(const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode, uint8_t* out,
                        int32_t* status, hipStream_t stream);

// audio.hip
struct WangWs {
    size_t frames = 0;          // upper bounds (the per-clip lengths of a batch live on the device)
    uint32_t n_sec = 0, n_seg = 0, seg = 0, n_clips = 0;
    size_t clips, seg_cnt, seg_base, sec_cnt, sec_base, seg_clip, sec_clip, out_off, cand_cnt, cand_t, cand_k, cand_p,
        sel_cnt, sel_off, sel_t, sel_k, sel_p, pt, pk, pp, pc, pair_cnt, pair_off, scan_tmp, total = 0;
};
size_t audio_resample_len(size_t n, uint32_t sr_in, uint32_t sr_out);
int launch_resample_linear(const float* in, size_t n, uint32_t sr_in, uint32_t sr_out, float* out,
                           hipStream_t stream);
size_t audio_stft_frames(size_t n, int N, int hop);
WangWs wang_ws_layout(size_t n_src_total, size_t n_clips, uint32_t sr_in, uint32_t pps);
int launch_wang_batch(const float* pcm, const uint64_t* d_offsets, size_t n_src_total, size_t n_clips, uint32_t sr_in,
                      uint32_t fan_out, uint32_t zone_t, uint32_t zone_f, uint32_t pps,
                      float floor_power, uint8_t* ws, const WangWs& w, uint32_t* out, size_t cap, uint64_t* d_out_off,
                      uint64_t* out_count, hipStream_t stream);
struct HaitsmaBatchWs {
    size_t n5_ub = 0, frames_ub = 0;      // upper bounds (the per-clip lengths of a batch live on the device)
    size_t edges, s5_off, fr_off, src_map, pcm5k, E, total = 0;
};
HaitsmaBatchWs haitsma_batch_ws(size_t n_total, size_t n_clips, uint32_t sr);
int launch_haitsma_batch(const float* pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips, uint32_t sr,
                         const uint32_t* h_edges, uint8_t* ws, const HaitsmaBatchWs& w, uint32_t* out, size_t cap_frames,
                         uint64_t* d_out_offsets, hipStream_t stream);
size_t haitsma_ws_bytes(size_t n5k);
int launch_haitsma(const float* pcm5k, size_t n, const uint32_t* h_edges, uint8_t* ws, uint32_t* out,
                   hipStream_t stream);

}  // namespace ucfp
