"""ctypes binding of libucfp_hip.so -- the C ABI declared in include/ucfp_hip.h.

There is no fallback: if the shared library is missing or no gfx950 device is present the
import of a compute entry point fails loudly (the product path never routes through CPU code).
"""
import ctypes as C
import os
import threading

from . import errors

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("UCFP_HIP_SO") or os.path.join(_HERE, "libucfp_hip.so")   # (UCFP_HIP_SO: a tuning build of the same library)

_lib = None
_lock = threading.Lock()


class WangConfig(C.Structure):
    """ucfp_wang_config (audiofp WangConfig; defaults src/server/algorithms_manifest.rs:553-592)."""
    _fields_ = [("fan_out", C.c_uint32), ("target_zone_t", C.c_uint32), ("target_zone_f", C.c_uint32),
                ("peaks_per_sec", C.c_uint32), ("min_anchor_mag_db", C.c_float)]


class HaitsmaConfig(C.Structure):
    """ucfp_haitsma_config (defaults manifest :655-672)."""
    _fields_ = [("fmin", C.c_float), ("fmax", C.c_float)]


class ImageItem(C.Structure):
    """ucfp_image_item: one decoded frame of a ragged batch."""
    _fields_ = [("offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("row_stride", C.c_uint32),
                ("pixfmt", C.c_int32)]


class UploadInfo(C.Structure):
    """ucfp_upload_info: what ucfp_image_probe says about one encoded upload."""
    _fields_ = [("format", C.c_int32), ("status", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pixfmt", C.c_int32), ("reserved", C.c_uint32)]


class ImagePreprocess(C.Structure):
    """ucfp_image_preprocess (imgfprint::PreprocessConfig guards)."""
    _fields_ = [("max_dimension", C.c_uint32), ("min_dimension", C.c_uint32)]


# name -> (restype, argtypes); every symbol of include/ucfp_hip.h appears here and
# tests/test_abi.py checks the two lists against each other.
SIGNATURES = {
    "ucfp_abi_version": (C.c_int, []),
    "ucfp_last_error": (C.c_char_p, []),
    "ucfp_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ucfp_ctx_destroy": (None, [C.c_void_p]),
    "ucfp_image_record_bytes": (C.c_size_t, [C.c_uint32]),
    "ucfp_image_hash_batch_dev": (C.c_int, [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t,
        C.c_size_t, C.c_int, C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p]),
    "ucfp_image_hash_batch": (C.c_int, [
        C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t,
        C.c_size_t, C.c_int, C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_image_hash_ragged_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(ImageItem), C.c_size_t,
                                             C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_image_hash_ragged": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(ImageItem), C.c_size_t,
                                         C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_image_probe": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(UploadInfo)]),
    "ucfp_image_probe_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_image_upload_hash_batch_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                   C.POINTER(UploadInfo), C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    "ucfp_image_upload_frames_bytes": (C.c_size_t, [C.POINTER(UploadInfo), C.c_size_t]),
    "ucfp_image_upload_decode_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                     C.POINTER(UploadInfo), C.c_void_p, C.c_size_t, C.POINTER(ImageItem),
                                                     C.c_void_p, C.c_void_p]),
    "ucfp_audio_wang_max_hashes": (C.c_size_t, [C.c_size_t, C.POINTER(WangConfig)]),
    "ucfp_audio_wang": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(WangConfig), C.c_void_p,
                                  C.c_size_t, C.POINTER(C.c_size_t)]),
    "ucfp_audio_wang_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(WangConfig),
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_audio_wang_batch_max_hashes": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(WangConfig)]),
    "ucfp_audio_wang_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                            C.POINTER(WangConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_audio_haitsma_frames": (C.c_size_t, [C.c_size_t, C.c_uint32]),
    "ucfp_audio_haitsma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(HaitsmaConfig),
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ucfp_audio_haitsma_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(HaitsmaConfig), C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    "ucfp_audio_resample_len": (C.c_size_t, [C.c_size_t, C.c_uint32, C.c_uint32]),
    "ucfp_audio_resample_linear_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "ucfp_text_minhash_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_text_minhash_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32,
                                          C.c_void_p, C.c_void_p]),
    "ucfp_text_simhash_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_text_simhash_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    "ucfp_text_lsh_band_keys_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_void_p]),
    "ucfp_lsh_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_lsh_destroy": (None, [C.c_void_p]),
    "ucfp_lsh_build_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ucfp_lsh_query_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "ucfp_image_record_codes_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_void_p]),
    "ucfp_index_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ucfp_index_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ucfp_index_create": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_index_destroy": (None, [C.c_void_p]),
    "ucfp_index_upsert": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ucfp_index_append_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ucfp_index_delete": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ucfp_index_size": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]),
    "ucfp_index_flush": (C.c_int, [C.c_void_p]),
    "ucfp_index_search": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_index_search_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_index_search_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_index_search_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_index_search_batcher_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.POINTER(C.c_uint32)]),
    "ucfp_index_search_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_topk_merge_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t,
                                      C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_topk_pack_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ucfp_topk_merge_packed_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_topk_merge_packed_ex_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_shard_unique_id": (C.c_int, [C.c_void_p]),
    "ucfp_shard_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ucfp_shard_comm_create_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32,
                                            C.POINTER(C.c_void_p)]),
    "ucfp_shard_comm_uses_rccl": (C.c_int, [C.c_void_p]),
    "ucfp_shard_comm_destroy": (None, [C.c_void_p]),
    "ucfp_shard_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "ucfp_shard_range": (None, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_index_search_sharded_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                                   C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.POINTER(C.c_uint64)]),
    "ucfp_index_search_sharded_collect": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "ucfp_index_search_sharded_missing": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ucfp_index_search_sharded_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_image_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                            C.POINTER(ImagePreprocess), C.c_size_t, C.c_uint32,
                                            C.POINTER(C.c_void_p)]),
    "ucfp_image_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_image_batcher_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int32)]),
    "ucfp_image_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_text_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_size_t, C.c_size_t,
                                           C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_text_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_text_batcher_submit": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int32)]),
    "ucfp_text_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_audio_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(WangConfig), C.c_size_t, C.c_size_t,
                                            C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_audio_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_audio_batcher_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "ucfp_audio_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_png_probe": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_int)]),
    "ucfp_image_png_decode_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                                  C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                  C.c_void_p]),
    "ucfp_image_png_hash_batch_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                C.c_uint32, C.c_uint32, C.c_int, C.POINTER(ImagePreprocess), C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "ucfp_jpeg_probe": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ucfp_image_jpeg_decode_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                                   C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_image_jpeg_hash_batch_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                 C.c_uint32, C.c_uint32, C.POINTER(ImagePreprocess), C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "ucfp_sidecar_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ucfp_sidecar_close": (None, [C.c_void_p]),
    "ucfp_sidecar_append_upsert": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint32, C.c_void_p,
                                             C.c_uint32, C.c_char_p, C.c_uint32]),
    "ucfp_sidecar_append_delete": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "ucfp_sidecar_sync": (C.c_int, [C.c_void_p]),
    "ucfp_sidecar_snapshot_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_sidecar_snapshot_close": (None, [C.c_void_p]),
    "ucfp_sidecar_snapshot_row": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "ucfp_sidecar_snapshot_gather_fingerprints": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ucfp_sidecar_snapshot_gather_vectors": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_uint64, C.POINTER(C.c_uint64)]),
    "ucfp_audio_haitsma_batch_max_frames": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_uint32]),
    "ucfp_audio_haitsma_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                               C.POINTER(HaitsmaConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_blake3_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ucfp_png_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(ImagePreprocess),
                                          C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_jpeg_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ImagePreprocess),
                                           C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_png_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_png_batcher_submit": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int32)]),
    "ucfp_png_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_upload_batcher_create": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(ImagePreprocess), C.c_size_t, C.c_size_t,
                                             C.c_uint32, C.POINTER(C.c_void_p)]),
    "ucfp_upload_batcher_destroy": (None, [C.c_void_p]),
    "ucfp_upload_batcher_submit": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int32)]),
    "ucfp_upload_batcher_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ucfp_blake3": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ucfp_image_synth_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                       C.c_uint32, C.c_size_t, C.c_void_p]),
}


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (SONAME
    libamdhip64.so.7) and resolves it by path; if /opt/rocm's copy were loaded first the process
    would hold two HIP/HSA runtimes and torch would lose the device.  Loading torch's copy first
    makes our DT_NEEDED libamdhip64.so.7 bind to the same runtime, so device pointers and
    hipStream_t handles are shared between torch (memory/stream plumbing) and our kernels."""
    try:
        import torch  # noqa: F401  (plumbing only; the kernels do not use torch)
    except ImportError:
        return


def load() -> C.CDLL:
    """dlopen the in-tree library and declare every signature. Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(SO_PATH):
                raise ImportError(
                    f"{SO_PATH} is missing: run `python -m ucfp_amd.build` (or "
                    "__graft_entry__.build()). ucfp_amd has no CPU fallback.")
            _preload_hip_runtime()
            lib = C.CDLL(SO_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ucfp_last_error().decode("utf-8", "replace")
        raise errors.from_status(rc, msg)


class Context:
    """Owns one ucfp_ctx (one per process per GPU). `device` is the HIP ordinal (LOCAL_RANK)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        check(self._lib.ucfp_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def current_device() -> int:
    """The device this process works on: torch's current device when torch has one selected (one process per GPU:
    the launcher did `torch.cuda.set_device(LOCAL_RANK)`), else LOCAL_RANK, else 0."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        return int(torch.cuda.current_device())
    return int(os.environ.get("LOCAL_RANK", "0"))


def current_context() -> "Context":
    """Default context of the CURRENT device (never silently device 0 on a rank > 0)."""
    return default_context(current_device())


def default_context(device: int = 0) -> Context:
    with _lock:
        ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _default_ctx[device] = ctx
    return ctx
