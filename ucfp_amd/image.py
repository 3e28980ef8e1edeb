"""Image fingerprinting -- host-side mirror of src/modality/image.rs.

Same names, argument meaning and error behaviour as the reference adapters:

    fingerprint(bytes, tenant_id, record_id)                      image.rs:56-58
    fingerprint_with(bytes, tenant_id, record_id, preprocess)     image.rs:62-88   (536-B bundle)
    fingerprint_phash / _dhash / _ahash(bytes, preprocess, t, r)  image.rs:112-162 (168 B each)

plus the batched form the GPU wants (`fingerprint_frames`), which the reference lacks
 and its per-request adapters are the n = 1 case of.  All hashing runs in the HIP
library through the C ABI; this module only decodes (Pillow -- decode is out of the hot path,
SURVEY 8f N4), enforces PreprocessConfig and wraps bytes into `Record`s.
"""
import ctypes as C
import io
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .core import Modality, Record
from .errors import ModalityError

# algorithm tags: src/modality/image.rs:38-48
ALGORITHM = "imgfprint-multihash-v1"
ALGORITHM_MULTIHASH = "imgfprint-multihash-v1"
ALGORITHM_PHASH = "imgfprint-phash-v1"
ALGORITHM_DHASH = "imgfprint-dhash-v1"
ALGORITHM_AHASH = "imgfprint-ahash-v1"

FORMAT_VERSION = 1  # imgfprint::FORMAT_VERSION as stored by image.rs:76

AHASH, PHASH, DHASH, MULTI = 1, 2, 4, 7
PIX_GRAY8, PIX_RGB8, PIX_RGBA8 = 0, 1, 2
_TAG = {AHASH: ALGORITHM_AHASH, PHASH: ALGORITHM_PHASH, DHASH: ALGORITHM_DHASH,
        MULTI: ALGORITHM_MULTIHASH}
_BPP = {PIX_GRAY8: 1, PIX_RGB8: 3, PIX_RGBA8: 4}


@dataclass
class PreprocessConfig:
    """imgfprint::PreprocessConfig; defaults per src/server/algorithms_manifest.rs:446-469,
    query mapping src/server/handlers.rs:307-319."""
    max_input_bytes: int = 50 * 1024 * 1024
    max_dimension: int = 8192
    min_dimension: int = 32

    def _c(self) -> _lib.ImagePreprocess:
        return _lib.ImagePreprocess(self.max_dimension, self.min_dimension)


def record_bytes(algo: int) -> int:
    return int(_lib.load().ucfp_image_record_bytes(algo))


# ------------------------------------------------------------------------------------------
# batched, decoded frames (device or host resident)
# ------------------------------------------------------------------------------------------

def fingerprint_frames_dev(frames_ptr: int, n: int, width: int, height: int, *, algo: int = MULTI,
                           pixfmt: int = PIX_GRAY8, row_stride: Optional[int] = None,
                           frame_stride: Optional[int] = None, exact_ptr: int = 0, out_ptr: int,
                           status_ptr: int = 0, stream: int = 0,
                           preprocess: Optional[PreprocessConfig] = None, ctx=None) -> None:
    """Enqueue hashing of `n` device-resident frames; raw device addresses, no sync.
    `stream` is a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""
    ctx = ctx or _lib.current_context()
    bpp = _BPP[pixfmt]
    rs = row_stride if row_stride is not None else width * bpp
    fs = frame_stride if frame_stride is not None else rs * height
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_hash_batch_dev(
        ctx.handle, algo, frames_ptr, n, width, height, rs, fs, pixfmt, C.byref(pre),
        exact_ptr or None, out_ptr, status_ptr or None, stream or None))


def record_codes_dev(records_ptr: int, n: int, codes_ptr: int, *, algo: int = MULTI, which: int = PHASH,
                     stream: int = 0, ctx=None) -> None:
    """Device records (168 / 536 B each) -> their 64-bit global hashes, ready for DeviceIndex.append_dev."""
    ctx = ctx or _lib.current_context()
    _lib.check(_lib.load().ucfp_image_record_codes_dev(ctx.handle, records_ptr, n, algo, which, codes_ptr,
                                                       stream or None))


def fingerprint_frames(frames: np.ndarray, *, algo: int = MULTI, pixfmt: int = PIX_GRAY8,
                       exact: Optional[np.ndarray] = None,
                       preprocess: Optional[PreprocessConfig] = None, ctx=None):
    """Hash host-resident decoded frames [n, h, w(, c)] uint8 through the host-pointer ABI.
    Returns (records uint8 [n, record_bytes], status int32 [n])."""
    ctx = ctx or _lib.current_context()
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim < 3:
        raise ModalityError("frames must be [n, h, w] or [n, h, w, c]")
    n, h, w = frames.shape[:3]
    bpp = _BPP[pixfmt]
    if frames.size != n * h * w * bpp:
        raise ModalityError(f"frames shape {frames.shape} does not match pixfmt {pixfmt}")
    rec = record_bytes(algo)
    out = np.zeros((n, max(rec, 1)), np.uint8)
    status = np.zeros(n, np.int32)
    ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        if ex.shape != (n, 32):
            raise ModalityError("exact must be [n, 32] bytes")
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_hash_batch(
        ctx.handle, algo, frames.ctypes.data, n, w, h, w * bpp, h * w * bpp, pixfmt,
        C.byref(pre), ex.ctypes.data if ex is not None else None, out.ctypes.data,
        status.ctypes.data))
    return out[:, :rec], status


# ------------------------------------------------------------------------------------------
# ragged batches: every frame its own geometry (the reference's route takes any upload, handlers.rs:232-302)
# ------------------------------------------------------------------------------------------

def make_items(geoms) -> "C.Array":
    """[(offset, width, height, row_stride, pixfmt), ...] -> ucfp_image_item array."""
    arr = (_lib.ImageItem * max(len(geoms), 1))()
    for i, (off, w, h, rs, fmt) in enumerate(geoms):
        arr[i] = _lib.ImageItem(int(off), int(w), int(h), int(rs), int(fmt))
    return arr


def fingerprint_frames_ragged_dev(frames_ptr: int, frames_bytes: int, geoms, *, algo: int = MULTI, exact_ptr: int = 0,
                                  out_ptr: int, status_ptr: int = 0, stream: int = 0,
                                  preprocess: Optional[PreprocessConfig] = None, ctx=None) -> None:
    """Enqueue hashing of device-resident frames of ANY mix of geometries: geoms[i] = (byte offset into the buffer at
    frames_ptr, width, height, row_stride, pixfmt).  One launch per form of row; no sync (ucfp_image_hash_ragged_dev)."""
    ctx = ctx or _lib.current_context()
    pre = (preprocess or PreprocessConfig())._c()
    items = geoms if not isinstance(geoms, (list, tuple)) else make_items(geoms)
    n = len(geoms)
    _lib.check(_lib.load().ucfp_image_hash_ragged_dev(ctx.handle, algo, frames_ptr, frames_bytes, items, n, C.byref(pre),
                                                      exact_ptr or None, out_ptr, status_ptr or None, stream or None))


def fingerprint_frames_ragged(frames: Sequence[np.ndarray], pixfmts: Sequence[int], *, algo: int = MULTI,
                              exact: Optional[np.ndarray] = None, preprocess: Optional[PreprocessConfig] = None, ctx=None):
    """Hash host-resident decoded frames of different sizes ([h, w] or [h, w, c] uint8 each) in one call through the
    host-pointer ABI (ucfp_image_hash_ragged).  Returns (records uint8 [n, record_bytes], status int32 [n])."""
    ctx = ctx or _lib.current_context()
    n = len(frames)
    rec = record_bytes(algo)
    geoms, parts, off = [], [], 0
    for f, fmt in zip(frames, pixfmts):
        f = np.ascontiguousarray(f, dtype=np.uint8)
        h, w = f.shape[:2]
        if f.size != h * w * _BPP[fmt]:
            raise ModalityError(f"frame shape {f.shape} does not match pixfmt {fmt}")
        geoms.append((off, w, h, w * _BPP[fmt], fmt))
        parts.append(f.reshape(-1))
        off += f.size
    blob = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    out = np.zeros((max(n, 1), rec), np.uint8)
    status = np.zeros(max(n, 1), np.int32)
    ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        if ex.shape != (n, 32):
            raise ModalityError("exact must be [n, 32] bytes")
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_hash_ragged(ctx.handle, algo, blob.ctypes.data, blob.size, make_items(geoms), n, C.byref(pre),
                                                  ex.ctypes.data if ex is not None else None, out.ctypes.data, status.ctypes.data))
    return out[:n], status[:n]


# ------------------------------------------------------------------------------------------
# encoded PNG files in: chunk walk, inflate and filter reconstruction on the GPU
# ------------------------------------------------------------------------------------------
NEEDS_HOST = 1


def png_probe(data: bytes):
    """IHDR of a PNG -> (status, width, height, pixfmt); status 0, NEEDS_HOST (a kind the device decoder hands back)
    or a negative UCFP_E_* (not a PNG)."""
    w, h, fmt = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
    rc = _lib.load().ucfp_png_probe(data, len(data), C.byref(w), C.byref(h), C.byref(fmt))
    return int(rc), int(w.value), int(h.value), int(fmt.value)


def _upload_pngs(pngs: Sequence[bytes], dev):
    import torch
    offs = np.zeros(len(pngs) + 1, np.int64)
    np.cumsum([len(p) for p in pngs], out=offs[1:])
    blob = np.frombuffer(b"".join(pngs) + b"\0" * 16, np.uint8)
    return torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offs).to(dev), int(offs[-1])


def decode_pngs(pngs: Sequence[bytes], width: int, height: int, pixfmt: int, ctx=None):
    """Decode a batch of PNG files announced as width x height, `pixfmt` on the GPU.
    -> (frames uint8 [n, h, w(, c)], status int32 [n]); frames of files with status != 0 are undefined."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n, bpp = len(pngs), _BPP[pixfmt]
    d_blob, d_off, total = _upload_pngs(pngs, dev)
    shape = (n, height, width) if bpp == 1 else (n, height, width, bpp)
    d_fr = torch.zeros(shape, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().ucfp_image_png_decode_batch_dev(
        ctx.handle, d_blob.data_ptr(), d_off.data_ptr(), n, total, width, height, pixfmt, d_fr.data_ptr(), width * bpp,
        width * bpp * height, d_st.data_ptr(), torch.cuda.current_stream().cuda_stream or None))
    return d_fr.cpu().numpy(), d_st[:n].cpu().numpy()


def fingerprint_pngs_dev(png_ptr: int, offsets_ptr: int, n: int, png_bytes: int, width: int, height: int, pixfmt: int, *,
                         algo: int = MULTI, exact_ptr: int = 0, out_ptr: int, status_ptr: int = 0, stream: int = 0,
                         preprocess: Optional[PreprocessConfig] = None, ctx=None) -> None:
    """Device-resident encoded files -> records; raw device addresses, no sync (ucfp_image_png_hash_batch_dev)."""
    ctx = ctx or _lib.current_context()
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_png_hash_batch_dev(
        ctx.handle, algo, png_ptr, offsets_ptr, n, png_bytes, width, height, pixfmt, C.byref(pre), exact_ptr or None,
        out_ptr, status_ptr or None, stream or None))


def fingerprint_pngs(pngs: Sequence[bytes], width: int, height: int, pixfmt: int, *, algo: int = MULTI,
                     exact: Optional[np.ndarray] = None, preprocess: Optional[PreprocessConfig] = None, ctx=None):
    """Host convenience: encoded files -> (records uint8 [n, record_bytes], status int32 [n])."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(pngs)
    rec = record_bytes(algo)
    d_blob, d_off, total = _upload_pngs(pngs, dev)
    d_out = torch.zeros((max(n, 1), rec), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    d_ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        if ex.shape != (n, 32):
            raise ModalityError("exact must be [n, 32] bytes")
        d_ex = torch.from_numpy(ex).to(dev)
    fingerprint_pngs_dev(d_blob.data_ptr(), d_off.data_ptr(), n, total, width, height, pixfmt, algo=algo,
                         exact_ptr=d_ex.data_ptr() if d_ex is not None else 0, out_ptr=d_out.data_ptr(),
                         status_ptr=d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                         preprocess=preprocess, ctx=ctx)
    return d_out[:n].cpu().numpy(), d_st[:n].cpu().numpy()


# ---- uploads of any kind and size in one batch (ucfp_image_probe / ucfp_image_upload_*; upload.hip) ----
UPLOAD_OTHER, UPLOAD_PNG, UPLOAD_JPEG = 0, 1, 2


def probe(data: bytes) -> "_lib.UploadInfo":
    """What an upload is, from its first bytes: .format (UPLOAD_*), .status (0: the device decodes it; NEEDS_HOST; < 0: no
    image), .width / .height / .pixfmt of the frame it decodes to."""
    info = _lib.UploadInfo()
    _lib.load().ucfp_image_probe(data, len(data), C.byref(info))
    return info


def _probe_all(files: Sequence[bytes]):
    arr = (_lib.UploadInfo * max(len(files), 1))()
    lib = _lib.load()
    for i, f in enumerate(files):
        lib.ucfp_image_probe(f, len(f), C.byref(arr[i]))
    return arr


def decode_uploads(files: Sequence[bytes], ctx=None):
    """Decode a batch of PNG / JPEG files of ANY sizes on the GPU -> (list of frames, status int32 [n]); frame i is uint8
    [h, w] or [h, w, c] (a JPEG: its luma plane), None where status[i] != 0."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(files)
    if n == 0:
        return [], np.zeros(0, np.int32)
    info = _probe_all(files)
    lib = _lib.load()
    need = int(lib.ucfp_image_upload_frames_bytes(info, n))
    d_blob, d_off, total = _upload_pngs(files, dev)
    d_fr = torch.zeros(need + 64, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    items = (_lib.ImageItem * n)()
    _lib.check(lib.ucfp_image_upload_decode_batch_dev(ctx.handle, d_blob.data_ptr(), d_off.data_ptr(), n, total, info,
                                                      d_fr.data_ptr(), need, items, d_st.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream or None))
    st = d_st.cpu().numpy()
    buf = d_fr.cpu().numpy()
    out = []
    for i in range(n):
        it = items[i]
        if st[i] != 0 or it.width == 0:
            out.append(None)
            continue
        bpp = _BPP[it.pixfmt]
        rows = np.lib.stride_tricks.as_strided(buf[it.offset:], shape=(it.height, it.width * bpp), strides=(it.row_stride, 1))
        fr = np.ascontiguousarray(rows)
        out.append(fr if bpp == 1 else fr.reshape(it.height, it.width, bpp))
    return out, st


def fingerprint_uploads_dev(blob_ptr: int, offsets_ptr: int, n: int, blob_bytes: int, info, *, algo: int = MULTI,
                            exact_ptr: int = 0, out_ptr: int, status_ptr: int = 0, stream: int = 0,
                            preprocess: Optional[PreprocessConfig] = None, ctx=None) -> None:
    """Device-resident encoded uploads of any kinds and sizes -> records (ucfp_image_upload_hash_batch_dev).  info: the
    UploadInfo array of `probe` results, or None: probed on the device (one synchronisation inside the call)."""
    ctx = ctx or _lib.current_context()
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_upload_hash_batch_dev(ctx.handle, algo, blob_ptr, offsets_ptr, n, blob_bytes, info,
                                                            C.byref(pre), exact_ptr or None, out_ptr, status_ptr or None,
                                                            stream or None))


def fingerprint_uploads(files: Sequence[bytes], *, algo: int = MULTI, exact: Optional[np.ndarray] = None,
                        preprocess: Optional[PreprocessConfig] = None, probe_on_device: bool = False, ctx=None):
    """Host convenience: encoded uploads (PNG, JPEG, anything) -> (records uint8 [n, record_bytes], status int32 [n]);
    status NEEDS_HOST: decode that upload on the host (`fingerprint_with`)."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(files)
    rec = record_bytes(algo)
    if n == 0:
        return np.zeros((0, rec), np.uint8), np.zeros(0, np.int32)
    d_blob, d_off, total = _upload_pngs(files, dev)
    d_out = torch.zeros((n, rec), dtype=torch.uint8, device=dev)
    d_st = torch.full((n,), 77, dtype=torch.int32, device=dev)
    d_ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        if ex.shape != (n, 32):
            raise ModalityError("exact must be [n, 32] bytes")
        d_ex = torch.from_numpy(ex).to(dev)
    fingerprint_uploads_dev(d_blob.data_ptr(), d_off.data_ptr(), n, total, None if probe_on_device else _probe_all(files),
                            algo=algo, exact_ptr=d_ex.data_ptr() if d_ex is not None else 0, out_ptr=d_out.data_ptr(),
                            status_ptr=d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, preprocess=preprocess,
                            ctx=ctx)
    return d_out.cpu().numpy(), d_st.cpu().numpy()


class UploadBatcher:
    """Host micro-batcher for uploads of ANY kind and size (ucfp_upload_batcher_*): request threads `submit` encoded bytes --
    PNG, JPEG, anything; nothing is announced at creation -- and the library coalesces what the device decodes into one
    copy + decode + ragged hash per flush.  status NEEDS_HOST: decode that upload on the host (`fingerprint_with`)."""

    def __init__(self, *, algo: int = MULTI, max_batch: int = 1024, max_bytes: int = 64 << 20, max_delay_us: int = 0,
                 preprocess: Optional[PreprocessConfig] = None, ctx=None):
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.rec = record_bytes(algo)
        pre = (preprocess or PreprocessConfig())._c()
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_upload_batcher_create(self.ctx.handle, algo, C.byref(pre), max_batch, max_bytes, max_delay_us,
                                                        C.byref(h)))
        self.handle = h

    def submit(self, data: bytes):
        out = (C.c_uint8 * self.rec)()
        st = C.c_int32(0)
        _lib.check(self._lib.ucfp_upload_batcher_submit(self.handle, data, len(data), out, C.byref(st)))
        return bytes(out), int(st.value)

    def stats(self):
        b, i = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_upload_batcher_stats(self.handle, C.byref(b), C.byref(i)))
        return int(b.value), int(i.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_upload_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def JpegBatcher(width: int, height: int, **kw):
    """Micro-batcher for JPEG uploads of one announced geometry (records of the files' luma planes, DESIGN J1)."""
    return PngBatcher(width, height, PIX_GRAY8, jpeg=True, **kw)


# ---- JPEG uploads: luma plane on the device (DESIGN J1) ----

def jpeg_probe(data: bytes):
    """Frame header of a JPEG -> (status, width, height); status 0, NEEDS_HOST (progressive, 12-bit, CMYK ...: the host's
    decoder takes it) or a negative UCFP_E_* (not a JPEG)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    rc = _lib.load().ucfp_jpeg_probe(data, len(data), C.byref(w), C.byref(h))
    return int(rc), int(w.value), int(h.value)


def decode_jpegs(jpgs: Sequence[bytes], width: int, height: int, ctx=None):
    """Decode the LUMA plane of a batch of baseline JPEG files announced as width x height on the GPU.
    -> (frames uint8 [n, h, w], status int32 [n]); frames of files with status != 0 are undefined."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(jpgs)
    d_blob, d_off, total = _upload_pngs(jpgs, dev)
    d_fr = torch.zeros((n, height, width), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().ucfp_image_jpeg_decode_batch_dev(
        ctx.handle, d_blob.data_ptr(), d_off.data_ptr(), n, total, width, height, d_fr.data_ptr(), width, width * height,
        d_st.data_ptr(), torch.cuda.current_stream().cuda_stream or None))
    return d_fr.cpu().numpy(), d_st[:n].cpu().numpy()


def fingerprint_jpegs_dev(jpg_ptr: int, offsets_ptr: int, n: int, jpg_bytes: int, width: int, height: int, *,
                          algo: int = MULTI, exact_ptr: int = 0, out_ptr: int, status_ptr: int = 0, stream: int = 0,
                          preprocess: Optional[PreprocessConfig] = None, ctx=None) -> None:
    """Device-resident encoded files -> records; raw device addresses, no sync (ucfp_image_jpeg_hash_batch_dev)."""
    ctx = ctx or _lib.current_context()
    pre = (preprocess or PreprocessConfig())._c()
    _lib.check(_lib.load().ucfp_image_jpeg_hash_batch_dev(
        ctx.handle, algo, jpg_ptr, offsets_ptr, n, jpg_bytes, width, height, C.byref(pre), exact_ptr or None, out_ptr,
        status_ptr or None, stream or None))


def fingerprint_jpegs(jpgs: Sequence[bytes], width: int, height: int, *, algo: int = MULTI,
                      exact: Optional[np.ndarray] = None, preprocess: Optional[PreprocessConfig] = None, ctx=None):
    """Host convenience: encoded JPEG files -> (records uint8 [n, record_bytes], status int32 [n])."""
    import torch
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(jpgs)
    rec = record_bytes(algo)
    d_blob, d_off, total = _upload_pngs(jpgs, dev)
    d_out = torch.zeros((max(n, 1), rec), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    d_ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        if ex.shape != (n, 32):
            raise ModalityError("exact must be [n, 32] bytes")
        d_ex = torch.from_numpy(ex).to(dev)
    fingerprint_jpegs_dev(d_blob.data_ptr(), d_off.data_ptr(), n, total, width, height, algo=algo,
                          exact_ptr=d_ex.data_ptr() if d_ex is not None else 0, out_ptr=d_out.data_ptr(),
                          status_ptr=d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                          preprocess=preprocess, ctx=ctx)
    return d_out[:n].cpu().numpy(), d_st[:n].cpu().numpy()


class PngBatcher:
    """Host micro-batcher for encoded uploads (SURVEY 8f N1 + N4): request threads `submit` PNG bytes of one announced
    geometry (or, with jpeg=True, JPEG bytes: `JpegBatcher`); the library copies them to the device together and decodes,
    BLAKE3-hashes and fingerprints them there."""

    def __init__(self, width: int, height: int, pixfmt: int = PIX_RGB8, *, algo: int = MULTI, max_batch: int = 1024,
                 max_bytes: int = 256 << 20, max_delay_us: int = 0, preprocess: Optional[PreprocessConfig] = None, ctx=None,
                 jpeg: bool = False):
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.rec = record_bytes(algo)
        pre = (preprocess or PreprocessConfig())._c()
        h = C.c_void_p()
        if jpeg:
            _lib.check(self._lib.ucfp_jpeg_batcher_create(self.ctx.handle, algo, width, height, C.byref(pre), max_batch,
                                                          max_bytes, max_delay_us, C.byref(h)))
        else:
            _lib.check(self._lib.ucfp_png_batcher_create(self.ctx.handle, algo, width, height, pixfmt, C.byref(pre), max_batch,
                                                         max_bytes, max_delay_us, C.byref(h)))
        self.handle = h

    def submit(self, png: bytes):
        """-> (record bytes, status); status NEEDS_HOST: decode this upload on the host (`fingerprint_with`)."""
        out = (C.c_uint8 * self.rec)()
        st = C.c_int32(0)
        _lib.check(self._lib.ucfp_png_batcher_submit(self.handle, png, len(png), out, C.byref(st)))
        return bytes(out), int(st.value)

    def stats(self):
        b, i = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_png_batcher_stats(self.handle, C.byref(b), C.byref(i)))
        return int(b.value), int(i.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_png_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# per-request adapters (encoded bytes in, Record out) -- the reference's call shape
# ------------------------------------------------------------------------------------------

def _decode(data: bytes, pre: PreprocessConfig):
    """Encoded image -> (array, pixfmt). Errors map to Error::Modality like image.rs:70."""
    if len(data) > pre.max_input_bytes:
        raise ModalityError(f"image payload {len(data)} B exceeds max_input_bytes {pre.max_input_bytes}")
    try:
        from PIL import Image  # decode only; not part of the hashing path
        img = Image.open(io.BytesIO(data))
        if img.format == "JPEG" and img.mode in ("RGB", "L"):
            # DESIGN J1: of a JPEG the LUMA component is what gets hashed -- libjpeg decodes just that plane (draft "L" =
            # out_color_space JCS_GRAYSCALE), exactly what the device front end (jpeg.hip) produces
            img.draft("L", img.size)
        img.load()
    except Exception as e:  # noqa: BLE001 - any decoder failure is a modality error
        raise ModalityError(f"image decode: {e}") from None
    if img.mode == "L":
        return np.asarray(img, dtype=np.uint8), PIX_GRAY8
    if img.mode == "RGBA":
        return np.asarray(img, dtype=np.uint8), PIX_RGBA8
    return np.asarray(img.convert("RGB"), dtype=np.uint8), PIX_RGB8


def _exact_digest(data: bytes) -> np.ndarray:
    from .blake3 import blake3_digest
    return np.frombuffer(blake3_digest(data), np.uint8).reshape(1, 32)


def _single(data: bytes, pre: PreprocessConfig, algo: int, tenant_id: int, record_id: int) -> Record:
    arr, pixfmt = _decode(data, pre)
    recs, status = fingerprint_frames(arr[None], algo=algo, pixfmt=pixfmt,
                                      exact=_exact_digest(data), preprocess=pre)
    if status[0] != 0:
        h, w = arr.shape[:2]
        raise ModalityError(
            f"image {w}x{h} outside [{pre.min_dimension}, {pre.max_dimension}] px")
    return Record(tenant_id=tenant_id, record_id=record_id, modality=Modality.Image,
                  format_version=FORMAT_VERSION, algorithm=_TAG[algo], config_hash=0,
                  fingerprint=recs[0].tobytes(), embedding=None, model_id=None, metadata=b"",
                  text=None)


def fingerprint(data: bytes, tenant_id: int, record_id: int) -> Record:
    return fingerprint_with(data, tenant_id, record_id, PreprocessConfig())


def fingerprint_with(data: bytes, tenant_id: int, record_id: int,
                     preprocess: PreprocessConfig) -> Record:
    return _single(data, preprocess, MULTI, tenant_id, record_id)


def fingerprint_multi_with(data: bytes, preprocess: PreprocessConfig, _multi_cfg, tenant_id: int,
                           record_id: int) -> Record:
    """image.rs:96-104: the MultiHashConfig is a compare-time setting and does not change the bytes."""
    return fingerprint_with(data, tenant_id, record_id, preprocess)


def fingerprint_phash(data: bytes, preprocess: PreprocessConfig, tenant_id: int, record_id: int) -> Record:
    return _single(data, preprocess, PHASH, tenant_id, record_id)


def fingerprint_dhash(data: bytes, preprocess: PreprocessConfig, tenant_id: int, record_id: int) -> Record:
    return _single(data, preprocess, DHASH, tenant_id, record_id)


def fingerprint_ahash(data: bytes, preprocess: PreprocessConfig, tenant_id: int, record_id: int) -> Record:
    return _single(data, preprocess, AHASH, tenant_id, record_id)


def global_hashes(record_bytes_: bytes) -> dict:
    """Extract the 64-bit global hashes from a 168-B or 536-B record (SURVEY 8f N2 offsets)."""
    b = bytes(record_bytes_)
    rd = lambda off: int.from_bytes(b[off:off + 8], "little")  # noqa: E731
    if len(b) == 536:
        return {"ahash": rd(32 + 32), "phash": rd(32 + 168 + 32), "dhash": rd(32 + 336 + 32)}
    if len(b) == 168:
        return {"global": rd(32)}
    raise ModalityError(f"not an image fingerprint: {len(b)} bytes")


class ImageBatcher:
    """Host micro-batcher: many request threads call `submit` concurrently, the
    library coalesces them into one GPU launch. `submit` blocks until this frame's record is ready."""

    def __init__(self, width: int, height: int, *, algo: int = MULTI, pixfmt: int = PIX_GRAY8,
                 max_batch: int = 512, max_delay_us: int = 200, preprocess: Optional[PreprocessConfig] = None,
                 ctx=None):
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.width, self.height, self.algo, self.pixfmt = width, height, algo, pixfmt
        self.rec = record_bytes(algo)
        pre = (preprocess or PreprocessConfig())._c()
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_image_batcher_create(self.ctx.handle, algo, width, height, pixfmt, C.byref(pre),
                                                       max_batch, max_delay_us, C.byref(h)))
        self.handle = h

    def submit(self, frame: np.ndarray, exact: Optional[bytes] = None):
        """frame: uint8 [h, w] or [h, w, c], C-contiguous. Returns (record bytes, status)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        out = (C.c_uint8 * self.rec)()
        st = C.c_int32(0)
        ex = (C.c_uint8 * 32).from_buffer_copy(exact) if exact is not None else None
        _lib.check(self._lib.ucfp_image_batcher_submit(self.handle, frame.ctypes.data,
                                                       self.width * _BPP[self.pixfmt], ex, out, C.byref(st)))
        return bytes(out), int(st.value)

    def stats(self):
        b, i = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_image_batcher_stats(self.handle, C.byref(b), C.byref(i)))
        return int(b.value), int(i.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_image_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
