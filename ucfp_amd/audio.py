"""Audio fingerprinting -- host-side mirror of src/modality/audio.rs.

    fingerprint_wang(samples, sample_rate, tenant_id, record_id)              audio.rs:46-60
    fingerprint_wang_with(samples, sample_rate, cfg, tenant_id, record_id)    audio.rs:64-98
    fingerprint_haitsma / fingerprint_haitsma_with                            audio.rs:164-224
    StreamingWangSession(sample_rate, tenant_id, record_id).push/.finalize    audio.rs:414-480

All DSP runs in the HIP library through the C ABI (ucfp_audio_*); this module validates arguments
the way the reference does and wraps bytes into `Record`s.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib
from .core import Modality, Record
from .errors import ModalityError

ALGORITHM_WANG = "audiofp-wang-v1"
ALGORITHM_HAITSMA = "audiofp-haitsma-v1"
WANG_SR, HAITSMA_SR = 8000, 5000


@dataclass
class WangConfig:
    """audiofp::classical::WangConfig; defaults src/server/algorithms_manifest.rs:553-592."""
    fan_out: int = 10
    target_zone_t: int = 63
    target_zone_f: int = 64
    peaks_per_sec: int = 30
    min_anchor_mag_db: float = -50.0

    def _c(self):
        return _lib.WangConfig(self.fan_out, self.target_zone_t, self.target_zone_f, self.peaks_per_sec,
                               self.min_anchor_mag_db)


@dataclass
class HaitsmaConfig:
    """audiofp::classical::HaitsmaConfig; defaults manifest :655-672."""
    fmin: float = 300.0
    fmax: float = 2000.0

    def _c(self):
        return _lib.HaitsmaConfig(self.fmin, self.fmax)


def _check_rate(sample_rate: int):
    if not (0 < int(sample_rate) <= 384_000):
        raise ModalityError(f"invalid sample rate {sample_rate}")   # audio.rs:74-75


def wang_hashes(samples, sample_rate: int, cfg: Optional[WangConfig] = None, ctx=None) -> np.ndarray:
    """-> uint32 [n, 2]: (packed hash, t_anchor) -- the byte image of audiofp's [WangHash]."""
    ctx = ctx or _lib.current_context()
    _check_rate(sample_rate)
    x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    c = (cfg or WangConfig())._c()
    lib = _lib.load()
    cap = max(1, int(lib.ucfp_audio_wang_max_hashes(x.size, C.byref(c))))
    out = np.zeros((cap, 2), np.uint32)
    n = C.c_size_t(0)
    _lib.check(lib.ucfp_audio_wang(ctx.handle, x.ctypes.data, x.size, sample_rate, C.byref(c), out.ctypes.data,
                                   cap, C.byref(n)))
    return out[: n.value].copy()


def wang_hashes_batch_dev(pcm_ptr: int, offsets_ptr: int, n_total: int, n_clips: int, sample_rate: int, out_ptr: int,
                          cap_hashes: int, out_offsets_ptr: int, cfg: Optional[WangConfig] = None, stream: int = 0,
                          ctx=None) -> None:
    """Ragged batch of clips, device pointers, no sync (ucfp_audio_wang_batch_dev): clip i = pcm[offsets[i] ..
    offsets[i+1]) at `sample_rate` (resampled to 8 kHz in the kernel unless it is 8000)."""
    ctx = ctx or _lib.current_context()
    c = (cfg or WangConfig())._c()
    _lib.check(_lib.load().ucfp_audio_wang_batch_dev(ctx.handle, pcm_ptr or None, offsets_ptr or None, n_total, n_clips,
                                                     sample_rate, C.byref(c), out_ptr or None, cap_hashes,
                                                     out_offsets_ptr, stream or None))


def wang_hashes_batch(clips, sample_rate: int, cfg: Optional[WangConfig] = None, ctx=None) -> List[np.ndarray]:
    """Host convenience over the batch entry: a list of mono f32 clips (all at `sample_rate`) -> one uint32 [n_i, 2]
    array per clip.  One launch sequence for the whole batch."""
    import torch
    ctx = ctx or _lib.current_context()
    _check_rate(sample_rate)
    arrs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in clips]
    if not arrs:
        return []
    offs = np.zeros(len(arrs) + 1, np.uint64)
    np.cumsum([a.size for a in arrs], out=offs[1:])
    blob = np.concatenate(arrs) if offs[-1] else np.zeros(1, np.float32)
    c = (cfg or WangConfig())._c()
    cap = max(1, int(_lib.load().ucfp_audio_wang_batch_max_hashes(int(offs[-1]), len(arrs), sample_rate, C.byref(c))))
    dev = f"cuda:{ctx.device}"
    d_pcm = torch.from_numpy(blob).to(dev)
    d_off = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
    d_oo = torch.zeros(len(arrs) + 1, dtype=torch.int64, device=dev)
    wang_hashes_batch_dev(d_pcm.data_ptr(), d_off.data_ptr(), int(offs[-1]), len(arrs), sample_rate, d_out.data_ptr(), cap,
                          d_oo.data_ptr(), cfg, torch.cuda.current_stream().cuda_stream, ctx)
    oo = d_oo.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32)
    if oo[-1] > cap:
        raise ModalityError(f"Wang batch produced {oo[-1]} hashes, buffer holds {cap}")
    return [out[oo[i]:oo[i + 1]].copy() for i in range(len(arrs))]


class WangBatcher:
    """Host micro-batcher for clips (handlers.rs:704-918 fingerprints one clip per request): concurrent
    `submit` calls become one ucfp_audio_wang_batch_dev launch sequence.  All clips at `sample_rate`."""

    def __init__(self, sample_rate: int = WANG_SR, cfg: Optional[WangConfig] = None, *, max_batch: int = 1024,
                 max_samples: int = 64 << 20, max_delay_us: int = 500, ctx=None):
        _check_rate(sample_rate)
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.sample_rate = sample_rate
        self._cfg = (cfg or WangConfig())._c()
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_audio_batcher_create(self.ctx.handle, sample_rate, C.byref(self._cfg), max_batch,
                                                       max_samples, max_delay_us, C.byref(h)))
        self.handle = h

    def submit(self, samples) -> np.ndarray:
        """-> uint32 [n, 2] hashes of this clip.  Blocks until they are ready."""
        x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        cap = max(1, int(self._lib.ucfp_audio_wang_batch_max_hashes(x.size, 1, self.sample_rate, C.byref(self._cfg))))
        out = np.zeros((cap, 2), np.uint32)
        n = C.c_size_t(0)
        _lib.check(self._lib.ucfp_audio_batcher_submit(self.handle, x.ctypes.data, x.size, out.ctypes.data, cap,
                                                       C.byref(n)))
        return out[: n.value].copy()

    def stats(self):
        b, i = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.ucfp_audio_batcher_stats(self.handle, C.byref(b), C.byref(i)))
        return int(b.value), int(i.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_audio_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def haitsma_frames(samples, sample_rate: int, cfg: Optional[HaitsmaConfig] = None, ctx=None) -> np.ndarray:
    ctx = ctx or _lib.current_context()
    _check_rate(sample_rate)
    x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    c = (cfg or HaitsmaConfig())._c()
    lib = _lib.load()
    cap = max(1, int(lib.ucfp_audio_haitsma_frames(x.size, sample_rate)))
    out = np.zeros(cap, np.uint32)
    n = C.c_size_t(0)
    _lib.check(lib.ucfp_audio_haitsma(ctx.handle, x.ctypes.data, x.size, sample_rate, C.byref(c), out.ctypes.data,
                                      cap, C.byref(n)))
    return out[: n.value].copy()


def haitsma_frames_batch(clips, sample_rate: int, cfg: Optional[HaitsmaConfig] = None, ctx=None) -> List[np.ndarray]:
    """A list of mono f32 clips (all at `sample_rate`) -> one uint32 [frames_i] array per clip, one launch sequence for
    the whole batch (ucfp_audio_haitsma_batch_dev)."""
    import torch
    ctx = ctx or _lib.current_context()
    _check_rate(sample_rate)
    arrs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in clips]
    if not arrs:
        return []
    offs = np.zeros(len(arrs) + 1, np.uint64)
    np.cumsum([a.size for a in arrs], out=offs[1:])
    blob = np.concatenate(arrs) if offs[-1] else np.zeros(1, np.float32)
    c = (cfg or HaitsmaConfig())._c()
    lib = _lib.load()
    cap = max(1, int(lib.ucfp_audio_haitsma_batch_max_frames(int(offs[-1]), len(arrs), sample_rate)))
    dev = f"cuda:{ctx.device}"
    d_pcm = torch.from_numpy(blob).to(dev)
    d_off = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_oo = torch.zeros(len(arrs) + 1, dtype=torch.int64, device=dev)
    _lib.check(lib.ucfp_audio_haitsma_batch_dev(ctx.handle, d_pcm.data_ptr(), d_off.data_ptr(), int(offs[-1]), len(arrs),
                                                sample_rate, C.byref(c), d_out.data_ptr(), cap, d_oo.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream or None))
    oo = d_oo.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32)
    return [out[oo[i]:oo[i + 1]].copy() for i in range(len(arrs))]


def _record(algo: str, payload: bytes, tenant_id: int, record_id: int) -> Record:
    # format_version 1, config_hash 0: audio.rs:89-91
    return Record(tenant_id=tenant_id, record_id=record_id, modality=Modality.Audio, format_version=1,
                  algorithm=algo, config_hash=0, fingerprint=payload, embedding=None, model_id=None, metadata=b"",
                  text=None)


def fingerprint_wang(samples, sample_rate: int, tenant_id: int, record_id: int) -> Record:
    return fingerprint_wang_with(samples, sample_rate, WangConfig(), tenant_id, record_id)


def fingerprint_wang_with(samples, sample_rate: int, cfg: WangConfig, tenant_id: int, record_id: int) -> Record:
    return _record(ALGORITHM_WANG, wang_hashes(samples, sample_rate, cfg).tobytes(), tenant_id, record_id)


def fingerprint_haitsma(samples, sample_rate: int, tenant_id: int, record_id: int) -> Record:
    return fingerprint_haitsma_with(samples, sample_rate, HaitsmaConfig(), tenant_id, record_id)


def fingerprint_haitsma_with(samples, sample_rate: int, cfg: HaitsmaConfig, tenant_id: int, record_id: int) -> Record:
    return _record(ALGORITHM_HAITSMA, haitsma_frames(samples, sample_rate, cfg).tobytes(), tenant_id, record_id)


class StreamingWangSession:
    """Push/finalize wrapper (audio.rs:414-480). Hashes are a function of the whole signal (per-second
    peak caps, forward target zones), so the session buffers PCM and emits at `finalize`; `push`
    returns no records -- allowed by the reference contract ("typically zero or one")."""

    def __init__(self, sample_rate: int, tenant_id: int, record_id: int):
        if sample_rate != WANG_SR:
            raise ModalityError(f"Wang requires 8 kHz mono input (got {sample_rate} Hz); resample upstream")
        self._chunks: List[np.ndarray] = []
        self.tenant_id, self.record_id = tenant_id, record_id

    def push(self, samples) -> List[Record]:
        self._chunks.append(np.asarray(samples, dtype=np.float32).reshape(-1))
        return []

    def finalize(self) -> List[Record]:
        if not self._chunks:
            return []
        x = np.concatenate(self._chunks)
        self._chunks = []
        h = wang_hashes(x, WANG_SR)
        if h.shape[0] == 0:
            return []
        return [_record(ALGORITHM_WANG, h.tobytes(), self.tenant_id, self.record_id)]
