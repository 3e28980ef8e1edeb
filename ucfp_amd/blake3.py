"""BLAKE3-256 of a host buffer via the C ABI (`ucfp_blake3`) -- the `exact` field of image records."""
import ctypes as C

from . import _lib


def blake3_digest(data: bytes) -> bytes:
    data = bytes(data)
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
    _lib.check(_lib.load().ucfp_blake3(buf, len(data), out))
    return bytes(out)


def blake3_batch(items, ctx=None):
    """BLAKE3 of every byte string in `items` on the GPU (ucfp_blake3_batch_dev) -> uint8 [n, 32]."""
    import ctypes as C

    import numpy as np
    import torch

    from . import _lib
    ctx = ctx or _lib.current_context()
    dev = f"cuda:{ctx.device}"
    n = len(items)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum([len(b) for b in items], out=offs[1:])
    blob = np.frombuffer(b"".join(bytes(b) for b in items) + b"\0" * 16, np.uint8).copy()
    d_blob, d_off = torch.from_numpy(blob).to(dev), torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((max(n, 1), 32), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().ucfp_blake3_batch_dev(ctx.handle, d_blob.data_ptr(), d_off.data_ptr(), n, int(offs[-1]),
                                                 d_out.data_ptr(), torch.cuda.current_stream().cuda_stream or None))
    return d_out[:n].cpu().numpy()
