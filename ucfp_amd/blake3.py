"""BLAKE3-256 of a host buffer via the C ABI (`ucfp_blake3`) -- the `exact` field of image records."""
import ctypes as C

from . import _lib


def blake3_digest(data: bytes) -> bytes:
    data = bytes(data)
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
    _lib.check(_lib.load().ucfp_blake3(buf, len(data), out))
    return bytes(out)
