"""Error taxonomy mirroring the reference's `Error` enum (src/error.rs:9-61) and its HTTP
mapping (src/server/error.rs:22-41)."""


class UcfpError(Exception):
    """Base class; `http_status` is what the reference's ApiError would answer."""
    http_status = 500
    code = 0


class ModalityError(UcfpError):      # Error::Modality(String) -> 400
    http_status = 400
    code = -1


class UnsupportedError(UcfpError):   # Error::Unsupported -> 501
    http_status = 501
    code = -2


class IndexError_(UcfpError):        # Error::Index(String) -> 500
    http_status = 500
    code = -3


class InvalidArgument(UcfpError):    # caller bug at the FFI boundary
    http_status = 500
    code = -4


class RecordNotFound(UcfpError):     # Error::RecordNotFound -> 404
    http_status = 404
    code = -5


_BY_CODE = {c.code: c for c in (ModalityError, UnsupportedError, IndexError_, InvalidArgument,
                                RecordNotFound)}


def from_status(code: int, message: str) -> UcfpError:
    return _BY_CODE.get(code, UcfpError)(message)
