"""Model-feed hand-off: keep the [N, L] int32 outputs in HBM and give them to a framework
without a copy.

The reference returns Python lists which its users stack into the arrays of `DataCollection`
(models/bert/dataset.py:7-28: input_ids, attention_mask, token_type_ids, ...).  `Tokenize.encode_to_device` returns
the same field names as `DeviceArray`s: buffers owned by this library that speak DLPack (`__dlpack__`,
`__dlpack_device__`, device type kDLROCM), so `torch.from_dlpack(out["input_ids"])` is zero-copy on a ROCm build,
and `.numpy()` copies to the host.  No torch import here.

The DLManagedTensor and its deleter live in the C library (gz_block_*): a consumer may drop its tensor while the
interpreter is shutting down, when no Python callback can run any more.  The HBM allocation is refcounted there and
freed when the DeviceArray and every tensor made from it are gone.

Process set-up when the consumer is PyTorch: let torch touch the GPU first (`torch.zeros(1, device="cuda")`) and only
then create a `Tokenize`; both resolve the HIP runtime by soname and the first one loaded serves both.
"""
import ctypes as C

import numpy as np

kDLROCM = 10
_kDLInt, _kDLUInt = 0, 1

_api = C.pythonapi
_api.PyCapsule_New.restype = C.py_object
_api.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]


class DeviceArray:
    """A C-contiguous array in HBM; owns (a reference to) its allocation."""

    def __init__(self, ctx, ptr: int, shape, dtype=np.int32):
        self._ctx, self.ptr, self.shape, self.dtype = ctx, int(ptr), tuple(int(x) for x in shape), np.dtype(dtype)
        blk = C.c_void_p()
        ctx._check(ctx.lib.gz_block_create(ctx.handle, C.c_void_p(self.ptr), C.byref(blk)))
        self._block = blk

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            self._ctx.d2h(out, self.ptr)
        return out

    def __dlpack_device__(self):
        return (kDLROCM, self._ctx.device)

    def __dlpack__(self, stream=None, **_):
        # the producing call was synchronised (gz_sync) before this object existed: nothing to wait for on `stream`
        if not self._block:
            raise RuntimeError("DeviceArray has been freed")
        shape = (C.c_int64 * max(1, len(self.shape)))(*self.shape)
        code = _kDLInt if self.dtype.kind == "i" else _kDLUInt
        mt = self._ctx.lib.gz_block_dlpack(self._block, len(self.shape), C.cast(shape, C.c_void_p), code, self.dtype.itemsize * 8)
        if not mt:
            raise MemoryError("gz_block_dlpack failed")
        # the capsule's destructor lives in the C library (a Python callback could run during interpreter shutdown):
        # a capsule that is never consumed gives its reference back, a consumed one is left to the consumer's deleter
        return _api.PyCapsule_New(mt, b"dltensor", C.cast(self._ctx.lib.gz_dlpack_capsule_destructor, C.c_void_p))

    def free(self):
        """Drop this object's reference; tensors already exported keep the memory alive."""
        if getattr(self, "_block", None):
            self._ctx.lib.gz_block_release(self._block)
            self._block = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass
