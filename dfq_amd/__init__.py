"""dfq_amd -- MI355X-native DFQ calibration engine (weight equalization, bias correction, fake-quant).

Drop-in for the calibration hot path of jakc4103/DFQ: the module layout mirrors the reference
(``dfq``, ``improve_dfq``, ``utils.quantize``, ``utils.layer_transform``, ``utils.relation``) so that
``main_cls.py`` only has to change its import prefix.  All arithmetic runs in hand-written HIP
kernels (gfx950) behind the C ABI declared in ``include/dfq_hip.h``.
"""
__version__ = '0.1.0'


def staging():
    """``with dfq_amd.staging(): ...`` -- one staging area for a sequence of entry-point calls on a CPU-resident model: every
    tensor crosses PCIe once each way for the whole sequence (see ``_ffi.staging``)."""
    from . import _ffi
    return _ffi.staging()


def release_staging():
    """Drop the device copies (and the references to the caller's tensors) that the drop-in entry points keep of the last
    CPU-resident model they calibrated, so that consecutive plain calls transfer it once (``_ffi.persistent_stage``)."""
    from . import _ffi
    _ffi.release_staging()


def pool_trim():
    """Return the device blocks that destroyed plans parked in the library's free lists to the driver (``dfq_pool_trim``);
    the number of bytes released.  For a process that needs the memory for something else: torch's allocator cannot see them."""
    from . import _ffi
    return int(_ffi.lib().dfq_pool_trim())
