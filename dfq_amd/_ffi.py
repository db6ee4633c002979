"""ctypes binding of ``libdfq_hip.so`` (C ABI: ``include/dfq_hip.h``) and the device-buffer staging
used by every engine call.

There is no CPU path: if the library or a ROCm GPU is missing, the first call raises.  PyTorch is
used only to own device memory and the HIP stream the kernels are enqueued on.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64,
                    c_size_t, c_void_p)

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFQ_HIP_LIB points at another build of the same library (kernel-tuning experiments); still no fallback
LIB_PATH = os.environ.get('DFQ_HIP_LIB') or os.path.join(_HERE, 'libdfq_hip.so')

c_float_p = c_void_p      # device pointers travel as opaque addresses
c_int32_dp = c_void_p
c_uint32_dp = c_void_p


# ---- structs of include/dfq_hip.h ---------------------------------------------------------------
class DfqLayer(Structure):
    _fields_ = [('weight', c_void_p), ('bias', c_void_p), ('out_ch', c_int32),
                ('in_per_group', c_int32), ('khkw', c_int32), ('groups', c_int32)]


class DfqRelation(Structure):
    _fields_ = [('first', c_int32), ('second', c_int32), ('bn_weight', c_void_p),
                ('bn_bias', c_void_p), ('scale_cum', c_void_p)]


class DfqLeConfig(Structure):
    _fields_ = [('s_lo', c_float), ('s_hi', c_float), ('inv_lo', c_float), ('inv_hi', c_float),
                ('hi_gt_lo', c_int32), ('eps', c_float), ('signed_range', c_int32),
                ('converge_thres', c_double), ('converge_count', c_int32), ('max_sweeps', c_int32)]


class DfqLeResult(Structure):
    _fields_ = [('sweeps', c_int32), ('stall_count', c_int32), ('diff', c_double),
                ('last_diff_tmp', c_double)]


class DfqSegment(Structure):
    _fields_ = [('data', c_void_p), ('n', c_int64), ('num_bits', c_int32), ('symmetric', c_int32),
                ('codes', c_void_p)]


class DfqBcSource(Structure):
    _fields_ = [('fake_weight', c_void_p), ('fake_bias', c_void_p), ('channels', c_int32),
                ('relu', c_int32), ('concat', c_int32)]


class DfqBcStep(Structure):
    _fields_ = [('layer', c_int32), ('source_begin', c_int32), ('source_count', c_int32),
                ('next_bn_bias', c_void_p), ('net', c_int32), ('reserved', c_int32)]


# every exported symbol: name -> (restype, argtypes).  tests/test_abi.py checks this table against
# include/dfq_hip.h and against the symbols the shared object really exports.
class DfqRebuildItem(Structure):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('s_out', c_void_p), ('s_in', c_void_p), ('rows', c_int32),
                ('cols', c_int32), ('khkw', c_int32), ('groups', c_int32), ('in_reciprocal', c_int32), ('reserved', c_int32)]


class DfqBnRangeReq(Structure):
    _fields_ = [('fake_weight', c_void_p), ('fake_bias', c_void_p), ('channels', c_int32), ('relu_mode', c_int32)]


SIGNATURES = {
    'dfq_version': (c_int32, []),
    'dfq_last_error': (c_char_p, []),
    'dfq_pool_trim': (ctypes.c_longlong, []),
    'dfq_device_count': (c_int32, []),
    'dfq_le_plan_create': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(DfqRelation), c_int32, POINTER(c_void_p)]),
    'dfq_le_plan_create_batch': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(c_int32), c_int32, POINTER(DfqRelation), c_int32,
                                           POINTER(c_void_p)]),
    'dfq_le_plan_create_replicated': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(DfqRelation), c_int32, POINTER(c_void_p), c_int32,
                                                POINTER(c_void_p)]),
    'dfq_le_plan_destroy': (None, [c_void_p]),
    'dfq_le_plan_nets': (c_int32, [c_void_p]),
    'dfq_le_query_all': (c_int32, [c_void_p, c_void_p, POINTER(DfqLeResult), POINTER(c_int32)]),
    'dfq_le_plan_resident_tiles': (c_int32, [c_void_p]),
    'dfq_le_plan_resident_reason': (ctypes.c_char_p, [c_void_p]),
    'dfq_le_plan_degraded': (c_int32, [c_void_p]),
    'dfq_le_resident_stats': (c_int32, [c_void_p, c_void_p, POINTER(c_int64)]),
    'dfq_le_resident_trace_words': (c_int64, [c_void_p]),
    'dfq_le_resident_trace': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_int32, c_void_p, c_void_p, c_int64]),
    'dfq_le_plan_uniform': (c_int32, [c_void_p]),
    'dfq_le_plan_levels': (c_int32, [c_void_p]),
    'dfq_le_plan_paired_elements': (c_int64, [c_void_p]),
    'dfq_le_plan_depth': (c_int32, [c_void_p]),
    'dfq_le_plan_rw_elements': (c_int64, [c_void_p]),
    'dfq_le_plan_ro_elements': (c_int64, [c_void_p]),
    'dfq_le_plan_deferred_elements': (c_int64, [c_void_p]),
    'dfq_le_plan_defer_depth': (c_int32, [c_void_p]),
    'dfq_le_plan_free_running_elements': (c_int64, [c_void_p]),
    'dfq_le_plan_free_running_group': (c_int32, [c_void_p]),
    'dfq_le_plan_lean_background': (c_int32, [c_void_p]),
    'dfq_le_plan_lean_tiles': (c_int32, [c_void_p]),
    'dfq_le_plan_lean_info': (c_int32, [c_void_p, c_int32, POINTER(c_int64)]),
    'dfq_le_plan_level_launches': (c_int32, [c_void_p, c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
    'dfq_le_plan_level_grid': (c_int32, [c_void_p, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    'dfq_le_enqueue': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_int32, c_int32, c_void_p]),
    'dfq_le_query': (c_int32, [c_void_p, c_void_p, POINTER(DfqLeResult), POINTER(c_int32)]),
    'dfq_le_run': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_void_p, POINTER(DfqLeResult)]),
    'dfq_le_plan_has_waits': (c_int32, [c_void_p]),
    'dfq_le_plan_set_safe_mode': (c_int32, [c_void_p]),
    'dfq_bc_plan_has_waits': (c_int32, [c_void_p]),
    'dfq_bc_plan_set_safe_mode': (c_int32, [c_void_p]),
    'dfq_le_set_diff_log': (c_int32, [c_void_p, c_void_p, c_int32]),
    'dfq_le_shared_verdict': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_double, c_int32, c_int32, c_void_p]),
    'dfq_le_profile': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_int32, c_void_p, POINTER(c_double), POINTER(c_double),
                                 POINTER(c_int32), POINTER(c_double), POINTER(c_double), POINTER(c_int32)]),
    'dfq_le_trace': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_int32, c_int32, c_void_p, POINTER(c_int64)]),
    'dfq_le_plan_sweep_workgroups': (c_int32, [c_void_p]),
    'dfq_le_plan_block_info': (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_int64)]),
    'dfq_le_trace_blocks': (c_int32, [c_void_p, POINTER(DfqLeConfig), c_int32, c_void_p, POINTER(c_int64), c_int64]),
    'dfq_tensor_minmax': (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'dfq_fake_quant': (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_double, c_double,
                                 c_void_p, c_void_p, c_void_p]),
    'dfq_sample_minmax_mean': (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dfq_quant_measure': (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    'dfq_quant_measure_fused_grid': (c_int32, [c_int32, c_int64]),
    'dfq_quant_measure_fused': (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    'dfq_quant_measure_fused_status': (c_int32, [c_void_p, c_int32, c_void_p]),
    'dfq_quant_plan_create': (c_int32, [POINTER(DfqSegment), c_int32, POINTER(c_void_p)]),
    'dfq_quant_plan_destroy': (None, [c_void_p]),
    'dfq_quant_plan_run': (c_int32, [c_void_p, c_void_p]),
    'dfq_quant_plan_measure': (c_int32, [c_void_p, c_void_p]),
    'dfq_quant_plan_minmax': (c_void_p, [c_void_p]),
    'dfq_bc_plan_create': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(DfqBcStep), c_int32,
                                     POINTER(DfqBcSource), c_int32, POINTER(c_void_p)]),
    'dfq_bc_plan_create_replicated': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(DfqBcStep), c_int32, POINTER(DfqBcSource), c_int32,
                                                POINTER(c_void_p), c_int32, POINTER(c_void_p)]),
    'dfq_bc_plan_destroy': (None, [c_void_p]),
    'dfq_bc_plan_run': (c_int32, [c_void_p, c_int32, c_void_p]),
    'dfq_bc_plan_status': (c_int32, [c_void_p, c_void_p]),
    'dfq_bc_plan_eps': (c_void_p, [c_void_p, c_int32]),
    'dfq_bc_plan_correction': (c_void_p, [c_void_p, c_int32]),
    'dfq_bc_plan_weight_elements': (c_int64, [c_void_p]),
    'dfq_bc_plan_tagged': (c_int32, [c_void_p]),
    'dfq_bc_plan_last_run_tagged': (c_int32, [c_void_p]),
    'dfq_bc_plan_one_launch': (c_int32, [c_void_p]),
    'dfq_bc_debug_trace': (c_int64, [c_void_p, c_int64]),
    'dfq_bc_plan_eps_elements': (c_int64, [c_void_p]),
    'dfq_bc_plan_folded': (c_int32, [c_void_p]),
    'dfq_bc_plan_chain_steps': (c_int32, [c_void_p]),
    'dfq_quant_error_scratch_bytes': (c_size_t, [c_int64, c_int64]),
    'dfq_quant_error': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'dfq_scale_rows': (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_int32, c_void_p]),
    'dfq_scale_cols': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    'dfq_vec_op': (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    'dfq_le_lazy_plan_create': (c_int32, [POINTER(DfqLayer), c_int32, POINTER(c_int32), c_int32, POINTER(DfqRelation), c_int32,
                                          POINTER(c_void_p)]),
    'dfq_le_lazy_plan_destroy': (None, [c_void_p]),
    'dfq_le_lazy_run': (c_int32, [c_void_p, POINTER(DfqLeConfig), POINTER(c_int32), c_void_p]),
    'dfq_le_lazy_plan_levels': (c_int32, [c_void_p]),
    'dfq_le_lazy_plan_paired_elements': (c_int64, [c_void_p]),
    'dfq_le_lazy_plan_weight_elements': (c_int64, [c_void_p]),
    'dfq_le_lazy_plan_sweep_elements': (c_int64, [c_void_p]),
    'dfq_rebuild_plan_create': (c_int32, [POINTER(DfqRebuildItem), c_int32, POINTER(c_void_p)]),
    'dfq_rebuild_plan_destroy': (None, [c_void_p]),
    'dfq_rebuild_plan_elements': (c_int64, [c_void_p]),
    'dfq_rebuild_plan_run': (c_int32, [c_void_p, c_void_p]),
    'dfq_clamp': (c_int32, [c_void_p, c_int64, c_float, c_float, c_void_p]),
    'dfq_fold_batchnorm': (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p, c_void_p, c_void_p]),
    'dfq_bias_absorb': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_float, c_void_p]),
    'dfq_row_range': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    'dfq_col_range': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'dfq_le_solve': (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    'dfq_le_apply': (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'dfq_le_pair': (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                              c_void_p, c_double, c_double, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    'dfq_absdiff_mean_scratch_bytes': (c_size_t, [c_int64]),
    'dfq_absdiff_mean': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'dfq_fake_quant_rows': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int32, c_int32,
                                      c_void_p, c_void_p, c_void_p]),
    'dfq_zeroq_quant_rows': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                       c_void_p]),
    'dfq_grouped_matvec': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'dfq_bn_ranges_scratch_bytes': (c_size_t, [c_int32]),
    'dfq_bn_ranges': (c_int32, [POINTER(DfqBnRangeReq), c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    'dfq_relu_moments': (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    'dfq_moments_after_add': (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    'dfq_moment_range': (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    'dfq_bn_stat_loss_scratch_bytes': (c_size_t, [c_int64]),
    'dfq_bn_stat_loss_forward': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    'dfq_bn_stat_loss_backward': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                            c_void_p, c_float, c_float, c_void_p, c_int32, c_void_p]),
    'dfq_bn_stat_loss_backward_dev': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    'dfq_bn_through_layer': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
}


def bind(cdll):
    """Attach restype/argtypes of every C-ABI symbol; raises AttributeError if one is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'dfq_amd: {} is missing -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C dfq_amd/csrc`).  There is no CPU implementation to fall back to.'.format(LIB_PATH))
        _lib = bind(ctypes.CDLL(LIB_PATH))
    return _lib


class DfqError(RuntimeError):
    code = 0


ERR_ABANDONED = -4          # DFQ_ERR_ABANDONED: a workgroup gave up a bounded in-launch wait (include/dfq_hip.h)


def check(rc):
    if rc != 0:
        msg = lib().dfq_last_error()
        err = DfqError('libdfq_hip error {}: {}'.format(rc, msg.decode() if msg else '?'))
        err.code = int(rc)
        raise err


def target_device():
    """The HIP device the engine runs on.  No GPU -> error (never a CPU fallback)."""
    if not torch.cuda.is_available():
        raise RuntimeError('dfq_amd: no ROCm GPU visible; the calibration engine has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def current_stream():
    """hipStream_t of torch's current stream, as an integer address."""
    return torch.cuda.current_stream().cuda_stream


def stream_arg():
    return c_void_p(current_stream())


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def _pinned(n):
    """float32 host buffer for a packed transfer: page-locked when the target is a GPU (torch's host allocator caches the
    blocks).  A synchronous copy between the device and PAGEABLE memory -- what `.to()` does -- stalled for 60-90 ms every
    few calls on the MI355X boxes, whatever its size (even the 64 KB of the scale vectors); copies through page-locked
    memory that are enqueued on the stream and waited for with a stream synchronisation do not."""
    return torch.empty(n, dtype=torch.float32, pin_memory=(target_device().type == 'cuda'))


_NP_OK = (torch.float32, torch.float64, torch.float16)


def _host_copy(pack_np, pack_t, t, to_pack):
    """One host-side copy between a caller's CPU tensor `t` and its slot of a packed buffer (converts dtype, follows strides).
    Through numpy where it can: torch's copy_ of a large tensor is an OpenMP parallel-for, and on a host whose cores are not
    all available at once (the GPU boxes are VMs) the team's barrier stalled for 60-280 ms every few calls; a plain
    single-threaded memcpy of a network's 14 MB is 2-3 ms."""
    if t.dtype in _NP_OK:
        a = t.numpy()
        if to_pack:
            np.copyto(pack_np, a, casting='same_kind')
        else:
            np.copyto(a, pack_np, casting='same_kind')
    elif to_pack:
        pack_t.copy_(t)
    else:
        t.copy_(pack_t)


_zero_row = None


def _zeros_row():
    global _zero_row
    if _zero_row is None:
        _zero_row = torch.zeros(Stage._ALIGN, dtype=torch.float32)
    return _zero_row


@contextlib.contextmanager
def _one_thread():
    """torch's CPU copies of large tensors are OpenMP parallel-fors; on a host whose cores are not all available at once (the
    GPU boxes are VMs) the team's barrier stalled for 60-280 ms every few calls.  The packed copies run on the calling thread."""
    n = torch.get_num_threads()
    if n > 1:
        torch.set_num_threads(1)
    try:
        yield
    finally:
        if n > 1:
            torch.set_num_threads(n)


_STAGE_THREADS = max(1, int(os.environ.get('DFQ_STAGE_THREADS', '1') or 1))
_stage_pool = None


def _host_copies(jobs, to_pack):
    """_host_copy for a list of (numpy slot, torch slot, caller's tensor) -- the path for tensors that need a dtype conversion or
    are not contiguous.  DFQ_STAGE_THREADS > 1 deals the tensors to plain threads by size (numpy's copy releases the GIL); off by
    default: for the ~250 small tensors of a network the hand-off costs more than it saves (measured: 1 thread 1.8 ms, 4 threads
    2.7 ms per 14 MB -- the time is per-tensor Python, not bandwidth)."""
    global _stage_pool
    total = sum(j[2].numel() for j in jobs)
    if _STAGE_THREADS == 1 or total < (1 << 18) or len(jobs) < 2 * _STAGE_THREADS:
        for a, b, t in jobs:
            _host_copy(a, b, t, to_pack)
        return
    if _stage_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _stage_pool = ThreadPoolExecutor(max_workers=_STAGE_THREADS, thread_name_prefix='dfq-stage')
    shares = [[] for _ in range(_STAGE_THREADS)]
    load = [0] * _STAGE_THREADS
    for j in sorted(jobs, key=lambda j: -j[2].numel()):           # largest first onto the lightest share
        k = load.index(min(load))
        shares[k].append(j)
        load[k] += j[2].numel()

    def run(share):
        for a, b, t in share:
            _host_copy(a, b, t, to_pack)
    for fut in [_stage_pool.submit(run, sh) for sh in shares if sh]:
        fut.result()


_ambient = threading.local()


@contextlib.contextmanager
def staging():
    """One staging area for a SEQUENCE of entry-point calls on a CPU-resident model:

        with dfq_amd.staging():
            cross_layer_equalization(graph, relations, targ_type)
            bias_absorption(graph, relations, bottoms)
            bias_correction(graph, bottoms, targ_type)

    Every tensor crosses PCIe once in each direction for the whole sequence -- copied to the device when the first call
    touches it, written back when the scope ends -- instead of once per call (a MobileNetV2 is 14 MB each way per entry point,
    plus the host-side packing).  Between the calls the device copies are the truth: host code inside the scope must not read
    or write the model's tensors (the reference's calibration section, main_cls.py:149-181, does not).  If the scope is left by
    an exception nothing is written back: the model keeps the values it had when the scope was entered.  Device-resident
    tensors are used in place as always; scopes do not nest (an inner scope joins the outer one).  Only the calibration
    entry points join the scope (entry_stage()); QuantMeasure / quantize() / the single-step primitives called inside it
    keep their one-call staging (their inputs change between calls)."""
    outer = getattr(_ambient, 'stage', None)
    if outer is not None:
        yield outer
        return
    # the scope works on the thread's persistent stage when there is one: what earlier plain calls (merge_batchnorm) have
    # uploaded is found again, and the plans keyed on those device copies come back from the cache
    st = persistent_stage() or Stage()
    st._scoped = True
    st._touched = set()
    _ambient.stage = st
    ok = False
    try:
        yield st
        ok = True
    finally:
        _ambient.stage = None
        st._scoped = False
        if ok:
            st._writeback()
        elif st._persistent:
            st.reset()                                    # the device copies are ahead of the caller's tensors: they go
        if getattr(st, '_late', False) and st.device.type == 'cuda':
            torch.cuda.current_stream().synchronize()     # results handed out as views of buffers still in flight


_PERSIST = os.environ.get('DFQ_STAGE_PERSIST', '1') != '0'


def persistent_stage():
    """The calling thread's persistent stage (see Stage.__init__), validated for a new call; None when switched off
    (DFQ_STAGE_PERSIST=0).  It makes the reference's own call sequence on a CPU-resident model -- merge_batchnorm,
    cross_layer_equalization, bias_correction, quantize_targ_layer, one plain call after the other (main_cls.py:149-188) --
    transfer the network to the device ONCE; every call still leaves its results in the caller's tensors.  It holds the last
    model's tensors and their device copies alive until another model arrives or release_staging() is called."""
    if not _PERSIST:
        return None
    st = getattr(_ambient, 'persist', None)
    if st is None:
        st = _ambient.persist = Stage()
        st._persistent = True
    return st.begin_call()


def release_staging():
    """Drop the persistent stage's device copies (and its references to the caller's tensors)."""
    st = getattr(_ambient, 'persist', None)
    if st is not None:
        st.reset(keep_buffers=False)


def scoped_stage():
    """The stage whose device copies outlive the call: the enclosing staging() scope's, else the thread's persistent stage
    (or None).  Plans built on such a stage are keyed on its device copies and come back from the plan cache."""
    return getattr(_ambient, 'stage', None) or persistent_stage()


def entry_stage():
    """The stage of a CALIBRATION entry point (merge_batchnorm, cross_layer_equalization, bias_absorption, clip_weight,
    bias_correction, quantize_targ_layer, set_quant_minmax): the enclosing staging() scope's, else a private one.  Nothing
    else shares the scope's stage: a per-call user -- QuantMeasure.forward, quantize(), the single-step primitives -- binds
    tensors whose host values change between calls (running ranges, activations), and a scope-long binding would hand every
    later call the first call's device copy and write that stale copy back when the scope ends."""
    return scoped_stage() or Stage()


def _to_device(host, device):
    if device.type != 'cuda':
        return host.to(device)
    flat = torch.empty(host.numel(), dtype=torch.float32, device=device)
    flat.copy_(host, non_blocking=True)          # ordered on the current stream before every kernel that reads it
    return flat


def _to_host(flat):
    if flat.device.type != 'cuda':
        return flat.to('cpu')
    host = _pinned(flat.numel())
    host.copy_(flat.reshape(-1), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host


class Stage:
    """Binds torch tensors to float32 contiguous device buffers for one engine call.

    A tensor already resident on the target device is used in place (the kernels then mutate the
    caller's storage directly, like the reference's ``mul_``/``add_``).  Anything else (the
    reference's default: a CPU model) is shadowed by a device copy and written back by
    ``writeback()`` -- the PCIe-inclusive way of calling the engine.
    """

    def __init__(self):
        self.device = target_device()
        self._bound = {}
        self._shadow = []
        self._packs = []          # (flat device buffer, [(caller's tensor, offset, numel)]) of prefetch()
        self._hosts = []
        self._scoped = False
        # the thread's PERSISTENT stage (persistent_stage()): device shadows of a CPU-resident model that outlive the call.  Every
        # entry point still writes its results back before it returns (the caller's tensors are the truth between calls, as in
        # the reference), but the next entry point finds the shadows of the previous one: no packing, no host-to-device copy, the
        # same device addresses -- so its plan comes from the cache.  _seen: id(tensor) -> (tensor._version, data_ptr) as of the
        # moment the shadow and the caller's tensor last held the same values
        self._persistent = False
        self._seen = {}
        self._touched = set()
        self._spare = {}
        self._whole = []          # (device buffer, layout key) of every transfer: what reset() keeps aside

    def _note(self, t):
        self._seen[id(t)] = (t._version, t.data_ptr())

    def begin_call(self):
        """Persistent stage, at the start of an entry point: a shadow whose tensor the caller has written since (torch bumps
        ``_version`` on every in-place write; a new storage shows in ``data_ptr``) is refreshed from the host, in place -- the
        device address stays, cached plans stay valid."""
        if not self._scoped:
            self._touched = set()                        # what THIS call binds: all its write-back has to bring home
        stale = [(t, buf) for (t, buf) in self._bound.values() if buf is not t and self._seen.get(id(t)) != (t._version, t.data_ptr())]
        if stale:
            with torch.no_grad():
                for t, buf in stale:
                    buf.copy_(t.detach().to(torch.float32).reshape(buf.shape))
                    self._note(t)
        return self

    def reset(self, keep_buffers=True):
        """Drop every shadow (a new model arrives, or release_staging()).  The packs' device buffers are kept aside by layout
        (the sizes of the tensors packed into them): the next model of the same architecture is uploaded INTO them, so its shadows
        have the addresses the previous model's had and the plans built on those come back from the plan cache."""
        spare = {}
        if keep_buffers:
            for flat, layout in self._whole:
                spare.setdefault(layout, flat)
        self._bound, self._shadow, self._packs, self._hosts, self._seen, self._whole = {}, [], [], [], {}, []
        self._spare = spare
        self._touched = set()

    _ALIGN = 64                   # floats: every packed tensor starts on a 256-byte boundary (the kernels' 16-byte vectors)
    _BIG = 4096                   # floats: tensors of at least this size go into the persistent stage's "large" pack

    def prefetch(self, tensors):
        """Shadow every tensor of `tensors` that is not usable in place with ONE host-to-device copy: the tensors are packed
        into one host buffer (float32, contiguous), copied once, and bound to views of one flat device buffer.  A CPU-resident
        MobileNetV2 is ~250 tensors; one synchronous copy each way per tensor was 40-150 ms per entry point, packed it is the
        14 MB over PCIe plus two host memcpys.  Later bind() calls find the tensors bound."""
        if self._persistent and self._bound:
            wanted = [t for t in tensors if t is not None and t.device.type == 'cpu']
            if wanted and not any(id(t) in self._bound and self._bound[id(t)][0] is t for t in wanted):
                self.reset()                              # none of these tensors is known: another model -- the old shadows go first
                                                          # (its device buffers return to the allocator: the new ones get their addresses)
        todo, seen = [], set()
        amb = getattr(_ambient, 'stage', None)
        amb_bound = amb._bound if (amb is not None and amb is not self) else {}
        if self._persistent:
            self._touched.update(id(t) for t in tensors if t is not None)
        for t in tensors:
            if t is None or id(t) in self._bound or id(t) in seen or id(t) in amb_bound:      # (the scope's copy is the truth: bind())
                continue
            if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous():
                continue
            if t.device.type != 'cpu' or t.numel() == 0:
                continue                                  # another device / empty: bind() handles it one by one
            seen.add(id(t))
            todo.append(t)
        if len(todo) < 2:
            return
        n_big = 0
        if self._persistent:
            # the large tensors (weights) first, then the small ones (biases, BN proxies, scales): ONE transfer, registered as TWO
            # packs over the two regions of the buffer -- an entry point that rewrites only vectors (bias_correction) then brings
            # back a few dozen KB instead of the whole network (writeback)
            big = [t for t in todo if t.numel() >= self._BIG]
            if len(big) >= 2 and len(todo) - len(big) >= 2:
                todo = big + [t for t in todo if t.numel() < self._BIG]
                n_big = len(big)
        offs, total = [], 0
        for t in todo:
            offs.append(total)
            total += -(-t.numel() // self._ALIGN) * self._ALIGN
        host = _pinned(total)
        fast = all(t.dtype is torch.float32 and t.is_contiguous() for t in todo)
        flats = None
        with torch.no_grad():
            if fast:
                # ONE call packs every tensor (torch.cat into the page-locked buffer, the alignment gaps filled from a row of
                # zeros): per tensor only the flat view is made in Python -- slicing numpy / torch views and one copy call per
                # tensor was 1.5 of the 1.8 ms this took for a MobileNetV2, the 14 MB themselves move in 0.2 ms
                views = [t.detach() if t.dim() == 1 else t.detach().view(-1) for t in todo]
                parts, sizes, where, zeros = [], [], [], _zeros_row()
                for v, o, nxt in zip(views, offs, offs[1:] + [total]):
                    where.append(len(parts))
                    parts.append(v)
                    sizes.append(v.numel())
                    gap = nxt - o - v.numel()
                    if gap:
                        parts.append(zeros[:gap])
                        sizes.append(gap)
                with _one_thread():
                    torch.cat(parts, out=host)
                flats = (views, sizes, where, [t.data_ptr() for t in todo])
            else:
                hn = host.numpy()
                _host_copies([(hn[o:o + t.numel()].reshape(t.shape), host[o:o + t.numel()].view(t.shape), t.detach()) for t, o in zip(todo, offs)],
                             to_pack=True)
        spare = getattr(self, '_spare', None)
        flat = spare.pop(tuple(t.numel() for t in todo), None) if (spare and flats is not None) else None
        if flat is not None and flat.numel() == total and flat.device == self.device:
            flat.copy_(host, non_blocking=True)   # the previous model's buffer of the same layout: same device addresses (see reset)
        else:
            flat = _to_device(host, self.device)
        self._hosts.append(host)                  # alive until the stage goes (the copy may still be in flight)
        items = []
        if flats is not None:                     # the device views: one split, a reshape only where the tensor is not 1-D
            pieces = flat.split(flats[1])
            bound = self._bound
            for t, o, w in zip(todo, offs, flats[2]):
                buf = pieces[w]
                if t.dim() != 1:
                    buf = buf.view(t.shape)
                bound[id(t)] = (t, buf)
                items.append((t, o, t.numel()))
        else:
            for t, o in zip(todo, offs):
                buf = flat[o:o + t.numel()].view(t.shape)
                self._bound[id(t)] = (t, buf)
                items.append((t, o, t.numel()))
        if n_big and flats is not None:
            # the two regions as packs of their own (views of the one buffer; the pieces' order is the tensors' order)
            a = offs[n_big]
            views, sizes, where, ptrs = flats
            k = where[n_big]                          # first piece of the small region
            self._packs.append((flat[:a], items[:n_big], (views[:n_big], sizes[:k], where[:n_big], ptrs[:n_big])))
            self._packs.append((flat[a:], [(t, o - a, n) for (t, o, n) in items[n_big:]],
                                (views[n_big:], sizes[k:], [w - k for w in where[n_big:]], ptrs[n_big:])))
            self._whole.append((flat, tuple(t.numel() for t in todo)))
        else:
            self._packs.append((flat, items, flats))
            self._whole.append((flat, tuple(t.numel() for t in todo)))
        if self._persistent:
            for t in todo:
                self._note(t)

    def new_flat(self, n):
        """A flat float32 device buffer that adopt() may be given later: the persistent stage hands out the previous model's
        buffer of the same size (same address: cached plans stay valid, see reset)."""
        flat = self._spare.pop(('flat', int(n)), None) if self._persistent else None
        return flat if flat is not None else torch.empty((int(n),), dtype=torch.float32, device=self.device)

    def adopt(self, flat, items):
        """A flat device buffer whose slices ARE the device values of host tensors that were just created from it (the BN proxies
        merge_batchnorm computes on the device and registers on the host): the stage takes it as one more pack, so the next entry
        point finds those tensors shadowed instead of uploading them.  items: [(host tensor, offset, numel)].  Stages whose
        shadows outlive the call only (a scope or the persistent stage)."""
        if not (self._scoped or self._persistent) or flat.device != self.device:
            return
        kept = []
        for t, o, n in sorted(items, key=lambda it: it[1]):
            if id(t) in self._bound or t.device.type != 'cpu' or t.dtype != torch.float32 or not t.is_contiguous():
                continue
            self._bound[id(t)] = (t, flat[o:o + n].view(t.shape))
            kept.append((t, o, n))
            if self._persistent:
                self._note(t)
        if kept:
            # the same description prefetch() makes of a pack (flat views of the caller's tensors, the sizes the host copy of the
            # buffer is split into, which piece is whose): the write-back is then one split and one multi-tensor copy
            views, sizes, where, at = [], [], [], 0
            for t, o, n in kept:
                if o > at:
                    sizes.append(o - at)
                where.append(len(sizes))
                sizes.append(n)
                views.append(t.view(-1))
                at = o + n
            if at < flat.numel():
                sizes.append(flat.numel() - at)
            self._packs.append((flat, kept, (views, sizes, where, [t.data_ptr() for t, _, _ in kept])))
            self._whole.append((flat, ('flat', flat.numel())))

    def bind(self, t):
        if t is None:
            return None
        key = id(t)
        if self._persistent:
            self._touched.add(key)
        hit = self._bound.get(key)
        if hit is not None:
            return hit[1]
        amb = getattr(_ambient, 'stage', None)
        if amb is not None and amb is not self:
            # a private stage INSIDE a staging() scope (merge_scale_into_layer, _layer_equalization, the single-step primitives):
            # a tensor the scope has already shadowed is bound to the SCOPE's device copy -- between the calls of a scope that
            # copy is the truth (the host value is stale), and what this call writes must not be overwritten by the scope's
            # write-back when it ends
            shared = amb._bound.get(key)
            if shared is not None and shared[1] is not t:
                return shared[1]
        if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous():
            # already where the kernels want it: the engine only takes its address (no torch op ever writes through this
            # handle), so the tensor itself will do -- no detach(), whose cost adds up over the ~250 tensors of a network
            buf = t
        else:
            d = t.detach()
            buf = d.to(device=self.device, dtype=torch.float32).contiguous()
            if buf.data_ptr() == d.data_ptr():      # .to() was a no-op view
                buf = buf.clone()
            self._shadow.append((t, buf))
            if self._persistent:
                self._note(t)
        self._bound[key] = (t, buf)     # keep `t` alive so id() stays unique
        return buf

    def new(self, shape, fill=None, dtype=torch.float32):
        if fill is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        return torch.full(shape, fill, dtype=dtype, device=self.device)

    def writeback(self, unchanged=()):
        """Bring the shadows' values back into the caller's tensors.  ``unchanged``: tensors this call is known not to have
        written (bias_correction: the weights) -- a pack made of nothing else is not transferred."""
        if self._scoped:
            return                                        # staging() writes everything back once, when the scope ends
        self._writeback(frozenset(id(t) for t in unchanged if t is not None) if unchanged else frozenset())

    def _writeback(self, skip=frozenset()):
        touched = self._touched if self._persistent else None      # persistent stage: only what this call has bound can have changed
        with torch.no_grad():
            for flat, items, flats in self._packs:        # one device-to-host copy per pack, then host-side copies
                if skip and all(id(t) in skip for (t, _, _) in items):
                    continue
                if touched is not None and not any(id(t) in touched for (t, _, _) in items):
                    continue
                host = _to_host(flat)
                # (persistent stage: of a pack that comes back, only the tensors this call has bound are copied into the caller's
                # tensors -- the others hold on the host what they hold on the device)
                want = (lambda t: id(t) in touched) if touched is not None else (lambda t: True)
                if flats is not None:                     # the flat views of prefetch(): one split, one multi-tensor copy
                    views, sizes, where, ptrs = flats
                    for i, (t, _, _) in enumerate(items):
                        if t.data_ptr() != ptrs[i]:       # the caller gave the tensor a new storage meanwhile: write there
                            views[i] = t.detach().reshape(-1) if t.is_contiguous() else None
                    pieces = host.split(sizes)
                    dst, src = [], []
                    for (t, _, _), w, v in zip(items, where, views):
                        if v is not None and want(t):
                            dst.append(v)
                            src.append(pieces[w])
                    if dst:
                        with _one_thread():
                            torch._foreach_copy_(dst, src)
                    for (t, o, n), v in zip(items, views):
                        if v is None and want(t):
                            t.detach().copy_(host[o:o + n].view(t.shape))
                    if self._persistent:                  # shadow and tensor hold the same values again: remember the versions
                        for (t, _, _) in items:
                            if want(t):
                                self._note(t)
                    continue
                hn = host.numpy()
                _host_copies([(hn[o:o + n].reshape(t.shape), host[o:o + n].view(t.shape), t.detach()) for t, o, n in items if want(t)], to_pack=False)
                if self._persistent:
                    for (t, _, _) in items:
                        if want(t):
                            self._note(t)
            for t, buf in self._shadow:
                if id(t) in skip or (touched is not None and id(t) not in touched):
                    continue
                t.data.copy_(buf.to(t.device, t.dtype) if (buf.device != t.device or buf.dtype != t.dtype) else buf)
                if self._persistent:
                    self._note(t)

    def out_like_many(self, t, bufs):
        """out_like for a list of device tensors with ONE transfer (independent tensors on the caller's device)."""
        if not bufs or bufs[0].device == t.device:
            return list(bufs)
        flat = torch.cat([b.reshape(-1) for b in bufs])
        if self._scoped and not self._persistent and t.device.type == 'cpu' and flat.device.type == 'cuda':
            # inside staging(): the copy is enqueued and awaited when the scope ends (host code does not read the model's
            # tensors before that); the results are views of the page-locked buffer it lands in
            host = _pinned(flat.numel())
            host.copy_(flat, non_blocking=True)
            self._hosts.append(host)
            self._late = True
            outs, at = [], 0
            for b in bufs:
                outs.append(host[at:at + b.numel()].view(b.shape))
                at += b.numel()
            return outs
        flat = _to_host(flat) if t.device.type == 'cpu' else flat.to(t.device)
        outs, at = [], 0
        for b in bufs:
            outs.append(flat[at:at + b.numel()].view(b.shape).clone())
            at += b.numel()
        return outs

    def out_like(self, t, buf):
        """Return `buf` on the device/dtype the caller's tensor `t` lives on."""
        if buf.device == t.device:
            return buf
        return buf.to(t.device)


def synchronize():
    torch.cuda.current_stream().synchronize()
