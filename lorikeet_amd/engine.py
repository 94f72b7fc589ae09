"""Thin object wrapper over the C ABI: one `HipPairHMMEngine` == one `phmm_handle`.

All compute happens in libphmm.so's gfx950 kernels; this file only moves pointers.
"""
import ctypes as C

import numpy as np

from . import _lib
from .batch import RegionBatch


class PhmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("phmm error %d: %s" % (code, msg))
        self.code = code


def _p(a, t):
    return a.ctypes.data_as(t)


class DevicePlan:
    """A `phmm_batch`: launch plan + device copies of the offset arrays."""

    def __init__(self, engine, batch):
        self.engine = engine
        self.batch = batch
        L = engine.lib
        self._b = L.phmm_batch_create(engine._h, batch.n_regions, _p(batch.region_read_off, _lib.u32p),
                                      _p(batch.region_hap_off, _lib.u32p), _p(batch.read_off, _lib.u32p),
                                      _p(batch.hap_off, _lib.u32p), _p(batch.out_off, _lib.u64p))
        if not self._b:
            raise PhmmError(_lib.PHMM_ERR_INVALID_ARG, engine.last_error())
        self._keep = None

    def bind_torch(self, tensors, out):
        """tensors: dict of uint8 CUDA tensors (read_bases, base_q, ins_q, del_q, gcp, hap_bases);
        out: float64 CUDA tensor with batch.n_out elements."""
        self._keep = (tensors, out)
        ptrs = [tensors[k].data_ptr() for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")]
        self.engine._check(self.engine.lib.phmm_batch_bind_device(self._b, *ptrs, out.data_ptr()))

    def upload(self):
        b = self.batch
        self.engine._check(self.engine.lib.phmm_batch_upload(
            self._b, _p(b.read_bases, _lib.u8p), _p(b.base_q, _lib.u8p), _p(b.ins_q, _lib.u8p),
            _p(b.del_q, _lib.u8p), _p(b.gcp, _lib.u8p), _p(b.hap_bases, _lib.u8p)))

    def launch(self, stream=None):
        """Enqueue the forward kernels.  stream: raw hipStream_t as int (e.g.
        torch.cuda.current_stream().cuda_stream) or None for the engine's own stream."""
        self.engine._check(self.engine.lib.phmm_batch_launch(self._b, C.c_void_p(stream) if stream else None))

    def download(self):
        out = np.empty(self.batch.n_out, dtype=np.float64)
        self.engine._check(self.engine.lib.phmm_batch_download(self._b, _p(out, _lib.f64p)))
        return out

    def status(self):
        self.engine._check(self.engine.lib.phmm_batch_status(self._b))

    @property
    def cells(self):
        return int(self.engine.lib.phmm_batch_cells(self._b))

    @property
    def executed_cells(self):
        return int(self.engine.lib.phmm_batch_executed_cells(self._b))

    @property
    def algorithmic_bytes(self):
        return int(self.engine.lib.phmm_batch_algorithmic_bytes(self._b))

    @property
    def num_launches(self):
        return int(self.engine.lib.phmm_batch_num_launches(self._b))

    @property
    def dominant_kernel(self):
        return self.engine.lib.phmm_batch_dominant_kernel(self._b).decode()

    def close(self):
        if self._b:
            self.engine.lib.phmm_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipPairHMMEngine:
    """phmm_create / phmm_destroy.  Raises if there is no HIP device (no CPU fallback)."""

    def __init__(self, device_id=0, do_not_use_tristate_correction=False, f32_first=False):
        self.lib = _lib.load()
        flags = _lib.PHMM_FLAG_NO_TRISTATE if do_not_use_tristate_correction else 0
        if f32_first:  # opt-in, see PHMM_FLAG_F32_FIRST in include/phmm.h
            flags |= _lib.PHMM_FLAG_F32_FIRST
        self._h = self.lib.phmm_create(int(device_id), flags)
        if not self._h:
            raise PhmmError(_lib.PHMM_ERR_NO_DEVICE, self.lib.phmm_last_error(None).decode())

    def last_error(self):
        return self.lib.phmm_last_error(self._h).decode()

    def _check(self, code):
        if code != _lib.PHMM_OK:
            raise PhmmError(code, self.last_error())

    def compute(self, batch: RegionBatch):
        """Synchronous phmm_compute on host arrays.  Returns out (float64, batch.n_out)."""
        out = np.empty(batch.n_out, dtype=np.float64)
        args = getattr(batch, "_abi_args", None)
        if args is None:  # pointer conversion is the slow part of a ctypes call: do it once per batch
            args = (batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                    _p(batch.read_off, _lib.u32p), _p(batch.read_bases, _lib.u8p), _p(batch.base_q, _lib.u8p),
                    _p(batch.ins_q, _lib.u8p), _p(batch.del_q, _lib.u8p), _p(batch.gcp, _lib.u8p),
                    _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p), _p(batch.out_off, _lib.u64p))
            batch._abi_args = args
        self._check(self.lib.phmm_compute(self._h, *args, _p(out, _lib.f64p)))
        return out

    @staticmethod
    def _abi_args(batch):
        args = getattr(batch, "_abi_args", None)
        if args is None:
            args = (batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                    _p(batch.read_off, _lib.u32p), _p(batch.read_bases, _lib.u8p), _p(batch.base_q, _lib.u8p),
                    _p(batch.ins_q, _lib.u8p), _p(batch.del_q, _lib.u8p), _p(batch.gcp, _lib.u8p),
                    _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p), _p(batch.out_off, _lib.u64p))
            batch._abi_args = args
        return args

    def submit(self, batch: RegionBatch):
        """phmm_submit: queue `batch` on this (shared, thread-safe for submit/wait) engine.  Returns (ticket, out);
        `out` holds the results once wait(ticket) has returned.  The batch must stay alive until then."""
        import ctypes as C
        out = np.empty(batch.n_out, dtype=np.float64)
        ticket = C.c_uint64(0)
        self._check(self.lib.phmm_submit(self._h, *self._abi_args(batch), _p(out, _lib.f64p), C.byref(ticket)))
        return ticket.value, out

    def wait(self, ticket):
        """phmm_wait: block until the submission's results are in its `out`; raises PhmmError with its own status."""
        self._check(self.lib.phmm_wait(self._h, int(ticket)))

    def submit_stats(self):
        """(flushes run, submissions they carried) of phmm_submit / phmm_wait on this engine."""
        import ctypes as C
        f, n = C.c_uint64(0), C.c_uint64(0)
        self.lib.phmm_submit_stats(self._h, C.byref(f), C.byref(n))
        return f.value, n.value

    def plan(self, batch: RegionBatch):
        return DevicePlan(self, batch)

    def set_switch(self, name, value):
        """phmm_set_switch: one developer switch of this handle (tests / A-B measurements; include/phmm.h)."""
        self._check(self.lib.phmm_set_switch(self._h, name.encode(), int(value)))

    def switches(self, **kw):
        """Context manager: set developer switches for the duration of a `with` block, then back to the planner's
        choice (force_L=0, force_chain=-1, force_streams=0, no_pipeline=0, ...)."""
        import contextlib
        defaults = {"force_L": 0, "force_chain": -1, "force_streams": 0, "no_pipeline": 0, "no_rescue": 0, "trace": 0}

        @contextlib.contextmanager
        def cm():
            for k, v in kw.items():
                self.set_switch(k, v)
            try:
                yield self
            finally:
                for k in kw:
                    self.set_switch(k, defaults[k])
        return cm()

    def stat(self, name):
        """phmm_get_stat: "staged_bytes", "rescue_passes"."""
        return int(self.lib.phmm_get_stat(self._h, name.encode()))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.phmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_describe(batch: RegionBatch, flags=0, concurrent_callers=1):
    """phmm_plan_describe: the launch plan of `batch` without touching a device (host only) -> _lib.PlanInfo."""
    lib = _lib.load()
    info = _lib.PlanInfo()
    code = lib.phmm_plan_describe(int(flags), int(concurrent_callers), batch.n_regions, _p(batch.region_read_off, _lib.u32p),
                                  _p(batch.region_hap_off, _lib.u32p), _p(batch.read_off, _lib.u32p), _p(batch.hap_off, _lib.u32p),
                                  C.byref(info))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, "phmm_plan_describe: invalid argument")
    return info


def assign_regions(batch: RegionBatch, n_parts):
    """phmm_assign_regions: greedy longest-processing-time assignment of whole regions to `n_parts` engines by
    cells(region).  Host only -- works without a device.  Returns uint32[n_regions]."""
    lib = _lib.load()
    part = np.zeros(batch.n_regions, np.uint32)
    code = lib.phmm_assign_regions(batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                                   _p(batch.read_off, _lib.u32p), _p(batch.hap_off, _lib.u32p), int(n_parts),
                                   _p(part, _lib.u32p))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, "phmm_assign_regions: invalid argument")
    return part


def split_regions(batch: RegionBatch, n_parts):
    """phmm_split_regions: boundaries of `n_parts` contiguous, cell-balanced ranges of regions (host only).
    Returns uint32[n_parts + 1]; identical to sharding.split_contiguous."""
    lib = _lib.load()
    first = np.zeros(int(n_parts) + 1, np.uint32)
    code = lib.phmm_split_regions(batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                                  _p(batch.read_off, _lib.u32p), _p(batch.hap_off, _lib.u32p), int(n_parts),
                                  _p(first, _lib.u32p))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, "phmm_split_regions: invalid argument")
    return first


def compute_multi(engines, batch: RegionBatch):
    """phmm_compute_multi: one batch over several engines (normally one per device) of this process; regions are
    sharded by assign_regions, every engine computes its share concurrently, no exchange between devices."""
    import ctypes as C
    lib = engines[0].lib
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    out = np.empty(batch.n_out, dtype=np.float64)
    code = lib.phmm_compute_multi(hs, len(engines), *HipPairHMMEngine._abi_args(batch), _p(out, _lib.f64p))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, engines[0].last_error())
    return out
