"""ctypes binding of libntt_mi355x.so -- the MI355X-native negacyclic NTT engine.

The product is the C-ABI shared library built from ``csrc/`` (HIP kernels for
gfx950 + a C host layer); this module is only a thin binding so that the Python
test-suite and ``bench.py`` can call it.  It mirrors the library's two surfaces:

* the batched, device-resident plan API of ``include/ntt_mi355x.h``
  (``Plan.fwd`` / ``Plan.inv`` / ``Plan.pointwise_mul`` / ``Plan.negacyclic_mul``);
* the reference's single-polynomial signatures on host arrays
  (``fwd_ntt_ref_harvey``, ``inv_ntt_ref_harvey``, ``fwd_ntt_radix4``,
  ``inv_ntt_radix4``, ``fwd_ntt_radix4x4``, ``fwd_ntt_ref_harvey_dbl``), which
  behave like reference include/ntt_reference.h:13-65, ntt_radix4.h:10-35 and
  ntt_radix4x4.h:10-28.

There is no CPU fallback: if the shared library is missing the import fails,
and without a HIP device every compute call raises ``NttError``.

The directory name contains hyphens (it is fixed by the project layout), so
import it with ``importlib`` -- see ``load()`` in the repository's ``ontt.py``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTT_LIB") or os.path.join(_HERE, "libntt_mi355x.so")  # NTT_LIB: A/B builds of the same library

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libntt_mi355x.so is not built (run `make lib` or __graft_entry__.build()); "
        "there is no pure-Python or CPU fallback for the NTT kernels")

_lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)

U64P = C.POINTER(C.c_uint64)
VOIDP = C.c_void_p

NTT_OK = 0
ARITH_AUTO, ARITH_U64, ARITH_F64, ARITH_U64_R4 = 0, 1, 2, 3
FLAG_INVERSE, FLAG_WIDE_IN, FLAG_LAZY_OUT = 1, 2, 4
MUL_LAZY_IN, MUL_B_BROADCAST, MUL_ACCUMULATE = 1, 2, 4
OPT_MAX_GRID, OPT_CHUNK_MIB, OPT_F64_CLASS, OPT_TWO_PHASE, OPT_FUSED_PRODUCT, OPT_BLOCK_LOG = 1, 2, 3, 4, 5, 6
OPT_XCD_LOCAL, OPT_XCD_LOCAL_LAG, OPT_XCD_LOCAL_WGS_PER_CU, OPT_INT_WIDE, OPT_BLOCK_OVERSUB = 7, 8, 9, 10, 11
OPT_RNS_LAUNCH, OPT_DOT_FUSED, OPT_MAX_BATCH_HINT, OPT_CTL_ALLOCATIONS, OPT_ONE_PASS = 12, 13, 14, 15, 16

#: every symbol include/ntt_mi355x.h and the reference-named headers declare
EXPORTED_SYMBOLS = [
    "ntt_last_error", "ntt_device_count", "ntt_version", "ntt_plan_create",
    "ntt_plan_create_from_tables", "ntt_plan_destroy", "ntt_plan_info", "ntt_plan_set_generic", "ntt_plan_set_option", "ntt_plan_get_option", "ntt_plan_reserve", "ntt_plan_export_table",
    "ntt_fwd_batch", "ntt_inv_batch", "ntt_fwd_batch_wide", "ntt_inv_batch_wide",
    "ntt_fwd_batch_lazy", "ntt_inv_batch_lazy", "ntt_transform_batch",
    "ntt_pointwise_mul_batch", "ntt_pointwise_mul_batch_lazy", "ntt_negacyclic_mul_batch", "ntt_rns_fwd_batch", "ntt_rns_inv_batch",
    "ntt_rns_negacyclic_mul_batch", "ntt_inv_product_batch", "ntt_inv_dot_batch", "ntt_mul_transformed_batch",
    "ntt_rns_inv_dot_batch", "ntt_rns_mul_transformed_batch", "ntt_fwd_mul_batch", "ntt_rns_fwd_mul_batch",
    "ntt_rns_fwd_batch_strided", "ntt_rns_inv_batch_strided", "ntt_rns_negacyclic_mul_batch_strided", "ntt_rns_inv_dot_batch_strided",
    "ntt_rns_mul_transformed_batch_strided", "ntt_rns_fwd_mul_batch_strided", "ntt_transform_batch_strided", "ntt_transform_ptrs", "ntt_rns_transform_ptrs", "ntt_transform_dev_ptrs", "ntt_rns_transform_dev_ptrs", "ntt_inv_dot_dev_ptrs", "ntt_fwd_mul_dev_ptrs", "ntt_negacyclic_mul_dev_ptrs",
    "ntt_rns_inv_dot_dev_ptrs", "ntt_rns_fwd_mul_dev_ptrs", "ntt_rns_negacyclic_mul_dev_ptrs", "ntt_dev_malloc", "ntt_dev_free", "ntt_dev_mem_info",
    "ntt_h2d", "ntt_d2h", "ntt_stream_create", "ntt_stream_destroy", "ntt_stream_sync",
    "ntt_event_create", "ntt_event_destroy", "ntt_event_record", "ntt_event_elapsed_ms",
    "ntt_fill_uniform", "ntt_poly_checksum", "ntt_rmw_probe", "ntt_shape_probe", "ntt_copy_probe", "ntt_batch_multi", "ntt_rns_mul_multi", "ntt_min_root", "ntt_find_prime",
    "ntt_compat_release", "ntt_compat_cached_plans",
    # reference signatures (include/ntt_reference.h, ntt_radix4.h, ntt_radix4x4.h, ntt_seal.h)
    "fwd_ntt_ref_harvey_lazy", "inv_ntt_ref_harvey", "fwd_ntt_ref_harvey_lazy_dbl",
    "fwd_ntt_radix4_lazy", "inv_ntt_radix4", "fwd_ntt_radix4x4_lazy", "fwd_ntt_seal_lazy",
    "inv_ntt_seal",
]


class NttError(RuntimeError):
    pass


class MulOp(C.Structure):
    """reference mul_op_t: two __uint128_t, passed by value (fast_mul_operators.h:10-13).

    __uint128_t is 16-byte aligned and ctypes has no 128-bit integer: the zero-length long double array
    contributes nothing but its 16-byte alignment, so libffi places the by-value copy where the C callee
    expects it even when other stack arguments precede it (tests/test_abi.py::test_mul_op_by_value_through_ctypes)."""
    _fields_ = [("_align16", C.c_longdouble * 0),
                ("op_lo", C.c_uint64), ("op_hi", C.c_uint64), ("con_lo", C.c_uint64), ("con_hi", C.c_uint64)]


def _sig(name, restype, *argtypes):
    if "NTT_LIB" in os.environ and not hasattr(_lib, name):
        return None   # an older build selected for a same-box A/B (tools/build_head.sh): entry points added since are simply absent
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("ntt_last_error", C.c_char_p)
_sig("ntt_version", C.c_char_p)
_sig("ntt_device_count", C.c_int)
_sig("ntt_plan_create", C.c_int, C.POINTER(VOIDP), C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int)
_sig("ntt_plan_create_from_tables", C.c_int, C.POINTER(VOIDP), C.c_int, C.c_uint64, C.c_uint64, U64P, U64P, C.c_int)
_sig("ntt_plan_destroy", None, VOIDP)
_sig("ntt_plan_info", C.c_int, VOIDP, U64P)
_sig("ntt_plan_set_generic", C.c_int, VOIDP, C.c_int)
_sig("ntt_plan_set_option", C.c_int, VOIDP, C.c_int, C.c_int64)
_sig("ntt_plan_reserve", C.c_int, VOIDP, VOIDP, C.c_uint64)
_sig("ntt_plan_get_option", C.c_int, VOIDP, C.c_int, C.POINTER(C.c_int64))
_sig("ntt_plan_export_table", C.c_int, VOIDP, C.c_int, VOIDP, C.c_size_t)
for _n in ("ntt_fwd_batch", "ntt_inv_batch", "ntt_fwd_batch_wide", "ntt_inv_batch_wide", "ntt_fwd_batch_lazy",
           "ntt_inv_batch_lazy"):
    _sig(_n, C.c_int, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_transform_batch", C.c_int, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_pointwise_mul_batch", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_pointwise_mul_batch_lazy", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_negacyclic_mul_batch", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_rns_fwd_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_uint64, VOIDP)
_sig("ntt_rns_inv_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_uint64, VOIDP)
_sig("ntt_rns_negacyclic_mul_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_inv_product_batch", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_inv_dot_batch", C.c_int, VOIDP, VOIDP, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_mul_transformed_batch", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_inv_dot_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64,
     C.c_uint, VOIDP)
_sig("ntt_rns_mul_transformed_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_fwd_mul_batch", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_fwd_mul_batch", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_fwd_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_rns_inv_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_rns_negacyclic_mul_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_rns_inv_dot_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64,
     C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_mul_transformed_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint64, C.c_uint64,
     C.c_uint, VOIDP)
_sig("ntt_rns_fwd_mul_batch_strided", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_transform_batch_strided", C.c_int, VOIDP, VOIDP, C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_transform_ptrs", C.c_int, VOIDP, C.POINTER(VOIDP), C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_transform_ptrs", C.c_int, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_transform_dev_ptrs", C.c_int, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_transform_dev_ptrs", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_inv_dot_dev_ptrs", C.c_int, VOIDP, VOIDP, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_fwd_mul_dev_ptrs", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_negacyclic_mul_dev_ptrs", C.c_int, VOIDP, VOIDP, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_rns_inv_dot_dev_ptrs", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_fwd_mul_dev_ptrs", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint64, C.c_uint, VOIDP)
_sig("ntt_rns_negacyclic_mul_dev_ptrs", C.c_int, C.c_int, C.POINTER(VOIDP), VOIDP, VOIDP, VOIDP, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_dev_malloc", C.c_int, C.c_int, C.POINTER(VOIDP), C.c_size_t)
_sig("ntt_dev_mem_info", C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
_sig("ntt_dev_free", C.c_int, C.c_int, VOIDP)
_sig("ntt_h2d", C.c_int, C.c_int, VOIDP, VOIDP, C.c_size_t)
_sig("ntt_d2h", C.c_int, C.c_int, VOIDP, VOIDP, C.c_size_t)
_sig("ntt_stream_create", C.c_int, C.c_int, C.POINTER(VOIDP))
_sig("ntt_stream_destroy", C.c_int, C.c_int, VOIDP)
_sig("ntt_stream_sync", C.c_int, C.c_int, VOIDP)
_sig("ntt_event_create", C.c_int, C.c_int, C.POINTER(VOIDP))
_sig("ntt_event_destroy", C.c_int, C.c_int, VOIDP)
_sig("ntt_event_record", C.c_int, C.c_int, VOIDP, VOIDP)
_sig("ntt_event_elapsed_ms", C.c_int, C.c_int, VOIDP, VOIDP, C.POINTER(C.c_float))
_sig("ntt_fill_uniform", C.c_int, C.c_int, VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_poly_checksum", C.c_int, C.c_int, VOIDP, VOIDP, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_rmw_probe", C.c_int, C.c_int, VOIDP, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_shape_probe", C.c_int, C.c_int, VOIDP, C.c_uint64, C.c_uint64, VOIDP)
_sig("ntt_copy_probe", C.c_int, C.c_int, VOIDP, VOIDP, C.c_uint64, VOIDP)
_sig("ntt_batch_multi", C.c_int, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), U64P, C.c_int)
_sig("ntt_rns_mul_multi", C.c_int, C.c_int, C.c_int, C.POINTER(VOIDP), C.POINTER(VOIDP), C.POINTER(VOIDP), C.POINTER(VOIDP), U64P)
_sig("ntt_compat_release", None)
_sig("ntt_compat_cached_plans", C.c_int)
_sig("ntt_min_root", C.c_uint64, C.c_uint64, C.c_uint64)
_sig("ntt_find_prime", C.c_uint64, C.c_uint, C.c_uint64, C.c_uint)
for _n in ("fwd_ntt_ref_harvey_lazy", "fwd_ntt_radix4_lazy", "fwd_ntt_radix4x4_lazy", "fwd_ntt_seal_lazy"):
    _sig(_n, None, U64P, C.c_uint64, C.c_uint64, U64P, U64P)
_sig("fwd_ntt_ref_harvey_lazy_dbl", None, U64P, U64P, C.c_uint64, C.c_uint64, U64P, U64P)
_sig("inv_ntt_ref_harvey", None, U64P, C.c_uint64, C.c_uint64, MulOp, C.c_uint64, U64P, U64P)
_sig("inv_ntt_radix4", None, U64P, C.c_uint64, C.c_uint64, MulOp, U64P, U64P)
_sig("inv_ntt_seal", None, U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, U64P, U64P)


def last_error():
    return _lib.ntt_last_error().decode()


def _check(rc):
    # ntt_last_error is thread-local in the library and only read here, on the failing thread, immediately
    # after the failing call
    if rc != NTT_OK:
        raise NttError("ntt status %d: %s" % (rc, last_error()))


def version():
    return _lib.ntt_version().decode()


def device_count():
    n = _lib.ntt_device_count()
    if n < 0:
        raise NttError(last_error())
    return n


def compat_release():
    _lib.ntt_compat_release()


def compat_cached_plans():
    return int(_lib.ntt_compat_cached_plans())


def min_root(q, n):
    return int(_lib.ntt_min_root(q, n))


def find_prime(bits, n, skip=0):
    return int(_lib.ntt_find_prime(bits, n, skip))


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


# --------------------------------------------------------------------------
# device memory (library-owned HIP allocations; torch tensors work as well:
# pass tensor.data_ptr() wherever a device pointer is expected)
# --------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, n_u64, device=0):
        self.device, self.n = device, int(n_u64)
        p = VOIDP()
        _check(_lib.ntt_dev_malloc(device, C.byref(p), self.n * 8))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        assert host.size <= self.n
        _check(_lib.ntt_h2d(self.device, self.ptr, host.ctypes.data, host.size * 8))
        return self

    def download(self, n=None, offset=0):
        n = self.n - offset if n is None else int(n)
        out = np.empty(n, dtype=np.uint64)
        _check(_lib.ntt_d2h(self.device, out.ctypes.data, self.ptr + 8 * offset, n * 8))
        return out

    def free(self):
        if self.ptr:
            _lib.ntt_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mem_info(device=0):
    """(free, total) bytes of device memory: hipMemGetInfo"""
    f, t = C.c_size_t(), C.c_size_t()
    _check(_lib.ntt_dev_mem_info(device, C.byref(f), C.byref(t)))
    return f.value, t.value


def fill_uniform(dptr, n, q, seed, offset=0, device=0, stream=None):
    _check(_lib.ntt_fill_uniform(device, dptr, n, q, seed, offset, stream))


def poly_checksum(dout, dptr, N, batch, device=0, stream=None):
    _check(_lib.ntt_poly_checksum(device, dout, dptr, N, batch, stream))


def rmw_probe(dptr, n, mask=0, device=0, stream=None):
    """in-place read-XOR-write of n words: the measured ceiling of the transform's memory shape"""
    _check(_lib.ntt_rmw_probe(device, dptr, n, mask, stream))


def shape_probe(dptr, n, mask=0, device=0, stream=None):
    """the same in the memory shape of the 2^14 block kernels (persistent workgroups, register prefetch, 16-byte accesses)"""
    _check(_lib.ntt_shape_probe(device, dptr, n, mask, stream))


def copy_probe(ddst, dsrc, n, device=0, stream=None):
    """out-of-place 16-byte-per-lane copy of n words (the guide's copy shape: 16 n bytes move)"""
    _check(_lib.ntt_copy_probe(device, ddst, dsrc, n, stream))


def stream_sync(device=0, stream=None):
    _check(_lib.ntt_stream_sync(device, stream))


class Event:
    def __init__(self, device=0):
        self.device = device
        e = VOIDP()
        _check(_lib.ntt_event_create(device, C.byref(e)))
        self.h = e.value

    def record(self, stream=None):
        _check(_lib.ntt_event_record(self.device, self.h, stream))

    def elapsed_ms_since(self, start):
        ms = C.c_float()
        _check(_lib.ntt_event_elapsed_ms(self.device, start.h, self.h, C.byref(ms)))
        return ms.value


# --------------------------------------------------------------------------
# plans
# --------------------------------------------------------------------------
class Plan:
    """Tables for one (device, N, q, root); see include/ntt_mi355x.h."""

    def __init__(self, N, q, root, device=0, arith=ARITH_AUTO):
        h = VOIDP()
        _check(_lib.ntt_plan_create(C.byref(h), device, N, q, root, arith))
        self.h, self.N, self.q, self.root, self.device = h.value, N, q, root, device

    def info(self):
        v = (C.c_uint64 * 8)()
        _check(_lib.ntt_plan_info(self.h, v))
        keys = ("N", "q", "log2N", "arith", "f64_class", "hbm_passes", "device", "root")
        return dict(zip(keys, [int(x) for x in v]))

    def set_generic(self, on):
        _check(_lib.ntt_plan_set_generic(self.h, int(on)))

    def export_table(self, which, n_records, dtype=np.uint64):
        """device table `which` (0 fwd, 1 inv, 2/3 compact fwd/inv) as an array of shape (n_records, 2) -- or
        (n_records,) for the compact tables -- of `dtype` (uint64 for the integer policies, float64 for FP64)"""
        width = 1 if which >= 2 else 2
        out = np.zeros(n_records * width, dtype=dtype)
        _check(_lib.ntt_plan_export_table(self.h, which, out.ctypes.data, out.nbytes))
        return out.reshape(n_records, width) if width == 2 else out

    def set_option(self, option, value):
        _check(_lib.ntt_plan_set_option(self.h, option, value))

    def get_option(self, option):
        v = C.c_int64()
        _check(_lib.ntt_plan_get_option(self.h, option, C.byref(v)))
        return v.value

    def reserve(self, polys, stream=None):
        """ntt_plan_reserve: the control blocks of the XCD-local launches on `stream`, sized for `polys` polynomials x limbs"""
        _check(_lib.ntt_plan_reserve(self.h, stream, polys))

    def fwd(self, dptr, batch, stream=None, wide=False, lazy=False):
        if wide and lazy:
            _check(_lib.ntt_transform_batch(self.h, dptr, batch, FLAG_WIDE_IN | FLAG_LAZY_OUT, stream))
            return
        f = _lib.ntt_fwd_batch_lazy if lazy else (_lib.ntt_fwd_batch_wide if wide else _lib.ntt_fwd_batch)
        _check(f(self.h, dptr, batch, stream))

    def inv(self, dptr, batch, stream=None, wide=False, lazy=False):
        if wide and lazy:
            _check(_lib.ntt_transform_batch(self.h, dptr, batch, FLAG_INVERSE | FLAG_WIDE_IN | FLAG_LAZY_OUT, stream))
            return
        f = _lib.ntt_inv_batch_lazy if lazy else (_lib.ntt_inv_batch_wide if wide else _lib.ntt_inv_batch)
        _check(f(self.h, dptr, batch, stream))

    def transform_strided(self, dptr, poly_stride, batch, flags=0, stream=None):
        """`batch` polynomials poly_stride words apart (ntt_transform_batch_strided); flags = FLAG_*"""
        _check(_lib.ntt_transform_batch_strided(self.h, dptr, poly_stride, batch, flags, stream))

    def transform_ptrs(self, ptrs, flags=0, stream=None):
        """one device pointer per polynomial (ntt_transform_ptrs)"""
        k = len(ptrs)
        _check(_lib.ntt_transform_ptrs(self.h, (VOIDP * k)(*ptrs), k, flags, stream))

    def transform_dev_ptrs(self, d_table, count, flags=0, stream=None):
        """the same with the pointers already in DEVICE memory (ntt_transform_dev_ptrs): d_table = device address of `count` pointers"""
        _check(_lib.ntt_transform_dev_ptrs(self.h, d_table, count, flags, stream))

    def inv_dot_dev_ptrs(self, c_tab, a_tabs, b_tabs, count, flags=0, stream=None):
        """ntt_inv_dot_dev_ptrs: c_tab = device table of `count` pointers; a_tabs / b_tabs = lists of k device tables (a broadcast b: device pointers to ONE polynomial each)"""
        k = len(a_tabs)
        _check(_lib.ntt_inv_dot_dev_ptrs(self.h, c_tab, k, (VOIDP * k)(*a_tabs), (VOIDP * k)(*b_tabs), count, flags, stream))

    def fwd_mul_dev_ptrs(self, c_tab, a_tab, b_tab, count, flags=0, stream=None):
        _check(_lib.ntt_fwd_mul_dev_ptrs(self.h, c_tab, a_tab, b_tab, count, flags, stream))

    def negacyclic_mul_dev_ptrs(self, c_tab, a_tab, b_tab, count, stream=None):
        _check(_lib.ntt_negacyclic_mul_dev_ptrs(self.h, c_tab, a_tab, b_tab, count, stream))

    def pointwise_mul(self, dc, da, db, batch, stream=None, lazy_in=False):
        f = _lib.ntt_pointwise_mul_batch_lazy if lazy_in else _lib.ntt_pointwise_mul_batch
        _check(f(self.h, dc, da, db, batch, stream))

    def negacyclic_mul(self, dc, da, db, batch, stream=None):
        _check(_lib.ntt_negacyclic_mul_batch(self.h, dc, da, db, batch, stream))

    # operands in the NTT domain (include/ntt_mi355x.h): flags = MUL_LAZY_IN | MUL_B_BROADCAST
    def inv_product(self, dc, dahat, dbhat, batch, flags=0, stream=None):
        _check(_lib.ntt_inv_product_batch(self.h, dc, dahat, dbhat, batch, flags, stream))

    def inv_dot(self, dc, dahats, dbhats, batch, flags=0, stream=None):
        k = len(dahats)
        assert k == len(dbhats)
        _check(_lib.ntt_inv_dot_batch(self.h, dc, k, (VOIDP * k)(*dahats), (VOIDP * k)(*dbhats), batch, flags, stream))

    def mul_transformed(self, dc, da, dbhat, batch, flags=0, stream=None):
        _check(_lib.ntt_mul_transformed_batch(self.h, dc, da, dbhat, batch, flags, stream))

    def fwd_mul(self, dc, da, dbhat, batch, flags=0, stream=None):
        """c^ = fwd(a) (.) b^ (MUL_ACCUMULATE: c^ += ...): the result stays in the NTT domain"""
        _check(_lib.ntt_fwd_mul_batch(self.h, dc, da, dbhat, batch, flags, stream))

    # host-array conveniences used by the parity tests
    def fwd_host(self, a, wide=False, lazy=False):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        batch = a.size // self.N
        buf = DeviceBuffer(a.size, self.device).upload(a)
        self.fwd(buf.ptr, batch, wide=wide, lazy=lazy)
        out = buf.download()
        buf.free()
        return out

    def inv_host(self, a, wide=False, lazy=False):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        batch = a.size // self.N
        buf = DeviceBuffer(a.size, self.device).upload(a)
        self.inv(buf.ptr, batch, wide=wide, lazy=lazy)
        out = buf.download()
        buf.free()
        return out

    def destroy(self):
        if self.h:
            _lib.ntt_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _plan_array(plans):
    return (VOIDP * len(plans))(*[p.h for p in plans])


def set_rns_launch(plans, mode):
    """how ntt_rns_* launch a run of compatible limbs (NTT_OPT_RNS_LAUNCH on every plan): 0 / "0" = one launch over the run
    wherever built, 1 / "1" = one launch chain per limb, None = the library's choice"""
    v = -1 if mode is None else int(mode)
    for p in plans:
        p.set_option(OPT_RNS_LAUNCH, v)


def rns_transform_ptrs(plans, ptrs, limb_stride, flags=0, stream=None):
    """one device pointer per RNS polynomial (its limbs limb_stride words apart): ntt_rns_transform_ptrs"""
    k = len(ptrs)
    _check(_lib.ntt_rns_transform_ptrs(len(plans), _plan_array(plans), (VOIDP * k)(*ptrs), k, limb_stride, flags, stream))


def rns_transform_dev_ptrs(plans, d_table, count, limb_stride, flags=0, stream=None):
    """ntt_rns_transform_dev_ptrs: the pointers (limb 0 of every RNS polynomial) in a DEVICE array"""
    _check(_lib.ntt_rns_transform_dev_ptrs(len(plans), _plan_array(plans), d_table, count, limb_stride, flags, stream))


def rns_inv_dot_dev_ptrs(plans, c_tab, a_tabs, b_tabs, count, limb_stride, flags=0, stream=None):
    k = len(a_tabs)
    _check(_lib.ntt_rns_inv_dot_dev_ptrs(len(plans), _plan_array(plans), c_tab, k, (VOIDP * k)(*a_tabs), (VOIDP * k)(*b_tabs), count, limb_stride, flags, stream))


def rns_fwd_mul_dev_ptrs(plans, c_tab, a_tab, b_tab, count, limb_stride, flags=0, stream=None):
    _check(_lib.ntt_rns_fwd_mul_dev_ptrs(len(plans), _plan_array(plans), c_tab, a_tab, b_tab, count, limb_stride, flags, stream))


def rns_negacyclic_mul_dev_ptrs(plans, c_tab, a_tab, b_tab, count, limb_stride, stream=None):
    _check(_lib.ntt_rns_negacyclic_mul_dev_ptrs(len(plans), _plan_array(plans), c_tab, a_tab, b_tab, count, limb_stride, stream))


def batch_major(plans):
    """(limb_stride, poly_stride) of SURVEY 8(d)'s [batch][prime][N] layout for this limb list"""
    return (plans[0].N, len(plans) * plans[0].N)


def rns_fwd(plans, dptr, batch, stream=None, layout=None):
    """limbs laid out [limb][batch][N]; layout = (limb_stride, poly_stride) in words for any other placement"""
    if layout: _check(_lib.ntt_rns_fwd_batch_strided(len(plans), _plan_array(plans), dptr, layout[0], layout[1], batch, stream))
    else: _check(_lib.ntt_rns_fwd_batch(len(plans), _plan_array(plans), dptr, batch, stream))


def rns_inv(plans, dptr, batch, stream=None, layout=None):
    if layout: _check(_lib.ntt_rns_inv_batch_strided(len(plans), _plan_array(plans), dptr, layout[0], layout[1], batch, stream))
    else: _check(_lib.ntt_rns_inv_batch(len(plans), _plan_array(plans), dptr, batch, stream))


def rns_negacyclic_mul(plans, dc, da, db, batch, stream=None, layout=None):
    if layout: _check(_lib.ntt_rns_negacyclic_mul_batch_strided(len(plans), _plan_array(plans), dc, da, db, layout[0], layout[1], batch, stream))
    else: _check(_lib.ntt_rns_negacyclic_mul_batch(len(plans), _plan_array(plans), dc, da, db, batch, stream))


def rns_inv_dot(plans, dc, dahats, dbhats, batch, flags=0, stream=None, layout=None):
    """operands laid out [limb][batch][N] (a broadcast b^: [limb][N])"""
    k = len(dahats)
    if layout:
        _check(_lib.ntt_rns_inv_dot_batch_strided(len(plans), _plan_array(plans), dc, k, (VOIDP * k)(*dahats), (VOIDP * k)(*dbhats), layout[0],
                                                  layout[1], batch, flags, stream))
    else:
        _check(_lib.ntt_rns_inv_dot_batch(len(plans), _plan_array(plans), dc, k, (VOIDP * k)(*dahats), (VOIDP * k)(*dbhats), batch,
                                          flags, stream))


def rns_fwd_mul(plans, dc, da, dbhat, batch, flags=0, stream=None, layout=None):
    if layout: _check(_lib.ntt_rns_fwd_mul_batch_strided(len(plans), _plan_array(plans), dc, da, dbhat, layout[0], layout[1], batch, flags, stream))
    else: _check(_lib.ntt_rns_fwd_mul_batch(len(plans), _plan_array(plans), dc, da, dbhat, batch, flags, stream))


def rns_mul_transformed(plans, dc, da, dbhat, batch, flags=0, stream=None, layout=None):
    if layout:
        _check(_lib.ntt_rns_mul_transformed_batch_strided(len(plans), _plan_array(plans), dc, da, dbhat, layout[0], layout[1], batch, flags, stream))
    else:
        _check(_lib.ntt_rns_mul_transformed_batch(len(plans), _plan_array(plans), dc, da, dbhat, batch, flags, stream))


def batch_multi(plans, dptrs, batches, inverse=False):
    n = len(plans)
    ph = (VOIDP * n)(*[p.h for p in plans])
    dp = (VOIDP * n)(*dptrs)
    bt = (C.c_uint64 * n)(*batches)
    _check(_lib.ntt_batch_multi(n, ph, dp, bt, int(inverse)))


def rns_mul_multi(plan_sets, dcs, das, dbs, batches):
    """RNS products across devices: plan_sets[g] = the limbs' plans on shard g's device"""
    ndev, nl = len(plan_sets), len(plan_sets[0])
    ph = (VOIDP * (ndev * nl))(*[p.h for ps in plan_sets for p in ps])
    _check(_lib.ntt_rns_mul_multi(ndev, nl, ph, (VOIDP * ndev)(*dcs), (VOIDP * ndev)(*das), (VOIDP * ndev)(*dbs),
                                  (C.c_uint64 * ndev)(*batches)))


# --------------------------------------------------------------------------
# reference-signature entry points on host numpy arrays (in place)
# --------------------------------------------------------------------------
def _mulop(op, con):
    m = MulOp()
    m.op_lo, m.op_hi, m.con_lo, m.con_hi = op & (2**64 - 1), op >> 64, con & (2**64 - 1), con >> 64
    return m


def _reduce(a, q, k):
    """the header-inline final reduction of the reference wrappers"""
    for m in ((4, 2, 1) if k == 8 else (2, 1)):
        np.subtract(a, np.uint64(m * q), out=a, where=a >= np.uint64(m * q))


def fwd_ntt_ref_harvey(a, N, q, w, w_con):
    _lib.fwd_ntt_ref_harvey_lazy(_ptr(a), N, q, _ptr(w), _ptr(w_con))
    _reduce(a, q, 4)


def fwd_ntt_ref_harvey_dbl(a1, a2, N, q, w, w_con):
    _lib.fwd_ntt_ref_harvey_lazy_dbl(_ptr(a1), _ptr(a2), N, q, _ptr(w), _ptr(w_con))
    _reduce(a1, q, 4)
    _reduce(a2, q, 4)


def inv_ntt_ref_harvey(a, N, q, n_inv, n_inv_con, word_size, w, w_con):
    _lib.inv_ntt_ref_harvey(_ptr(a), N, q, _mulop(n_inv, n_inv_con), word_size, _ptr(w), _ptr(w_con))


def fwd_ntt_radix4(a, N, q, w, w_con):
    _lib.fwd_ntt_radix4_lazy(_ptr(a), N, q, _ptr(w), _ptr(w_con))
    _reduce(a, q, 8)


def fwd_ntt_radix4x4(a, N, q, w, w_con):
    _lib.fwd_ntt_radix4x4_lazy(_ptr(a), N, q, _ptr(w), _ptr(w_con))
    _reduce(a, q, 8)


def inv_ntt_radix4(a, N, q, n_inv, n_inv_con, w, w_con):
    _lib.inv_ntt_radix4(_ptr(a), N, q, _mulop(n_inv, n_inv_con), _ptr(w), _ptr(w_con))


def fwd_ntt_seal(a, N, q, w, w_con):
    _lib.fwd_ntt_seal_lazy(_ptr(a), N, q, _ptr(w), _ptr(w_con))
    _reduce(a, q, 4)


def inv_ntt_seal(a, N, q, n_inv, n_inv_con, w, w_con):
    _lib.inv_ntt_seal(_ptr(a), N, q, n_inv, n_inv_con, _ptr(w), _ptr(w_con))
