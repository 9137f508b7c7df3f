"""ctypes binding of oracle/libntt_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
U64P = C.POINTER(C.c_uint64)


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


class OrcCtx(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("N", "q", "root", "root_inv", "ninv", "ninv_con")] + \
               [("m", C.c_uint)] + \
               [(n, U64P) for n in ("w", "wcon", "winv", "winv_con", "e", "econ", "einv", "einv_con")]


class Ctx:
    def __init__(self, orc, N, q, root):
        self.orc, self.N, self.q, self.root = orc, N, q, root
        self.h = orc.lib.orc_ctx_new(N, q, root)
        assert self.h
        self.c = C.cast(self.h, C.POINTER(OrcCtx)).contents

    def table(self, name):
        n = self.N if name in ("w", "wcon", "winv", "winv_con") else 2 * self.N
        return np.ctypeslib.as_array(getattr(self.c, name), (n,)).copy()

    def _batch(self, fn, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        assert a.size % self.N == 0
        fn(ptr(a), a.size // self.N, self.h)
        return a

    def fwd(self, a):
        return self._batch(self.orc.lib.orc_fwd_r2_batch, a)

    def fwd_r4(self, a):
        return self._batch(self.orc.lib.orc_fwd_r4_batch, a)

    def _lazy(self, fn, tab, con, a):
        """single-polynomial *_lazy entry points of the oracle, applied to every polynomial of a"""
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        assert a.size % self.N == 0
        t, c = self.table(tab), self.table(con)
        for p in range(a.size // self.N):
            v = a[p * self.N:(p + 1) * self.N]
            fn(ptr(v), self.N, self.q, ptr(t), ptr(c))
        return a

    def fwd_lazy(self, a):
        """fwd_ntt_ref_harvey_lazy: outputs in [0,4q) (oracle restatement of src/ntt_reference.c:11-31)"""
        return self._lazy(self.orc.lib.orc_fwd_r2_lazy, "w", "wcon", a)

    def fwd_r4_lazy(self, a):
        """fwd_ntt_radix4_lazy: outputs in [0,8q) (src/ntt_radix4.c:27-62)"""
        return self._lazy(self.orc.lib.orc_fwd_r4_lazy, "e", "econ", a)

    def fwd_r4x4_lazy(self, a):
        """fwd_ntt_radix4x4_lazy (src/ntt_radix4x4.c:41-114): fwd_r4_lazy's words unless log2 N = 4k+3"""
        return self._lazy(self.orc.lib.orc_fwd_r4x4_lazy, "e", "econ", a)

    def inv(self, a):
        return self._batch(self.orc.lib.orc_inv_r2_batch, a)

    def inv_r4(self, a):
        return self._batch(self.orc.lib.orc_inv_r4_batch, a)

    def __del__(self):
        try:
            self.orc.lib.orc_ctx_free(self.h)
        except Exception:
            pass


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(ODIR, "libntt_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ODIR, "all"])
        L = self.lib = C.CDLL(path)
        L.orc_ctx_new.restype = C.c_void_p
        L.orc_ctx_new.argtypes = [C.c_uint64] * 3
        L.orc_ctx_free.argtypes = [C.c_void_p]
        for f in ("orc_fwd_r2_batch", "orc_fwd_r4_batch", "orc_inv_r2_batch", "orc_inv_r4_batch"):
            getattr(L, f).argtypes = [U64P, C.c_uint64, C.c_void_p]
        for f in ("orc_fwd_r2_lazy", "orc_fwd_r4_lazy", "orc_fwd_r4x4_lazy"):
            getattr(L, f).argtypes = [U64P, C.c_uint64, C.c_uint64, U64P, U64P]
        L.orc_fill_uniform.argtypes = [U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [U64P, C.c_uint64]
        L.orc_min_root.restype = C.c_uint64
        L.orc_min_root.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_find_prime.restype = C.c_uint64
        L.orc_find_prime.argtypes = [C.c_uint, C.c_uint64, C.c_uint]
        L.orc_is_prime.argtypes = [C.c_uint64]
        L.orc_pointwise.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_uint64]
        L.orc_negacyclic_schoolbook.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_uint64]
        L.orc_fwd_naive.argtypes = [U64P, U64P, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_powmod.restype = C.c_uint64
        L.orc_powmod.argtypes = [C.c_uint64] * 3

    def ctx(self, N, q, root):
        return Ctx(self, N, q, root)

    def fill_uniform(self, n, q, seed, offset=0):
        a = np.zeros(n, dtype=np.uint64)
        self.lib.orc_fill_uniform(ptr(a), n, q, seed, offset)
        return a

    def fnv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return "%016x" % self.lib.orc_fnv1a64(ptr(a), a.size)

    def min_root(self, q, N):
        return int(self.lib.orc_min_root(q, N))

    def find_prime(self, bits, N, skip=0):
        return int(self.lib.orc_find_prime(bits, N, skip))

    def pointwise(self, a, b, q):
        c = np.zeros_like(a)
        self.lib.orc_pointwise(ptr(c), ptr(a), ptr(b), a.size, q)
        return c

    def dot(self, a_list, b_list, q, N=None, bcast=False):
        """sum_i a_i (.) b_i mod q, element-wise (the reference's fast_mul_mod_q semantics per product, include/internal/
        fast_mul_operators.h:56-60, here with 128-bit products); bcast: every b_i is one polynomial of N words shared by
        the batch.  Operand words may be lazy (any 64-bit value): they are reduced first."""
        acc = None
        for a, b in zip(a_list, b_list):
            a = np.ascontiguousarray(a, dtype=np.uint64) % np.uint64(q)
            b = np.ascontiguousarray(b, dtype=np.uint64) % np.uint64(q)
            if bcast:
                b = np.tile(b, a.size // N)
            t = self.pointwise(a, b, q)
            acc = t if acc is None else (acc + t) % np.uint64(q)
        return acc

    def schoolbook(self, a, b, N, q):
        c = np.zeros(N, dtype=np.uint64)
        self.lib.orc_negacyclic_schoolbook(ptr(c), ptr(a), ptr(b), N, q)
        return c

    def fwd_naive(self, a, N, q, root):
        out = np.zeros(N, dtype=np.uint64)
        self.lib.orc_fwd_naive(ptr(out), ptr(a), N, q, root)
        return out

    def checksum(self, poly):
        """host restatement of ntt_poly_checksum: sum_i splitmix64(i)*a[i] mod 2^64"""
        n = poly.size
        idx = np.arange(n, dtype=np.uint64)
        with np.errstate(over="ignore"):
            x = idx + np.uint64(0x9e3779b97f4a7c15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
            x = x ^ (x >> np.uint64(31))
            return int((x * poly).sum(dtype=np.uint64))
