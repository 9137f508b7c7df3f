"""ctypes binding of tests/emu/libntt_emu.so: the kernel templates of csrc/
compiled for the CPU (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDIR = os.path.join(ROOT, "tests", "emu")
U64P = C.POINTER(C.c_uint64)


class Emu:
    def __init__(self):
        path = os.path.join(EDIR, "libntt_emu.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-j8", "-C", EDIR])
        L = self.lib = C.CDLL(path)
        L.emu_transform.argtypes = [U64P, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64] + [C.c_int] * 5
        L.emu_plan_info.argtypes = [C.c_int, U64P]
        L.emu_pointwise.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_uint64, C.c_int]
        L.emu_chk_stats.argtypes = [U64P, C.c_int]
        L.emu_pointwise_lazy.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_uint64, C.c_int]
        L.emu_set_lazy.argtypes = [C.c_int]
        L.emu_set_poly_stride.argtypes = [C.c_uint64]
        L.emu_set_poly_table.argtypes = [C.c_void_p]
        L.emu_set_one_pass.argtypes = [C.c_int]
        L.emu_set_operand_stride.argtypes = [C.c_uint64]
        L.emu_set_u64x_worst.argtypes = [C.c_int]
        L.emu_u64x_schedule.restype = C.c_uint32
        L.emu_u64x_schedule.argtypes = [C.c_int] * 3
        L.emu_set_product_both.argtypes = [C.c_int]
        L.emu_fused_product14.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.emu_expand_radix4.argtypes = [U64P, U64P, C.c_uint64, C.c_uint64]
        L.emu_fused_product_chk.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]
        L.emu_fwd_mul.argtypes = [U64P, U64P, U64P, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_team_decode.argtypes = [C.c_uint32] * 7 + [C.POINTER(C.c_uint32)]
        L.emu_fwd_r4x4_layers.argtypes = [U64P, C.c_uint64, C.c_uint64, U64P, U64P]
        L.emu_inv_dot.argtypes = [U64P, C.c_int, U64P, U64P, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]

    def transform(self, a, m, q, root, arith, inverse=False, generic=False, wide=False, ksh=-1, lazy=False):
        """arith: 0 = integer radix-2, 1 = FP64, 2 = checked FP64, 3 = integer radix-4 (expanded table), 4 / 5 = FP64 up to 2^52
        (5: checked), 6 = the wide integer policy ArithU64X<ksh>, checked"""
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.emu_set_lazy(int(lazy))
        rc = self.lib.emu_transform(a.ctypes.data_as(U64P), a.size >> m, m, q, root, arith, int(inverse),
                                    int(generic), int(wide), ksh)
        self.lib.emu_set_lazy(0)
        return rc, a

    def set_one_pass(self, mode):
        """N = 2^15, FP64 policies: 1 / -1 = the one-pass route of onepass_kernel (the library's default), 0 = the two-pass route"""
        self.lib.emu_set_one_pass(int(mode))

    def transform_ptrs(self, buf, offsets, m, q, root, arith, inverse=False, generic=False, ksh=-1):
        """a pointer batch: the polynomials of `buf` that start at the given word offsets (any order, any spacing), in place, through
        the table addressing of the transform kernels (csrc/ntt_core.h poly_offset: entries are the polynomials' addresses, the data
        pointer is null); returns the status"""
        assert buf.dtype == np.uint64 and buf.flags["C_CONTIGUOUS"]
        table = np.array([buf.ctypes.data + 8 * int(o) for o in offsets], dtype=np.uint64)
        self.lib.emu_set_poly_table(table.ctypes.data)
        rc = self.lib.emu_transform(C.cast(0, U64P), len(offsets), m, q, root, arith, int(inverse), int(generic), 0, ksh)
        self.lib.emu_set_poly_table(None)
        return rc

    def transform_limb(self, buf, limb, nlimbs, batch, m, q, root, arith, inverse=False, ksh=-1):
        """limb `limb` of a [batch][nlimbs][N] buffer (SURVEY 8(d)'s layout: polynomials nlimbs * N words apart), in place, through
        the block_offset addressing of the kernels (csrc/ntt_core.h); returns the status"""
        n = 1 << m
        assert buf.dtype == np.uint64 and buf.flags["C_CONTIGUOUS"] and buf.size == batch * nlimbs * n
        self.lib.emu_set_poly_stride(nlimbs * n)
        ptr = C.cast(buf.ctypes.data + 8 * limb * n, U64P)
        rc = self.lib.emu_transform(ptr, batch, m, q, root, arith, int(inverse), 0, 0, ksh)
        self.lib.emu_set_poly_stride(0)
        return rc

    def inv_dot_limb(self, out, a, b, limb, nlimbs, k, batch, m, q, root, arith=1, lazy=False):
        """limb `limb` of c = inv(sum_i a_i (.) b_i) with a, b laid out [k][batch][nlimbs][N] and out [batch][nlimbs][N]"""
        n = 1 << m
        self.lib.emu_set_poly_stride(nlimbs * n)
        self.lib.emu_set_operand_stride(batch * nlimbs * n)
        off = 8 * limb * n
        rc = self.lib.emu_inv_dot(C.cast(out.ctypes.data + off, U64P), k, C.cast(a.ctypes.data + off, U64P), C.cast(b.ctypes.data + off, U64P),
                                  batch, m, q, root, arith, int(lazy), 0)
        self.lib.emu_set_poly_stride(0)
        self.lib.emu_set_operand_stride(0)
        return rc

    def fwd_mul_limb(self, out, a, bhat, limb, nlimbs, batch, m, q, root, arith=1, acc=False):
        """limb `limb` of c^ (+)= fwd(a) (.) b^, all three laid out [batch][nlimbs][N] (a is scratch above 2^14)"""
        n = 1 << m
        self.lib.emu_set_poly_stride(nlimbs * n)
        off = 8 * limb * n
        rc = self.lib.emu_fwd_mul(C.cast(out.ctypes.data + off, U64P), C.cast(a.ctypes.data + off, U64P), C.cast(bhat.ctypes.data + off, U64P),
                                  batch, m, q, root, arith, 0, 0, int(acc))
        self.lib.emu_set_poly_stride(0)
        return rc

    def fused_product14(self, ahat, b, q, root, a_lazy=False, chk=False, both=False):
        """inv(fwd(b) * ahat) as the fused product kernel computes it (N = 2^14); both: `ahat` holds a's COEFFICIENTS and
        the kernel takes them through the forward stages too (its BOTH variant)"""
        ahat = np.ascontiguousarray(ahat, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(b)
        self.lib.emu_set_product_both(int(both))
        rc = self.lib.emu_fused_product14(out.ctypes.data_as(U64P), ahat.ctypes.data_as(U64P), b.ctypes.data_as(U64P),
                                          b.size >> 14, q, root, int(a_lazy), int(chk))
        self.lib.emu_set_product_both(0)
        return rc, out

    def fused_product_chk(self, ahat, b, m, q, root, both=False):
        """inv(fwd(b) * ahat) as the product kernels for N = 2^8 .. 2^13 compute it, checked policy (both: as above)"""
        ahat = np.ascontiguousarray(ahat, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(b)
        self.lib.emu_set_product_both(int(both))
        rc = self.lib.emu_fused_product_chk(out.ctypes.data_as(U64P), ahat.ctypes.data_as(U64P), b.ctypes.data_as(U64P),
                                            b.size >> m, m, q, root)
        self.lib.emu_set_product_both(0)
        return rc, out

    def inv_dot(self, a_list, b_list, m, q, root, arith=1, lazy=False, bcast=False):
        """inv(sum_i a_i (.) b_i) for operands in the NTT domain as dot_inv_kernel (+ the inverse's column passes above 2^14)
        computes it.  arith 0: integer radix-2, 1: the checked FP64 policy of q's class."""
        k = len(a_list)
        a = np.ascontiguousarray(np.concatenate(a_list), dtype=np.uint64)
        b = np.ascontiguousarray(np.concatenate(b_list), dtype=np.uint64)
        batch = a_list[0].size >> m
        out = np.zeros(batch << m, dtype=np.uint64)
        rc = self.lib.emu_inv_dot(out.ctypes.data_as(U64P), k, a.ctypes.data_as(U64P), b.ctypes.data_as(U64P), batch, m, q, root,
                                  arith, int(lazy), int(bcast))
        return rc, out

    def fwd_mul(self, a, bhat, m, q, root, arith=1, lazy=False, bcast=False, acc=None):
        """c^ = fwd(a) (.) b^ (+ acc) as fwd_mul_kernel computes it (checked FP64 policy, or arith 0: integer)"""
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        bhat = np.ascontiguousarray(bhat, dtype=np.uint64)
        out = np.zeros_like(a) if acc is None else np.ascontiguousarray(acc, dtype=np.uint64).copy()
        rc = self.lib.emu_fwd_mul(out.ctypes.data_as(U64P), a.ctypes.data_as(U64P), bhat.ctypes.data_as(U64P), a.size >> m, m, q, root,
                                  arith, int(lazy), int(bcast), int(acc is not None))
        return rc, out

    def team_decode(self, k, q, total, lag, n0, n1, n2):
        """queue entry k of queue q -> (stop, valid, pass, item, polynomial): csrc/ntt_core.h team_decode, the function the
        XCD-local kernels call"""
        out = (C.c_uint32 * 5)()
        self.lib.emu_team_decode(k, q, total, lag, n0, n1, n2, out)
        return tuple(int(x) for x in out)

    def fwd_r4x4_layers(self, a, q, e, econ):
        """fwd_ntt_radix4x4_lazy at log2 N = 4k+3 through csrc/ntt_core.h r4x4_layer_r4 / r4x4_layer_r2 (the functions the
        layer kernels of ntt_host.hip call), on the caller's expanded table"""
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        e, econ = np.ascontiguousarray(e, dtype=np.uint64), np.ascontiguousarray(econ, dtype=np.uint64)
        assert e.size == 2 * a.size == econ.size
        self.lib.emu_fwd_r4x4_layers(a.ctypes.data_as(U64P), a.size, q, e.ctypes.data_as(U64P), econ.ctypes.data_as(U64P))
        return a

    def expand_radix4(self, w, q):
        w = np.ascontiguousarray(w, dtype=np.uint64)
        e = np.zeros(2 * w.size, dtype=np.uint64)
        self.lib.emu_expand_radix4(e.ctypes.data_as(U64P), w.ctypes.data_as(U64P), w.size, q)
        return e

    def plan_info(self, logn):
        v = np.zeros(10, dtype=np.uint64)
        assert self.lib.emu_plan_info(logn, v.ctypes.data_as(U64P)) == 0
        keys = ("NG", "R0", "RL", "T", "ROW", "LDS_ELEMS", "wave_local", "fmask", "imask", "conflict_free")
        return dict(zip(keys, [int(x) for x in v]))

    def u64x_schedule(self, inverse, nstages, k):
        """csrc/ntt_arith.h u64x_schedule: bit s set = the stage processed at position s folds the growing operand"""
        return int(self.lib.emu_u64x_schedule(int(inverse), nstages, k))

    def set_u64x_worst(self, on):
        """arith=6 (checked ArithU64X): products and reduce_any return the largest representative their claims allow, so
        that every value grows as fast as the compile-time schedule assumes"""
        self.lib.emu_set_u64x_worst(int(on))

    def chk_stats(self, reset=True):
        """(violations, max |v|/q, max |product|/q) recorded by the checked FP64 policy (arith=2)"""
        v = np.zeros(3, dtype=np.uint64)
        self.lib.emu_chk_stats(v.ctypes.data_as(U64P), int(reset))
        return int(v[0]), v[1] / 1e6, v[2] / 1e6

    def pointwise_lazy(self, a, b, q, arith):
        c = np.zeros_like(a)
        rc = self.lib.emu_pointwise_lazy(c.ctypes.data_as(U64P), a.ctypes.data_as(U64P), b.ctypes.data_as(U64P),
                                         a.size, q, arith)
        return rc, c

    def pointwise(self, a, b, q, arith):
        c = np.zeros_like(a)
        rc = self.lib.emu_pointwise(c.ctypes.data_as(U64P), a.ctypes.data_as(U64P), b.ctypes.data_as(U64P),
                                    a.size, q, arith)
        return rc, c
