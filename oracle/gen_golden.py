#!/usr/bin/env python3
"""Pin the oracle against the compiled reference and freeze golden fixtures.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where /root/reference
exists and `make -C oracle ref` has produced oracle/_ref/libntt_ref.so (the
reference's own C sources compiled in place).  It

  1. replays the reference's correctness procedure (tests/test_correctness.c:256-285):
     cases 0..18 in order, input = unseeded glibc rand() % q, out = fwd_ntt_ref_harvey;
  2. checks that every reference variant agrees (radix-4, radix-4x4, seal, dbl, inverses);
  3. checks the oracle restatement bit-for-bit against the reference on those inputs,
     on full-range splitmix inputs, on edge inputs and on every table;
  4. writes tests/golden/kat.json (parameters, digests, heads),
     tests/golden/case0_vectors.json (complete N=256 input/output vectors) and
     tests/golden/lazy_words.json (digests of the three *_lazy formulations' UNREDUCED outputs on
     the uniform input of every case: what the reference returns before its header-inline reduction).

The fixtures are data (inputs / expected outputs); no reference text is stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
U64P = C.POINTER(C.c_uint64)
UNI_SEED = 0x5EED5EED


def ptr(a):
    return a.ctypes.data_as(U64P)


def load():
    ref = C.CDLL(os.path.join(HERE, "_ref", "libntt_ref.so"))
    orc = C.CDLL(os.path.join(HERE, "libntt_oracle.so"))
    ref.ref_case_table.restype = U64P
    ref.ref_case_table.argtypes = [C.c_int, C.c_int]
    ref.ref_case_ninv_con.restype = C.c_uint64
    ref.ref_case_ninv_con.argtypes = [C.c_int]
    ref.ref_case_params.argtypes = [C.c_int, U64P]
    ref.ref_random_buf.argtypes = [U64P, C.c_uint64, C.c_uint64]
    ref.ref_fwd.argtypes = [C.c_int, C.c_int, U64P]
    ref.ref_fwd_lazy.argtypes = [C.c_int, C.c_int, U64P]
    ref.ref_inv.argtypes = [C.c_int, C.c_int, U64P]
    ref.ref_fwd_dbl.argtypes = [C.c_int, U64P, U64P]
    orc.orc_fnv1a64.restype = C.c_uint64
    orc.orc_fnv1a64.argtypes = [U64P, C.c_uint64]
    orc.orc_min_root.restype = C.c_uint64
    orc.orc_min_root.argtypes = [C.c_uint64, C.c_uint64]
    orc.orc_invmod.restype = C.c_uint64
    orc.orc_invmod.argtypes = [C.c_uint64, C.c_uint64]
    orc.orc_ctx_new.restype = C.c_void_p
    orc.orc_ctx_new.argtypes = [C.c_uint64] * 3
    orc.orc_ctx_free.argtypes = [C.c_void_p]
    for f in ("orc_fwd_r2", "orc_fwd_r4", "orc_fwd_r2_lazy", "orc_fwd_r4_lazy", "orc_fwd_r4x4_lazy"):
        getattr(orc, f).argtypes = [U64P, C.c_uint64, C.c_uint64, U64P, U64P]
    orc.orc_inv_r2.argtypes = [U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                               C.c_uint, U64P, U64P]
    orc.orc_inv_r4.argtypes = [U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                               U64P, U64P]
    orc.orc_fill_uniform.argtypes = [U64P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    orc.orc_fwd_naive.argtypes = [U64P, U64P, C.c_uint64, C.c_uint64, C.c_uint64]
    return ref, orc


class OrcCtx(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("N", "q", "root", "root_inv", "ninv", "ninv_con")] + \
               [("m", C.c_uint)] + \
               [(n, U64P) for n in ("w", "wcon", "winv", "winv_con", "e", "econ", "einv", "einv_con")]


def fnv(orc, a):
    return "%016x" % orc.orc_fnv1a64(ptr(a), a.size)


def main():
    ref, orc = load()
    ref.ref_init()
    ncases = ref.ref_num_cases()
    assert ncases == 19, ncases
    out = {"procedure": "cases in order; in = glibc rand()%q (unseeded); out = fwd_ntt_ref_harvey; "
                        "fnv = FNV-1a-64 over little-endian bytes",
           "uniform_seed": UNI_SEED, "cases": []}
    vectors = {}
    lazy = {"procedure": "uniform input of kat.json (orc_fill_uniform, uniform_seed, stream case << 32); digest = FNV-1a-64 of the "
                         "words fwd_ntt_ref_harvey_lazy / fwd_ntt_radix4_lazy / fwd_ntt_radix4x4_lazy of the compiled reference "
                         "leave in the buffer (expanded tables for the radix-4 forms)", "cases": []}
    for i in range(ncases):
        p = (C.c_uint64 * 5)()
        ref.ref_case_params(i, C.cast(p, U64P))
        m, q, w, w_inv, n_inv = [int(x) for x in p]
        n = 1 << m
        ninv_con = int(ref.ref_case_ninv_con(i))
        # --- parameter sanity (tests/test_cases.h:113-142 rule) ---
        assert pow(w, n, q) == q - 1 and (w * w_inv) % q == 1 and (n * n_inv) % q == 1
        assert orc.orc_min_root(q, n) == w, (i, orc.orc_min_root(q, n), w)
        # --- tables: oracle vs reference ---
        cx = C.cast(orc.orc_ctx_new(n, q, w), C.POINTER(OrcCtx)).contents
        assert cx.ninv == n_inv and cx.ninv_con == ninv_con and cx.root_inv == w_inv
        tbl_fnv = {}
        names = ["w", "wcon", "winv", "winv_con", "e", "econ", "einv", "einv_con"]
        for which, name in enumerate(names):
            ln = n if which < 4 else 2 * n
            rt = np.ctypeslib.as_array(ref.ref_case_table(i, which), (ln,))
            ot = np.ctypeslib.as_array(getattr(cx, name), (ln,))
            assert np.array_equal(rt, ot), (i, name)
            tbl_fnv[name] = fnv(orc, np.ascontiguousarray(rt))
        # --- the reference's own input stream ---
        a = np.zeros(n, dtype=np.uint64)
        ref.ref_random_buf(ptr(a), n, q)
        ref_out = a.copy()
        ref.ref_fwd(i, 0, ptr(ref_out))
        for variant in (1, 2, 3):
            t = a.copy()
            ref.ref_fwd(i, variant, ptr(t))
            assert np.array_equal(t, ref_out), (i, "ref variant", variant)
        t, t2 = a.copy(), a.copy()
        ref.ref_fwd_dbl(i, ptr(t), ptr(t2))
        assert np.array_equal(t, ref_out) and np.array_equal(t2, ref_out)
        for variant in (0, 1, 3):
            t = ref_out.copy()
            ref.ref_inv(i, variant, ptr(t))
            assert np.array_equal(t, a), (i, "ref inv", variant)

        def check_oracle(inp, expect):
            t = inp.copy()
            orc.orc_fwd_r2(ptr(t), n, q, cx.w, cx.wcon)
            assert np.array_equal(t, expect), (i, "orc r2")
            t = inp.copy()
            orc.orc_fwd_r4(ptr(t), n, q, cx.e, cx.econ)
            assert np.array_equal(t, expect), (i, "orc r4")
            t = expect.copy()
            orc.orc_inv_r2(ptr(t), n, q, cx.ninv, cx.ninv_con, 64, cx.winv, cx.winv_con)
            assert np.array_equal(t, inp % np.uint64(q)), (i, "orc inv r2")
            t = expect.copy()
            orc.orc_inv_r4(ptr(t), n, q, cx.ninv, cx.ninv_con, cx.einv, cx.einv_con)
            assert np.array_equal(t, inp % np.uint64(q)), (i, "orc inv r4")
            # lazy outputs: reference and oracle must agree exactly too (same algorithm)
            digests = {}
            for variant, fn, tb, tc in ((0, orc.orc_fwd_r2_lazy, cx.w, cx.wcon),
                                        (1, orc.orc_fwd_r4_lazy, cx.e, cx.econ),
                                        (2, orc.orc_fwd_r4x4_lazy, cx.e, cx.econ)):
                r, o = inp.copy(), inp.copy()
                ref.ref_fwd_lazy(i, variant, ptr(r))
                fn(ptr(o), n, q, tb, tc)
                assert np.array_equal(r, o), (i, "lazy", variant)
                # radix-4 butterflies leave [0,8q); a trailing radix-2 stage (log2 N odd, but not the 4k+3 order of
                # the radix-4x4 formulation, which ends on a radix-4 layer) leaves [0,4q)
                ends_r2 = variant == 0 or (m % 2 == 1 and not (variant == 2 and m % 4 == 3))
                assert int(r.max()) < (4 * q if ends_r2 else 8 * q)
                assert np.array_equal(r % np.uint64(q), expect)
                digests[("ref_harvey", "radix4", "radix4x4")[variant]] = fnv(orc, r)
            return digests

        check_oracle(a, ref_out)
        # --- full-range uniform inputs (the reference never exercises these) ---
        u = np.zeros(n, dtype=np.uint64)
        orc.orc_fill_uniform(ptr(u), n, q, UNI_SEED, i << 32)
        u_out = u.copy()
        ref.ref_fwd(i, 0, ptr(u_out))
        lazy_fnv = check_oracle(u, u_out)
        lazy["cases"].append({"case": i, "m": m, "q": q, "uni_in_fnv": fnv(orc, u), "lazy_out_fnv": lazy_fnv,
                              "radix4x4_differs_from_radix4": lazy_fnv["radix4"] != lazy_fnv["radix4x4"]})
        t = u_out.copy()
        ref.ref_inv(i, 1, ptr(t))
        assert np.array_equal(t, u)
        # --- edge inputs ---
        edges = {"zero": np.zeros(n, dtype=np.uint64),
                 "qm1": np.full(n, q - 1, dtype=np.uint64),
                 "delta0": np.eye(1, n, 0, dtype=np.uint64)[0],
                 "delta1": np.eye(1, n, 1, dtype=np.uint64)[0],
                 "deltaN1": np.eye(1, n, n - 1, dtype=np.uint64)[0]}
        edge_fnv = {}
        for name, e in edges.items():
            e = np.ascontiguousarray(e)
            r = e.copy()
            ref.ref_fwd(i, 0, ptr(r))
            check_oracle(e, r)
            edge_fnv[name] = fnv(orc, r)
        # --- definition check on small sizes ---
        if m <= 10:
            nv = np.zeros(n, dtype=np.uint64)
            orc.orc_fwd_naive(ptr(nv), ptr(u), n, q, w)
            assert np.array_equal(nv, u_out), (i, "naive definition")
        out["cases"].append({
            "case": i, "m": m, "q": q, "w": w, "w_inv": w_inv, "n_inv": n_inv, "n_inv_con": ninv_con,
            "in_head": [int(x) for x in a[:3]], "in_fnv": fnv(orc, a),
            "out_head": [int(x) for x in ref_out[:3]], "out_fnv": fnv(orc, ref_out),
            "uni_in_fnv": fnv(orc, u), "uni_out_head": [int(x) for x in u_out[:3]],
            "uni_out_fnv": fnv(orc, u_out), "edge_out_fnv": edge_fnv, "table_fnv": tbl_fnv,
            "w_powers_1": int(cx.w[1]), "w_powers_con_1": int(cx.wcon[1])})
        if i == 0:
            vectors = {"case": 0, "m": m, "q": q, "w": w,
                       "rand_in": [int(x) for x in a], "rand_out": [int(x) for x in ref_out],
                       "uni_in": [int(x) for x in u], "uni_out": [int(x) for x in u_out]}
        orc.orc_ctx_free(C.byref(cx))
        print("case %2d m=%2d q=%#x pinned: in %s out %s" % (i, m, q, out["cases"][-1]["in_fnv"],
                                                           out["cases"][-1]["out_fnv"]))
    gd = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gd, exist_ok=True)
    with open(os.path.join(gd, "kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(gd, "case0_vectors.json"), "w") as f:
        json.dump(vectors, f)
    with open(os.path.join(gd, "lazy_words.json"), "w") as f:
        json.dump(lazy, f, indent=1)
    print("wrote", gd)


if __name__ == "__main__":
    sys.exit(main())
