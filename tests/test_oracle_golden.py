"""CPU: the travelling oracle reproduces every golden vector captured from the
compiled reference (oracle/gen_golden.py), without the reference."""
import numpy as np
import pytest

UNI_SEED = 0x5EED5EED


def _ctx(oracle, c):
    return oracle.ctx(1 << c["m"], c["q"], c["w"])


def test_nineteen_cases_present(kat):
    assert len(kat["cases"]) == 19
    assert [c["m"] for c in kat["cases"]] == [8, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14, 14, 14, 14, 15, 15, 16, 16, 17]


@pytest.mark.parametrize("i", range(19))
def test_tables_and_parameters(kat, oracle, i):
    c = kat["cases"][i]
    n, q = 1 << c["m"], c["q"]
    cx = _ctx(oracle, c)
    assert cx.c.root_inv == c["w_inv"] and cx.c.ninv == c["n_inv"] and cx.c.ninv_con == c["n_inv_con"]
    assert oracle.min_root(q, n) == c["w"]  # "minimum root" rule, tests/test_cases.h:113-142
    for name, digest in c["table_fnv"].items():
        assert oracle.fnv(cx.table(name)) == digest, name
    assert int(cx.table("w")[1]) == c["w_powers_1"] and int(cx.table("wcon")[1]) == c["w_powers_con_1"]


@pytest.mark.parametrize("i", range(19))
def test_uniform_kat(kat, oracle, i):
    c = kat["cases"][i]
    n, q = 1 << c["m"], c["q"]
    cx = _ctx(oracle, c)
    u = oracle.fill_uniform(n, q, UNI_SEED, i << 32)
    assert oracle.fnv(u) == c["uni_in_fnv"]
    out = cx.fwd(u)
    assert [int(x) for x in out[:3]] == c["uni_out_head"]
    assert oracle.fnv(out) == c["uni_out_fnv"]
    assert np.array_equal(cx.fwd_r4(u), out)
    assert np.array_equal(cx.inv(out), u) and np.array_equal(cx.inv_r4(out), u)


@pytest.mark.parametrize("i", range(19))
def test_lazy_words_of_the_three_formulations(kat, lazy_words, oracle, i):
    """the UNREDUCED words the reference's *_lazy functions return (digests captured from the compiled reference,
    oracle/gen_golden.py): radix-2, radix-4, and the radix-4x4 formulation, whose words differ from radix-4's exactly
    when log2 N = 4k+3 (src/ntt_radix4x4.c:91-111: radix-2 stage BEFORE the last radix-4 layer)"""
    c, lz = kat["cases"][i], lazy_words["cases"][i]
    n, q = 1 << c["m"], c["q"]
    assert (lz["case"], lz["m"], lz["q"]) == (i, c["m"], q) and lz["uni_in_fnv"] == c["uni_in_fnv"]
    cx = _ctx(oracle, c)
    u = oracle.fill_uniform(n, q, UNI_SEED, i << 32)
    words = {"ref_harvey": cx.fwd_lazy(u), "radix4": cx.fwd_r4_lazy(u), "radix4x4": cx.fwd_r4x4_lazy(u)}
    for name, x in words.items():
        assert oracle.fnv(x) == lz["lazy_out_fnv"][name], name
        assert np.array_equal(x % np.uint64(q), cx.fwd(u)), name
    assert lz["radix4x4_differs_from_radix4"] == (c["m"] % 4 == 3)
    assert np.array_equal(words["radix4"], words["radix4x4"]) == (c["m"] % 4 != 3)


@pytest.mark.parametrize("m", range(1, 14))
def test_radix4x4_lazy_words_at_sizes_without_a_reference_case(oracle, m):
    """sizes 2 .. 2^13 (the reference's cases start at 2^8 and hold one 4k+3 size below 2^15): the restatement against
    the compiled reference where this container has it (oracle/_ref); residues against the radix-2 path everywhere"""
    import ctypes as C
    import os
    n = 1 << m
    q = oracle.find_prime(30 + m, n)
    cx = oracle.ctx(n, q, oracle.min_root(q, n))
    a = oracle.fill_uniform(n, q, 900 + m)
    a[: min(n, 8)] = q - 1                                   # large leading coefficients: the 8q -> 4q steps fire
    x = cx.fwd_r4x4_lazy(a)
    assert int(x.max()) < 8 * q and np.array_equal(x % np.uint64(q), cx.fwd(a))
    if m % 4 != 3:
        assert np.array_equal(x, cx.fwd_r4_lazy(a))
    ref_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libntt_ref.so")
    if os.path.exists(ref_path):
        ref = C.CDLL(ref_path)
        if hasattr(ref, "ref_fwd_r4_lazy_generic"):
            U64P = C.POINTER(C.c_uint64)
            e, ec = cx.table("e"), cx.table("econ")
            for variant, want in ((1, cx.fwd_r4_lazy(a)), (2, x)):
                r = a.copy()
                ref.ref_fwd_r4_lazy_generic(variant, r.ctypes.data_as(U64P), C.c_uint64(n), C.c_uint64(q), e.ctypes.data_as(U64P),
                                            ec.ctypes.data_as(U64P))
                assert np.array_equal(r, want), variant


@pytest.mark.parametrize("i", range(19))
def test_edge_kats(kat, oracle, i):
    c = kat["cases"][i]
    n, q = 1 << c["m"], c["q"]
    cx = _ctx(oracle, c)
    edges = {"zero": np.zeros(n, dtype=np.uint64), "qm1": np.full(n, q - 1, dtype=np.uint64)}
    for name, pos in (("delta0", 0), ("delta1", 1), ("deltaN1", n - 1)):
        e = np.zeros(n, dtype=np.uint64)
        e[pos] = 1
        edges[name] = e
    for name, e in edges.items():
        assert oracle.fnv(cx.fwd(e)) == c["edge_out_fnv"][name], name


def test_case0_full_vectors(case0_vectors, oracle):
    v = case0_vectors
    cx = oracle.ctx(1 << v["m"], v["q"], v["w"])
    for a, b in (("rand_in", "rand_out"), ("uni_in", "uni_out")):
        x = np.array(v[a], dtype=np.uint64)
        y = np.array(v[b], dtype=np.uint64)
        assert np.array_equal(cx.fwd(x), y) and np.array_equal(cx.fwd_r4(x), y)
        assert np.array_equal(cx.inv(y), x) and np.array_equal(cx.inv_r4(y), x)


def test_reference_rand_stream_digests(kat, case0_vectors, oracle):
    """inputs of the reference's own run (glibc rand()%q): the N=256 case is stored
    in full, so its digests must match App. B"""
    c = kat["cases"][0]
    x = np.array(case0_vectors["rand_in"], dtype=np.uint64)
    assert oracle.fnv(x) == c["in_fnv"] == "e8a3b71c74d71cfb"
    assert [int(v) for v in x[:3]] == c["in_head"] == [7121, 783, 6956]
    y = oracle.ctx(256, c["q"], c["w"]).fwd(x)
    assert oracle.fnv(y) == c["out_fnv"] == "8ae456206dc728b0"


@pytest.mark.parametrize("m,q,w", [(4, 0x10001, None), (6, 0x1e01, None), (8, 0x1e01, 62), (10, 0x10001, 33)])
def test_definition(oracle, m, q, w):
    """out[bitrev(i)] = sum_j a_j w^{(2i+1)j}, O(N^2) evaluation"""
    n = 1 << m
    w = w or oracle.min_root(q, n)
    a = oracle.fill_uniform(n, q, 7, 0)
    assert np.array_equal(oracle.ctx(n, q, w).fwd(a), oracle.fwd_naive(a, n, q, w))


def test_negacyclic_product(oracle):
    n, q = 64, 0x1e01
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    a, b = oracle.fill_uniform(n, q, 1), oracle.fill_uniform(n, q, 2)
    c = cx.inv(oracle.pointwise(cx.fwd(a), cx.fwd(b), q))
    assert np.array_equal(c, oracle.schoolbook(a, b, n, q))


def test_prime_search(oracle):
    p = oracle.find_prime(50, 1 << 14)
    assert p < 2**50 and p % (1 << 15) == 1 and oracle.lib.orc_is_prime(p)
    assert oracle.min_root(p, 1 << 14) != 0
