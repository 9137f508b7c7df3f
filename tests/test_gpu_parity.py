"""GPU (-m gpu): the HIP library, called through its C ABI, is bit-exact against
the oracle and the golden vectors on every reference parameter set; the
reference-signature entry points behave like the reference's; full-size runs
are checked through round trips, linearity and per-polynomial checksums."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNI_SEED = 0x5EED5EED


def _arith_modes(lib, q):
    return [lib.ARITH_U64] + ([lib.ARITH_F64] if q <= (1 << 51) + (1 << 41) else [])


def _inputs(oracle, n, q, batch, seed):
    a = oracle.fill_uniform(batch * n, q, seed)
    a[:4] = q - 1
    a[4:8] = 0
    a[-1] = q - 1
    return a


def test_device_present(lib):
    assert lib.device_count() >= 1


@pytest.mark.parametrize("i", range(19))
def test_reference_cases_batched(lib, oracle, kat, i):
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    batch = 3
    a = _inputs(oracle, n, q, batch, 4242 + i)
    a[:n] = oracle.fill_uniform(n, q, UNI_SEED, i << 32)     # polynomial 0 = the golden KAT input
    expect = cx.fwd(a)
    assert oracle.fnv(expect[:n]) == c["uni_out_fnv"]
    for arith in _arith_modes(lib, q):
        plan = lib.Plan(n, q, w, arith=arith)
        got = plan.fwd_host(a)
        assert np.array_equal(got, expect), (i, arith, "fwd")
        assert oracle.fnv(got[:n]) == c["uni_out_fnv"]       # golden digest straight from the GPU
        back = plan.inv_host(got)
        assert np.array_equal(back, a), (i, arith, "inv")
        plan.set_generic(True)                               # strided multi-pass self-check path
        assert np.array_equal(plan.fwd_host(a), expect), (i, arith, "generic fwd")
        assert np.array_equal(plan.inv_host(expect), a), (i, arith, "generic inv")
        plan.destroy()


@pytest.mark.parametrize("i", range(19))
def test_edge_inputs_match_golden(lib, oracle, kat, i):
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    rows = [np.zeros(n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64)]
    for pos in (0, 1, n - 1):
        e = np.zeros(n, dtype=np.uint64)
        e[pos] = 1
        rows.append(e)
    names = ["zero", "qm1", "delta0", "delta1", "deltaN1"]
    plan = lib.Plan(n, q, w)
    got = plan.fwd_host(np.concatenate(rows))
    for k, name in enumerate(names):
        assert oracle.fnv(got[k * n:(k + 1) * n]) == c["edge_out_fnv"][name], name
    plan.destroy()


def test_case0_full_vectors(lib, case0_vectors):
    v = case0_vectors
    plan = lib.Plan(1 << v["m"], v["q"], v["w"])
    x = np.array(v["rand_in"], dtype=np.uint64)
    y = np.array(v["rand_out"], dtype=np.uint64)
    assert np.array_equal(plan.fwd_host(x), y)
    assert np.array_equal(plan.inv_host(y), x)


@pytest.mark.parametrize("i", [0, 1, 2, 5, 9, 12, 13, 14, 17, 18])
def test_reference_signatures(lib, oracle, kat, i):
    """mirror of reference tests/test_correctness.c:23-111 through the exported
    fwd_ntt_*/inv_ntt_* symbols (host pointers, reference table layouts)"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    tw, twc, twi, twic = (cx.table(k) for k in ("w", "wcon", "winv", "winv_con"))
    te, tec, tei, teic = (cx.table(k) for k in ("e", "econ", "einv", "einv_con"))
    a_orig = oracle.fill_uniform(n, q, 99 + i)
    a_ntt = cx.fwd(a_orig)
    a = a_orig.copy()
    lib.fwd_ntt_ref_harvey(a, n, q, tw, twc)
    assert np.array_equal(a, a_ntt)
    lib.inv_ntt_ref_harvey(a, n, q, c["n_inv"], c["n_inv_con"], 64, twi, twic)
    assert np.array_equal(a, a_orig)
    a, b = a_orig.copy(), a_orig.copy()
    lib.fwd_ntt_ref_harvey_dbl(a, b, n, q, tw, twc)
    assert np.array_equal(a, a_ntt) and np.array_equal(b, a_ntt)
    a = a_orig.copy()
    lib.fwd_ntt_seal(a, n, q, tw, twc)
    assert np.array_equal(a, a_ntt)
    lib.inv_ntt_seal(a, n, q, c["n_inv"], c["n_inv_con"], twi, twic)
    assert np.array_equal(a, a_orig)
    a = a_orig.copy()
    lib.fwd_ntt_radix4(a, n, q, te, tec)
    assert np.array_equal(a, a_ntt)
    lib.inv_ntt_radix4(a, n, q, c["n_inv"], c["n_inv_con"], tei, teic)
    assert np.array_equal(a, a_orig)
    a = a_orig.copy()
    lib.fwd_ntt_radix4x4(a, n, q, te, tec)
    assert np.array_equal(a, a_ntt)
    # lazy-range inputs, as the reference's bench feeds them back (tests/bench.c:123-137)
    lazy = a_orig + np.uint64(q) * (oracle.fill_uniform(n, 8, 3) % np.uint64(8))
    a = lazy.copy()
    lib.fwd_ntt_radix4(a, n, q, te, tec)
    assert np.array_equal(a, a_ntt)
    a = (a_ntt + np.uint64(q) * (oracle.fill_uniform(n, 8, 4) % np.uint64(8))).astype(np.uint64)
    lib.inv_ntt_radix4(a, n, q, c["n_inv"], c["n_inv_con"], tei, teic)
    assert np.array_equal(a, a_orig)


@pytest.mark.parametrize("bits,m", [(50, 14), (50, 12), (52, 16), (49, 13), (40, 11), (31, 10), (60, 12), (45, 6), (20, 3), (33, 7), (50, 18), (45, 20)])
def test_generated_parameters(lib, oracle, bits, m):
    n = 1 << m
    q = lib.find_prime(bits, n)
    w = lib.min_root(q, n)
    assert q == oracle.find_prime(bits, n) and w == oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    batch = 5   # ragged: not a multiple of the blocks one workgroup holds
    a = _inputs(oracle, n, q, batch, bits * 131 + m)
    expect = cx.fwd(a)
    for arith in _arith_modes(lib, q):
        plan = lib.Plan(n, q, w, arith=arith)
        assert np.array_equal(plan.fwd_host(a), expect)
        assert np.array_equal(plan.inv_host(expect), a)
        plan.destroy()


def test_randomised_sizes_moduli_batches(lib, oracle):
    """seeded sweep over block sizes 2^6..2^15, the three FP64 headroom classes and the integer
    policy, ragged batches and lazy ([0,8q)) inputs: every kernel variant and every LDS twiddle
    table layout against the oracle"""
    rng = np.random.default_rng(20261002)
    checked = 0
    for m in range(6, 16):
        n = 1 << m
        for bits in (20 + int(rng.integers(0, 12)), 34 + int(rng.integers(0, 15)), 50, 51):
            q = lib.find_prime(bits, n, int(rng.integers(0, 3)))
            if q == 0:
                continue
            w = lib.min_root(q, n)
            cx = oracle.ctx(n, q, w)
            batch = int(rng.integers(1, 8))
            a = _inputs(oracle, n, q, batch, 77 * m + bits)
            expect = cx.fwd(a)
            lazy = a + np.uint64(q) * rng.integers(0, 8, size=a.shape, dtype=np.uint64)   # same residues, in [0,8q)
            for arith in _arith_modes(lib, q):
                plan = lib.Plan(n, q, w, arith=arith)
                assert np.array_equal(plan.fwd_host(a), expect), (m, hex(q), arith, "fwd")
                assert np.array_equal(plan.fwd_host(lazy, wide=True), expect), (m, hex(q), arith, "fwd wide")
                assert np.array_equal(plan.inv_host(expect), a), (m, hex(q), arith, "inv")
                lazy_e = expect + np.uint64(q) * rng.integers(0, 8, size=a.shape, dtype=np.uint64)
                assert np.array_equal(plan.inv_host(lazy_e, wide=True), a), (m, hex(q), arith, "inv wide")
                plan.destroy()
                checked += 1
    assert checked >= 60


def test_empty_batch_and_bad_arguments(lib):
    plan = lib.Plan(256, 0x1e01, 62)
    plan.fwd(None, 0)
    plan.inv(None, 0)
    with pytest.raises(lib.NttError):
        lib.Plan(256, 0x1e01, 63)             # not a primitive 2N-th root
    with pytest.raises(lib.NttError):
        lib.Plan(300, 0x1e01, 62)             # N not a power of two
    with pytest.raises(lib.NttError):
        lib.Plan(1 << 10, (1 << 55) - 54783, 3, arith=lib.ARITH_F64)


@pytest.mark.parametrize("q,m", [(0x1e01, 6), (0x10001, 8), (0x7fffffffe0001, 7), (0x80000001c0001, 6)])
def test_negacyclic_product_vs_schoolbook(lib, oracle, q, m):
    n = 1 << m
    w = lib.min_root(q, n)
    batch = 4
    a, b = _inputs(oracle, n, q, batch, 21), _inputs(oracle, n, q, batch, 22)
    for arith in _arith_modes(lib, q):
        plan = lib.Plan(n, q, w, arith=arith)
        da, db = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b)
        plan.negacyclic_mul(da.ptr, da.ptr, db.ptr, batch)
        got = da.download()
        for p in range(batch):
            s = slice(p * n, (p + 1) * n)
            assert np.array_equal(got[s], oracle.schoolbook(a[s].copy(), b[s].copy(), n, q))
        plan.destroy()


def test_pointwise_large_moduli(lib, oracle):
    for q, n in ((0x7fffffffe0001, 1 << 14), ((1 << 60) - 93 * (1 << 15) + 1, 1 << 10)):
        q = q if q % (2 * n) == 1 and oracle.lib.orc_is_prime(q) else lib.find_prime(60, n)
        w = lib.min_root(q, n)
        plan = lib.Plan(n, q, w)
        a, b = _inputs(oracle, n, q, 2, 31), _inputs(oracle, n, q, 2, 32)
        da, db = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b)
        plan.pointwise_mul(da.ptr, da.ptr, db.ptr, 2)
        assert np.array_equal(da.download(), oracle.pointwise(a, b, q))
        plan.destroy()


def _full_size_roundtrip(lib, oracle, m, q, w, batch, sample):
    """device-generated inputs; fwd; sampled polynomials vs oracle; inv; checksum
    of every polynomial must equal the checksum of its input"""
    n = 1 << m
    plan = lib.Plan(n, q, w)
    buf = lib.DeviceBuffer(batch * n)
    cs0, cs1 = lib.DeviceBuffer(batch), lib.DeviceBuffer(batch)
    lib.fill_uniform(buf.ptr, batch * n, q, UNI_SEED, 0)
    lib.poly_checksum(cs0.ptr, buf.ptr, n, batch)
    plan.fwd(buf.ptr, batch)
    cx = oracle.ctx(n, q, w)
    for p in sample:
        got = buf.download(n, p * n)
        inp = oracle.fill_uniform(n, q, UNI_SEED, p * n)
        assert np.array_equal(got, cx.fwd(inp)), p
    plan.inv(buf.ptr, batch)
    lib.poly_checksum(cs1.ptr, buf.ptr, n, batch)
    a0, a1 = cs0.download(), cs1.download()
    assert np.array_equal(a0, a1)
    p = sample[-1]
    assert oracle.checksum(oracle.fill_uniform(n, q, UNI_SEED, p * n)) == int(a0[p])
    for b in (buf, cs0, cs1):
        b.free()
    plan.destroy()


def test_full_size_config2_n4096(lib, oracle):
    """BASELINE config 2: N=4096, 50-bit q, batch=65536"""
    q = lib.find_prime(50, 4096)
    _full_size_roundtrip(lib, oracle, 12, q, lib.min_root(q, 4096), 65536, [0, 1, 4095, 32768, 65535])


@pytest.mark.parametrize("m,bits,batch", [(14, 57, 32768), (14, 61, 16384), (16, 59, 4096)])
def test_full_size_integer_moduli(lib, oracle, m, bits, batch):
    """config-4- and config-3-shaped slabs (4 GiB / 2 GiB) with moduli the FP64 policies cannot serve: the wide integer policy's
    three headroom classes at full persistent grids -- sampled polynomials against the oracle, every polynomial's round trip by
    checksum"""
    n = 1 << m
    q = lib.find_prime(bits, n)
    p = lib.Plan(n, q, lib.min_root(q, n))
    assert p.info()["f64_class"] == 100 + (3 if bits <= 58 else (1 if bits <= 60 else 0))
    p.destroy()
    _full_size_roundtrip(lib, oracle, m, q, lib.min_root(q, n), batch, [0, 1, batch // 2 + 1, batch - 1])


def test_full_size_config3_n65536(lib, oracle, kat):
    """BASELINE config 3: N=65536, 51-bit q (reference case 17), batch=8192, fwd+inv round trip"""
    c = kat["cases"][17]
    _full_size_roundtrip(lib, oracle, 16, c["q"], c["w"], 8192, [0, 1, 4097, 8191])


def test_full_size_config3_n65536_true_52_bit_prime(lib, oracle):
    """BASELINE config 3 to the letter: N=65536, a 52-bit q (the largest prime below 2^52 with 2N | q-1), batch=8192,
    fwd+inv round trip -- served by the FP64 policy for moduli up to 2^52"""
    q = lib.find_prime(52, 65536)
    assert q >> 51 == 1
    _full_size_roundtrip(lib, oracle, 16, q, lib.min_root(q, 65536), 8192, [0, 1, 4097, 8191])


def test_full_size_config4_share_n16384(lib, oracle, kat):
    """BASELINE config 4, one GPU's share: N=16384, 51-bit q (reference case 12), batch=131072"""
    c = kat["cases"][12]
    _full_size_roundtrip(lib, oracle, 14, c["q"], c["w"], 131072, [0, 77, 65536, 131071])


def test_linearity_full_batch(lib, oracle, kat):
    """fwd(a+b) == fwd(a)+fwd(b) mod q on a 1 GiB batch (checksummed per polynomial)"""
    c = kat["cases"][13]
    n, q, w, batch = 1 << 14, c["q"], c["w"], 4096
    plan = lib.Plan(n, q, w)
    a = oracle.fill_uniform(batch * n, q, 5)
    b = oracle.fill_uniform(batch * n, q, 6)
    s = (a + b) % np.uint64(q)
    fa, fb, fs = plan.fwd_host(a), plan.fwd_host(b), plan.fwd_host(s)
    assert np.array_equal((fa + fb) % np.uint64(q), fs)
    plan.destroy()


def _rns_primes(lib, n, count=4, bits=50):
    return [lib.find_prime(bits, n, skip) for skip in range(count)]


def test_rns_pipeline_small(lib, oracle):
    """fwd -> pointwise -> inv over 4 RNS limbs (layout [limb][batch][N]) vs the oracle, N=2^11"""
    n, batch = 1 << 11, 3
    qs = _rns_primes(lib, n)
    assert len(set(qs)) == 4 and all(q % (2 * n) == 1 for q in qs)
    plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
    a = np.concatenate([_inputs(oracle, n, q, batch, 300 + l) for l, q in enumerate(qs)])
    b = np.concatenate([_inputs(oracle, n, q, batch, 400 + l) for l, q in enumerate(qs)])
    da, db = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b)
    lib.rns_negacyclic_mul(plans, da.ptr, da.ptr, db.ptr, batch)
    got = da.download()
    for l, q in enumerate(qs):
        cx = oracle.ctx(n, q, lib.min_root(q, n))
        s = slice(l * batch * n, (l + 1) * batch * n)
        expect = cx.inv(oracle.pointwise(cx.fwd(a[s]), cx.fwd(b[s]), q))
        assert np.array_equal(got[s], expect), l
    # schoolbook cross-check of one polynomial of one limb
    q = qs[2]
    s0 = 2 * batch * n
    assert np.array_equal(got[s0:s0 + n], oracle.schoolbook(a[s0:s0 + n].copy(), b[s0:s0 + n].copy(), n, q))


@pytest.mark.parametrize("bits", [50, 52, 31])
@pytest.mark.parametrize("m", [15, 16, 17])
def test_xcd_local_two_pass_kernel(lib, oracle, m, bits):
    """NTT_OPT_XCD_LOCAL 1: both passes of a 2^15..2^17 transform as items of ONE launch, every polynomial handled by the
    workgroups of one XCD (per-XCD queues, per-polynomial hand-off counters, intermediate kept in that XCD's L2).
    Forward and inverse against the oracle on every polynomial, for batches that leave queues ragged (batch mod 8 != 0),
    for every lag / residency setting, and word for word against the one-launch-per-pass path."""
    _xcd_local_check(lib, oracle, m, bits, lib.ARITH_F64, ((64, 0, 0), (67, 1, 1), (100, 2, 3), (131, 9, 2), (77, 3, 4)))


@pytest.mark.parametrize("m,bits", [(15, 57), (16, 60), (17, 57), (16, 61)])
def test_xcd_local_two_pass_kernel_integer_moduli(lib, oracle, m, bits):
    """the same launch with the wide integer policy's row and column items (moduli of 2^52 and more, N = 2^15..2^17: the
    sizes and primes of bootstrappable CKKS parameter sets): words between the passes are lazy, the queue protocol and the
    cache policies are the FP64 launch's"""
    _xcd_local_check(lib, oracle, m, bits, lib.ARITH_AUTO, ((64, 0, 0), (67, 1, 1), (131, 9, 2), (77, 3, 4)))


def _xcd_local_check(lib, oracle, m, bits, arith, settings):
    n = 1 << m
    q = lib.find_prime(bits, n, 1)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith=arith)
    ref = lib.Plan(n, q, w, arith=arith)
    ref.set_option(lib.OPT_XCD_LOCAL, 0)
    plan.set_option(lib.OPT_XCD_LOCAL, 1)
    for batch, lag, wpc in settings:
        plan.set_option(lib.OPT_XCD_LOCAL_LAG, lag)
        plan.set_option(lib.OPT_XCD_LOCAL_WGS_PER_CU, wpc)
        a = _inputs(oracle, n, q, batch, 1200 + batch)
        buf = lib.DeviceBuffer(a.size).upload(a)
        plan.fwd(buf.ptr, batch)
        f = buf.download()
        check = [0, 1, 7, 8, batch // 2, batch - 2, batch - 1]
        pick = np.concatenate([np.arange(p * n, (p + 1) * n) for p in check])
        assert np.array_equal(f[pick], cx.fwd(a[pick])), (batch, lag, wpc)
        assert np.array_equal(f, ref.fwd_host(a)), (batch, lag, wpc)      # every polynomial, against the per-pass path
        plan.inv(buf.ptr, batch)
        assert np.array_equal(buf.download(), a), (batch, lag, wpc)
        # the inverse alone, from oracle-made inputs
        fo = cx.fwd(a[pick])
        b2 = lib.DeviceBuffer(batch * n).upload(f)
        plan.inv(b2.ptr, batch)
        assert np.array_equal(b2.download()[pick], cx.inv(fo)), (batch, lag, wpc)
        buf.free(), b2.free()
    plan.destroy(), ref.destroy()


@pytest.mark.parametrize("m,bits", [(15, 50), (16, 52), (17, 50), (16, 31)])
def test_xcd_local_product_kernel(lib, oracle, m, bits):
    """products of 512+ polynomials at N = 2^15..2^17 as ONE launch (team_product_kernel): column stages of b and a, block
    products (both blocks through their twelve stages, product, inverse stages), inverse column stages of c -- and its
    three-pass form (NTT_OPT_FUSED_PRODUCT 2: a^ = fwd(a) by a launch of its own first).  All aliasing forms, ragged
    batches, sampled polynomials against the oracle and every polynomial against the four-transform chain."""
    n = 1 << m
    q = lib.find_prime(bits, n, 2)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan, three, chain = lib.Plan(n, q, w), lib.Plan(n, q, w), lib.Plan(n, q, w)
    three.set_option(lib.OPT_FUSED_PRODUCT, 2)
    chain.set_option(lib.OPT_FUSED_PRODUCT, 0)
    chain.set_option(lib.OPT_XCD_LOCAL, 0)
    for batch, alias, lag in ((512, "c", 0), (519, "a", 3), (777 if m < 17 else 521, "b", 1)):
        a = oracle.fill_uniform(batch * n, q, 1300 + batch)
        b = oracle.fill_uniform(batch * n, q, 1400 + batch)
        b[:8] = [0, 1, q - 1, q - 2, 2, 3, q // 2, q // 2 + 1]
        a[:8] = [q - 1, 0, 1, q - 2, q // 2, 3, 2, q // 2 + 1]
        da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(a.size).upload(b), lib.DeviceBuffer(a.size)
        sample = [0, 1, 7, 8, batch // 2, batch - 2, batch - 1]
        pick = np.concatenate([np.arange(p * n, (p + 1) * n) for p in sample])
        want = cx.inv(oracle.pointwise(cx.fwd(a[pick]), cx.fwd(b[pick]), q))
        got = None
        for pl in (plan, three):
            pl.set_option(lib.OPT_XCD_LOCAL_LAG, lag)
            da.upload(a), db.upload(b)
            out = {"a": da, "b": db, "c": dc}[alias]
            pl.negacyclic_mul(out.ptr, da.ptr, db.ptr, batch)
            res = out.download()
            assert np.array_equal(res[pick], want), (batch, alias, pl is three)
            assert got is None or np.array_equal(res, got), (batch, alias)
            got = res
        da.upload(a), db.upload(b)
        chain.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), got), (batch, alias)
        for x in (da, db, dc):
            x.free()
    plan.destroy(), three.destroy(), chain.destroy()


def test_xcd_local_full_size_round_trip_and_cross_check(lib, oracle):
    """BASELINE config 3's shape (N = 2^16, batch 8192) through the XCD-local kernel: per-polynomial checksums equal the
    per-pass path's after the forward transform, and the inverse restores the input exactly"""
    n, batch = 1 << 16, 8192
    q = lib.find_prime(52, n)
    w = lib.min_root(q, n)
    plan, ref = lib.Plan(n, q, w), lib.Plan(n, q, w)
    plan.set_option(lib.OPT_XCD_LOCAL, 1)
    ref.set_option(lib.OPT_XCD_LOCAL, 0)
    da, db = lib.DeviceBuffer(batch * n), lib.DeviceBuffer(batch * n)
    cs = [lib.DeviceBuffer(batch) for _ in range(3)]
    lib.fill_uniform(da.ptr, batch * n, q, UNI_SEED)
    lib.fill_uniform(db.ptr, batch * n, q, UNI_SEED)
    lib.poly_checksum(cs[0].ptr, da.ptr, n, batch)
    plan.fwd(da.ptr, batch)
    ref.fwd(db.ptr, batch)
    lib.poly_checksum(cs[1].ptr, da.ptr, n, batch)
    lib.poly_checksum(cs[2].ptr, db.ptr, n, batch)
    assert np.array_equal(cs[1].download(), cs[2].download())
    cx = oracle.ctx(n, q, w)
    for p in (0, 4097, 8191):
        assert np.array_equal(da.download(n, p * n), cx.fwd(oracle.fill_uniform(n, q, UNI_SEED, p * n))), p
    plan.inv(da.ptr, batch)
    lib.poly_checksum(cs[1].ptr, da.ptr, n, batch)
    assert np.array_equal(cs[1].download(), cs[0].download())
    plan.destroy(), ref.destroy()


@pytest.mark.parametrize("logn", [14, 16])
@pytest.mark.parametrize("nlimbs", [4, 16])
@pytest.mark.parametrize("batch", [1, 2, 8])
def test_rns_one_launch_over_all_limbs(lib, oracle, logn, nlimbs, batch, monkeypatch):
    """ntt_rns_{fwd,inv,negacyclic_mul}_batch with a small per-limb batch: ONE launch (per pass) serves every limb
    (kernel variants MULTI: the workgroup picks its limb's tables and constants from an array in the kernel arguments).
    Every limb of every polynomial against the oracle; the same calls with NTT_OPT_RNS_LAUNCH 1 (one launch chain per prime,
    the single-set kernels) must give the same words."""
    _rns_one_launch_check(lib, oracle, monkeypatch, logn, nlimbs, batch, 50)


@pytest.mark.parametrize("logn,nlimbs,batch,bits", [(14, 4, 2, 57), (12, 16, 1, 60), (16, 3, 2, 57), (13, 5, 3, 61), (8, 4, 9, 58)])
def test_rns_one_launch_over_integer_limbs(lib, oracle, logn, nlimbs, batch, bits, monkeypatch):
    """the same for limbs the FP64 policies cannot serve (54..60-bit primes, the sizes FHE libraries default to): the wide
    integer policy's kernels have MULTI variants too, and the product of such a set is three launches (both forward
    transforms, the products inside the inverse's first pass) instead of four per prime"""
    _rns_one_launch_check(lib, oracle, monkeypatch, logn, nlimbs, batch, bits)


def _rns_one_launch_check(lib, oracle, monkeypatch, logn, nlimbs, batch, bits):
    n = 1 << logn
    qs = [lib.find_prime(bits, n, k) for k in range(nlimbs)]
    assert len(set(qs)) == nlimbs
    roots = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    assert len({p.info()["f64_class"] for p in plans}) == 1
    a = np.concatenate([_inputs(oracle, n, q, batch, 900 + l) for l, q in enumerate(qs)])
    b = np.concatenate([_inputs(oracle, n, q, batch, 950 + l) for l, q in enumerate(qs)])
    ctxs = [oracle.ctx(n, q, w) for q, w in zip(qs, roots)]
    sl = [slice(l * batch * n, (l + 1) * batch * n) for l in range(nlimbs)]
    exp_f = np.concatenate([cx.fwd(a[s]) for cx, s in zip(ctxs, sl)])
    exp_p = np.concatenate([cx.inv(oracle.pointwise(cx.fwd(a[s]), cx.fwd(b[s]), q)) for cx, s, q in zip(ctxs, sl, qs)])
    results = {}
    for loop in ("0", "1"):
        lib.set_rns_launch(plans, loop)
        da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
        lib.rns_fwd(plans, da.ptr, batch)
        f = da.download()
        lib.rns_inv(plans, da.ptr, batch)
        back = da.download()
        lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
        results[loop] = (f, back, dc.download())
        for x in (da, db, dc):
            x.free()
    for loop, (f, back, prod) in results.items():
        assert np.array_equal(f, exp_f), loop
        assert np.array_equal(back, a), loop
        assert np.array_equal(prod, exp_p), loop
    for p in plans:
        p.destroy()


def test_rns_one_launch_more_limbs_than_one_launch_holds(lib, oracle, monkeypatch):
    """20 limbs: the kernel arguments hold 16 records, the set is served by two launches (16 + 4)"""
    n, nlimbs, batch = 1 << 12, 20, 3
    qs = [lib.find_prime(49, n, k) for k in range(nlimbs)]
    roots = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    a = np.concatenate([_inputs(oracle, n, q, batch, 700 + l) for l, q in enumerate(qs)])
    b = np.concatenate([_inputs(oracle, n, q, batch, 750 + l) for l, q in enumerate(qs)])
    lib.set_rns_launch(plans, 0)
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
    lib.rns_fwd(plans, da.ptr, batch)
    got_f = da.download()
    da.upload(a)
    lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
    got_p = dc.download()
    for l, (q, w) in enumerate(zip(qs, roots)):
        cx = oracle.ctx(n, q, w)
        s = slice(l * batch * n, (l + 1) * batch * n)
        assert np.array_equal(got_f[s], cx.fwd(a[s])), l
        assert np.array_equal(got_p[s], cx.inv(oracle.pointwise(cx.fwd(a[s]), cx.fwd(b[s]), q))), l


@pytest.mark.parametrize("sizes", [(30, 51, 45), (57, 60, 59, 61, 53), (52, 52, 50)])
def test_rns_mixed_headroom_classes_share_a_launch(lib, oracle, sizes, monkeypatch):
    """limbs of ONE policy whose primes fall into different headroom classes (a 30-bit, a 51-bit and a 45-bit prime: FP64
    classes 18, 0, 1; 57- to 61-bit primes: integer classes 3, 1, 1, 0, 3) share a launch in the coarsest class of the run;
    the 52-bit policy does not mix with the scheduled one (two runs).  Same results as limb by limb."""
    n, batch = 1 << 10, 2
    seen = {}
    qs = []
    for bts in sizes:
        qs.append(lib.find_prime(bts, n, seen.get(bts, 0)))
        seen[bts] = seen.get(bts, 0) + 1
    roots = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    lib.set_rns_launch(plans, 0)
    a = np.concatenate([_inputs(oracle, n, q, batch, 600 + l) for l, q in enumerate(qs)])
    da = lib.DeviceBuffer(a.size).upload(a)
    lib.rns_fwd(plans, da.ptr, batch)
    got = da.download()
    lib.rns_inv(plans, da.ptr, batch)
    back = da.download()
    db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(a.size)
    lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch)        # the squares
    sq = dc.download()
    for l, (q, w) in enumerate(zip(qs, roots)):
        s = slice(l * batch * n, (l + 1) * batch * n)
        cx = oracle.ctx(n, q, w)
        assert np.array_equal(got[s], cx.fwd(a[s])), l
        assert np.array_equal(back[s], a[s]), l
        assert np.array_equal(sq[s], cx.inv(oracle.pointwise(got[s], got[s], q))), l
    for x in (da, db, dc):
        x.free()
    for p in plans:
        p.destroy()


@pytest.mark.parametrize("m,batch", [(12, 2), (14, 1), (16, 2)])
def test_rns_modulus_chain_with_primes_of_several_sizes(lib, oracle, m, batch, monkeypatch):
    """a modulus chain as FHE libraries build them -- a 60-bit first prime, three 50-bit primes, two 57-bit primes, a 30-bit
    one: the limb list is served as maximal RUNS of consecutive compatible limbs (one launch per pass and run, single limbs
    by themselves), in both forced forms; every entry point of the family against the oracle, limb by limb"""
    n = 1 << m
    qs = [lib.find_prime(60, n)] + [lib.find_prime(50, n, i) for i in range(3)] + [lib.find_prime(57, n, i) for i in range(2)] + [lib.find_prime(30, n)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    nl, slab = len(qs), batch * n
    a = np.concatenate([oracle.fill_uniform(slab, q, 8100 + l) for l, q in enumerate(qs)])
    b = np.concatenate([oracle.fill_uniform(slab, q, 8200 + l) for l, q in enumerate(qs)])
    ctx = [oracle.ctx(n, q, w) for q, w in zip(qs, ws)]
    sl = [slice(l * slab, (l + 1) * slab) for l in range(nl)]
    fa = np.concatenate([c.fwd(a[x]) for c, x in zip(ctx, sl)])
    fb = np.concatenate([c.fwd(b[x]) for c, x in zip(ctx, sl)])
    prod = np.concatenate([c.inv(oracle.pointwise(fa[x], fb[x], q)) for c, x, q in zip(ctx, sl, qs)])
    had = np.concatenate([oracle.pointwise(fa[x], fb[x], q) for x, q in zip(sl, qs)])
    da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
    for loop in ("0", "1", None):
        if loop is None:
            lib.set_rns_launch(plans, None)
        else:
            lib.set_rns_launch(plans, loop)
        da.upload(a)
        lib.rns_fwd(plans, da.ptr, batch)
        assert np.array_equal(da.download(), fa), loop
        lib.rns_inv(plans, da.ptr, batch)
        assert np.array_equal(da.download(), a), loop
        da.upload(a), db.upload(b)
        lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), prod), loop
        da.upload(a), db.upload(fb)
        lib.rns_mul_transformed(plans, dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), prod), loop
        da.upload(fa), db.upload(fb)
        lib.rns_inv_dot(plans, dc.ptr, [da.ptr], [db.ptr], batch)
        assert np.array_equal(dc.download(), prod), loop
        da.upload(a), db.upload(fb)
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), had), loop
    for x in (da, db, dc):
        x.free()
    for p in plans:
        p.destroy()


def test_full_size_config5_share_rns_n131072(lib, oracle):
    """BASELINE config 5, one GPU's share: N=2^17, 4-prime RNS, 512 polynomials per GPU
    (2 GiB per operand): fwd/pointwise/inv pipeline, sampled polynomials vs the oracle,
    and fwd->inv round trip checksums over the whole slab"""
    n, batch = 1 << 17, 512
    qs = _rns_primes(lib, n)
    roots = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    slab = batch * n
    da, db = lib.DeviceBuffer(4 * slab), lib.DeviceBuffer(4 * slab)
    for l, q in enumerate(qs):
        lib.fill_uniform(da.ptr + 8 * l * slab, slab, q, UNI_SEED, l * slab)
        lib.fill_uniform(db.ptr + 8 * l * slab, slab, q, UNI_SEED + 1, l * slab)
    cs0, cs1 = lib.DeviceBuffer(4 * batch), lib.DeviceBuffer(4 * batch)
    lib.poly_checksum(cs0.ptr, da.ptr, n, 4 * batch)
    lib.rns_fwd(plans, da.ptr, batch)
    lib.rns_inv(plans, da.ptr, batch)
    lib.poly_checksum(cs1.ptr, da.ptr, n, 4 * batch)
    assert np.array_equal(cs0.download(), cs1.download())
    lib.rns_negacyclic_mul(plans, da.ptr, da.ptr, db.ptr, batch)
    for l, p in ((0, 0), (3, 511), (1, 77)):
        q, w = qs[l], roots[l]
        cx = oracle.ctx(n, q, w)
        ia = oracle.fill_uniform(n, q, UNI_SEED, l * slab + p * n)
        ib = oracle.fill_uniform(n, q, UNI_SEED + 1, l * slab + p * n)
        expect = cx.inv(oracle.pointwise(cx.fwd(ia), cx.fwd(ib), q))
        assert np.array_equal(da.download(n, l * slab + p * n), expect), (l, p)
    # the WHOLE slab, not only the three sampled polynomials: per-polynomial checksums of the fused product path against
    # two independent routes through the library -- the four-transform chain (NTT_OPT_FUSED_PRODUCT 0: forward, forward,
    # pointwise, inverse launches) and the generic column-pass implementation (no fused block kernel at all)
    lib.poly_checksum(cs0.ptr, da.ptr, n, 4 * batch)
    fused_sums = cs0.download()

    def refill():
        for l, q in enumerate(qs):
            lib.fill_uniform(da.ptr + 8 * l * slab, slab, q, UNI_SEED, l * slab)
            lib.fill_uniform(db.ptr + 8 * l * slab, slab, q, UNI_SEED + 1, l * slab)
    for route in ("chain", "generic"):
        refill()
        for p in plans:
            p.set_option(lib.OPT_FUSED_PRODUCT, 0)
            p.set_generic(route == "generic")
        lib.rns_negacyclic_mul(plans, da.ptr, da.ptr, db.ptr, batch)
        lib.poly_checksum(cs1.ptr, da.ptr, n, 4 * batch)
        assert np.array_equal(cs1.download(), fused_sums), route
    for p in plans:
        p.destroy()


def test_multi_device_call(lib, oracle, kat):
    """ntt_batch_multi with however many devices are visible (1 on the test box)"""
    c = kat["cases"][9]
    n, q, w = 1 << 14, c["q"], c["w"]
    ndev = lib.device_count()
    plans = [lib.Plan(n, q, w, device=d) for d in range(ndev)]
    batches = [7 + d for d in range(ndev)]
    host = [oracle.fill_uniform(bt * n, q, 50 + d) for d, bt in enumerate(batches)]
    bufs = [lib.DeviceBuffer(h.size, device=d).upload(h) for d, h in enumerate(host)]
    lib.batch_multi(plans, [b.ptr for b in bufs], batches)
    cx = oracle.ctx(n, q, w)
    for d in range(ndev):
        assert np.array_equal(bufs[d].download(), cx.fwd(host[d]))
    lib.batch_multi(plans, [b.ptr for b in bufs], batches, inverse=True)
    for d in range(ndev):
        assert np.array_equal(bufs[d].download(), host[d])


def test_reference_test_driver_drop_in():
    """the reference's own tests/main.c + tests/test_correctness.c, compiled
    unchanged against include/ and linked to libntt_mi355x.so (oracle/Makefile
    `dropin`), run here on the GPU: 19 cases, zero 'Bad results'"""
    exe = os.path.join(ROOT, "oracle", "_ref", "ntt-variants-dropin")
    # a GPU box that lacks the binary has lost the boundary row's only through-the-reference-driver
    # check: that is a failure, not a skip (the binary is built here by `make -C oracle dropin` and
    # travels with the snapshot; oracle/_ref is git-ignored but not gpurun-ignored)
    assert os.path.exists(exe), "oracle/_ref/ntt-variants-dropin did not travel to the GPU box"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("Test ") == 19
    assert "Bad results" not in out.stdout
    assert out.stdout.count("Running ") == 19 * 8   # 8 variants per case on a non-IFMA, non-s390x build


def test_bench_two_ranks_folded_on_one_gpu():
    """bench.py's N>1 path end to end (torch.distributed.run, rank sharding, barrier, MAX-reduction,
    rank-0 JSON) with two ranks folded onto the one GPU of the test box (gloo for the control plane:
    RCCL refuses two ranks on one device; the data path has no collective anyway)"""
    import json
    import sys
    env = dict(os.environ, NTT_BENCH_DEVICE_MOD="1", NTT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8192", "--cpu-budget-s", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 2 * 8192
    # every rank checked its own shard against the oracle; the CPU baseline rides on every line, N > 1 included (north_star: "in the same run")
    assert d["value"] > 1e5 and d["parity"] == dict(d["parity"], shards_checked=2, of=2)
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port")


def test_bench_single_process_two_shards_folded_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: ONE process drives both shards (one stream and one
    pair of events each, one host clock), here folded onto the one GPU of the test box"""
    import json
    import sys
    env = dict(os.environ, NTT_BENCH_DEVICE_MOD="1")
    env.pop("WORLD_SIZE", None)
    for scaling, per_gpu in (("weak", 4096), ("strong", None)):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--scaling", scaling, "--no-cpu-baseline"]
        if per_gpu:
            cmd += ["--batch", str(per_gpu)]
        else:
            cmd += ["--logn", "8"]          # strong: 2^20 * 2^14 / 2^8 polynomials of 2^8 split over 2 (same bytes)
            cmd += ["--batch", str(1 << 16)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling
        assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"]
        assert len(d["roofline"]["kernel_ms_per_gpu"]) == 2 and d["value"] > 1e5


def test_bench_eight_shards_folded_on_one_gpu():
    """the shape of the driver's 8-GPU run, folded onto the one GPU of the test box: `bench.py --gpus 8` in one process (eight
    streams, eight chains of events, one host clock) for config 4 and for config 2 (0.8 ms steps: sixteen host calls per step),
    every shard parity-checked, the CPU baseline on the line; then the same eight shards as eight ranks under
    torch.distributed.run (gloo for the control plane).  The aggregate rate against one shard of the same total batch is a
    measurement, kept in profiles/r06/folded_8_shards.txt; here only a loose bound guards against a serialising host loop."""
    import json
    import sys
    env = dict(os.environ, NTT_BENCH_DEVICE_MOD="1")
    env.pop("WORLD_SIZE", None)

    def run(extra, launcher=False, e=env):
        cmd = [sys.executable]
        if launcher:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29547"]
        cmd += [os.path.join(ROOT, "bench.py")] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for cfg, batch in ((4, 8192), (2, 16384)):
        one = run(["--gpus", "1", "--config", str(cfg), "--batch", str(8 * batch), "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--headline-only"])
        d = run(["--gpus", "8", "--config", str(cfg), "--batch", str(batch), "--steps", "10", "--warmup", "4", "--cpu-budget-s", "1", "--headline-only"])
        assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * batch and len(d["roofline"]["kernel_ms_per_gpu"]) == 8
        assert d["parity"]["shards_checked"] == 8 and d["parity"]["of"] == 8
        assert d["cpu_baseline"]["value"] > 0
        assert d["value"] > 0.8 * one["value"], (cfg, d["value"], one["value"])
    d = run(["--gpus", "8", "--batch", "8192", "--steps", "5", "--warmup", "2", "--cpu-budget-s", "1", "--headline-only"], launcher=True,
            e=dict(env, NTT_BENCH_BACKEND="gloo"))
    assert d["n_gpus"] == 8 and d["parity"]["shards_checked"] == 8 and d["cpu_baseline"]["value"] > 0
    assert len(d["roofline"]["kernel_ms_per_gpu"]) == 8


def test_bench_refuses_more_gpus_than_devices():
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("NTT_BENCH_DEVICE_MOD", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0", "--batch", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)


@pytest.mark.parametrize("q,config", [("0xfff88001", 4), ("0xffffffffffc0001", 4), ("0x80000001c0001", 2)])
def test_bench_with_another_modulus(q, config):
    """bench.py --q: the chosen configuration's transforms over another of the reference's moduli (tests/test_cases.h: a 32-bit,
    a 60-bit and the 51-bit one of cases 13-15), small batch -- the line's parity check (first and last polynomial against the oracle)
    runs on that modulus, the metric string says it is not BASELINE's"""
    import json
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--q", q, "--steps", "2", "--warmup", "1", "--batch", "96",
           "--no-cpu-baseline", "--headline-only"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert "--q " + hex(int(q, 0)) in line["metric"] and line["parity"]["shards_checked"] == 1 and line["value"] > 0


def test_squaring_aliases_both_operands(lib, oracle):
    """negacyclic_mul with d_a == d_b is a*a (the shared buffer used to be transformed twice)"""
    n, batch = 1 << 10, 3
    q = lib.find_prime(50, n)
    w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w)
    a = _inputs(oracle, n, q, batch, 77)
    da, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(a.size)
    plan.negacyclic_mul(dc.ptr, da.ptr, da.ptr, batch)
    got = dc.download()
    for p in range(batch):
        ap = a[p * n:(p + 1) * n].copy()
        assert np.array_equal(got[p * n:(p + 1) * n], oracle.schoolbook(ap, ap.copy(), n, q)), p
    # fully in place as well: c == a == b
    da.upload(a)
    plan.negacyclic_mul(da.ptr, da.ptr, da.ptr, batch)
    assert np.array_equal(da.download(), got)
    plan.destroy()


def test_caller_n_inv_is_honoured(lib, oracle, kat):
    """the by-value mul_op_t really reaches the library: a sentinel n_inv = 2*N^-1 doubles every output"""
    c = kat["cases"][4]
    n, q, w = 1 << c["m"], c["q"], c["w"]
    cx = oracle.ctx(n, q, w)
    a = oracle.fill_uniform(n, q, 99)
    fa = cx.fwd(a)
    ninv2 = (2 * cx.c.ninv) % q
    con = (ninv2 << 64) // q
    for fn, tab, tcon in ((lib.inv_ntt_ref_harvey, "winv", "winv_con"), (lib.inv_ntt_radix4, "einv", "einv_con")):
        x = fa.copy()
        if fn is lib.inv_ntt_ref_harvey:
            fn(x, n, q, ninv2, con, 64, cx.table(tab), cx.table(tcon))
        else:
            fn(x, n, q, ninv2, con, cx.table(tab), cx.table(tcon))
        assert np.array_equal(x, (a * np.uint64(2)) % np.uint64(q))


def test_compat_cache_sees_every_table_entry(lib, oracle, kat):
    """the reference-signature shims key their cached plans on ALL table entries: editing one unsampled slot in
    place must change the result (a sampled digest kept serving the stale plan), and the cache is bounded"""
    c = kat["cases"][6]
    n, q, w = 1 << c["m"], c["q"], c["w"]
    cx = oracle.ctx(n, q, w)
    tab, con = cx.table("w"), cx.table("wcon")
    a = oracle.fill_uniform(n, q, 5)
    x = a.copy()
    lib.fwd_ntt_ref_harvey(x, n, q, tab, con)
    assert np.array_equal(x, cx.fwd(a))
    k = 3 * n // 4 + 1                                    # a slot no 256-stride sample would hit
    assert k % max(n // 256, 1) != 0
    tab2 = tab.copy()
    tab2[k] = (int(tab2[k]) + 1) % q
    y = a.copy()
    lib.fwd_ntt_ref_harvey(y, n, q, tab2, con)
    assert not np.array_equal(y, x)
    # the same ARRAYS edited in place between two calls (same pointers: the call starts speculatively on the cached entry while it
    # hashes the tables, round 5): the stale entry must not be served, and restoring the entry restores the result
    x2 = a.copy()
    lib.fwd_ntt_ref_harvey(x2, n, q, tab, con)            # second call with these pointers: served speculatively
    assert np.array_equal(x2, x)
    saved = int(tab[k])
    tab[k] = (saved + 1) % q
    y2 = a.copy()
    lib.fwd_ntt_ref_harvey(y2, n, q, tab, con)
    assert np.array_equal(y2, y)                           # = the result of the edited copy above, not the cached plan's
    tab[k] = saved
    x3 = a.copy()
    lib.fwd_ntt_ref_harvey(x3, n, q, tab, con)
    assert np.array_equal(x3, x)
    con_saved = int(con[k])
    con[k] = (con_saved + 1) & ((1 << 64) - 1)            # the precomputed quotients are part of the key too
    y3 = a.copy()
    lib.fwd_ntt_ref_harvey(y3, n, q, tab, con)
    con[k] = con_saved
    z3 = a.copy()
    lib.fwd_ntt_ref_harvey(z3, n, q, tab, con)
    assert np.array_equal(z3, x)
    lib.compat_release()
    assert lib.compat_cached_plans() == 0
    for i in range(40):                                   # 40 distinct tables: the cache stays at <= 32 plans
        t = tab.copy()
        t[2] = (int(t[2]) + 1 + i) % q
        z = a.copy()
        lib.fwd_ntt_ref_harvey(z, n, q, t, con)
    assert 0 < lib.compat_cached_plans() <= 32
    z = a.copy()
    lib.fwd_ntt_ref_harvey(z, n, q, tab, con)             # the original table still gives the right answer
    assert np.array_equal(z, cx.fwd(a))
    lib.compat_release()


def test_current_device_is_restored(lib):
    """entry points leave the caller's current HIP device as they found it (torch interoperability)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    cur = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(cur)) == 0
    before = cur.value
    plan = lib.Plan(256, 0x1e01, 62, device=0)
    buf = lib.DeviceBuffer(256)
    plan.fwd(buf.ptr, 1)
    lib.stream_sync(0, None)
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == before
    plan.destroy()


# ---------------------------------------------------------------------------------------------
# radix-4 integer policy on the device, lazy outputs, plan options (round 2)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("i", range(19))
def test_radix4_device_policy(lib, oracle, kat, i):
    """NTT_ARITH_U64_R4: the reference's radix-4 butterflies / shared-quotient double products
    (fast_mul_operators.h:62-70,108-149) and its 5-twiddle fetch (src/ntt_radix4.c:7-25) as HIP code.  Lazy
    outputs equal fwd_ntt_radix4_lazy bit for bit; the inverse equals inv_ntt_radix4."""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    if m < 6:
        with pytest.raises(lib.NttError):
            lib.Plan(n, q, w, arith=lib.ARITH_U64_R4)
        return
    batch = 5
    a = oracle.fill_uniform(batch * n, q, 1300 + i)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith=lib.ARITH_U64_R4)
    assert plan.info()["arith"] == lib.ARITH_U64_R4
    lazy = plan.fwd_host(a, lazy=True)
    assert np.array_equal(lazy, cx.fwd_r4_lazy(a))
    assert int(lazy.max()) < (8 if m % 2 == 0 else 4) * q
    assert np.array_equal(plan.fwd_host(a), cx.fwd(a))
    assert np.array_equal(plan.fwd_host(lazy, wide=True), cx.fwd(lazy % np.uint64(q)))
    # (reference cases 14-18, N = 2^15..2^17: a column pass of one or two radix-4 levels before / after the blocks)
    assert np.array_equal(plan.fwd_host(lazy, wide=True, lazy=True), cx.fwd_r4_lazy(lazy))
    assert np.array_equal(plan.inv_host(cx.fwd(a)), a)
    assert np.array_equal(plan.inv_host(lazy, wide=True), a)
    lz = plan.inv_host(cx.fwd(a), lazy=True)
    assert int(lz.max()) < 2 * q and np.array_equal(lz % np.uint64(q), a)
    # products through the radix-4 formulation end to end (fwd_ntt_radix4 x 2, pointwise, inv_ntt_radix4); c aliasing a
    b = oracle.fill_uniform(batch * n, q, 1350 + i)
    want = cx.inv(oracle.pointwise(cx.fwd(a), cx.fwd(b), q))
    da, db = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b)
    plan.negacyclic_mul(da.ptr, da.ptr, db.ptr, batch)
    assert np.array_equal(da.download(), want)
    da.free(); db.free()
    plan.destroy()


def test_radix4_device_policy_large_moduli(lib, oracle):
    for bits, m in ((59, 8), (59, 14), (52, 14), (52, 11), (59, 15), (59, 16), (52, 17), (59, 18)):
        n = 1 << m
        q = oracle.find_prime(bits, n)
        w = oracle.min_root(q, n)
        a = oracle.fill_uniform(3 * n, q, bits * 10 + m)
        cx = oracle.ctx(n, q, w)
        plan = lib.Plan(n, q, w, arith=lib.ARITH_U64_R4)
        lazy = plan.fwd_host(a, lazy=True)
        assert np.array_equal(lazy, cx.fwd_r4_lazy(a)), (bits, m)
        assert np.array_equal(plan.inv_host(lazy, wide=True), a), (bits, m)
        plan.destroy()
    with pytest.raises(lib.NttError):
        lib.Plan(1 << 19, oracle.find_prime(50, 1 << 19), oracle.min_root(oracle.find_prime(50, 1 << 19), 1 << 19), arith=lib.ARITH_U64_R4)


@pytest.mark.parametrize("i", range(19))
def test_lazy_outputs(lib, oracle, kat, i):
    """ntt_fwd_batch_lazy / ntt_inv_batch_lazy: integer policy = the reference's lazy values bit for
    bit; FP64 policy = values in [0,4q) congruent to them; lazy -> wide chains without any reduction pass"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    a = oracle.fill_uniform(3 * n, q, 1500 + i)
    cx = oracle.ctx(n, q, w)
    pu = lib.Plan(n, q, w, arith=lib.ARITH_U64)
    lz = pu.fwd_host(a, lazy=True)
    assert np.array_equal(lz, cx.fwd_lazy(a))
    back = pu.inv_host(lz, wide=True, lazy=True)
    assert int(back.max()) < 2 * q and np.array_equal(back % np.uint64(q), a)
    pu.destroy()
    if q <= (1 << 51) + (1 << 41):
        pf = lib.Plan(n, q, w, arith=lib.ARITH_F64)
        lf = pf.fwd_host(a, lazy=True)
        assert int(lf.max()) < 4 * q and np.array_equal(lf % np.uint64(q), cx.fwd(a))
        assert np.array_equal(pf.inv_host(lf, wide=True, lazy=True), a)
        pf.destroy()


@pytest.mark.parametrize("i", range(19))
def test_reference_lazy_signatures_are_bit_exact(lib, oracle, kat, lazy_words, i):
    """the *_lazy entry points return exactly what the reference returns BEFORE its header-inline final
    reduction: the integer policies run the reference's own butterflies on the caller's w AND w_con"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    a = oracle.fill_uniform(n, q, 1700 + i)
    U64P = lib.U64P
    tw, twc, e, ec = cx.table("w"), cx.table("wcon"), cx.table("e"), cx.table("econ")
    x = a.copy()
    lib._lib.fwd_ntt_ref_harvey_lazy(x.ctypes.data_as(U64P), n, q, tw.ctypes.data_as(U64P), twc.ctypes.data_as(U64P))
    assert np.array_equal(x, cx.fwd_lazy(a))
    x = a.copy()
    lib._lib.fwd_ntt_seal_lazy(x.ctypes.data_as(U64P), n, q, tw.ctypes.data_as(U64P), twc.ctypes.data_as(U64P))
    assert int(x.max()) < 4 * q and np.array_equal(x % np.uint64(q), cx.fwd(a))
    # radix-4 and radix-4x4: each formulation's OWN words (they differ when log2 N = 4k+3: cases 3, 14, 15), on this input
    # and on the uniform input whose digests tests/golden/lazy_words.json holds straight from the compiled reference
    u = oracle.fill_uniform(n, q, 0x5EED5EED, i << 32)
    lz = lazy_words["cases"][i]
    assert oracle.fnv(u) == lz["uni_in_fnv"]
    for fn, want, name in ((lib._lib.fwd_ntt_radix4_lazy, cx.fwd_r4_lazy, "radix4"),
                           (lib._lib.fwd_ntt_radix4x4_lazy, cx.fwd_r4x4_lazy, "radix4x4")):
        x = a.copy()
        fn(x.ctypes.data_as(U64P), n, q, e.ctypes.data_as(U64P), ec.ctypes.data_as(U64P))
        assert np.array_equal(x, want(a)), name             # every reference case incl. 14-18 (N = 2^15..2^17, two passes)
        x = u.copy()
        fn(x.ctypes.data_as(U64P), n, q, e.ctypes.data_as(U64P), ec.ctypes.data_as(U64P))
        assert oracle.fnv(x) == lz["lazy_out_fnv"][name], name
    x = u.copy()
    lib._lib.fwd_ntt_ref_harvey_lazy(x.ctypes.data_as(U64P), n, q, tw.ctypes.data_as(U64P), twc.ctypes.data_as(U64P))
    assert oracle.fnv(x) == lz["lazy_out_fnv"]["ref_harvey"]
    lib.compat_release()


@pytest.mark.parametrize("m", [6, 7, 11])
def test_radix4x4_lazy_words_at_other_sizes(lib, oracle, m):
    """2^7 and 2^11 with a 59-bit and a 31-bit modulus (the reference's only 4k+3 cases below 2^15 use q = 65537), and 2^6
    (shared engine): the radix-4x4 symbol returns the oracle's words, large leading coefficients included"""
    n = 1 << m
    U64P = lib.U64P
    for bits in (59, 31):
        q = oracle.find_prime(bits, n)
        cx = oracle.ctx(n, q, oracle.min_root(q, n))
        a = oracle.fill_uniform(n, q, 1900 + m)
        a[:16] = q - 1
        e, ec = cx.table("e"), cx.table("econ")
        x = a.copy()
        lib._lib.fwd_ntt_radix4x4_lazy(x.ctypes.data_as(U64P), n, q, e.ctypes.data_as(U64P), ec.ctypes.data_as(U64P))
        assert np.array_equal(x, cx.fwd_r4x4_lazy(a))
        assert np.array_equal(x, cx.fwd_r4_lazy(a)) == (m % 4 != 3)
    lib.compat_release()


def test_plan_options(lib, oracle, kat):
    """ntt_plan_set_option replaces the environment variables of round 1: every knob gives identical results"""
    c = kat["cases"][17]                       # m = 16, q = 0x7fffffffe0001: multi-pass, chunked
    n, q, w = 1 << c["m"], c["q"], c["w"]
    a = oracle.fill_uniform(6 * n, q, 4242)
    expect = oracle.ctx(n, q, w).fwd(a)
    plan = lib.Plan(n, q, w)
    for opt, val in ((lib.OPT_CHUNK_MIB, 1), (lib.OPT_MAX_GRID, 64), (lib.OPT_F64_CLASS, 0), (lib.OPT_CHUNK_MIB, 256),
                     (lib.OPT_MAX_GRID, 0), (lib.OPT_TWO_PHASE, 0), (lib.OPT_MAX_GRID, 3), (lib.OPT_TWO_PHASE, 1), (lib.OPT_MAX_GRID, 0)):
        plan.set_option(opt, val)
        assert np.array_equal(plan.fwd_host(a), expect), (opt, val)
        assert np.array_equal(plan.inv_host(expect), a), (opt, val)
    with pytest.raises(lib.NttError):
        plan.set_option(lib.OPT_F64_CLASS, 18)   # q is a 51-bit prime: class 18 is not valid for it
    with pytest.raises(lib.NttError):
        plan.set_option(99, 1)
    plan.destroy()


def test_reference_test_driver_drop_in_fp64_engine():
    """the same unchanged reference driver with NTT_COMPAT_ARITH=f64: the reference-signature entry points are
    then served by the FP64 throughput kernels wherever q <= 2^51"""
    exe = os.path.join(ROOT, "oracle", "_ref", "ntt-variants-dropin")
    assert os.path.exists(exe), "oracle/_ref/ntt-variants-dropin did not travel to the GPU box"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, NTT_COMPAT_ARITH="f64"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("Test ") == 19 and "Bad results" not in out.stdout


def test_reference_test_driver_drop_in_device_staging():
    """the same driver with NTT_COMPAT_ZERO_COPY=0: single-pass transforms staged through device memory (H2D, kernel, D2H) as in
    rounds 1-4 instead of running in place on the pinned, device-mapped host buffer (the default since round 5)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "ntt-variants-dropin")
    assert os.path.exists(exe), "oracle/_ref/ntt-variants-dropin did not travel to the GPU box"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, NTT_COMPAT_ZERO_COPY="0"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("Test ") == 19 and "Bad results" not in out.stdout


@pytest.mark.parametrize("i", (14, 15, 16, 17, 18))
@pytest.mark.parametrize("arith", ("u64", "f64"))
def test_two_phase_equals_per_pass_path(lib, oracle, kat, i, arith):
    """N = 2^15..2^17: the one-launch two-phase kernel and the round-1 per-pass launches (NTT_OPT_TWO_PHASE 0) give
    the oracle's result, forward and inverse, wide and lazy, for batches around the grid size"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    ar = lib.ARITH_U64 if arith == "u64" else lib.ARITH_F64
    if ar == lib.ARITH_F64 and q > (1 << 51) + (1 << 41):
        pytest.skip("modulus above the FP64 range")
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith=ar)
    for batch in (1, 5):
        a = oracle.fill_uniform(batch * n, q, 2000 + i + batch)
        expect = cx.fwd(a)
        for two in (1, 0):
            plan.set_option(lib.OPT_TWO_PHASE, two)
            assert np.array_equal(plan.fwd_host(a), expect), (batch, two)
            assert np.array_equal(plan.inv_host(expect), a), (batch, two)
            lz = plan.fwd_host(a, lazy=True)
            assert int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), expect)
            assert np.array_equal(plan.inv_host(lz, wide=True), a)
    plan.set_option(lib.OPT_TWO_PHASE, 1)
    plan.set_option(lib.OPT_MAX_GRID, 2)          # more polynomials than workgroups: the persistent loop wraps
    a = oracle.fill_uniform(7 * n, q, 77)
    assert np.array_equal(plan.fwd_host(a), cx.fwd(a))
    plan.destroy()


@pytest.mark.parametrize("m,bits,batch", [(12, 50, 65536), (14, 51, 131072), (16, 51, 8192)])
def test_full_batch_forward_cross_check(lib, oracle, kat, m, bits, batch):
    """every polynomial of the config-2/3/4 shares, not a sample: the fused kernels and the independent
    column-pass path (ntt_plan_set_generic) must give the same per-polynomial checksums of the FORWARD
    transform (closes the gap a cancelling fwd/inv defect could hide in)"""
    n = 1 << m
    if bits == 51:
        c = kat["cases"][12 if m == 14 else 17]
        q, w = c["q"], c["w"]
    else:
        q = lib.find_prime(bits, n)
        w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w)
    buf = lib.DeviceBuffer(batch * n)
    cs = [lib.DeviceBuffer(batch) for _ in range(2)]
    for k, generic in enumerate((0, 1)):
        lib.fill_uniform(buf.ptr, batch * n, q, UNI_SEED, 0)
        plan.set_generic(generic)
        plan.fwd(buf.ptr, batch)
        lib.poly_checksum(cs[k].ptr, buf.ptr, n, batch)
    a, b = cs[0].download(), cs[1].download()
    assert np.array_equal(a, b)
    # and the checksum itself is the oracle's for one polynomial
    p = batch // 3
    assert oracle.checksum(oracle.ctx(n, q, w).fwd(oracle.fill_uniform(n, q, UNI_SEED, p * n))) == int(b[p])
    for x in (buf, *cs):
        x.free()
    plan.destroy()


@pytest.mark.parametrize("m", [14, 13, 12, 11, 10, 9, 8])
@pytest.mark.parametrize("q", [0x7fffffffe0001, 0x80000001c0001, 0x3ffffffdf0001, 0x7ffe0001, 0xffffffff00001])
def test_fused_product_kernel(lib, oracle, q, m):
    """N = 2^8 .. 2^14, FP64: negacyclic_mul = ONE kernel (a and b through the forward stages, product in registers,
    inverse; neither transform ever leaves the CU) -- or, NTT_OPT_FUSED_PRODUCT 2, fwd(a) by a launch of its own + ONE
    kernel (fwd(b) * a^ -> inverse); both equal the oracle's inv(fwd(a) . fwd(b)), the four-launch chain
    (NTT_OPT_FUSED_PRODUCT 0) and, for one polynomial, the schoolbook product; all aliasing forms; batches around the
    persistent grid.  (0xffffffff00001: a 52-bit prime, served by the reduce-both-operands FP64 policy.)"""
    n = 1 << m
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith=lib.ARITH_F64)
    for batch in (1, 3, 300):
        a = oracle.fill_uniform(batch * n, q, 61)
        b = oracle.fill_uniform(batch * n, q, 62)
        if batch == 3:
            b[:8] = [0, 1, q - 1, q - 2, 2, 3, q // 2, q // 2 + 1]
        # the oracle on every polynomial of the small batches and on a sample of the large one (first, last, and the
        # polynomials either side of the persistent grid's wrap-around at 256 workgroups)
        sample = list(range(batch)) if batch < 300 else [0, 1, 127, 255, 256, 257, 298, 299]
        pick = np.concatenate([np.arange(p * n, (p + 1) * n) for p in sample])
        expect = cx.inv(oracle.pointwise(cx.fwd(a[pick]), cx.fwd(b[pick]), q))
        da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
        outs = []
        for fused in (1, 2, 0):
            plan.set_option(lib.OPT_FUSED_PRODUCT, fused)
            da.upload(a), db.upload(b)
            plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
            outs.append(dc.download())
            if fused == 1:
                assert np.array_equal(da.download(), a), batch      # the one-launch form only reads a
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), batch
        assert np.array_equal(outs[0][pick], expect), batch
        for fused in (1, 2):
            plan.set_option(lib.OPT_FUSED_PRODUCT, fused)
            for alias in ("a", "b"):
                da.upload(a), db.upload(b)
                plan.negacyclic_mul((da if alias == "a" else db).ptr, da.ptr, db.ptr, batch)
                assert np.array_equal((da if alias == "a" else db).download(), outs[0]), (batch, alias, fused)
        plan.set_option(lib.OPT_FUSED_PRODUCT, 1)
        da.upload(a)
        plan.negacyclic_mul(dc.ptr, da.ptr, da.ptr, batch)          # squaring takes the four-launch chain
        if batch == 1:
            assert np.array_equal(dc.download(), oracle.schoolbook(a.copy(), a.copy(), n, q))
        for x in (da, db, dc):
            x.free()
    a1, b1 = oracle.fill_uniform(n, q, 71), oracle.fill_uniform(n, q, 72)
    da, db = lib.DeviceBuffer(n).upload(a1), lib.DeviceBuffer(n).upload(b1)
    plan.negacyclic_mul(da.ptr, da.ptr, db.ptr, 1)
    assert np.array_equal(da.download(), oracle.schoolbook(a1.copy(), b1.copy(), n, q))
    plan.destroy()


@pytest.mark.parametrize("bits", [50, 52])
@pytest.mark.parametrize("m", [15, 16, 17])
def test_fused_product_kernel_large(lib, oracle, m, bits):
    """N = 2^15..2^17: the product fuses block by block (column stages on b, ONE launch fwd block * a^ block -> inverse
    block, inverse column stages): equals the oracle and the four-transform chain, also chunk by chunk"""
    n = 1 << m
    q = lib.find_prime(bits, n, 1)       # 52 bits: the reduce-both-operands FP64 policy
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith=lib.ARITH_F64)
    assert plan.info()["f64_class"] == (52 if bits == 52 else 1)
    batch = 5
    a = oracle.fill_uniform(batch * n, q, 81)
    b = oracle.fill_uniform(batch * n, q, 82)
    expect = cx.inv(oracle.pointwise(cx.fwd(a), cx.fwd(b), q))
    da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
    for fused, chunk, blk in ((1, 256, 0), (1, 1, 0), (0, 256, 0), (1, 256, 14), (1, 1, 12 if m < 17 else 14)):
        plan.set_option(lib.OPT_FUSED_PRODUCT, fused)
        plan.set_option(lib.OPT_CHUNK_MIB, chunk)
        plan.set_option(lib.OPT_BLOCK_LOG, blk)      # blocks of the fused launch (and of the transforms): 2^12 or 2^14
        da.upload(a), db.upload(b)
        plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), expect), (fused, chunk, blk)
    plan.set_option(lib.OPT_BLOCK_LOG, 0)
    plan.set_option(lib.OPT_FUSED_PRODUCT, 1)
    da.upload(a), db.upload(b)
    plan.negacyclic_mul(db.ptr, da.ptr, db.ptr, batch)           # c aliases b
    assert np.array_equal(db.download(), expect)
    for chunk in (256, 1):                                       # c aliases a: whole batch in one chunk, and chunk by chunk
        plan.set_option(lib.OPT_CHUNK_MIB, chunk)
        da.upload(a), db.upload(b)
        plan.negacyclic_mul(da.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(da.download(), expect), chunk
    plan.destroy()


@pytest.mark.parametrize("i", [0, 4, 9, 12, 13, 17, 18])
def test_device_built_tables(lib, oracle, kat, i):
    """ntt_plan_create builds every table ON THE DEVICE: powers, 128-by-64-bit Shoup quotients, the
    expanded radix-4 table and the folded N^-1 records equal the oracle's (= the reference's pre_compute.h) tables
    bit for bit; the FP64 tables hold the balanced residues exactly and their quotients to within one rounding"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    pu = lib.Plan(n, q, w, arith=lib.ARITH_U64)
    for which, tab, con in ((0, "w", "wcon"), (1, "winv", "winv_con")):
        t = pu.export_table(which, n)
        assert np.array_equal(t[:, 0], cx.table(tab)) and np.array_equal(t[:, 1], cx.table(con)), which
    ext = pu.export_table(1, n + 16)[n:]
    winv = cx.table("winv")
    for k in range(min(16, n)):
        v = (int(cx.c.ninv) * int(winv[k])) % q
        assert int(ext[k, 0]) == v and int(ext[k, 1]) == (v << 64) // q
    pu.destroy()
    if 6 <= m <= 14:
        pr = lib.Plan(n, q, w, arith=lib.ARITH_U64_R4)
        for which, tab, con in ((0, "e", "econ"), (1, "einv", "einv_con")):
            t = pr.export_table(which, 2 * n)
            assert np.array_equal(t[:, 0], cx.table(tab)) and np.array_equal(t[:, 1], cx.table(con)), which
        pr.destroy()
    if q <= (1 << 51) + (1 << 41):
        pf = lib.Plan(n, q, w, arith=lib.ARITH_F64)
        for which, tab in ((0, "w"), (1, "winv")):
            t = pf.export_table(which, n, dtype=np.float64)
            wt = cx.table(tab).astype(np.int64)
            bal = np.where(wt > q // 2, wt - q, wt).astype(np.float64)
            assert np.array_equal(t[:, 0], bal), which
            assert np.all(np.abs(t[:, 1] - bal / q) <= np.abs(bal / q) * 2.3e-16), which
            assert np.array_equal(pf.export_table(which + 2, n, dtype=np.float64), bal), which
        pf.destroy()
    # a plan built from caller tables (host route) behaves identically
    a = oracle.fill_uniform(2 * n, q, 4)
    p2 = lib.Plan.__new__(lib.Plan)
    import ctypes as C
    h = C.c_void_p()
    tw, twi = cx.table("w"), cx.table("winv")
    lib._check(lib._lib.ntt_plan_create_from_tables(C.byref(h), 0, n, q, tw.ctypes.data_as(lib.U64P), twi.ctypes.data_as(lib.U64P), 0))
    p2.h, p2.N, p2.q, p2.root, p2.device = h.value, n, q, 0, 0
    assert np.array_equal(p2.fwd_host(a), cx.fwd(a)) and np.array_equal(p2.inv_host(cx.fwd(a)), a)
    p2.destroy()


@pytest.mark.parametrize("m", [8, 11, 12, 13, 14, 15, 16, 17])
def test_wide_fp64_policy_52_bit_moduli(lib, oracle, m):
    """moduli between 2^51(1+2^-10) and 2^52 now run in FP64 (ArithF64W: both operands of every butterfly reduced)
    instead of the integer policy: AUTO picks it, results equal the oracle's and the integer policy's; config 3's
    literal wording -- N = 65536 with a true 52-bit prime -- is the m = 16 case"""
    n = 1 << m
    q = oracle.find_prime(52, n, 1)
    assert (1 << 51) + (1 << 41) < q < (1 << 52)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    info = plan.info()
    assert info["arith"] == lib.ARITH_F64 and info["f64_class"] == 52
    batch = 3
    a = oracle.fill_uniform(batch * n, q, 5200 + m)
    a[:6] = [0, 1, q - 1, q - 2, q // 2, q // 2 + 1]
    expect = cx.fwd(a)
    assert np.array_equal(plan.fwd_host(a), expect)
    assert np.array_equal(plan.inv_host(expect), a)
    assert np.array_equal(plan.fwd_host(a, lazy=True), expect)          # no lazy form below 2^53: reduced values
    assert np.array_equal(plan.inv_host(expect + np.uint64(5 * q), wide=True), a)
    pu = lib.Plan(n, q, w, arith=lib.ARITH_U64)
    assert np.array_equal(pu.fwd_host(a), expect)
    b = oracle.fill_uniform(batch * n, q, 5300 + m)
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
    plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
    assert np.array_equal(dc.download(), cx.inv(oracle.pointwise(expect, cx.fwd(b), q)))
    if m in (12, 16):
        plan.set_generic(1)
        assert np.array_equal(plan.fwd_host(a), expect)
    with pytest.raises(lib.NttError):
        plan.set_option(lib.OPT_F64_CLASS, 0)
    with pytest.raises(lib.NttError):
        lib.Plan(n, oracle.find_prime(53, n), 3, arith=lib.ARITH_F64)
    plan.destroy(), pu.destroy()


def _ntt_prime_near(oracle, bound, n, below=True):
    step = 2 * n
    q = (bound // step) * step + 1
    if below and q > bound:
        q -= step
    if not below and q <= bound:
        q += step
    while not oracle.lib.orc_is_prime(q):
        q += -step if below else step
    return q


@pytest.mark.parametrize("m", [6, 8, 11, 12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize("k,top", [(3, 58), (1, 60), (0, 61)])
def test_wide_integer_policy(lib, oracle, m, k, top):
    """moduli the FP64 policies cannot serve (q >= 2^52): NTT_ARITH_AUTO plans run the transforms with ArithU64X<K>
    (estimated Shoup quotient, no conditional subtraction per butterfly; csrc/ntt_arith.h) -- the largest prime of each
    headroom class and a 53-bit one against the oracle: forward, inverse, lazy and wide words, products and the
    NTT-domain entry points (which keep the reference's butterflies on the same tables), every narrower class forced,
    and the reference's butterflies (NTT_OPT_INT_WIDE 0 / a plan created with NTT_ARITH_U64) giving the same results"""
    n = 1 << m
    for q in (_ntt_prime_near(oracle, (1 << top) - 1, n), _ntt_prime_near(oracle, 1 << 52, n, below=False)):
        w = oracle.min_root(q, n)
        cx = oracle.ctx(n, q, w)
        plan = lib.Plan(n, q, w)
        best = 3 if q < (1 << 58) else (1 if q < (1 << 60) else 0)
        info = plan.info()
        assert info["arith"] == lib.ARITH_U64 and info["f64_class"] == 100 + best and (best == k or q < (1 << 53))
        batch = 3
        a = oracle.fill_uniform(batch * n, q, 6100 + m + k)
        a[:6] = [0, 1, q - 1, q - 2, q // 2, q // 2 + 1]
        expect = cx.fwd(a)
        b = oracle.fill_uniform(batch * n, q, 6200 + m)
        bhat = cx.fwd(b)
        prod = cx.inv(oracle.pointwise(expect, bhat, q))
        for opt in [1] + [10 + c for c in (0, 1, 3) if c <= best] + [0]:
            plan.set_option(lib.OPT_INT_WIDE, opt)
            assert plan.info()["f64_class"] == (0 if opt == 0 else 100 + (best if opt == 1 else opt - 10))
            assert np.array_equal(plan.fwd_host(a), expect), (hex(q), opt)
            assert np.array_equal(plan.inv_host(expect), a), (hex(q), opt)
            lz = plan.fwd_host(a + np.uint64(3 * q), lazy=True)                    # lazy words in, lazy words out
            assert int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), expect)
            lb = plan.inv_host(expect + np.uint64(q if opt == 0 else 3 * q), lazy=True)   # (the reference's inverse takes [0,2q))
            assert int(lb.max()) < 2 * q and np.array_equal(lb % np.uint64(q), a), (hex(q), opt)
            assert np.array_equal(plan.fwd_host(a + np.uint64(7 * q), wide=True), expect)
            assert np.array_equal(plan.inv_host(expect + np.uint64(7 * q), wide=True), a)
            da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
            plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
            assert np.array_equal(dc.download(), prod), (hex(q), opt)
            da.upload(expect), db.upload(bhat)
            plan.inv_product(dc.ptr, da.ptr, db.ptr, batch)
            assert np.array_equal(dc.download(), prod), (hex(q), opt)
            da.upload(a), db.upload(bhat)
            plan.fwd_mul(dc.ptr, da.ptr, db.ptr, batch)
            assert np.array_equal(dc.download(), oracle.pointwise(expect, bhat, q)), (hex(q), opt)
            da.free(), db.free(), dc.free()
        if m in (12, 16):
            plan.set_option(lib.OPT_INT_WIDE, 1)
            plan.set_generic(1)                                                      # column passes only: the reference's butterflies
            assert np.array_equal(plan.fwd_host(a), expect) and np.array_equal(plan.inv_host(expect), a)
        with pytest.raises(lib.NttError):
            plan.set_option(lib.OPT_INT_WIDE, 12)
        if best < 3:
            with pytest.raises(lib.NttError):
                plan.set_option(lib.OPT_INT_WIDE, 13)
        pu = lib.Plan(n, q, w, arith=lib.ARITH_U64)                                  # explicit: the reference's butterflies and lazy words
        assert pu.info()["f64_class"] == 0
        assert np.array_equal(pu.fwd_host(a, lazy=True), cx.fwd_lazy(a))
        plan.destroy(), pu.destroy()
    small = lib.Plan(1 << 8, 0x1e01, oracle.min_root(0x1e01, 1 << 8), arith=lib.ARITH_U64)
    with pytest.raises(lib.NttError):
        small.set_option(lib.OPT_INT_WIDE, 1)                                        # below 2^40: not served
    with pytest.raises(lib.NttError):
        lib.Plan(n, oracle.find_prime(50, n), oracle.min_root(oracle.find_prime(50, n), n)).set_option(lib.OPT_INT_WIDE, 1)   # an FP64 plan
    small.destroy()


def test_compat_device_selection_env():
    """NTT_DEVICE (with NTT_COMPAT_ARITH and NTT_COMPAT_ZERO_COPY the only environment the library reads): the reference-signature entry points
    run on that device, and a device that does not exist makes them fail loudly (stderr + abort), never silently"""
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, ontt\nfrom oracle_binding import Oracle\n"
            "lib = ontt.load(); o = Oracle(); n, q = 256, 0x1e01; w = o.min_root(q, n); cx = o.ctx(n, q, w)\n"
            "a = o.fill_uniform(n, q, 1); x = a.copy(); lib.fwd_ntt_ref_harvey(x, n, q, cx.table('w'), cx.table('wcon'))\n"
            "assert np.array_equal(x, cx.fwd(a)); print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, NTT_DEVICE="0"))
    assert ok.returncode == 0 and "ok" in ok.stdout, ok.stderr[-2000:]
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, NTT_DEVICE="63"))
    assert bad.returncode != 0 and "libntt_mi355x" in bad.stderr and "ok" not in bad.stdout


def test_c_example_runs():
    """the plain-C caller of the batched API (examples/batched_product.c): built here if it did not travel, run on the
    GPU, exit code 0 = its product coefficient equals the schoolbook value"""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "batched_product")
    libdir = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "batched_product.c"), "-L" + libdir, "-lntt_mi355x",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "schoolbook" in out.stdout


def test_c_example_pointer_batch_runs():
    """examples/pointer_batch.c: 96 polynomials allocated ONE BY ONE (ntt_dev_malloc each), transformed in one launch through
    ntt_transform_ptrs and ntt_transform_dev_ptrs, multiplied through ntt_negacyclic_mul_dev_ptrs -- from plain C; exit code 0 =
    equal to the contiguous slab word for word, product coefficient equal to the schoolbook value"""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "pointer_batch")
    libdir = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "pointer_batch.c"), "-L" + libdir, "-lntt_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("equal to the slab") == 2 and "schoolbook" in out.stdout


def test_c_example_rns_ciphertext_tensor_runs():
    """examples/rns_ciphertext_tensor.c: the tensor step of a ciphertext multiplication over four SEPARATELY ALLOCATED RNS polynomials
    (eight 50-bit primes, N = 2^14) from plain C -- ntt_rns_transform_dev_ptrs, ntt_rns_inv_dot_dev_ptrs (k = 1 over two pairs, k = 2),
    ntt_rns_negacyclic_mul_dev_ptrs, every call one launch over all limbs; one coefficient of every limb of e0, e1, e2 against the
    schoolbook value"""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "rns_ciphertext_tensor")
    libdir = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "rns_ciphertext_tensor.c"), "-L" + libdir, "-lntt_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("equal to the schoolbook value") == 3 and "word for word" in out.stdout


def test_c_example_rns_modulus_chain_runs():
    """examples/rns_chain_product.c: the RNS entry points from plain C over a chain of a 60-bit, three 50-bit and two 57-bit
    primes (runs of compatible limbs), every limb's product coefficient against the schoolbook value"""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "rns_chain_product")
    libdir = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "rns_chain_product.c"), "-L" + libdir, "-lntt_mi355x",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("schoolbook") == 6 and out.stdout.count("same coefficients") == 2 and "round trip exact" in out.stdout
    assert "MISMATCH" not in out.stdout


def test_c_example_graph_replay_runs():
    """examples/graph_replay.c: a key-switching step (three forward-side multiply-accumulates with a broadcast key, one NTT-domain
    product: four XCD-local launches at N = 2^16) captured into a HIP graph from plain C after ntt_plan_reserve, replayed four times
    on new inputs, each replay equal to the direct calls on every polynomial; the capture and the replays allocate nothing"""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "graph_replay")
    libdir = os.path.join(ROOT, "optimized-number-theoretic-transform-implementations_amd")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "graph_replay.c"), "-L" + libdir, "-lntt_mi355x", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("graph == direct calls") == 4 and "MISMATCH" not in out.stdout


def test_torch_tensors_and_streams_interoperate():
    """PyTorch is plumbing here (device memory, streams): a CUDA int64 tensor's data_ptr and a torch stream go straight
    into the C ABI; the library leaves torch's current device alone.  Run in a fresh process with torch imported FIRST:
    the torch wheel bundles its own HIP runtime under the same SONAME, and the library must bind to the copy torch
    loaded (loaded the other way round the process holds two runtimes and the second one finds no device --
    profiles/r02/torch_runtime_coexistence.txt).  This is also the order of bench.py's torch.distributed path."""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
assert torch.cuda.is_available()
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, q, w, batch = 1 << 14, 0x7fffffffe0001, 83051296654, 16
a = orc.fill_uniform(batch * n, q, 2024)
t = torch.from_numpy(a.view(np.int64)).to("cuda:0")
plan = lib.Plan(n, q, w, device=0)
s = torch.cuda.Stream(device=0)
with torch.cuda.stream(s):
    plan.fwd(t.data_ptr(), batch, stream=s.cuda_stream)
    y = t.clone()                      # torch work queued behind the transform on the same stream
    plan.inv(t.data_ptr(), batch, stream=s.cuda_stream)
s.synchronize()
assert torch.cuda.current_device() == 0
assert np.array_equal(y.cpu().numpy().view(np.uint64), orc.ctx(n, q, w).fwd(a))
assert np.array_equal(t.cpu().numpy().view(np.uint64), a)
print("interop ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "interop ok" in out.stdout, out.stderr[-3000:]


def test_bench_rccl_control_plane_single_rank():
    """bench.py's torch.distributed path with the real backend (nccl = RCCL): torch first, then the library, process group
    on the GPU, barrier, MAX-reduction of the elapsed time and all_gather of the kernel times on device tensors -- with
    the one rank a 1-GPU box allows (the 2-rank form runs folded with gloo: RCCL refuses two ranks on one device)"""
    import json
    import sys
    env = dict(os.environ, NTT_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "8192", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e5 and len(d["roofline"]["kernel_ms_per_gpu"]) == 1


def test_fp64_class_boundary_moduli_on_device(lib, oracle):
    """moduli at the edges of the FP64 headroom classes, adversarial inputs, every kernel family (2^9, 2^12, 2^14, 2^16):
    FP64 result == integer-policy result == oracle"""
    from test_emu import _prime_near
    for m in (9, 12, 14, 16):
        n = 1 << m
        for bound, below in (((1 << 51) + (1 << 41), True), ((1 << 51) + (1 << 41), False), ((1 << 50) + (1 << 40), True),
                             ((1 << 50) + (1 << 40), False), ((1 << 33) + (1 << 23), True), ((1 << 33) + (1 << 23), False),
                             ((1 << 52) - 1, True)):
            q = _prime_near(oracle, bound, n, below)
            w = oracle.min_root(q, n)
            a = oracle.fill_uniform(3 * n, q, m)
            a[n:2 * n] = q - 1
            a[2 * n:3 * n:2] = 0
            a[2 * n + 1:3 * n:2] = q - 1
            pf, pu = lib.Plan(n, q, w, arith=lib.ARITH_F64), lib.Plan(n, q, w, arith=lib.ARITH_U64)
            f = pf.fwd_host(a)
            assert np.array_equal(f, pu.fwd_host(a)), (m, hex(q))
            if m <= 14:
                assert np.array_equal(f, oracle.ctx(n, q, w).fwd(a)), (m, hex(q))
            assert np.array_equal(pf.inv_host(f), a), (m, hex(q))
            pf.destroy(), pu.destroy()


def test_rmw_probe_touches_every_word_and_nothing_else(lib, oracle):
    """bench.py's measured memory ceiling: read, XOR, write back in place -- mask 0 leaves the data alone, a mask
    applied twice restores it, the words behind the range are not touched, bad arguments are refused"""
    n, mask = (1 << 16) + 2, 0x0123456789abcdef
    a = oracle.fill_uniform(n + 2, (1 << 61) - 1, 99)
    d = lib.DeviceBuffer(a.size).upload(a)
    lib.rmw_probe(d.ptr, n, 0)
    assert np.array_equal(d.download(), a)
    lib.rmw_probe(d.ptr, n, mask)
    want = a.copy()
    want[:n] ^= np.uint64(mask)
    assert np.array_equal(d.download(), want)
    lib.rmw_probe(d.ptr, n, mask)
    assert np.array_equal(d.download(), a)
    with pytest.raises(lib.NttError):
        lib.rmw_probe(d.ptr, n + 1, 0)          # odd length
    with pytest.raises(lib.NttError):
        lib.rmw_probe(d.ptr + 8, n, 0)          # not 16-byte aligned
    lib.rmw_probe(d.ptr, 0, 0)
    d.free()


def test_product_pipeline_captured_in_a_hip_graph():
    """Launch-bound use (small batches, serving): the whole fwd -> fused multiply+inverse chain of a 2-limb RNS product
    is captured ONCE into a HIP graph (torch.cuda.graph on a side stream; the library only enqueues kernels on the
    stream it is given -- no allocation, synchronisation or device query inside the batched entry points) and replayed
    on new inputs; every replay equals the oracle's schoolbook-free product (fwd, pointwise, inv)."""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch = 1 << 14, 4
qs = [0x7fffffffe0001, 0x3ffffffdf0001]
ws = [lib.min_root(q, n) for q in qs]
plans = [lib.Plan(n, q, w, device=0) for q, w in zip(qs, ws)]
ta = torch.zeros(len(qs) * batch * n, dtype=torch.int64, device="cuda:0")
tb, tc = torch.zeros_like(ta), torch.zeros_like(ta)
sa, sb = ta.clone(), tb.clone()                      # static inputs the graph reads from (the chain works in place)
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=s):
    ta.copy_(sa); tb.copy_(sb)
    lib.rns_negacyclic_mul(plans, tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=torch.cuda.current_stream().cuda_stream)
for seed in (1, 2, 3):
    a = np.concatenate([orc.fill_uniform(batch * n, q, 100 * seed + i) for i, q in enumerate(qs)])
    b = np.concatenate([orc.fill_uniform(batch * n, q, 200 * seed + i) for i, q in enumerate(qs)])
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay()
    torch.cuda.synchronize()
    got = tc.cpu().numpy().view(np.uint64)
    for i, (q, w) in enumerate(zip(qs, ws)):
        cx, sl = orc.ctx(n, q, w), slice(i * batch * n, (i + 1) * batch * n)
        want = cx.inv(orc.pointwise(cx.fwd(a[sl]), cx.fwd(b[sl]), q))
        assert np.array_equal(got[sl], want), (seed, i)
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_single_launch_two_pass_transform_captured_in_a_hip_graph():
    """The single-launch two-pass transform (N = 2^15, 512 polynomials: its default range) keeps queue heads and counters
    in a buffer the plan allocates per stream on first use.  Allocation cannot be captured, so (cold) a capture on a stream
    the plan has not seen takes the per-pass launches, and (warm) a capture on a stream that already owns a buffer records
    the clearing kernel + the one launch; so does one on a stream ntt_plan_reserve prepared.  All three graphs, replayed on fresh inputs,
    equal the oracle on sampled polynomials and each other on all of them; so does the product chain."""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch, q = 1 << 15, 512, 0x7fffffffe0001
w = lib.min_root(q, n)
cx = orc.ctx(n, q, w)
outs = {}
for mode in ("cold", "warm", "reserved"):
    plan = lib.Plan(n, q, w, device=0)
    ta = torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
    tb, tc, sa, sb = torch.zeros_like(ta), torch.zeros_like(ta), torch.zeros_like(ta), torch.zeros_like(ta)
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
    s.wait_stream(torch.cuda.current_stream())
    if mode == "reserved":
        plan.reserve(batch, stream=s.cuda_stream)         # ntt_plan_reserve: the control blocks exist before the capture begins
    if mode == "warm":
        with torch.cuda.stream(s):
            plan.fwd(ta.data_ptr(), batch, stream=s.cuda_stream)
            plan.negacyclic_mul(tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=s.cuda_stream)
        s.synchronize()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        ta.copy_(sa); tb.copy_(sb)
        plan.fwd(ta.data_ptr(), batch, stream=st)          # ta = fwd(a)
        tc.copy_(ta)
        plan.inv(tc.data_ptr(), batch, stream=st)          # tc = a again
        ta.copy_(sa)
        plan.negacyclic_mul(tb.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)   # tb = a * b (aliases b)
    for seed in (1, 2):
        a = orc.fill_uniform(batch * n, q, 10 * seed); b = orc.fill_uniform(batch * n, q, 10 * seed + 1)
        sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
        g.replay(); torch.cuda.synchronize()
        back, prod = tc.cpu().numpy().view(np.uint64), tb.cpu().numpy().view(np.uint64)
        assert np.array_equal(back, a), (mode, seed)
        for j in (0, 1, 255, 256, 511):
            sl = slice(j * n, (j + 1) * n)
            want = cx.inv(orc.pointwise(cx.fwd(a[sl]), cx.fwd(b[sl]), q))
            assert np.array_equal(prod[sl], want), (mode, seed, j)
        outs[(mode, seed)] = prod.copy()
for seed in (1, 2):
    assert np.array_equal(outs[("cold", seed)], outs[("warm", seed)]), seed
    assert np.array_equal(outs[("cold", seed)], outs[("reserved", seed)]), seed
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_captured_xcd_local_launches_replayed_between_other_work():
    """Round 5 found that a captured hipMemsetAsync in front of a captured XCD-local launch does not reliably take effect before the
    kernel's first workgroups read the control block when the graph is replayed between other GPU work (the round-4 library returns
    96 of 96 wrong polynomials here from the second replay on: profiles/r05/graph_replay_control_block_clear.txt); the block is now
    cleared by a kernel.  A graph that BEGINS with the XCD-local launches (forward transform, product), six replays with 2 GiB of
    unrelated traffic between them, every polynomial against the oracle."""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch, q = 1 << 16, 96, 0x7fffffffe0001
w = lib.min_root(q, n)
cx = orc.ctx(n, q, w)
plan = lib.Plan(n, q, w, device=0)
plan.set_option(lib.OPT_XCD_LOCAL, 1)
z = lambda: torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
ta, tb, tc = z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
plan.reserve(batch, stream=s.cuda_stream)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    plan.fwd(ta.data_ptr(), batch, stream=st)                                               # ta = fwd(a)
    plan.negacyclic_mul(tc.data_ptr(), tb.data_ptr(), tb.data_ptr(), batch, stream=st)      # tc = b * b
for seed in (1, 2, 3, 1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 30 * seed); b = orc.fill_uniform(batch * n, q, 30 * seed + 1)
    ta.copy_(torch.from_numpy(a.view(np.int64))); tb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay(); torch.cuda.synchronize()
    got_f, got_p = ta.cpu().numpy().view(np.uint64), tc.cpu().numpy().view(np.uint64)
    for j in range(batch):
        sl = slice(j * n, (j + 1) * n)
        fb = cx.fwd(b[sl].copy())
        assert np.array_equal(got_f[sl], cx.fwd(a[sl].copy())), ("forward", seed, j)
        assert np.array_equal(got_p[sl], cx.inv(orc.pointwise(fb, fb, q))), ("product", seed, j)
    big = torch.empty(1 << 28, dtype=torch.int64, device="cuda:0"); big.fill_(1); torch.cuda.synchronize(); del big
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_rns_xcd_local_launches_over_limbs_captured_in_a_hip_graph():
    """the same for ONE launch over the limbs of an RNS set in the caller-native layout [batch][limb][N] (MULTI variants: the limb in the
    queue entry, control block of the run's first plan): product, NTT-domain inner product and forward-side product captured after
    ntt_plan_reserve on the first plan, replayed between unrelated traffic, every word against the uncaptured per-limb per-chunk
    calls, samples against the oracle"""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, nl, batch = 1 << 15, 3, 40
qs = [lib.find_prime(50, n, i) for i in range(nl)]
ws = [lib.min_root(q, n) for q in qs]
plans = [lib.Plan(n, q, w, device=0) for q, w in zip(qs, ws)]
refs = [lib.Plan(n, q, w, device=0) for q, w in zip(qs, ws)]
for p in plans: p.set_option(lib.OPT_XCD_LOCAL, 1)
for p in refs: p.set_option(lib.OPT_XCD_LOCAL, 0)
lib.set_rns_launch(plans, 0); lib.set_rns_launch(refs, 1)
lay = (n, nl * n)                                    # [batch][limb][N]
words = nl * batch * n
z = lambda: torch.zeros(words, dtype=torch.int64, device="cuda:0")
sa, sb, ta, tb, tc, td, te = z(), z(), z(), z(), z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
plans[0].reserve(nl * batch, stream=s.cuda_stream)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    lib.rns_inv_dot(plans, tc.data_ptr(), [sa.data_ptr(), sb.data_ptr()], [sb.data_ptr(), sa.data_ptr()], batch, stream=st, layout=lay)   # inv(2 a^ . b^)
    ta.copy_(sa); tb.copy_(sb)
    lib.rns_negacyclic_mul(plans, td.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st, layout=lay)                              # a * b
    ta.copy_(sa)
    lib.rns_fwd_mul(plans, te.data_ptr(), ta.data_ptr(), sb.data_ptr(), batch, stream=st, layout=lay)                                      # fwd(a) . b^
rng = np.random.default_rng(5)
for rep in range(3):
    a = rng.integers(0, min(qs), size=words, dtype=np.uint64); b = rng.integers(0, min(qs), size=words, dtype=np.uint64)
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay(); torch.cuda.synchronize()
    got = [x.cpu().numpy().view(np.uint64).copy() for x in (tc, td, te)]
    da, db, dc = lib.DeviceBuffer(words).upload(a), lib.DeviceBuffer(words).upload(b), lib.DeviceBuffer(words)
    lib.rns_inv_dot(refs, dc.ptr, [da.ptr, db.ptr], [db.ptr, da.ptr], batch, layout=lay)
    assert np.array_equal(got[0], dc.download()), ("inv_dot", rep)
    lib.rns_negacyclic_mul(refs, dc.ptr, da.ptr, db.ptr, batch, layout=lay)
    assert np.array_equal(got[1], dc.download()), ("mul", rep)
    da.upload(a); db.upload(b)                         # (a product leaves scratch in both operands above 2^14)
    lib.rns_fwd_mul(refs, dc.ptr, da.ptr, db.ptr, batch, layout=lay)
    assert np.array_equal(got[2], dc.download()), ("fwd_mul", rep)
    l, p_ = nl - 1, batch - 1
    cx = orc.ctx(n, qs[l], ws[l])
    sl = slice((p_ * nl + l) * n, (p_ * nl + l + 1) * n)
    assert np.array_equal(got[1][sl], cx.inv(orc.pointwise(cx.fwd(a[sl].copy()), cx.fwd(b[sl].copy()), qs[l])))
    assert np.array_equal(got[2][sl], orc.pointwise(cx.fwd(a[sl].copy()), b[sl], qs[l]))
    for x in (da, db, dc): x.free()
    big = torch.empty(1 << 27, dtype=torch.int64, device="cuda:0"); big.fill_(1); torch.cuda.synchronize(); del big
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_one_launch_ntt_domain_products_captured_in_a_hip_graph():
    """round 5's one-launch kernels (team_dot_kernel, team_mul_kernel) inside a HIP graph: control blocks from ntt_plan_reserve, the
    capture records the clearing kernel + one launch per call; two replays on fresh inputs with direct (uncaptured, per-chunk) calls
    between them -- the interleaving that exposed the unreliable captured memset -- every word, and one polynomial against the oracle"""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch, q = 1 << 16, 96, 0x7fffffffe0001
w = lib.min_root(q, n)
cx = orc.ctx(n, q, w)
plan, ref = lib.Plan(n, q, w, device=0), lib.Plan(n, q, w, device=0)
plan.set_option(lib.OPT_XCD_LOCAL, 1); ref.set_option(lib.OPT_XCD_LOCAL, 0)
z = lambda: torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
sa, sb, ta, tb, tc, td = z(), z(), z(), z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
plan.reserve(batch, stream=s.cuda_stream)
before = plan.get_option(lib.OPT_CTL_ALLOCATIONS)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    ta.copy_(sa); tb.copy_(sb)
    plan.inv_product(tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)     # tc = inv(a^ . b^)
    plan.fwd_mul(td.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)         # td = fwd(a) . b^   (ta is scratch)
assert plan.get_option(lib.OPT_CTL_ALLOCATIONS) == before
for seed in (1, 2):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay(); torch.cuda.synchronize()
    got_c, got_d = tc.cpu().numpy().view(np.uint64).copy(), td.cpu().numpy().view(np.uint64).copy()
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
    ref.inv_product(dc.ptr, da.ptr, db.ptr, batch)
    assert np.array_equal(got_c, dc.download()), seed
    ref.fwd_mul(dc.ptr, da.ptr, db.ptr, batch)
    assert np.array_equal(got_d, dc.download()), seed
    j = batch - 1
    sl = slice(j * n, (j + 1) * n)
    assert np.array_equal(got_c[sl], cx.inv(orc.pointwise(a[sl], b[sl], q))) and np.array_equal(got_d[sl], orc.pointwise(cx.fwd(a[sl]), b[sl], q))
    for x in (da, db, dc): x.free()
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_concurrent_host_threads_on_their_own_streams(lib, oracle, kat):
    """the batched API is re-entrant like the reference's functions (no statics on that path, thread-local error
    text): four host threads, each with its own plan, stream and buffers, transform concurrently"""
    import ctypes as C
    import threading
    cases = [kat["cases"][i] for i in (4, 9, 12, 13)]
    errors, results = [], {}

    def work(idx, c):
        try:
            n, q, w = 1 << c["m"], c["q"], c["w"]
            plan = lib.Plan(n, q, w)
            h = C.c_void_p()
            lib._check(lib._lib.ntt_stream_create(0, C.byref(h)))
            a = oracle.fill_uniform(24 * n, q, 1000 + idx)
            d = lib.DeviceBuffer(a.size).upload(a)
            for _ in range(20):
                plan.fwd(d.ptr, 24, stream=h.value)
                plan.inv(d.ptr, 24, stream=h.value)
            plan.fwd(d.ptr, 24, stream=h.value)
            lib.stream_sync(0, h.value)
            results[idx] = (d.download(), a, n, q, w)
            d.free()
            plan.destroy()
            lib._lib.ntt_stream_destroy(0, h.value)
        except Exception as e:                      # noqa: BLE001 -- reported by the main thread
            errors.append((idx, repr(e)))

    threads = [threading.Thread(target=work, args=(i, c)) for i, c in enumerate(cases)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for idx, (got, a, n, q, w) in results.items():
        assert np.array_equal(got, oracle.ctx(n, q, w).fwd(a)), idx


def test_one_plan_shared_by_host_threads_on_their_own_streams(lib, oracle):
    """ONE plan, three host threads, each with a stream of its own, all issuing XCD-local launches at once (transforms, an NTT-domain
    product, a forward-side product: team_kernel, team_dot_kernel, team_mul_kernel): the (plan, stream) pairs own separate control
    blocks, created under the plan's mutex by whichever call comes first; every result against the oracle"""
    import ctypes as C
    import threading
    n, q, batch = 1 << 15, 0x7fffffffe0001, 80
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    plan.set_option(lib.OPT_XCD_LOCAL, 1)
    errors, results = [], {}

    def work(idx):
        try:
            h = C.c_void_p()
            lib._check(lib._lib.ntt_stream_create(0, C.byref(h)))
            a = oracle.fill_uniform(batch * n, q, 8300 + idx)
            b = oracle.fill_uniform(batch * n, q, 8400 + idx)
            da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
            for _ in range(8):
                if idx == 0:
                    plan.fwd(da.ptr, batch, stream=h.value)
                    plan.inv(da.ptr, batch, stream=h.value)
                elif idx == 1:
                    plan.inv_product(dc.ptr, da.ptr, db.ptr, batch, stream=h.value)
                else:
                    lib.stream_sync(0, h.value)                    # (the upload is a blocking copy on the null stream: the streams
                    da.upload(a)                                   #  here do not synchronise with it; a is scratch above 2^14)
                    plan.fwd_mul(dc.ptr, da.ptr, db.ptr, batch, stream=h.value)
            if idx == 0:
                plan.fwd(da.ptr, batch, stream=h.value)
            lib.stream_sync(0, h.value)
            results[idx] = ((da if idx == 0 else dc).download(), a, b)
            for x in (da, db, dc):
                x.free()
            lib._lib.ntt_stream_destroy(0, h.value)
        except Exception as e:                      # noqa: BLE001 -- reported by the main thread
            errors.append((idx, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert plan.get_option(lib.OPT_CTL_ALLOCATIONS) == 6       # three streams x (direct block + graph block)
    for idx, (got, a, b) in results.items():
        for j in (0, batch // 2, batch - 1):
            sl = slice(j * n, (j + 1) * n)
            exp = cx.fwd(a[sl].copy()) if idx == 0 else (cx.inv(oracle.pointwise(a[sl], b[sl], q)) if idx == 1 else
                                                          oracle.pointwise(cx.fwd(a[sl].copy()), b[sl], q))
            assert np.array_equal(got[sl], exp), (idx, j)
    plan.destroy()


@pytest.mark.parametrize("m", [15, 16])
@pytest.mark.parametrize("arith", ["u64", "f64", "f64_52bit"])
def test_block_sizes_below_the_column_pass_agree(lib, oracle, m, arith):
    """N = 2^15, 2^16: 3 or 4 column stages over 2^12-point blocks (the default where it measured faster) and 1 or 2 over
    2^14-point blocks (NTT_OPT_BLOCK_LOG) are the same transform: both equal the oracle, forward and inverse, lazy
    outputs of the integer policy bit for bit, also chunk by chunk; 2^17 refuses the small blocks (5 leading stages)"""
    n = 1 << m
    q = lib.find_prime(52 if arith == "f64_52bit" else 50, n, 0)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    batch = 5
    a = _inputs(oracle, n, q, batch, 900 + m)
    want = cx.fwd(a)
    plan = lib.Plan(n, q, w, arith=lib.ARITH_U64 if arith == "u64" else lib.ARITH_F64)
    plan.set_option(lib.OPT_TWO_PHASE, 0)
    lazies = []
    for blk, chunk in ((0, 256), (12, 256), (14, 256), (12, 1), (14, 1)):
        plan.set_option(lib.OPT_BLOCK_LOG, blk)
        plan.set_option(lib.OPT_CHUNK_MIB, chunk)
        assert np.array_equal(plan.fwd_host(a), want), (blk, chunk, "fwd")
        assert np.array_equal(plan.inv_host(want), a), (blk, chunk, "inv")
        lz = plan.fwd_host(a, lazy=True)
        assert int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), want)
        lazies.append(lz)
    if arith == "u64":
        assert all(np.array_equal(lazies[0], x) for x in lazies) and np.array_equal(lazies[0], cx.fwd_lazy(a))
    with pytest.raises(lib.NttError):
        plan.set_option(lib.OPT_BLOCK_LOG, 13)
    plan.destroy()
    if m == 16:
        n17 = 1 << 17
        q17 = lib.find_prime(50, n17, 0)
        p17 = lib.Plan(n17, q17, lib.min_root(q17, n17))
        with pytest.raises(lib.NttError):
            p17.set_option(lib.OPT_BLOCK_LOG, 12)
        p17.destroy()


# ---- operands in the NTT domain (second half): ntt_inv_product_batch / ntt_inv_dot_batch / ntt_mul_transformed_batch
def _ntt_domain_operands(oracle, n, q, batch, k, seed, lazy, bcast):
    """k operand pairs of `batch` polynomials in the NTT domain: uniform words with the extreme residues in the first
    slots; lazy: random multiples of q added so that the words cover [0,4q) (4q - 1 included)"""
    rng = np.random.default_rng(seed)
    a_list, b_list = [], []
    for i in range(k):
        a = oracle.fill_uniform(batch * n, q, seed + 2 * i)
        b = oracle.fill_uniform((1 if bcast else batch) * n, q, seed + 2 * i + 1)
        a[:3], b[:3] = q - 1, q - 1
        a[3:6], b[3:5] = 0, q // 2
        if lazy:
            a = a + rng.integers(0, 4, a.size).astype(np.uint64) * np.uint64(q)
            b = b + rng.integers(0, 4, b.size).astype(np.uint64) * np.uint64(q)
            a[0], b[0] = 4 * q - 1, 4 * q - 1
        a_list.append(a)
        b_list.append(b)
    return a_list, b_list


_DOT_MODULI = {"f64_class0": (51, 0), "f64_class1": (50, 0), "f64_52bit": (52, 0), "f64_small": (31, 0), "u64_60bit": (60, 0)}


@pytest.mark.parametrize("cls", sorted(_DOT_MODULI))
@pytest.mark.parametrize("m", [8, 12, 13, 14, 16, 17])
def test_inv_dot_in_the_ntt_domain(lib, oracle, m, cls):
    """c = inv(sum_{i<k} a_i^ (.) b_i^), operands given in the NTT domain: ONE launch up to 2^14 (the products are formed where
    the inverse transform would convert its input words), the product riding in the inverse's first pass above; against
    the oracle's inverse of the 128-bit element-wise products' sum, for k = 1, 2, 3, 8, canonical and lazy ([0,4q)) operand
    words, per-polynomial and broadcast b, c aliasing a^ or b^ (k = 1), ragged batches either side of the persistent grid,
    chunk by chunk above 2^14; modulus classes: scheduled FP64 class 0 / 1, reduce-both-operands (52 bits), integer (60 bits)"""
    n = 1 << m
    bits, skip = _DOT_MODULI[cls]
    q = lib.find_prime(bits, n, skip)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    want = {"f64_class0": (lib.ARITH_F64, 0), "f64_class1": (lib.ARITH_F64, 1), "f64_52bit": (lib.ARITH_F64, 52),
            "f64_small": (lib.ARITH_F64, 18), "u64_60bit": (lib.ARITH_U64, 0)}[cls]
    info = plan.info()
    assert (info["arith"], info["f64_class"] if want[0] == lib.ARITH_F64 else 0) == want
    big = 300 if m <= 14 else 6
    for k, batch, lazy, bcast in ((1, 1, False, False), (1, big, False, False), (1, 3, True, False), (2, 3, False, False),
                                  (3, big if m <= 12 else 5, True, False), (8, 2, False, True), (2, 5, True, True),
                                  (1, 5, False, True)):
        a_list, b_list = _ntt_domain_operands(oracle, n, q, batch, k, 5000 + 7 * m + k, lazy, bcast)
        expect = cx.inv(oracle.dot(a_list, b_list, q, n, bcast))
        flags = (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0)
        da = [lib.DeviceBuffer(a.size).upload(a) for a in a_list]
        db = [lib.DeviceBuffer(b.size).upload(b) for b in b_list]
        dc = lib.DeviceBuffer(batch * n)
        for chunk in ((256, 1) if m > 14 else (256,)):
            plan.set_option(lib.OPT_CHUNK_MIB, chunk)
            if k == 1:
                plan.inv_product(dc.ptr, da[0].ptr, db[0].ptr, batch, flags)
            else:
                plan.inv_dot(dc.ptr, [x.ptr for x in da], [x.ptr for x in db], batch, flags)
            assert np.array_equal(dc.download(), expect), (k, batch, lazy, bcast, chunk)
        if k == 1:
            plan.inv_product(da[0].ptr, da[0].ptr, db[0].ptr, batch, flags)         # c aliases a^
            assert np.array_equal(da[0].download(), expect), ("alias a", batch, lazy, bcast)
            if not bcast:
                da[0].upload(a_list[0])
                plan.inv_product(db[0].ptr, da[0].ptr, db[0].ptr, batch, flags)     # c aliases b^
                assert np.array_equal(db[0].download(), expect), ("alias b", batch, lazy)
        for x in da + db + [dc]:
            x.free()
    plan.destroy()


@pytest.mark.parametrize("m,arith", [(4, "auto"), (10, "generic"), (12, "r4"), (15, "generic"), (16, "u64"), (10, "generic52"),
                                     (12, "unfused52"), (15, "unfused52")])
def test_inv_dot_plans_without_the_fused_kernel(lib, oracle, m, arith):
    """plans the fused kernel is not built for (N < 2^6, column-pass-only plans, the radix-4 formulation) accumulate the
    products with pointwise launches and run their own inverse; the integer policy above 2^14 takes the kernel's
    two-pass form: same results.  *52: a 52-bit modulus, whose lazy words ([0,4q)) exceed 2^53 -- the pointwise kernels fold
    them with integer operations before the conversion (ArithF64::mulmod_full_lazy4); unfused52: NTT_OPT_DOT_FUSED = 0"""
    n = 1 << m
    q = lib.find_prime(52 if arith.endswith("52") else 50 if arith != "u64" else 59, n)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w, arith={"r4": lib.ARITH_U64_R4, "u64": lib.ARITH_U64}.get(arith, lib.ARITH_AUTO))
    if arith.startswith("generic"):
        plan.set_generic(True)
    if arith == "unfused52":
        plan.set_option(lib.OPT_DOT_FUSED, 0)
    for k, batch, lazy, bcast in ((1, 3, False, False), (3, 2, True, False), (4, 3, False, True)):
        a_list, b_list = _ntt_domain_operands(oracle, n, q, batch, k, 5100 + m, lazy, bcast)
        expect = cx.inv(oracle.dot(a_list, b_list, q, n, bcast))
        da = [lib.DeviceBuffer(a.size).upload(a) for a in a_list]
        db = [lib.DeviceBuffer(b.size).upload(b) for b in b_list]
        dc = lib.DeviceBuffer(batch * n)
        plan.inv_dot(dc.ptr, [x.ptr for x in da], [x.ptr for x in db], batch,
                     (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0))
        assert np.array_equal(dc.download(), expect), (k, batch, lazy, bcast)
        for x in da + db + [dc]:
            x.free()
    plan.destroy()


@pytest.mark.parametrize("m,batch,k,bits", [(15, 640, 1, 51), (16, 520, 2, 52), (17, 513, 1, 60)])
def test_ntt_domain_products_automatic_form_at_large_batches(lib, oracle, m, batch, k, bits):
    """batches of 512 polynomials and more at N >= 2^15 take the one-launch form by themselves (dot_team_pays) with the measured
    default lags (24 / 12 / 8 polynomials: queues much longer than the lag here, unlike the ragged cases above); the control
    block comes from NTT_OPT_MAX_BATCH_HINT, not from the first call.  Every polynomial against the forced per-chunk form,
    samples against the oracle."""
    n = 1 << m
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    plan.set_option(lib.OPT_MAX_BATCH_HINT, batch)
    da = [lib.DeviceBuffer(batch * n) for _ in range(k)]
    db = [lib.DeviceBuffer(batch * n) for _ in range(k)]
    for i in range(k):
        lib.fill_uniform(da[i].ptr, batch * n, q, 7100 + i)
        lib.fill_uniform(db[i].ptr, batch * n, q, 7200 + i)
    dc = lib.DeviceBuffer(batch * n)
    res = {}
    for form in (-1, 0):
        plan.set_option(lib.OPT_XCD_LOCAL, form)
        dc.upload(np.zeros(batch * n, dtype=np.uint64))
        plan.inv_dot(dc.ptr, [x.ptr for x in da], [x.ptr for x in db], batch, 0)
        res[form] = dc.download()
    assert np.array_equal(res[-1], res[0])
    for p_ in (0, 7, batch // 2, batch - 9, batch - 1):
        a_l = [x.download(n, p_ * n) for x in da]
        b_l = [x.download(n, p_ * n) for x in db]
        assert np.array_equal(res[-1][p_ * n:(p_ + 1) * n], cx.inv(oracle.dot(a_l, b_l, q, n))), p_
    # the forward-side counterpart (team_mul_kernel; automatic from 2^26 coefficients per operand on): c^ = fwd(a) (.) b^
    res = {}
    for form in (-1, 0):
        plan.set_option(lib.OPT_XCD_LOCAL, form)
        lib.fill_uniform(da[0].ptr, batch * n, q, 7300)
        plan.fwd_mul(dc.ptr, da[0].ptr, db[0].ptr, batch, 0)
        res[form] = dc.download()
    assert np.array_equal(res[-1], res[0])
    lib.fill_uniform(da[0].ptr, batch * n, q, 7300)
    for p_ in (0, batch // 2, batch - 1):
        a_ = da[0].download(n, p_ * n)
        assert np.array_equal(res[-1][p_ * n:(p_ + 1) * n], oracle.pointwise(cx.fwd(a_), db[0].download(n, p_ * n), q)), p_
    for x in da + db + [dc]:
        x.free()
    plan.destroy()


def test_no_batched_call_allocates_after_reserve(lib, oracle):
    """ntt_plan_reserve / NTT_OPT_MAX_BATCH_HINT (round 5): once the control blocks of the XCD-local launches are sized, no batched
    call allocates -- read from the plan's own counter (NTT_OPT_CTL_ALLOCATIONS; hipMemGetInfo does not move for allocations of a
    few KiB) -- transforms, products, NTT-domain products, on the null stream and on a stream of its own, growing batches up to the
    reserved size; an unreserved plan allocates its two blocks at its first XCD-local call"""
    import ctypes as C
    n, q = 1 << 15, 0x7fffffffe0001
    w = lib.min_root(q, n)
    batch = 600
    bufs = [lib.DeviceBuffer(batch * n) for _ in range(3)]
    for i, b in enumerate(bufs):
        lib.fill_uniform(b.ptr, batch * n, q, 8100 + i)
    h = C.c_void_p()
    lib._check(lib._lib.ntt_stream_create(0, C.byref(h)))
    plan = lib.Plan(n, q, w)
    plan.set_option(lib.OPT_XCD_LOCAL, 1)
    assert plan.get_option(lib.OPT_CTL_ALLOCATIONS) == 0 and plan.get_option(lib.OPT_XCD_LOCAL) == 1
    plan.set_option(lib.OPT_MAX_BATCH_HINT, batch)          # the null stream: direct block + graph block
    plan.reserve(batch, stream=h.value)                     # and this one
    reserved = plan.get_option(lib.OPT_CTL_ALLOCATIONS)
    assert reserved == 4 and plan.get_option(lib.OPT_MAX_BATCH_HINT) == batch
    for st in (None, h.value):
        for nb in (64, 200, batch):
            plan.fwd(bufs[0].ptr, nb, stream=st)
            plan.inv(bufs[0].ptr, nb, stream=st)
            plan.negacyclic_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, nb, stream=st)
            plan.inv_product(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, nb, stream=st)
            plan.fwd_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, nb, stream=st)
            assert plan.get_option(lib.OPT_CTL_ALLOCATIONS) == reserved, (st, nb)
    lib.stream_sync(0, h.value)
    plan2 = lib.Plan(n, q, w)
    plan2.set_option(lib.OPT_XCD_LOCAL, 1)
    plan2.fwd(bufs[0].ptr, 64)
    first = plan2.get_option(lib.OPT_CTL_ALLOCATIONS)
    assert first == 2                                       # the first XCD-local call of an unreserved plan allocates its two blocks
    plan2.fwd(bufs[0].ptr, 100)
    assert plan2.get_option(lib.OPT_CTL_ALLOCATIONS) == first        # (sized with a factor of two: no regrowth yet; regrowth itself:
    #                                                                   test_captured_xcd_local_launch_survives_regrowth_and_concurrent_direct_calls)
    lib.stream_sync(0, None)
    plan.destroy(), plan2.destroy()
    lib._lib.ntt_stream_destroy(0, h.value)
    for b in bufs:
        b.free()


def test_inv_dot_bad_arguments(lib, oracle):
    n, q = 256, 0x1e01
    plan = lib.Plan(n, q, lib.min_root(q, n))
    d = lib.DeviceBuffer(n)
    with pytest.raises(lib.NttError):
        plan.inv_dot(d.ptr, [d.ptr] * 33, [d.ptr] * 33, 1)          # more pairs than one launch carries
    with pytest.raises(lib.NttError):
        plan.inv_product(d.ptr, d.ptr, d.ptr, 1, flags=64)          # unknown flag
    with pytest.raises(lib.NttError):
        plan.inv_product(0, d.ptr, d.ptr, 1)
    plan.inv_product(d.ptr, d.ptr, d.ptr, 0)                        # empty batch: nothing to do
    d.free()
    plan.destroy()


@pytest.mark.parametrize("bits", [51, 50, 52])
@pytest.mark.parametrize("m", [8, 11, 12, 13, 14, 15, 16, 17])
def test_mul_by_a_transformed_operand(lib, oracle, m, bits):
    """c = inv(fwd(a) (.) b^) with b^ transformed beforehand (a key / plaintext kept in the NTT domain): the product
    kernels' form with one operand given in the NTT domain, canonical or lazy words, every aliasing form, small batches
    (block launches) and -- N >= 2^15 -- the batch size from which the whole chain is one launch"""
    n = 1 << m
    q = lib.find_prime(bits, n)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    batches = (1, 5, 300) if m <= 14 else (3, (1 << 23) >> m)
    for batch in batches:
        a = oracle.fill_uniform(batch * n, q, 91)
        bhat = oracle.fill_uniform(batch * n, q, 92)
        bhat[:4] = [0, q - 1, 1, q // 2]
        sample = list(range(batch)) if batch <= 8 else sorted({0, 1, batch // 2, batch - 2, batch - 1, min(batch - 1, 255), min(batch - 1, 256)})
        pick = np.concatenate([np.arange(p * n, (p + 1) * n) for p in sample])
        expect = cx.inv(oracle.pointwise(cx.fwd(a[pick]), bhat[pick], q))
        lazy_b = bhat + np.uint64(q) * (np.arange(bhat.size, dtype=np.uint64) % np.uint64(3 if 4 * q <= 2 ** 53 else 1))
        da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
        for words, flags in ((bhat, 0), (lazy_b, lib.MUL_LAZY_IN)):
            da.upload(a), db.upload(words)
            plan.mul_transformed(dc.ptr, da.ptr, db.ptr, batch, flags)
            got = dc.download()
            assert np.array_equal(got[pick], expect), (batch, flags)
            assert np.array_equal(db.download(), words)                 # the transformed operand is only read
            if m <= 14:
                assert np.array_equal(da.download(), a)                  # ... and so is a up to 2^14
            da.upload(a)
            plan.mul_transformed(da.ptr, da.ptr, db.ptr, batch, flags)     # c aliases a
            assert np.array_equal(da.download(), got), (batch, flags, "alias a")
            da.upload(a)
            plan.mul_transformed(db.ptr, da.ptr, db.ptr, batch, flags)     # c aliases b^
            assert np.array_equal(db.download(), got), (batch, flags, "alias b^")
        for x in (da, db, dc):
            x.free()
    plan.destroy()


@pytest.mark.parametrize("logn,batch", [(12, 2), (14, 1), (14, 600), (16, 2), (16, 70)])
def test_rns_products_in_the_ntt_domain(lib, oracle, logn, batch, monkeypatch):
    """the NTT-domain products over RNS limbs ([limb][batch][N]; a broadcast key: [limb][N]): one launch over the limbs
    when a limb's share cannot fill the chip, limb by limb otherwise -- both forms forced in turn, against the oracle"""
    _rns_ntt_domain_check(lib, oracle, monkeypatch, logn, batch, 50)


@pytest.mark.parametrize("logn,batch,bits", [(12, 2, 57), (14, 1, 60), (16, 2, 57), (14, 3, 61)])
def test_rns_products_in_the_ntt_domain_integer_limbs(lib, oracle, logn, batch, bits, monkeypatch):
    """... and over limbs of the wide integer policy"""
    _rns_ntt_domain_check(lib, oracle, monkeypatch, logn, batch, bits)


def _rns_ntt_domain_check(lib, oracle, monkeypatch, logn, batch, bits):
    n = 1 << logn
    nl, k = 4, 3
    qs = [lib.find_prime(bits, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    slab = batch * n
    rng_ops = [[_ntt_domain_operands(oracle, n, q, batch, k, 5200 + 11 * l, False, True) for l, q in enumerate(qs)]]
    ops = rng_ops[0]
    a_all = [np.concatenate([ops[l][0][i] for l in range(nl)]) for i in range(k)]        # [limb][batch][N]
    key_all = [np.concatenate([ops[l][1][i] for l in range(nl)]) for i in range(k)]      # [limb][N]
    sample = sorted({0, batch // 2, batch - 1})
    expect = []
    for l, (q, w) in enumerate(zip(qs, ws)):
        cx = oracle.ctx(n, q, w)
        full = cx.inv(oracle.dot(ops[l][0], ops[l][1], q, n, True)) if batch <= 8 else None
        for p_ in sample:
            if full is not None:
                expect.append(full[p_ * n:(p_ + 1) * n])
            else:
                sl = slice(p_ * n, (p_ + 1) * n)
                expect.append(cx.inv(oracle.dot([x[sl] for x in ops[l][0]], ops[l][1], q, n, True)))
    da = [lib.DeviceBuffer(x.size).upload(x) for x in a_all]
    dk = [lib.DeviceBuffer(x.size).upload(x) for x in key_all]
    dc = lib.DeviceBuffer(nl * slab)
    outs = []
    for loop in ("0", "1"):
        lib.set_rns_launch(plans, loop)
        lib.rns_inv_dot(plans, dc.ptr, [x.ptr for x in da], [x.ptr for x in dk], batch, lib.MUL_B_BROADCAST)
        got = dc.download()
        outs.append(got)
        j = 0
        for l in range(nl):
            for p_ in sample:
                assert np.array_equal(got[l * slab + p_ * n: l * slab + (p_ + 1) * n], expect[j]), (loop, l, p_)
                j += 1
    assert np.array_equal(outs[0], outs[1])
    # c = inv(fwd(a) (.) b^) over the limbs, b^ per polynomial
    a_co = np.concatenate([oracle.fill_uniform(slab, q, 5300 + l) for l, q in enumerate(qs)])
    bh = np.concatenate([oracle.fill_uniform(slab, q, 5400 + l) for l, q in enumerate(qs)])
    dco, dbh = lib.DeviceBuffer(a_co.size), lib.DeviceBuffer(bh.size).upload(bh)
    outs = []
    for loop in ("0", "1"):
        lib.set_rns_launch(plans, loop)
        dco.upload(a_co)
        lib.rns_mul_transformed(plans, dc.ptr, dco.ptr, dbh.ptr, batch)
        outs.append(dc.download())
    assert np.array_equal(outs[0], outs[1])
    for l, (q, w) in enumerate(zip(qs, ws)):
        cx = oracle.ctx(n, q, w)
        for p_ in sample:
            sl = slice(l * slab + p_ * n, l * slab + (p_ + 1) * n)
            assert np.array_equal(outs[0][sl], cx.inv(oracle.pointwise(cx.fwd(a_co[sl]), bh[sl], q))), (l, p_)
    for x in da + dk + [dc, dco, dbh]:
        x.free()
    for p in plans:
        p.destroy()


@pytest.mark.parametrize("m,nl,batch,bits", [(15, 4, 70, 50), (16, 3, 64, 50), (17, 4, 33, 50), (16, 16, 9, 50), (16, 3, 64, 57), (17, 4, 33, 60)])
def test_rns_one_xcd_local_launch_over_the_limbs(lib, oracle, m, nl, batch, bits, monkeypatch):
    """N = 2^15..2^17: the XCD-local launches take the limb as part of the queue entry, so a whole RNS set -- forward
    transforms, the product (both forms: both operands in coefficients; one operand transformed beforehand) -- is ONE launch
    over all limbs' polynomials.  Word for word the per-limb launches (NTT_OPT_RNS_LAUNCH 1), samples against the oracle; ragged
    per-limb batches (the queues run over limb * batch + polynomial).  57- / 60-bit limbs: the wide integer policy's transforms
    take the same launch (its products are forward transforms + the products inside the inverse)."""
    n = 1 << m
    qs = [lib.find_prime(bits, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    for p in plans:
        p.set_option(lib.OPT_XCD_LOCAL, 1)          # (the automatic choice needs 512 polynomials for a plain transform)
    slab = batch * n
    a = np.concatenate([oracle.fill_uniform(slab, q, 6100 + l) for l, q in enumerate(qs)])
    b = np.concatenate([oracle.fill_uniform(slab, q, 6200 + l) for l, q in enumerate(qs)])
    da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
    res = {}
    for loop in ("1", "0"):
        lib.set_rns_launch(plans, loop)
        da.upload(a)
        lib.rns_fwd(plans, da.ptr, batch)
        res["fwd", loop] = da.download()
        da.upload(a), db.upload(b)
        lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
        res["mul", loop] = dc.download()
        da.upload(a), db.upload(res["fwd", loop])
        lib.rns_mul_transformed(plans, dc.ptr, da.ptr, db.ptr, batch)        # c = inv(fwd(a) (.) b^), b^ = fwd(a): the square
        res["mult", loop] = dc.download()
    for kind in ("fwd", "mul", "mult"):
        assert np.array_equal(res[kind, "0"], res[kind, "1"]), kind
    for l in (0, nl - 1):
        cx = oracle.ctx(n, qs[l], ws[l])
        for p_ in (0, batch - 1):
            sl = slice(l * slab + p_ * n, l * slab + (p_ + 1) * n)
            fa = cx.fwd(a[sl])
            assert np.array_equal(res["fwd", "0"][sl], fa), (l, p_)
            assert np.array_equal(res["mul", "0"][sl], cx.inv(oracle.pointwise(fa, cx.fwd(b[sl]), qs[l]))), (l, p_)
            assert np.array_equal(res["mult", "0"][sl], cx.inv(oracle.pointwise(fa, fa, qs[l]))), (l, p_)
    for x in (da, db, dc):
        x.free()
    for p in plans:
        p.destroy()


def test_rns_products_across_devices_from_one_call(lib, oracle):
    """ntt_rns_mul_multi: BASELINE config 5's shape -- RNS products sharded over the visible devices (1 on the test box, so
    two shards are folded onto it: each shard has its own plans and therefore its own stream) from ONE C call"""
    n, nl = 1 << 15, 3
    ndev = lib.device_count()
    shards = max(2, ndev)
    qs = [lib.find_prime(50, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plan_sets = [[lib.Plan(n, q, w, device=g % ndev) for q, w in zip(qs, ws)] for g in range(shards)]
    batches = [5 + 60 * g for g in range(shards)]          # a small shard (block launches) and one of one-launch size
    hosts_a, hosts_b, da, db, dc = [], [], [], [], []
    for g, bt in enumerate(batches):
        a = np.concatenate([oracle.fill_uniform(bt * n, q, 6300 + 10 * g + l) for l, q in enumerate(qs)])
        b = np.concatenate([oracle.fill_uniform(bt * n, q, 6400 + 10 * g + l) for l, q in enumerate(qs)])
        hosts_a.append(a), hosts_b.append(b)
        da.append(lib.DeviceBuffer(a.size, device=g % ndev).upload(a))
        db.append(lib.DeviceBuffer(b.size, device=g % ndev).upload(b))
        dc.append(lib.DeviceBuffer(a.size, device=g % ndev))
    lib.rns_mul_multi(plan_sets, [x.ptr for x in dc], [x.ptr for x in da], [x.ptr for x in db], batches)
    for g, bt in enumerate(batches):
        got = dc[g].download()
        for l in (0, nl - 1):
            cx = oracle.ctx(n, qs[l], ws[l])
            for p_ in (0, bt - 1):
                sl = slice(l * bt * n + p_ * n, l * bt * n + (p_ + 1) * n)
                exp = cx.inv(oracle.pointwise(cx.fwd(hosts_a[g][sl]), cx.fwd(hosts_b[g][sl]), qs[l]))
                assert np.array_equal(got[sl], exp), (g, l, p_)
    for x in da + db + dc:
        x.free()
    for ps in plan_sets:
        for p in ps:
            p.destroy()


def test_captured_xcd_local_launch_survives_regrowth_and_concurrent_direct_calls():
    """a captured XCD-local launch bakes its queue/counter block into the graph.  The plan now keeps a
    block of its own for captured launches: (i) a later direct call with a LARGER batch on the capture stream regrows the
    direct block only -- the graph, replayed afterwards, still runs on live memory and equals the oracle; (ii) the graph
    replayed on ANOTHER stream while direct calls of the same plan run on the capture stream: both results right (separate
    counters).  The NTT-domain products (no workspace at all) are captured beside the transform."""
    import sys
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch, big, q = 1 << 15, 512, 1536, 0x7fffffffe0001
w = lib.min_root(q, n)
cx = orc.ctx(n, q, w)
plan = lib.Plan(n, q, w, device=0)
ta = torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
sa, tk, tc = torch.zeros_like(ta), torch.zeros(n, dtype=torch.int64, device="cuda:0"), torch.zeros_like(ta)
tbig = torch.zeros(big * n, dtype=torch.int64, device="cuda:0")
g, s, s2 = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.fwd(ta.data_ptr(), batch, stream=s.cuda_stream)          # warm: the (plan, stream) pair gets its two blocks
s.synchronize()
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    ta.copy_(sa)
    plan.fwd(ta.data_ptr(), batch, stream=st)                     # one XCD-local launch, on the graph block
    plan.inv_product(tc.data_ptr(), ta.data_ptr(), tk.data_ptr(), batch, lib.MUL_B_BROADCAST, stream=st)   # c = inv(a^ . key^)
key = orc.fill_uniform(n, q, 77)
tk.copy_(torch.from_numpy(cx.fwd(key).view(np.int64)))
def check(seed):
    a = orc.fill_uniform(batch * n, q, seed)
    sa.copy_(torch.from_numpy(a.view(np.int64)))
    torch.cuda.synchronize()
    return a
def verify(a, tag):
    f, c = ta.cpu().numpy().view(np.uint64), tc.cpu().numpy().view(np.uint64)
    for j in (0, 255, 256, 511):
        sl = slice(j * n, (j + 1) * n)
        fa = cx.fwd(a[sl])
        assert np.array_equal(f[sl], fa), (tag, "fwd", j)
        assert np.array_equal(c[sl], cx.inv(orc.pointwise(fa, cx.fwd(key), q))), (tag, "product", j)
a = check(1); g.replay(); torch.cuda.synchronize(); verify(a, "first replay")
# (i) a direct call with a larger batch on the capture stream: the direct block is outgrown and replaced
bigv = orc.fill_uniform(big * n, q, 5)
tbig.copy_(torch.from_numpy(bigv.view(np.int64))); torch.cuda.synchronize()
with torch.cuda.stream(s):
    plan.fwd(tbig.data_ptr(), big, stream=s.cuda_stream)
s.synchronize()
got = tbig.cpu().numpy().view(np.uint64)
for j in (0, 1535):
    assert np.array_equal(got[j * n:(j + 1) * n], cx.fwd(bigv[j * n:(j + 1) * n])), ("direct big", j)
a = check(2); g.replay(); torch.cuda.synchronize(); verify(a, "replay after regrowth")
# (ii) the graph on another stream while direct calls of the same plan run on the capture stream
a = check(3)
tbig.copy_(torch.from_numpy(bigv.view(np.int64))); torch.cuda.synchronize()
with torch.cuda.stream(s2):
    g.replay()
with torch.cuda.stream(s):
    plan.fwd(tbig.data_ptr(), big, stream=s.cuda_stream)
torch.cuda.synchronize()
verify(a, "concurrent replay")
got = tbig.cpu().numpy().view(np.uint64)
for j in (0, 700, 1535):
    assert np.array_equal(got[j * n:(j + 1) * n], cx.fwd(bigv[j * n:(j + 1) * n])), ("concurrent direct", j)
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


@pytest.mark.parametrize("cls", sorted(_DOT_MODULI))
@pytest.mark.parametrize("m", [8, 11, 12, 13, 14, 15, 17])
def test_forward_transform_times_a_transformed_operand(lib, oracle, m, cls):
    """c^ = fwd(a) (.) b^ and c^ += fwd(a) (.) b^ (the result stays in the NTT domain): ONE launch up to 2^14 (the product where the
    forward transform would reduce and store its outputs), riding in the block pass above; against the oracle's forward
    transform and 128-bit products; canonical and lazy b^, per-polynomial and broadcast b^, accumulating or not, c aliasing a or
    b^, ragged batches either side of the persistent grid, chunk by chunk above 2^14; every modulus class"""
    n = 1 << m
    bits, skip = _DOT_MODULI[cls]
    q = lib.find_prime(bits, n, skip)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    rng = np.random.default_rng(m)
    for batch in ((1, 5, 300) if m <= 14 else (1, 6)):
        a = oracle.fill_uniform(batch * n, q, 7000 + m)
        a[:3] = [q - 1, 0, q // 2]
        sample = list(range(batch)) if batch <= 8 else [0, 1, 127, 255, 256, 257, 298, 299]
        pick = np.concatenate([np.arange(p_ * n, (p_ + 1) * n) for p_ in sample])
        fa = cx.fwd(a[pick])
        da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
        for lazy, bcast, acc in ((False, False, False), (True, False, True), (False, True, True), (True, True, False), (False, False, True)):
            b = oracle.fill_uniform((1 if bcast else batch) * n, q, 7100 + m)
            b[:4] = [q - 1, 0, q // 2, 1]
            bw = b + (rng.integers(0, 4, b.size).astype(np.uint64) * np.uint64(q) if lazy else np.uint64(0))
            c0 = oracle.fill_uniform(batch * n, q, 7200 + m)
            exp = oracle.pointwise(fa, np.tile(b, len(sample)) if bcast else b[pick], q)
            if acc:
                exp = (exp + c0[pick]) % np.uint64(q)
            flags = (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0) | (lib.MUL_ACCUMULATE if acc else 0)
            for chunk in ((256, 1) if m > 14 else (256,)):
                plan.set_option(lib.OPT_CHUNK_MIB, chunk)
                da.upload(a), db.upload(bw), dc.upload(c0)
                plan.fwd_mul(dc.ptr, da.ptr, db.ptr, batch, flags)
                assert np.array_equal(dc.download()[pick], exp), (batch, lazy, bcast, acc, chunk)
                if m <= 14:
                    assert np.array_equal(da.download(), a)                         # a is only read up to 2^14
            assert np.array_equal(db.download(bw.size), bw)                          # b^ is only read
            if not acc:
                da.upload(a)
                plan.fwd_mul(da.ptr, da.ptr, db.ptr, batch, flags)                  # c aliases a
                assert np.array_equal(da.download()[pick], exp), (batch, lazy, bcast, "alias a")
            if not bcast:
                da.upload(a), db.upload(bw)
                plan.fwd_mul(db.ptr, da.ptr, db.ptr, batch, flags & ~lib.MUL_ACCUMULATE)   # c aliases b^
                want = oracle.pointwise(fa, b[pick], q)
                assert np.array_equal(db.download()[pick], want), (batch, lazy, "alias b^")
        for x in (da, db, dc):
            x.free()
    plan.destroy()


def test_key_switching_inner_product_digit_by_digit(lib, oracle):
    """the shape an FHE caller issues: acc^ = sum_i fwd(digit_i) (.) key_i^ over the limbs of an RNS set with a broadcast key
    ([limb][N]), digit by digit with NTT_MUL_ACCUMULATE, then ONE inverse -- against the oracle; also on plans without the fused
    kernels (column-only engine, radix-4 formulation)"""
    n, nl, batch, k = 1 << 13, 3, 5, 3
    qs = [lib.find_prime(50, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    slab = batch * n
    digits = [np.concatenate([oracle.fill_uniform(slab, q, 7300 + 10 * i + l) for l, q in enumerate(qs)]) for i in range(k)]
    keys = [np.concatenate([oracle.fill_uniform(n, q, 7400 + 10 * i + l) for l, q in enumerate(qs)]) for i in range(k)]   # [limb][N], NTT domain
    dacc = lib.DeviceBuffer(nl * slab)
    dd, dk = lib.DeviceBuffer(nl * slab), lib.DeviceBuffer(nl * n)
    for i in range(k):
        dd.upload(digits[i]), dk.upload(keys[i])
        lib.rns_fwd_mul(plans, dacc.ptr, dd.ptr, dk.ptr, batch, lib.MUL_B_BROADCAST | (lib.MUL_ACCUMULATE if i else 0))
    got_hat = dacc.download()
    lib.rns_inv(plans, dacc.ptr, batch)
    got = dacc.download()
    for l, (q, w) in enumerate(zip(qs, ws)):
        cx = oracle.ctx(n, q, w)
        sl, kl = slice(l * slab, (l + 1) * slab), slice(l * n, (l + 1) * n)
        exp_hat = oracle.dot([cx.fwd(d[sl]) for d in digits], [kk[kl] for kk in keys], q, n, True)
        assert np.array_equal(got_hat[sl], exp_hat), l
        assert np.array_equal(got[sl], cx.inv(exp_hat)), l
    for p in plans:
        p.destroy()
    # plans the kernel is not built for
    q, w = qs[0], ws[0]
    cx = oracle.ctx(n, q, w)
    a, b, c0 = digits[0][:slab], keys[0][:n], oracle.fill_uniform(slab, q, 7500)
    exp = (oracle.pointwise(cx.fwd(a), np.tile(b, batch), q) + c0) % np.uint64(q)
    for kind in ("generic", "r4"):
        plan = lib.Plan(n, q, w, arith=lib.ARITH_U64_R4 if kind == "r4" else lib.ARITH_AUTO)
        if kind == "generic":
            plan.set_generic(True)
        dd.upload(a), dk.upload(b), dacc.upload(c0)
        plan.fwd_mul(dacc.ptr, dd.ptr, dk.ptr, batch, lib.MUL_B_BROADCAST | lib.MUL_ACCUMULATE)
        assert np.array_equal(dacc.download(slab), exp), kind
        plan.destroy()
    for x in (dacc, dd, dk):
        x.free()


@pytest.mark.parametrize("cls", ["f64_class0", "f64_52bit", "u64_57bit", "u64_60bit"])
@pytest.mark.parametrize("m", [15, 16, 17])
def test_xcd_local_ntt_domain_products(lib, oracle, m, cls):
    """N = 2^15..2^17: c = inv(sum a_i^ (.) b_i^) with BOTH passes as items of ONE launch (team_dot_kernel: row items = the products
    and the inverse block stages, column items behind a per-polynomial counter; round 5) against the per-chunk launches
    (NTT_OPT_XCD_LOCAL 0), every polynomial, and samples against the oracle: k = 1 and 3, canonical and lazy words, a broadcast
    key, c aliasing a^ and b^, ragged batches (queues of unequal length)"""
    n = 1 << m
    bits = {"f64_class0": 51, "f64_52bit": 52, "u64_57bit": 57, "u64_60bit": 60}[cls]
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    for k, batch, lazy, bcast in ((1, 67, False, False), (3, 70, True, False), (2, 65, False, True), (1, 64, True, False)):
        a_list, b_list = _ntt_domain_operands(oracle, n, q, batch, k, 5600 + 5 * m + k, lazy, bcast)
        flags = (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0)
        da = [lib.DeviceBuffer(a.size).upload(a) for a in a_list]
        db = [lib.DeviceBuffer(b.size).upload(b) for b in b_list]
        dc = lib.DeviceBuffer(batch * n)
        res = {}
        for form in (0, 1):
            plan.set_option(lib.OPT_XCD_LOCAL, form)
            dc.upload(np.zeros(batch * n, dtype=np.uint64))
            plan.inv_dot(dc.ptr, [x.ptr for x in da], [x.ptr for x in db], batch, flags)
            res[form] = dc.download()
        assert np.array_equal(res[0], res[1]), (k, batch, lazy, bcast)
        for p_ in (0, batch // 2, batch - 1):
            sl = slice(p_ * n, (p_ + 1) * n)
            exp = cx.inv(oracle.dot([a[sl] for a in a_list], [b if bcast else b[sl] for b in b_list], q, n, bcast))
            assert np.array_equal(res[1][sl], exp), (k, batch, lazy, bcast, p_)
        if k == 1:
            plan.set_option(lib.OPT_XCD_LOCAL, 1)
            plan.inv_product(da[0].ptr, da[0].ptr, db[0].ptr, batch, flags)          # c aliases a^
            assert np.array_equal(da[0].download(), res[0]), ("alias a", batch, lazy)
            da[0].upload(a_list[0])
            plan.inv_product(db[0].ptr, da[0].ptr, db[0].ptr, batch, flags)          # c aliases b^
            assert np.array_equal(db[0].download(), res[0]), ("alias b", batch, lazy)
        for x in da + db + [dc]:
            x.free()
    plan.destroy()


@pytest.mark.parametrize("cls", ["f64_class0", "f64_class1", "f64_52bit", "u64_57bit", "u64_60bit"])
@pytest.mark.parametrize("m", [15, 16, 17])
def test_xcd_local_forward_transform_times_a_transformed_operand(lib, oracle, m, cls):
    """N = 2^15..2^17: c^ = fwd(a) (.) b^ (+ c^) with the forward column stages of a and the blocks with the product as items of ONE
    launch (team_mul_kernel, round 5) against the two launches per chunk (NTT_OPT_XCD_LOCAL 0), every word, and samples against
    the oracle: canonical / lazy b^, a broadcast key, the accumulator, c^ aliasing a and b^, ragged batches (queues of unequal
    length, much shorter than the lag)"""
    n = 1 << m
    bits = {"f64_class0": 51, "f64_class1": 50, "f64_52bit": 52, "u64_57bit": 57, "u64_60bit": 60}[cls]
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    rng = np.random.default_rng(100 + m)
    for batch, lazy, bcast, acc in ((67, False, False, False), (70, True, False, True), (65, False, True, True), (64, True, True, False)):
        a = oracle.fill_uniform(batch * n, q, 7600 + m)
        a[:3] = [q - 1, 0, q // 2]
        b = oracle.fill_uniform((1 if bcast else batch) * n, q, 7700 + m)
        b[:4] = [q - 1, 0, q // 2, 1]
        lz = 4 if (4 * q < (1 << 64)) else 1
        bw = b + (rng.integers(0, lz, b.size).astype(np.uint64) * np.uint64(q) if lazy else np.uint64(0))
        c0 = oracle.fill_uniform(batch * n, q, 7800 + m)
        flags = (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0) | (lib.MUL_ACCUMULATE if acc else 0)
        da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(bw.size), lib.DeviceBuffer(a.size)
        res = {}
        for form in (0, 1):
            plan.set_option(lib.OPT_XCD_LOCAL, form)
            da.upload(a), db.upload(bw), dc.upload(c0)
            plan.fwd_mul(dc.ptr, da.ptr, db.ptr, batch, flags)
            res[form] = dc.download()
            assert np.array_equal(db.download(), bw)                                   # b^ is only read
        assert np.array_equal(res[0], res[1]), (batch, lazy, bcast, acc)
        for p_ in (0, batch // 2, batch - 1):
            sl = slice(p_ * n, (p_ + 1) * n)
            exp = oracle.pointwise(cx.fwd(a[sl].copy()), b if bcast else b[sl], q)
            if acc:
                exp = (exp + c0[sl]) % np.uint64(q)
            assert np.array_equal(res[1][sl], exp), (batch, lazy, bcast, acc, p_)
        plan.set_option(lib.OPT_XCD_LOCAL, 1)
        if not acc:
            da.upload(a)
            plan.fwd_mul(da.ptr, da.ptr, db.ptr, batch, flags)                          # c^ aliases a
            assert np.array_equal(da.download(), res[0]), ("alias a", batch, lazy, bcast)
        if not bcast:
            da.upload(a), db.upload(bw)
            plan.fwd_mul(db.ptr, da.ptr, db.ptr, batch, flags & ~lib.MUL_ACCUMULATE)     # c^ aliases b^
            plan.set_option(lib.OPT_XCD_LOCAL, 0)
            da.upload(a), dc.upload(c0)
            db2 = lib.DeviceBuffer(bw.size).upload(bw)
            plan.fwd_mul(dc.ptr, da.ptr, db2.ptr, batch, flags & ~lib.MUL_ACCUMULATE)
            assert np.array_equal(db.download(), dc.download()), ("alias b^", batch, lazy)
            db2.free()
        for x in (da, db, dc):
            x.free()
    plan.destroy()


@pytest.mark.parametrize("m,nl,batch,bits,layout", [(15, 3, 30, 50, None), (16, 4, 20, 50, "bm"), (17, 2, 40, 57, "bm"), (16, 16, 5, 52, None)])
def test_xcd_local_fwd_mul_over_rns_limbs(lib, oracle, m, nl, batch, bits, layout):
    """the same over the limbs of an RNS set (the limb in the queue entry), [limb][batch][N] and [batch][limb][N], a broadcast key
    ([limb][N]) accumulated onto c^: against the per-limb per-chunk form, every word; samples against the oracle"""
    n = 1 << m
    qs = [lib.find_prime(bits, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    rng = np.random.default_rng(200 + m)
    a = rng.integers(0, min(qs), size=(nl, batch, n), dtype=np.uint64)
    key = rng.integers(0, min(qs), size=(nl, n), dtype=np.uint64)
    c0 = rng.integers(0, min(qs), size=(nl, batch, n), dtype=np.uint64)
    lay = (n, nl * n) if layout == "bm" else None

    def place(x):
        return np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(-1) if layout == "bm" else x.reshape(-1)

    da, dk, dc = lib.DeviceBuffer(nl * batch * n), lib.DeviceBuffer(nl * n).upload(key.reshape(-1)), lib.DeviceBuffer(nl * batch * n)
    res = {}
    for form in (0, 1):
        for p in plans:
            p.set_option(lib.OPT_XCD_LOCAL, form)
        lib.set_rns_launch(plans, 1 - form)
        da.upload(place(a)), dc.upload(place(c0))
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, dk.ptr, batch, lib.MUL_B_BROADCAST | lib.MUL_ACCUMULATE, layout=lay)
        res[form] = dc.download()
    assert np.array_equal(res[0], res[1])
    got = res[1].reshape(batch, nl, n).transpose(1, 0, 2) if layout == "bm" else res[1].reshape(nl, batch, n)
    for l in (0, nl - 1):
        cx = oracle.ctx(n, qs[l], ws[l])
        for p_ in (0, batch - 1):
            exp = (oracle.pointwise(cx.fwd(a[l, p_].copy()), key[l].copy(), qs[l]) + c0[l, p_]) % np.uint64(qs[l])
            assert np.array_equal(got[l, p_], exp), (l, p_)
    # a per-polynomial b^ laid out like a (its limb stride is the layout's), lazy words, no accumulator
    bh = rng.integers(0, min(qs), size=(nl, batch, n), dtype=np.uint64)
    db = lib.DeviceBuffer(nl * batch * n)
    res = {}
    for form in (0, 1):
        for p in plans:
            p.set_option(lib.OPT_XCD_LOCAL, form)
        lib.set_rns_launch(plans, 1 - form)
        da.upload(place(a)), db.upload(place(bh))
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch, lib.MUL_LAZY_IN, layout=lay)
        res[form] = dc.download()
    assert np.array_equal(res[0], res[1])
    got = res[1].reshape(batch, nl, n).transpose(1, 0, 2) if layout == "bm" else res[1].reshape(nl, batch, n)
    l, p_ = nl - 1, batch // 2
    cx = oracle.ctx(n, qs[l], ws[l])
    assert np.array_equal(got[l, p_], oracle.pointwise(cx.fwd(a[l, p_].copy()), bh[l, p_].copy(), qs[l]))
    db.free()
    for x in (da, dk, dc):
        x.free()
    for p in plans:
        p.destroy()


@pytest.mark.parametrize("m,nl,batch,bits,layout", [(15, 3, 30, 50, None), (16, 4, 20, 50, "bm"), (17, 2, 40, 57, "bm"), (16, 16, 5, 52, None)])
def test_xcd_local_ntt_domain_products_over_rns_limbs(lib, oracle, m, nl, batch, bits, layout):
    """the same over the limbs of an RNS set (the limb in the queue entry: MULTI variant), in [limb][batch][N] and in
    [batch][limb][N] layout, against the per-limb per-chunk form, every word; samples against the oracle"""
    n = 1 << m
    qs = [lib.find_prime(bits, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    k = 2
    rng = np.random.default_rng(m)
    # operands [k][limb][batch][n] dense, values below the smallest modulus
    ops_a = rng.integers(0, min(qs), size=(k, nl, batch, n), dtype=np.uint64)
    ops_b = rng.integers(0, min(qs), size=(k, nl, batch, n), dtype=np.uint64)
    lay = (n, nl * n) if layout == "bm" else None

    def place(x):       # [limb][batch][n] -> the layout's image
        return np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(-1) if layout == "bm" else x.reshape(-1)

    da = [lib.DeviceBuffer(nl * batch * n).upload(place(ops_a[i])) for i in range(k)]
    db = [lib.DeviceBuffer(nl * batch * n).upload(place(ops_b[i])) for i in range(k)]
    dc = lib.DeviceBuffer(nl * batch * n)
    res = {}
    for form in (0, 1):
        for p in plans:
            p.set_option(lib.OPT_XCD_LOCAL, form)
        lib.set_rns_launch(plans, 1 - form)            # per limb and per chunk  /  one launch over the limbs
        lib.rns_inv_dot(plans, dc.ptr, [x.ptr for x in da], [x.ptr for x in db], batch, layout=lay)
        res[form] = dc.download()
    assert np.array_equal(res[0], res[1])
    got = res[1].reshape(batch, nl, n).transpose(1, 0, 2) if layout == "bm" else res[1].reshape(nl, batch, n)
    for l in (0, nl - 1):
        cx = oracle.ctx(n, qs[l], ws[l])
        for p_ in (0, batch - 1):
            exp = cx.inv(oracle.dot([ops_a[i, l, p_].copy() for i in range(k)], [ops_b[i, l, p_].copy() for i in range(k)], qs[l], n))
            assert np.array_equal(got[l, p_], exp), (l, p_)
    for x in da + db + [dc]:
        x.free()
    for p in plans:
        p.destroy()


def test_graph_control_block_follows_a_larger_reserve(lib, oracle):
    """advisor r05 (medium): the control block that CAPTURED XCD-local launches use was sized once, by the first direct call or
    reserve on a (plan, stream) pair; a later, larger ntt_plan_reserve grew only the direct block, and a capture of the larger batch
    silently took the per-pass launches.  Now: reserve(small), reserve(big), capture -- the graph holds the TWO kernel nodes of the
    one-launch form (control-block clear + team_kernel), not the 76 of the per-pass launches over 38 chunks of 8 MiB; the counter of
    control-block allocations moves with the second reserve and not with the capture; replays are bit-exact."""
    hip = C.CDLL("libamdhip64.so.7")
    n, batch = 1 << 16, 600        # (the first reserve's block -- doubled on first allocation -- holds 545 entries: 600 outgrows it)
    q = lib.find_prime(50, n, 0)
    w = lib.min_root(q, n)
    plan, cx = lib.Plan(n, q, w), oracle.ctx(n, q, w)
    plan.set_option(lib.OPT_XCD_LOCAL, 1)
    plan.set_option(lib.OPT_CHUNK_MIB, 8)
    st = C.c_void_p()
    lib._check(lib._lib.ntt_stream_create(0, C.byref(st)))
    plan.reserve(8, stream=st.value)
    a0 = plan.get_option(lib.OPT_CTL_ALLOCATIONS)
    plan.reserve(batch, stream=st.value)
    a1 = plan.get_option(lib.OPT_CTL_ALLOCATIONS)
    assert a1 > a0
    a = oracle.fill_uniform(batch * n, q, 4)
    d = lib.DeviceBuffer(a.size).upload(a)
    graph, exe, count = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    assert hip.hipStreamBeginCapture(st, 2) == 0          # hipStreamCaptureModeRelaxed
    plan.fwd(d.ptr, batch, stream=st.value)
    assert hip.hipStreamEndCapture(st, C.byref(graph)) == 0
    assert plan.get_option(lib.OPT_CTL_ALLOCATIONS) == a1          # nothing allocated while capturing
    assert hip.hipGraphGetNodes(graph, None, C.byref(count)) == 0
    assert count.value == 2, "captured %d nodes: the per-pass launches, not the one-launch form" % count.value
    assert hip.hipGraphInstantiate(C.byref(exe), graph, None, None, C.c_size_t(0)) == 0
    for rep in range(3):
        d.upload(a)
        assert hip.hipGraphLaunch(exe, st) == 0
        lib.stream_sync(0, st.value)
        got = d.download()
        for p in (0, 341, batch - 1):
            assert np.array_equal(got[p * n:(p + 1) * n], cx.fwd(a[p * n:(p + 1) * n].copy())), (rep, p)
    hip.hipGraphExecDestroy(exe), hip.hipGraphDestroy(graph)
    d.free(), plan.destroy()
    lib._lib.ntt_stream_destroy(0, st)
