"""CPU: the HIP kernel templates (csrc/ntt_core.h, ntt_arith.h, ntt_passplan.h),
compiled for the host and run thread-by-thread, are bit-exact against the
oracle: index maps, LDS exchange layouts, twiddle addressing, the FP64 exactness
argument and its reduction schedule, and the multi-pass split for N > 2^14."""
import numpy as np
import pytest

from emu_binding import Emu

U64, F64 = 0, 1


@pytest.fixture(scope="module")
def emu():
    return Emu()


def _inputs(oracle, n, q, batch, seed):
    a = oracle.fill_uniform(batch * n, q, seed)
    a[:4] = q - 1          # extreme residues
    a[4:8] = 0
    a[n - 1] = q - 1
    return a


@pytest.mark.parametrize("logn", range(6, 15))
def test_plan_layouts(emu, logn):
    info = emu.plan_info(logn)
    assert info["conflict_free"] == 1, "ds_write_b64 bank conflicts in an LDS exchange"
    assert info["T"] == 1 << (logn - 4) and info["ROW"] == info["T"] + 1
    assert info["LDS_ELEMS"] * 8 * (256 // min(256, info["T"]) if info["T"] < 256 else 1) <= 160 * 1024
    if logn == 14:
        # q ~ 2^51: forward reduces the pass-through operand in 5 of 14 stages
        assert bin(info["fmask"]).count("1") == 5


@pytest.mark.parametrize("i", range(19))
@pytest.mark.parametrize("arith", [U64, F64])
def test_all_reference_cases(kat, oracle, emu, i, arith):
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    a = _inputs(oracle, n, q, 2, 1000 + i)
    expect = cx.fwd(a)
    rc, got = emu.transform(a, m, q, w, arith)
    assert rc == 0 and np.array_equal(got, expect)
    rc, back = emu.transform(expect, m, q, w, arith, inverse=True)
    assert rc == 0 and np.array_equal(back, a)


@pytest.mark.parametrize("i", [0, 4, 9, 12, 13, 15])
def test_generic_multipass_and_forced_class(kat, oracle, emu, i):
    """column-pass-only path (the library's self-check mode) and the most
    conservative FP64 schedule on small moduli"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    a = _inputs(oracle, n, q, 1, 77 + i)
    expect = cx.fwd(a)
    for arith in (U64, F64):
        rc, got = emu.transform(a, m, q, w, arith, generic=True)
        assert rc == 0 and np.array_equal(got, expect)
        rc, back = emu.transform(expect, m, q, w, arith, inverse=True, generic=True)
        assert rc == 0 and np.array_equal(back, a)
    rc, got = emu.transform(a, m, q, w, F64, ksh=0)
    assert rc == 0 and np.array_equal(got, expect)


@pytest.mark.parametrize("i", [2, 6, 12, 13])
def test_wide_inputs(kat, oracle, emu, i):
    """lazy inputs in [0,8q) (what the reference's *_lazy outputs look like)"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    a = _inputs(oracle, n, q, 1, 5)
    k = (oracle.fill_uniform(n, 8, 9) % np.uint64(8)).astype(np.uint64)
    lazy = a + k * np.uint64(q)
    for arith in (U64, F64):
        rc, got = emu.transform(lazy, m, q, w, arith, wide=True)
        assert rc == 0 and np.array_equal(got, cx.fwd(a))
        rc, got = emu.transform(lazy, m, q, w, arith, inverse=True, wide=True)
        assert rc == 0 and np.array_equal(got, cx.inv(a))


@pytest.mark.parametrize("bits,m", [(50, 14), (50, 12), (49, 13), (40, 11), (31, 10), (51, 14), (45, 6), (33, 7)])
def test_generated_primes(oracle, emu, bits, m):
    """moduli the reference table does not hold (configs 2/3/5 need them)"""
    n = 1 << m
    q = oracle.find_prime(bits, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    a = _inputs(oracle, n, q, 1, bits * 100 + m)
    expect = cx.fwd(a)
    for arith in (U64, F64):
        rc, got = emu.transform(a, m, q, w, arith)
        assert rc == 0 and np.array_equal(got, expect), (bits, m, arith)
        rc, back = emu.transform(expect, m, q, w, arith, inverse=True)
        assert rc == 0 and np.array_equal(back, a)


@pytest.mark.parametrize("i", [0, 6, 12, 13, 15, 17])
def test_f64_exactness_claims_hold(kat, oracle, emu, i):
    """DESIGN.md section 4, executed: every FP64 value is an integer below 2^53, every
    product equals y*w mod q exactly (checked with 128-bit integers), the bound
    B*q < 2^53 holds with margin, on random and adversarial inputs"""
    CHK = 2
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    cx = oracle.ctx(n, q, w)
    rnd = _inputs(oracle, n, q, 1, 900 + i)
    sign = np.where(np.arange(n) % 2 == 0, q - 1, 1).astype(np.uint64)          # +-1 pattern
    half = np.full(n, q // 2, dtype=np.uint64)                                      # largest balanced magnitude
    allmax = np.full(n, q - 1, dtype=np.uint64)
    emu.chk_stats(reset=True)
    for a in (rnd, sign, half, allmax):
        rc, got = emu.transform(a, m, q, w, CHK)
        assert rc == 0 and np.array_equal(got, cx.fwd(a))
        rc, back = emu.transform(got, m, q, w, CHK, inverse=True)
        assert rc == 0 and np.array_equal(back, a)
        # the inverse's per-slot reduction plan on inputs it does not get from a forward transform
        rc, inv = emu.transform(a, m, q, w, CHK, inverse=True)
        assert rc == 0 and np.array_equal(inv, cx.inv(a))
    fails, maxb, maxr = emu.chk_stats()
    assert fails == 0
    lim = 2.0 ** 53 / q
    assert maxb < lim * (1 - 1 / 64), (maxb, lim)
    assert maxr < 0.5 + maxb * q / 2.0 ** 53 * 1.01 + 0.01


def test_f64_refuses_large_modulus(oracle, emu):
    q = oracle.find_prime(55, 1 << 10)
    w = oracle.min_root(q, 1 << 10)
    a = oracle.fill_uniform(1 << 10, q, 3)
    rc, _ = emu.transform(a, 10, q, w, F64)
    assert rc == -2
    rc, got = emu.transform(a, 10, q, w, U64)
    assert rc == 0 and np.array_equal(got, oracle.ctx(1 << 10, q, w).fwd(a))


@pytest.mark.parametrize("q", [0x1e01, 0x10001, 0x7ffe0001, 0x7fffffffe0001, 0x80000001c0001])
def test_pointwise(oracle, emu, q):
    n = 4096
    a, b = oracle.fill_uniform(n, q, 11), oracle.fill_uniform(n, q, 12)
    a[:3] = q - 1
    b[:3] = q - 1
    a[3], b[3] = 0, q - 1
    expect = oracle.pointwise(a, b, q)
    for arith in (U64, F64):
        rc, got = emu.pointwise(a, b, q, arith)
        assert rc == 0 and np.array_equal(got, expect)
