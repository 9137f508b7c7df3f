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


# ---------------------------------------------------------------------------------------------
# radix-4 integer policy (reference fast_mul_operators.h:62-70,108-149, src/ntt_radix4.c:7-114)
# and lazy outputs (include/ntt_reference.h:13-17)
# ---------------------------------------------------------------------------------------------
R4_CASES = [i for i in range(19)]


@pytest.mark.parametrize("i", R4_CASES)
def test_radix4_policy_matches_reference_lazy_values(kat, oracle, emu, i):
    """the radix-4 kernels apply the reference's butterflies to the same operands in the same order: even the
    LAZY outputs equal fwd_ntt_radix4_lazy bit for bit; the inverse (fully reduced) equals inv_ntt_radix4"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    a = oracle.fill_uniform(3 * n, q, 700 + i)
    cx = oracle.ctx(n, q, w)
    # (N = 2^15..2^17, reference cases 14-18: a column pass of one or two radix-4 levels before / after blocks that end / begin
    # with the reference's radix-2 stage when m is odd -- the same checks as for the single-pass sizes)
    if m < 6:
        assert emu.transform(a, m, q, w, 3)[0] == -4          # below the block range: the library refuses too
        return
    rc, lazy = emu.transform(a, m, q, w, 3, lazy=True)
    assert rc == 0
    assert np.array_equal(lazy, cx.fwd_r4_lazy(a))
    assert int(lazy.max()) < (8 if m % 2 == 0 else 4) * q
    rc, red = emu.transform(a, m, q, w, 3)
    assert rc == 0 and np.array_equal(red, cx.fwd(a))
    # lazy values fed straight back in (tests/bench.c:123-137): same lazy outputs as the reference again
    rc, lazy2 = emu.transform(lazy, m, q, w, 3, lazy=True, wide=True)
    assert rc == 0 and np.array_equal(lazy2, cx.fwd_r4_lazy(lazy))
    rc, back = emu.transform(red, m, q, w, 3, inverse=True)
    assert rc == 0 and np.array_equal(back, a)
    rc, back = emu.transform(lazy, m, q, w, 3, inverse=True, wide=True)
    assert rc == 0 and np.array_equal(back, a)
    rc, lz = emu.transform(red, m, q, w, 3, inverse=True, lazy=True)
    assert rc == 0 and int(lz.max()) < 2 * q and np.array_equal(lz % np.uint64(q), a)


def test_radix4_policy_large_moduli(oracle, emu):
    """59- and 52-bit primes: the 128-bit double product really needs its carry"""
    for bits, m in ((59, 8), (59, 13), (52, 14), (52, 11), (59, 15), (59, 16), (52, 17), (59, 18)):
        n = 1 << m
        q = oracle.find_prime(bits, n)
        w = oracle.min_root(q, n)
        a = oracle.fill_uniform(2 * n, q, bits * 100 + m)
        cx = oracle.ctx(n, q, w)
        rc, lazy = emu.transform(a, m, q, w, 3, lazy=True)
        assert rc == 0 and np.array_equal(lazy, cx.fwd_r4_lazy(a)), (bits, m)
        rc, back = emu.transform(lazy, m, q, w, 3, inverse=True, wide=True)
        assert rc == 0 and np.array_equal(back, a), (bits, m)


def test_expanded_table_builder(kat, oracle, emu):
    for i in (0, 4, 9, 12):
        c = kat["cases"][i]
        cx = oracle.ctx(1 << c["m"], c["q"], c["w"])
        assert np.array_equal(emu.expand_radix4(cx.table("w"), c["q"]), cx.table("e"))
        assert np.array_equal(emu.expand_radix4(cx.table("winv"), c["q"]), cx.table("einv"))


@pytest.mark.parametrize("i", range(19))
def test_lazy_outputs_radix2(kat, oracle, emu, i):
    """integer policy: lazy forward outputs are the reference's fwd_ntt_ref_harvey_lazy values bit for bit
    ([0,4q)), lazy inverse outputs lie in [0,2q); FP64 policy: [0,4q) forward, congruent to the oracle"""
    c = kat["cases"][i]
    m, q, w = c["m"], c["q"], c["w"]
    n = 1 << m
    a = oracle.fill_uniform(2 * n, q, 900 + i)
    cx = oracle.ctx(n, q, w)
    expect = cx.fwd(a)
    rc, lz = emu.transform(a, m, q, w, 0, lazy=True)
    assert rc == 0 and np.array_equal(lz, cx.fwd_lazy(a)) and int(lz.max()) < 4 * q
    rc, back = emu.transform(lz, m, q, w, 0, inverse=True, wide=True, lazy=True)
    assert rc == 0 and int(back.max()) < 2 * q and np.array_equal(back % np.uint64(q), a)
    if q <= (1 << 51) + (1 << 41):
        for arith in (1, 2):
            rc, lf = emu.transform(a, m, q, w, arith, lazy=True)
            assert rc == 0 and int(lf.max()) < 4 * q and np.array_equal(lf % np.uint64(q), expect), arith
            rc, back = emu.transform(lf, m, q, w, arith, inverse=True, wide=True, lazy=True)
            assert rc == 0 and np.array_equal(back, a)
        if i == 12:
            assert emu.chk_stats()[0] == 0


def test_lazy_schedule_bounds_last_stage(emu):
    """the LAZY forward schedule reduces late enough that |v| < 2q at the end (checked policy, worst class)"""
    info = emu.plan_info(14)
    assert info["fmask"] != 0


@pytest.mark.parametrize("q", [0x7fffffffe0001, 0x80000001c0001, 0x3ffffffdf0001, 0x7ffe0001, 0x10001])
def test_pointwise_lazy_operands(oracle, emu, q):
    """a, b anywhere in [0,4q) (extremes included): the product equals the exact one of the residues"""
    rng = np.random.default_rng(q & 0xffff)
    n = 4096
    a = rng.integers(0, 4 * q, n, dtype=np.uint64)
    b = rng.integers(0, 4 * q, n, dtype=np.uint64)
    ext = np.array([0, 1, q - 1, q, q + 1, 2 * q - 1, 2 * q, 2 * q + 1, 3 * q, 4 * q - 1], dtype=np.uint64)
    a[:100] = np.repeat(ext, 10)
    b[:100] = np.tile(ext, 10)
    expect = oracle.pointwise(a % np.uint64(q), b % np.uint64(q), q)
    for arith in (0, 1):
        rc, c = emu.pointwise_lazy(a, b, q, arith)
        if arith == 1 and q > (1 << 51) + (1 << 41):
            assert rc == -2
            continue
        assert rc == 0 and np.array_equal(c, expect), arith


@pytest.mark.parametrize("q", [0x7fffffffe0001, 0x80000001c0001, 0x3ffffffdf0001, 0x7ffe0001])
def test_fused_product_kernel_logic(oracle, emu, q):
    """fused_product_kernel step by step on the CPU: forward of b, product with a^ in registers, inverse whose
    per-lane group reads the FORWARD table mirrored (w^-1[2^s+j] = -w[2^(s+1)-1-j]); checked policy: every value
    an integer below 2^53, every product exact, |product| <= 0.75 q"""
    n = 1 << 14
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    a = oracle.fill_uniform(2 * n, q, 31)
    b = oracle.fill_uniform(2 * n, q, 32)
    b[:8] = [0, 1, q - 1, q - 2, 2, 3, q // 2, q // 2 + 1]
    expect = cx.inv(oracle.pointwise(cx.fwd(a), cx.fwd(b), q))
    assert np.array_equal(expect[:n], oracle.schoolbook(a[:n].copy(), b[:n].copy(), n, q))
    emu.chk_stats()
    for lazy in (False, True):
        rc, ahat = emu.transform(a, 14, q, w, 1, lazy=lazy)
        assert rc == 0
        for chk in (True, False):
            rc, c = emu.fused_product14(ahat, b, q, w, a_lazy=lazy, chk=chk)
            assert rc == 0 and np.array_equal(c, expect), (lazy, chk)
    # the BOTH variant: a's coefficients go through the forward stages inside the kernel, the factors meet in registers
    for chk in (True, False):
        rc, c = emu.fused_product14(a, b, q, w, a_lazy=True, chk=chk, both=True)
        assert rc == 0 and np.array_equal(c, expect), ("both", chk)
    fails, maxb, maxr = emu.chk_stats()
    assert fails == 0


@pytest.mark.parametrize("m", [8, 9, 10, 11, 12, 13])
def test_fused_product_kernels_other_sizes_logic(oracle, emu, m):
    """the product kernels below 2^14 step by step on the CPU with the checked policy: fused_product_kernel at 2^12 and
    2^13 (one LDS table, last group's twiddles preloaded), fused_product_small_kernel at 2^8..2^11 (EVERY per-lane group,
    the last one included, from its forward table -- the inverse half mirrored); 51-, 50- and 52-bit moduli (scheduled
    classes 0 and 1, reduce-both-operands policy); lazy a^ as the forward transform leaves it"""
    n = 1 << m
    for q in (0x7fffffffe0001, oracle.find_prime(50, n, 0), oracle.find_prime(52, n, 1)):
        w = oracle.min_root(q, n)
        cx = oracle.ctx(n, q, w)
        a = oracle.fill_uniform(3 * n, q, 131 + m)
        b = oracle.fill_uniform(3 * n, q, 132 + m)
        b[:8] = [0, 1, q - 1, q - 2, 2, 3, q // 2, q // 2 + 1]
        expect = cx.inv(oracle.pointwise(cx.fwd(a), cx.fwd(b), q))
        if m <= 10:
            assert np.array_equal(expect[:n], oracle.schoolbook(a[:n].copy(), b[:n].copy(), n, q))
        emu.chk_stats()
        arith = 4 if q > (1 << 51) + (1 << 41) else 1
        rc, ahat = emu.transform(a, m, q, w, arith, lazy=True)
        assert rc == 0
        rc, c = emu.fused_product_chk(ahat, b, m, q, w)
        assert rc == 0 and np.array_equal(c, expect), (m, hex(q))
        rc, c = emu.fused_product_chk(a, b, m, q, w, both=True)     # both forward transforms inside the kernel
        assert rc == 0 and np.array_equal(c, expect), ("both", m, hex(q))
        fails, maxb, maxr = emu.chk_stats()
        assert fails == 0, (m, hex(q))


@pytest.mark.parametrize("m", [6, 9, 12, 13, 14, 15])
def test_wide_fp64_policy_52_bit_moduli(oracle, emu, m):
    """ArithF64W: moduli between 2^51(1+2^-10) and 2^52 (less than one bit below 2^53/2): the pass-through operand of
    every butterfly is reduced, the multiplied one where the compile-time schedule says so (forward: f64w_fwd_schedule;
    inverse: always); bit-exact against the oracle, every value an integer below 2^53 and every product exact (checked
    policy), for the largest 52-bit primes and for a 51-bit one, on random and adversarial inputs"""
    n = 1 << m
    for q in (oracle.find_prime(52, n, 0), oracle.find_prime(52, n, 3), 0x7fffffffe0001):
        if (q - 1) % (2 * n):
            continue
        assert q < (1 << 52)
        w = oracle.min_root(q, n)
        cx = oracle.ctx(n, q, w)
        a = oracle.fill_uniform(2 * n, q, 52 + m)
        a[:6] = [0, 1, q - 1, q - 2, q // 2, q // 2 + 1]
        expect = cx.fwd(a)
        emu.chk_stats()
        for arith in (4, 5):
            rc, got = emu.transform(a, m, q, w, arith)
            assert rc == 0 and np.array_equal(got, expect), (hex(q), arith)
            rc, back = emu.transform(got, m, q, w, arith, inverse=True)
            assert rc == 0 and np.array_equal(back, a), (hex(q), arith)
            rc, lz = emu.transform(a, m, q, w, arith, lazy=True)                 # no lazy form: reduced values
            assert rc == 0 and np.array_equal(lz, expect)
            rc, back = emu.transform(expect + np.uint64(3 * q), m, q, w, arith, inverse=True, wide=True)
            assert rc == 0 and np.array_equal(back, a)
        if m <= 12:
            rc, got = emu.transform(a, m, q, w, 5, generic=True)
            assert rc == 0 and np.array_equal(got, expect)
        # inputs that drive the unreduced multiplied operands to the edge of the schedule's bounds: all-(q-1), +-1 pattern,
        # the largest balanced magnitude
        for adv in (np.full(n, q - 1, dtype=np.uint64), np.where(np.arange(n) % 2 == 0, q - 1, 1).astype(np.uint64),
                    np.full(n, q // 2, dtype=np.uint64), np.where(np.arange(n) % 3 == 0, q // 2 + 1, q - 1).astype(np.uint64)):
            rc, got = emu.transform(adv, m, q, w, 5)
            assert rc == 0 and np.array_equal(got, cx.fwd(adv)), hex(q)
            rc, got = emu.transform(adv, m, q, w, 5, inverse=True)
            assert rc == 0 and np.array_equal(got, cx.inv(adv)), hex(q)
        fails, maxb, maxr = emu.chk_stats()
        # (products of unreduced operands: |m| <= (1/2 + 1.5 B theta2) q with B < 2, theta2 < 1/2)
        assert fails == 0 and maxb < 2.0 and maxr < 1.99, (hex(q), fails, maxb, maxr)
    assert emu.transform(a, m, (1 << 52) + 1, 3, 4)[0] == -2


def _prime_near(oracle, bound, n, below=True):
    """an NTT-friendly prime (q = 1 mod 2n) closest to `bound` from below / above"""
    step = 2 * n
    q = (bound // step) * step + 1
    if below and q > bound:
        q -= step
    if not below and q <= bound:
        q += step
    while not oracle.lib.orc_is_prime(q):
        q += -step if below else step
    return q


@pytest.mark.parametrize("m", [12, 14])
def test_fp64_class_boundary_moduli(oracle, emu, m):
    """moduli sitting right at the edges of the FP64 headroom classes (where the compile-time bounds are tightest):
    just below and above 2^51(1+2^-10) (scheduled policy <-> reduce-both policy), 2^50(1+2^-10) (class 0 <-> 1),
    2^33(1+2^-10) (class 1 <-> 18), just below 2^52; adversarial inputs (all q-1, alternating 0 / q-1); checked
    policy: no value ever leaves the exactly representable range"""
    n = 1 << m
    edges = [((1 << 51) + (1 << 41), True), ((1 << 51) + (1 << 41), False), ((1 << 50) + (1 << 40), True),
             ((1 << 50) + (1 << 40), False), ((1 << 33) + (1 << 23), True), ((1 << 33) + (1 << 23), False), ((1 << 52) - 1, True)]
    for bound, below in edges:
        q = _prime_near(oracle, bound, n, below)
        w = oracle.min_root(q, n)
        cx = oracle.ctx(n, q, w)
        wide = q > (1 << 51) + (1 << 41)
        a = oracle.fill_uniform(3 * n, q, bound & 0xffff)
        a[n:2 * n] = q - 1
        a[2 * n:3 * n:2] = 0
        a[2 * n + 1:3 * n:2] = q - 1
        expect = cx.fwd(a)
        emu.chk_stats()
        for arith in ((4, 5) if wide else (1, 2)):
            rc, got = emu.transform(a, m, q, w, arith)
            assert rc == 0 and np.array_equal(got, expect), (hex(q), arith)
            rc, back = emu.transform(got, m, q, w, arith, inverse=True)
            assert rc == 0 and np.array_equal(back, a), (hex(q), arith)
        assert emu.chk_stats()[0] == 0, hex(q)
        if not wide:
            assert emu.transform(a, m, q, w, 4)[0] == 0          # the wide policy is valid for every q < 2^52 as well


# ---- products of operands in the NTT domain: dot_inv_kernel's logic and arithmetic on the CPU -------------------------
def _dot_operands(oracle, n, q, batch, k, seed, lazy, bcast):
    """k operand pairs in the NTT domain: uniform words, with the extreme residues in the first slots; lazy: random
    multiples of q added so that the words cover [0,4q)"""
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


@pytest.mark.parametrize("k", [1, 2, 3, 8])
@pytest.mark.parametrize("m,q", [(12, 0x7fffffffe0001), (14, 0x7fffffffe0001), (14, 0x3ffffffdf0001), (8, 0x7ffe0001),
                                 (13, 0xffffffff00001), (14, 0xffffffff00001), (11, 0x7fffffffe0001)])
def test_dot_kernel_logic_checked_policy(oracle, emu, m, q, k):
    """c = inv(sum_i a_i^ (.) b_i^) as dot_inv_kernel forms it -- products where the inverse transform would convert its
    inputs, running sums folded every kDotEvery terms -- with the CHECKED FP64 policy (every value an integer below 2^53,
    every product exact, every bound of ntt_arith.h held), canonical and lazy operand words, per-polynomial and
    broadcast b; three modulus classes: scheduled class 0 (51 bits), class 1 (50 bits / small), reduce-both-operands (52 bits)"""
    n = 1 << m
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    for lazy, bcast in ((False, False), (True, False), (False, True), (True, True)):
        a_list, b_list = _dot_operands(oracle, n, q, 2, k, 4000 + 10 * m + k, lazy, bcast)
        emu.chk_stats(reset=True)
        rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=1, lazy=lazy, bcast=bcast)
        assert rc == 0
        exp = cx.inv(oracle.dot(a_list, b_list, q, n, bcast))
        assert np.array_equal(got, exp), (m, hex(q), k, lazy, bcast)
        fails, maxb, _ = emu.chk_stats()
        assert fails == 0
        assert maxb < 2.0 ** 53 / q * (1 - 1 / 64)


def test_dot_kernel_logic_worst_case_sums(oracle, emu):
    """all-(q-1) and alternating-sign operands drive every term to its largest magnitude and every running sum to the
    edge of its folding schedule: still exact, k up to the launch limit"""
    m, q = 12, 0x7fffffffe0001
    n = 1 << m
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    for k in (4, 7, 32):
        hi = np.full(n, q - 1, dtype=np.uint64)
        mid = np.full(n, q // 2, dtype=np.uint64)
        a_list = [hi if i % 2 == 0 else mid for i in range(k)]
        b_list = [mid if i % 3 == 0 else hi for i in range(k)]
        emu.chk_stats(reset=True)
        rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=1)
        assert rc == 0 and np.array_equal(got, cx.inv(oracle.dot(a_list, b_list, q)))
        assert emu.chk_stats()[0] == 0


@pytest.mark.parametrize("m,q", [(15, 0x7fffffffe0001), (16, 0xffffffff00001), (17, 0x3ffffffdf0001)])
def test_dot_kernel_logic_two_pass_sizes(oracle, emu, m, q):
    """above 2^14 the product rides in the blocks of the inverse's first pass (2^12-point blocks at 2^15 / 2^16, 2^14 at 2^17),
    the column passes follow: same result as inverse(sum of products)"""
    n = 1 << m
    if (q - 1) % (2 * n):
        q = oracle.find_prime(50, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    a_list, b_list = _dot_operands(oracle, n, q, 1, 2, 4100 + m, False, False)
    rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=1)
    assert rc == 0 and np.array_equal(got, cx.inv(oracle.dot(a_list, b_list, q)))


@pytest.mark.parametrize("m,q", [(6, 0x1e01), (10, 0x10001), (14, 0x1000000000b00001), (12, 0xffffffffffc0001), (15, 0xffffffffffc0001)])
def test_dot_kernel_logic_integer_policy(oracle, emu, m, q):
    """the same kernel with the reference's integer arithmetic (fast_mul_mod_q per product, one conditional subtract per
    term), moduli up to 60 bits, lazy words included"""
    n = 1 << m
    if (q - 1) % (2 * n) or not oracle.lib.orc_is_prime(q):
        q = oracle.find_prime(60, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    for k, lazy, bcast in ((1, False, False), (3, True, False), (5, False, True)):
        a_list, b_list = _dot_operands(oracle, n, q, 2, k, 4200 + m, lazy, bcast)
        rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=0, lazy=lazy, bcast=bcast)
        assert rc == 0 and np.array_equal(got, cx.inv(oracle.dot(a_list, b_list, q, n, bcast))), (m, hex(q), k)


# ---- forward transform with the product at its output: fwd_mul_kernel's logic and arithmetic on the CPU ------------------
@pytest.mark.parametrize("m,q,arith", [(12, 0x7fffffffe0001, 1), (14, 0x7fffffffe0001, 1), (14, 0x3ffffffdf0001, 1), (8, 0x7ffe0001, 1),
                                       (13, 0xffffffff00001, 1), (14, 0xffffffff00001, 1), (15, 0x7fffffffe0001, 1), (15, 0xffffffff00001, 1), (16, 0xffffffff00001, 1),
                                       (10, 0x10001, 0), (14, 0xffffffffffc0001, 0), (15, 0xffffffffffc0001, 0)])
def test_fwd_mul_kernel_logic(oracle, emu, m, q, arith):
    """c^ = fwd(a) (.) b^ and c^ += fwd(a) (.) b^ as fwd_mul_kernel forms them -- the product where the forward transform would
    reduce and store its outputs -- with the CHECKED FP64 policy (every value an integer below 2^53, every product exact, the sum
    with the accumulator exact) or the integer policy; canonical and lazy b^, per-polynomial and broadcast; above 2^14 the
    product rides in the block pass behind the column passes"""
    n = 1 << m
    if (q - 1) % (2 * n) or not oracle.lib.orc_is_prime(q):
        q = oracle.find_prime(60 if arith == 0 else 50, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    batch = 2 if m <= 14 else 1
    a = _inputs(oracle, n, q, batch, 4300 + m)
    fa = cx.fwd(a)
    rng = np.random.default_rng(m)
    for lazy, bcast, acc in ((False, False, False), (True, False, True), (False, True, True), (True, True, False)):
        b = oracle.fill_uniform((1 if bcast else batch) * n, q, 4400 + m)
        b[:4] = [q - 1, 0, q // 2, 1]
        bw = b + (rng.integers(0, 4, b.size).astype(np.uint64) * np.uint64(q) if lazy else np.uint64(0))
        c0 = oracle.fill_uniform(batch * n, q, 4500 + m)
        c0[:2] = [q - 1, 0]
        exp = oracle.pointwise(fa, np.tile(b, batch) if bcast else b, q)
        if acc:
            exp = (exp + c0) % np.uint64(q)
        # N = 2^15, FP64 policies: the one-pass kernel with the product at its output (round 6: onepass_mul_kernel) AND the two-pass route
        for route in ((1, 0) if m == 15 and arith == 1 else (-1,)):
            emu.set_one_pass(route)
            emu.chk_stats(reset=True)
            rc, got = emu.fwd_mul(a, bw, m, q, w, arith=arith, lazy=lazy, bcast=bcast, acc=c0 if acc else None)
            assert rc == 0 and np.array_equal(got, exp), (m, hex(q), lazy, bcast, acc, route)
            assert emu.chk_stats()[0] == 0
        emu.set_one_pass(-1)


@pytest.mark.parametrize("m", [3, 7, 11, 15])
def test_radix4x4_remainder_layers_logic(oracle, emu, m):
    """log2 N = 4k+3: the layer functions behind fwd_ntt_radix4x4_lazy (2k radix-4 layers, the radix-2 stage with the
    reference's reduction of a[group counter], the last radix-4 layer; src/ntt_radix4x4.c:53-111) leave the words of the
    oracle's restatement, which oracle/gen_golden.py pins against the compiled reference"""
    n = 1 << m
    for q in (oracle.find_prime(59, n), oracle.find_prime(31, n), 0x10001 if m <= 15 else None):
        if q is None:
            continue
        cx = oracle.ctx(n, q, oracle.min_root(q, n))
        a = oracle.fill_uniform(n, q, 7100 + m)
        a[: min(n, 16)] = q - 1
        got = emu.fwd_r4x4_layers(a, q, cx.table("e"), cx.table("econ"))
        assert np.array_equal(got, cx.fwd_r4x4_lazy(a))
        assert not np.array_equal(got, cx.fwd_r4_lazy(a)) or m == 3


@pytest.mark.parametrize("k,top", [(3, 58), (1, 60), (0, 61)])
@pytest.mark.parametrize("m", [6, 9, 12, 13, 14, 16])
def test_wide_integer_policy(oracle, emu, m, k, top):
    """ArithU64X<K> (ntt_arith.h): estimated Shoup quotient, no conditional subtraction per butterfly, folds where
    u64x_schedule says.  Bit-exact against the oracle for the largest prime of each headroom class and a 53-bit one,
    forward / inverse / lazy / wide inputs / column-pass form; the checked policy asserts every range claim against 128-bit
    arithmetic (no wrap-around, products below 4q, reduce_any below 2.01 q, sums below (B/2) q at every inverse stage) --
    once with the words the arithmetic produces and once with every product and fold replaced by the LARGEST representative
    its claim allows, which makes the values grow as fast as the schedule assumes"""
    n = 1 << m
    B = 8 << k
    for q in (_prime_near(oracle, (1 << top) - 1, n), _prime_near(oracle, 1 << 52, n, below=False)):
        assert (1 << 40) <= q < (1 << top)
        w = oracle.min_root(q, n)
        cx = oracle.ctx(n, q, w)
        a = oracle.fill_uniform(2 * n, q, 64 + m + k)
        a[:6] = [0, 1, q - 1, q - 2, q // 2, q // 2 + 1]
        expect = cx.fwd(a)
        emu.chk_stats()
        for worst in (0, 1):
            emu.set_u64x_worst(worst)
            rc, got = emu.transform(a, m, q, w, 6, ksh=k)
            assert rc == 0 and np.array_equal(got, expect), (hex(q), worst)
            rc, back = emu.transform(got, m, q, w, 6, ksh=k, inverse=True)
            assert rc == 0 and np.array_equal(back, a), (hex(q), worst)
            rc, lz = emu.transform(a + np.uint64(3 * q), m, q, w, 6, ksh=k, lazy=True)     # lazy words in, lazy words out
            assert rc == 0 and int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), expect)
            rc, lb = emu.transform(expect + np.uint64(3 * q), m, q, w, 6, ksh=k, inverse=True, lazy=True)
            assert rc == 0 and int(lb.max()) < 2 * q and np.array_equal(lb % np.uint64(q), a)
            rc, back = emu.transform(expect + np.uint64(7 * q), m, q, w, 6, ksh=k, inverse=True, wide=True)
            assert rc == 0 and np.array_equal(back, a)
            rc, got = emu.transform(a + np.uint64(7 * q), m, q, w, 6, ksh=k, wide=True)
            assert rc == 0 and np.array_equal(got, expect)
            if m <= 12:
                rc, got = emu.transform(a, m, q, w, 6, ksh=k, generic=True)
                assert rc == 0 and np.array_equal(got, expect)
                rc, back = emu.transform(expect, m, q, w, 6, ksh=k, generic=True, inverse=True)
                assert rc == 0 and np.array_equal(back, a)
            fails, maxb, _ = emu.chk_stats()
            assert fails == 0 and maxb < B, (hex(q), worst, fails, maxb)
            if worst and m >= 12 and k == 3:
                assert maxb > 40, maxb        # the injection works: values do climb towards 4 + 4 * stages
        emu.set_u64x_worst(0)
    assert emu.transform(a, m, (1 << top) + 1, 3, 6, ksh=k)[0] == -2


@pytest.mark.parametrize("top", [58, 60, 61])
def test_inner_products_wide_integer_policy_fold_schedule(oracle, emu, top):
    """32 operand pairs, lazy words, worst-case representatives: the running sums are folded every 20 / 4 / 1 terms
    (kDotEvery of the three headroom classes) and never pass B q (checked policy)"""
    m, n = 8, 256
    q = _prime_near(oracle, (1 << top) - 1, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    for lazy in (False, True):
        a_list, b_list = _dot_operands(oracle, n, q, 2, 32, 7700 + top, lazy, False)
        for worst in (0, 1):
            emu.set_u64x_worst(worst)
            emu.chk_stats(reset=True)
            rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=6, lazy=lazy, bcast=False)
            assert rc == 0 and np.array_equal(got, cx.inv(oracle.dot(a_list, b_list, q, n, False))), (top, lazy, worst)
            fails, maxb, _ = emu.chk_stats()
            assert fails == 0 and maxb < (8 << {58: 3, 60: 1, 61: 0}[top]), (top, lazy, worst, fails, maxb)
    emu.set_u64x_worst(0)


@pytest.mark.parametrize("m,top", [(6, 58), (12, 58), (14, 58), (14, 61), (15, 58), (16, 61), (12, 60)])
def test_ntt_domain_products_wide_integer_policy(oracle, emu, m, top):
    """dot_inv_kernel and fwd_mul_kernel with ArithU64X's stages around fast_mul_mod_q's products (class 3 below 2^58,
    class 0 below 2^61), the checked policy with and without worst-case representatives: the inverse stages start from
    canonical sums, the products at the forward transform's output take its unfolded values (any 64-bit word)"""
    n = 1 << m
    q = _prime_near(oracle, (1 << top) - 1, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    batch = 2 if m <= 14 else 1
    a = _inputs(oracle, n, q, batch, 7300 + m)
    fa = cx.fwd(a)
    for worst in (0, 1):
        emu.set_u64x_worst(worst)
        emu.chk_stats(reset=True)
        for k, lazy, bcast in ((1, False, False), (3, True, False), (5, False, True)):
            a_list, b_list = _dot_operands(oracle, n, q, batch, k, 7400 + m, lazy, bcast)
            rc, got = emu.inv_dot(a_list, b_list, m, q, w, arith=6, lazy=lazy, bcast=bcast)
            assert rc == 0 and np.array_equal(got, cx.inv(oracle.dot(a_list, b_list, q, n, bcast))), (m, hex(q), k, worst)
        for lazy, bcast, acc in ((False, False, False), (True, False, True), (False, True, True)):
            b = oracle.fill_uniform((1 if bcast else batch) * n, q, 7500 + m)
            bw = b + (np.uint64(3 * q) if lazy else np.uint64(0))
            c0 = oracle.fill_uniform(batch * n, q, 7600 + m)
            exp = oracle.pointwise(fa, np.tile(b, batch) if bcast else b, q)
            if acc:
                exp = (exp + c0) % np.uint64(q)
            rc, got = emu.fwd_mul(a, bw, m, q, w, arith=6, lazy=lazy, bcast=bcast, acc=c0 if acc else None)
            assert rc == 0 and np.array_equal(got, exp), (m, hex(q), lazy, bcast, acc, worst)
        assert emu.chk_stats()[0] == 0
    emu.set_u64x_worst(0)


@pytest.mark.parametrize("k", [0, 1, 3])
def test_wide_integer_fold_schedule_bounds(emu, k):
    """u64x_schedule, simulated with the worst-case growth of every stage: values never reach B q (B = 8, 16, 64), the
    inverse's difference offset (B/2) q covers its subtrahend, every inverse pass ends on a folding stage, and the
    documented shapes hold (K = 3: a 14-stage forward block never folds; its inverse folds every fourth stage)"""
    B = 8 << k
    for n in range(1, 18):
        m = emu.u64x_schedule(False, n, k)
        b = 4.0                                     # pass inputs: canonical words, lazy words below 4q, the previous pass's words
        for s in range(n):
            if (m >> s) & 1:
                b = max(b - B / 2, B / 2)           # x >= (B/2) q ? x - (B/2) q : x
            b += 4                                  # x + m and x + 4q - m with m < 4q
            assert b <= B, (k, n, s, b)
        mi = emu.u64x_schedule(True, n, k)
        assert (mi >> (n - 1)) & 1                  # a pass hands on words below 4q
        b = 4.0
        for s in range(n):
            assert 2 * b <= B, (k, n, s, b)         # the sum fits, (B/2) q >= y
            b = 4.0 if (mi >> s) & 1 else 2 * b     # folded sums below 2.01 q, products below 4q
    assert emu.u64x_schedule(False, 14, 3) == 0 and emu.u64x_schedule(True, 12, 3) == 0b100010001000
    assert emu.u64x_schedule(False, 14, 0) == 0x3ffe and emu.u64x_schedule(True, 14, 0) == 0x3fff


@pytest.mark.parametrize("m,q", [(8, 0x7fffffffe0001), (12, 0x80000001c0001), (13, 0x3ffffffdf0001), (15, 0x7fffffffe0001)])
def test_caller_native_layout_block_offsets(oracle, emu, m, q):
    """SURVEY 8(d)'s [batch][prime][N] layout through csrc/ntt_core.h block_offset -- the address computation the kernels and the
    library's *_strided entry points share: limb l of a [batch][3][N] buffer is transformed in place with polynomials 3 N words
    apart (whole-polynomial blocks, blocks below a column pass at 2^15, the column pass itself); the other limbs' words must
    stay untouched and every polynomial must equal the oracle's transform.  Forward, inverse, the NTT-domain inner product and
    the forward transform times a transformed operand."""
    n, batch, nl = 1 << m, 3, 3
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    rng = np.random.default_rng(m)
    buf = (rng.integers(0, q, size=(batch, nl, n), dtype=np.uint64)).copy()
    buf[0, 1, :4] = q - 1
    orig = buf.copy()
    limb = 1
    assert emu.transform_limb(buf.reshape(-1), limb, nl, batch, m, q, w, F64) == 0
    for l in range(nl):
        for p in range(batch):
            exp = cx.fwd(orig[p, l].copy()) if l == limb else orig[p, l]
            assert np.array_equal(buf[p, l], exp), (p, l)
    fwd = buf.copy()
    assert emu.transform_limb(buf.reshape(-1), limb, nl, batch, m, q, w, F64, inverse=True) == 0
    assert np.array_equal(buf, orig)
    # c = inv(a0^ b0^ + a1^ b1^) on the limb, operands [k][batch][nl][N]
    k = 2
    a = rng.integers(0, q, size=(k, batch, nl, n), dtype=np.uint64)
    b = rng.integers(0, q, size=(k, batch, nl, n), dtype=np.uint64)
    out = np.full((batch, nl, n), 7, dtype=np.uint64)
    assert emu.inv_dot_limb(out.reshape(-1), a.reshape(-1), b.reshape(-1), limb, nl, k, batch, m, q, w) == 0
    for p in range(batch):
        s = oracle.pointwise(a[0, p, limb].copy(), b[0, p, limb].copy(), q)
        s = (s + oracle.pointwise(a[1, p, limb].copy(), b[1, p, limb].copy(), q)) % np.uint64(q)
        assert np.array_equal(out[p, limb], cx.inv(s)), p
        assert (out[p, 0] == 7).all() and (out[p, 2] == 7).all()
    # c^ = fwd(a) (.) b^ on the limb
    acoef = orig.copy()
    out2 = np.full((batch, nl, n), 9, dtype=np.uint64)
    assert emu.fwd_mul_limb(out2.reshape(-1), acoef.reshape(-1), b[0].copy().reshape(-1), limb, nl, batch, m, q, w) == 0
    for p in range(batch):
        assert np.array_equal(out2[p, limb], oracle.pointwise(fwd[p, limb].copy(), b[0, p, limb].copy(), q)), p
        assert (out2[p, 0] == 9).all()


@pytest.mark.parametrize("m,q,arith", [(6, 0x7ffe0001, 1), (10, 0x7ffe0001, 1), (12, 0x7fffffffe0001, 1), (13, 0x7fffffffe0001, 1),
                                       (15, 0x7fffffffe0001, 1), (12, 0x7fffffffe0001, 0), (3, 0x7ffe0001, 0)])
def test_pointer_batch_table_addressing(oracle, emu, m, q, arith):
    """round 6: a batch of separately held polynomials (the reference's own batch form, fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...),
    include/ntt_reference.h:44-49) through the table addressing of the transform kernels -- csrc/ntt_core.h poly_offset /
    block_offset with Params::ptab, the functions fused_kernel, column_kernel and team_kernel call -- executed on the CPU: polynomials
    at irregular, unsorted places of one buffer (odd word offsets included), every one against the oracle, every other word
    untouched; forward and inverse, block passes and column passes (m = 15: both; m = 3: columns only)"""
    n = 1 << m
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    guard = np.uint64(0xA5A5A5A5A5A5A5A5)
    offs = [5 * n + 3, 0, 2 * n + 1, 9 * n + 8, 7 * n + 2]          # no progression, not sorted
    buf = np.full(11 * n, guard, dtype=np.uint64)
    polys = [oracle.fill_uniform(n, q, 700 + i) for i in range(len(offs))]
    for o, a in zip(offs, polys):
        buf[o:o + n] = a
    mask = np.ones(buf.size, dtype=bool)
    for o in offs:
        mask[o:o + n] = False
    assert emu.transform_ptrs(buf, offs, m, q, w, arith) == 0
    for o, a in zip(offs, polys):
        assert np.array_equal(buf[o:o + n], cx.fwd(a.copy())), o
    assert (buf[mask] == guard).all()
    assert emu.transform_ptrs(buf, offs[::-1], m, q, w, arith, inverse=True) == 0
    for o, a in zip(offs, polys):
        assert np.array_equal(buf[o:o + n], a), o
    assert (buf[mask] == guard).all()


@pytest.mark.parametrize("arith,bits,ksh", [(2, 51, -1), (2, 50, -1), (2, 33, -1), (2, 33, 0), (5, 52, -1), (1, 51, -1), (4, 52, -1)])
def test_one_pass_2p15_against_the_oracle_and_the_two_pass_route(oracle, emu, arith, bits, ksh):
    """round 6: N = 2^15 in ONE pass (csrc/ntt_kernels_block.h onepass_kernel) executed on the CPU -- the pair stage thread-local, both
    halves through the 2^14-point block stages at block positions 0 and 1, forward with ONE reduction schedule over all fifteen stages
    (ntt_core.h onepass_fwd_mask), inverse with both inputs of the folded last stage reduced first.  The CHECKED policies (arith 2, 5)
    assert every exactness bound of DESIGN 4 with 128-bit integers, on random and on extreme inputs, for every headroom class (51, 50
    and 33-bit moduli: classes 0, 1 and 18; a 33-bit modulus forced through class 0); results against the oracle and bit for bit
    against the two-pass route."""
    m, n = 15, 1 << 15
    q = oracle.find_prime(bits, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    rnd = oracle.fill_uniform(2 * n, q, 4242)
    edge = np.full(2 * n, q - 1, dtype=np.uint64)                      # first polynomial: every coefficient maximal
    edge[n + 1::2] = 1                                                 # second: the +-1 pattern
    half = np.full(2 * n, q // 2, dtype=np.uint64)                     # largest balanced magnitude
    emu.chk_stats(reset=True)
    for data in (rnd, edge, half):
        emu.set_one_pass(1)
        rc, f1 = emu.transform(data, m, q, w, arith, ksh=ksh)
        assert rc == 0 and np.array_equal(f1, cx.fwd(data.copy()))
        emu.set_one_pass(0)
        rc, f0 = emu.transform(data, m, q, w, arith, ksh=ksh)
        assert rc == 0 and np.array_equal(f0, f1)
        emu.set_one_pass(1)
        rc, b1 = emu.transform(f1, m, q, w, arith, inverse=True, ksh=ksh)
        assert rc == 0 and np.array_equal(b1, data)
        rc, inv = emu.transform(data, m, q, w, arith, inverse=True, ksh=ksh)       # inputs the inverse does not get from a forward transform
        assert rc == 0 and np.array_equal(inv, cx.inv(data.copy()))
        if q < (1 << 51):
            # lazy (wide) inputs through the one-pass route: the words the reference's *_lazy entry points emit
            rc, b2 = emu.transform(f1 + np.uint64(3 * q), m, q, w, arith, inverse=True, wide=True, ksh=ksh)
            assert rc == 0 and np.array_equal(b2, data)
            rc, f2 = emu.transform(data + np.uint64(7 * q), m, q, w, arith, wide=True, ksh=ksh)
            assert rc == 0 and np.array_equal(f2, f1)
    emu.set_one_pass(-1)
    fails, maxb, maxr = emu.chk_stats()
    assert fails == 0
    if arith == 2:
        assert maxb < 2.0 ** 53 / q * (1 - 1 / 64), (maxb, q)


def test_52_bit_inverse_input_contract(oracle, emu):
    """advisor r05: plans for 2^51 < q < 2^52 multiply the unreduced difference of two INPUTS in their first inverse stage (round 5's
    per-slot plan, kCanonInFlag) -- exact for canonical words only.  include/ntt_mi355x.h states the contract; this pins both sides of
    it on the CPU with the checked policy: words in [q, 2q) through the strict inverse trip the exactness checks (or give wrong
    words), the same words through the wide form -- where callers with lazy words belong -- are exact and right."""
    m, n = 12, 1 << 12
    q = oracle.find_prime(52, n)
    w = oracle.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    a = oracle.fill_uniform(2 * n, q, 77)
    fa = cx.fwd(a.copy())
    lazy = fa.copy()
    lazy[::3] += np.uint64(q)                 # every third word in [q, 2q)
    emu.chk_stats(reset=True)
    rc, back = emu.transform(lazy, m, q, w, 5, inverse=True, wide=True)
    assert rc == 0 and np.array_equal(back, a) and emu.chk_stats()[0] == 0
    emu.chk_stats(reset=True)
    rc, strict = emu.transform(lazy, m, q, w, 5, inverse=True)
    assert rc == 0 and (emu.chk_stats()[0] > 0 or not np.array_equal(strict, a)), "the strict inverse is documented NOT to take lazy words"
    emu.chk_stats(reset=True)
