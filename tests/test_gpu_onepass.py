"""N = 2^15 in ONE pass (round 6; review r05 item 5): onepass_kernel holds a whole polynomial in the registers of one 1024-thread
workgroup -- the stage on pairs 2^14 apart runs thread-locally, the halves go through the 2^14-point block stages one after the
other -- so every coefficient crosses HBM exactly twice (reference cases 14-15 are m = 15: tests/test_cases.h:145-208).  Through
the C ABI against the oracle and, whole slabs, bit for bit against the two-pass routes (NTT_OPT_ONE_PASS 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M, N = 15, 1 << 15


def _plan(lib, bits, skip=0):
    q = lib.find_prime(bits, N, skip)
    w = lib.min_root(q, N)
    return lib.Plan(N, q, w), q, w


@pytest.mark.parametrize("bits", [51, 50, 33, 52])
@pytest.mark.parametrize("batch", [300, 517])
def test_one_pass_transforms_against_oracle_and_two_pass_routes(lib, oracle, bits, batch):
    """every FP64 class (51-, 50-, 33-bit moduli: classes 0, 1, 18; 52-bit: the reduce-both-operands policy), batches above the CU
    count (the automatic choice) that are not a multiple of it: forward, inverse, lazy (wide) inputs"""
    plan, q, w = _plan(lib, bits)
    cx = oracle.ctx(N, q, w)
    assert plan.get_option(lib.OPT_ONE_PASS) == -1 and plan.info()["hbm_passes"] == 1
    d = lib.DeviceBuffer(batch * N)
    lib.fill_uniform(d.ptr, batch * N, q, 77, 0)
    a = d.download()
    plan.fwd(d.ptr, batch)
    one = d.download()
    for p in (0, 1, batch // 2, batch - 1):
        assert np.array_equal(one[p * N:(p + 1) * N], cx.fwd(a[p * N:(p + 1) * N].copy())), p
    plan.set_option(lib.OPT_ONE_PASS, 0)
    for xcd in (0, 1):
        plan.set_option(lib.OPT_XCD_LOCAL, xcd)
        d.upload(a)
        plan.fwd(d.ptr, batch)
        assert np.array_equal(d.download(), one), "two-pass route, xcd_local=%d" % xcd
    plan.set_option(lib.OPT_ONE_PASS, -1)
    plan.inv(d.ptr, batch)
    assert np.array_equal(d.download(), a)
    # extreme inputs: every coefficient q - 1; the +-1 pattern; q // 2
    edge = np.full(batch * N, q - 1, dtype=np.uint64)
    edge[N + 1:2 * N:2] = 1
    edge[2 * N:3 * N] = q // 2
    d.upload(edge)
    plan.fwd(d.ptr, batch)
    got = d.download()
    for p in (0, 1, 2, batch - 1):
        assert np.array_equal(got[p * N:(p + 1) * N], cx.fwd(edge[p * N:(p + 1) * N].copy())), p
    plan.inv(d.ptr, batch)
    assert np.array_equal(d.download(), edge)
    if bits < 52:
        # the reference's lazy ranges in: [0,8q) forward, [0,4q)-style words into the inverse
        d.upload(a + np.uint64(7 * q))
        plan.fwd(d.ptr, batch, wide=True)
        assert np.array_equal(d.download(), one)
        d.upload(one + np.uint64(3 * q))
        plan.inv(d.ptr, batch, wide=True)
        assert np.array_equal(d.download(), a)
    # lazy OUTPUTS: the one-pass kernel's canonical words satisfy the lazy contract
    d.upload(a)
    plan.fwd(d.ptr, batch, lazy=True)
    lz = d.download()
    assert int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), one)
    d.free(), plan.destroy()


def test_one_pass_choice_and_forced_small_batches(lib, oracle):
    """the automatic choice wants a polynomial for every CU; NTT_OPT_ONE_PASS 1 forces the kernel for any batch (1, 3, 255 polynomials:
    fewer workgroups than CUs, a workgroup's loop of one iteration), 0 switches it off"""
    plan, q, w = _plan(lib, 51)
    cx = oracle.ctx(N, q, w)
    for batch in (1, 3, 255, 256, 257):
        a = oracle.fill_uniform(batch * N, q, 9 + batch)
        d = lib.DeviceBuffer(batch * N).upload(a)
        for mode in (1, 0, -1):
            plan.set_option(lib.OPT_ONE_PASS, mode)
            d.upload(a)
            plan.fwd(d.ptr, batch)
            got = d.download()
            for p in {0, batch - 1}:
                assert np.array_equal(got[p * N:(p + 1) * N], cx.fwd(a[p * N:(p + 1) * N].copy())), (batch, mode, p)
            plan.inv(d.ptr, batch)
            assert np.array_equal(d.download(), a), (batch, mode)
        d.free()
    plan.destroy()


def test_one_pass_layouts_pointer_tables_and_rns_sets(lib, oracle):
    """the kernel behind every transform entry point: padded strides, shuffled pointer batches (the polynomial's address from the
    table), RNS sets in one launch over the limbs (MULTI: blockIdx.y is the limb) in both layouts"""
    plan, q, w = _plan(lib, 50)
    cx = oracle.ctx(N, q, w)
    plan.set_option(lib.OPT_ONE_PASS, 1)
    GUARD = np.uint64(0xA5A5A5A5A5A5A5A5)
    # strided
    batch, stride = 9, N + 24
    img = np.full(batch * stride, GUARD, dtype=np.uint64)
    polys = oracle.fill_uniform(batch * N, q, 5).reshape(batch, N)
    for p in range(batch):
        img[p * stride:p * stride + N] = polys[p]
    d = lib.DeviceBuffer(img.size).upload(img)
    plan.transform_strided(d.ptr, stride, batch)
    got = d.download()
    for p in range(batch):
        assert np.array_equal(got[p * stride:p * stride + N], cx.fwd(polys[p].copy())), p
        assert (got[p * stride + N:(p + 1) * stride] == GUARD).all()
    plan.transform_strided(d.ptr, stride, batch, lib.FLAG_INVERSE)
    assert np.array_equal(d.download(), img)
    # pointer table, shuffled, every second polynomial at an odd word offset (only 8-byte aligned)
    order = [4, 0, 7, 2, 8, 1, 6, 3, 5]
    offs = [p * stride + (3 if p % 2 else 0) for p in order]
    img = np.full(batch * stride, GUARD, dtype=np.uint64)
    for o, a in zip(offs, polys):
        img[o:o + N] = a
    d.upload(img)
    tab = lib.DeviceBuffer(batch).upload(np.array([d.ptr + 8 * o for o in offs], dtype=np.uint64))
    plan.transform_dev_ptrs(tab.ptr, batch)
    got = d.download()
    mask = np.ones(img.size, dtype=bool)
    for o, a in zip(offs, polys):
        assert np.array_equal(got[o:o + N], cx.fwd(a.copy())), o
        mask[o:o + N] = False
    assert (got[mask] == GUARD).all()
    plan.transform_ptrs([d.ptr + 8 * o for o in offs], lib.FLAG_INVERSE)
    assert np.array_equal(d.download(), img)
    tab.free(), d.free(), plan.destroy()
    # RNS sets
    nl = 3
    qs = [lib.find_prime(50, N, k) for k in range(nl)]
    ws = [lib.min_root(x, N) for x in qs]
    plans = [lib.Plan(N, x, y) for x, y in zip(qs, ws)]
    ctx = [oracle.ctx(N, x, y) for x, y in zip(qs, ws)]
    for p in plans:
        p.set_option(lib.OPT_ONE_PASS, 1)
    for batch, launch in ((5, "0"), (5, "1"), (120, None)):
        lib.set_rns_launch(plans, launch)
        a = np.concatenate([oracle.fill_uniform(batch * N, x, 70 + l) for l, x in enumerate(qs)])
        d = lib.DeviceBuffer(a.size).upload(a)
        lib.rns_fwd(plans, d.ptr, batch)
        got = d.download()
        for l in range(nl):
            for p in (0, batch - 1):
                sl = slice((l * batch + p) * N, (l * batch + p + 1) * N)
                assert np.array_equal(got[sl], ctx[l].fwd(a[sl].copy())), (batch, launch, l, p)
        lib.rns_inv(plans, d.ptr, batch)
        assert np.array_equal(d.download(), a), (batch, launch)
        # [batch][limb][N]
        bm = a.reshape(nl, batch, N).transpose(1, 0, 2).copy().reshape(-1)
        d.upload(bm)
        lib.rns_fwd(plans, d.ptr, batch, layout=lib.batch_major(plans))
        lib.rns_inv(plans, d.ptr, batch, layout=lib.batch_major(plans))
        assert np.array_equal(d.download(), bm), (batch, launch, "batch-major")
        d.free()
    for p in plans:
        p.destroy()


@pytest.mark.parametrize("bits", [51, 50, 33, 52])
def test_one_pass_forward_with_the_product_at_its_output(lib, oracle, bits):
    """onepass_mul_kernel: c^ = fwd(a) . b^ and c^ += fwd(a) . key^ at N = 2^15 in one pass (the one-pass forward transform with
    fwd_mul_kernel's epilogue where a half would be reduced and stored): canonical / lazy b^, per-polynomial / broadcast, accumulating
    or not, against the oracle on sampled polynomials and bit for bit against the two-pass forms over the whole slab; a is left as it
    was; c^ may alias a or b^"""
    plan, q, w = _plan(lib, bits)
    cx = oracle.ctx(N, q, w)
    batch = 261
    a = oracle.fill_uniform(batch * N, q, 11)
    b = oracle.fill_uniform(batch * N, q, 12)
    c0 = oracle.fill_uniform(batch * N, q, 13)
    fb = np.concatenate([cx.fwd(b[p * N:(p + 1) * N].copy()) for p in (0, 1, batch - 1)])
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size), lib.DeviceBuffer(a.size)
    plan.fwd(db.upload(b).ptr, batch)
    bhat = db.download()
    assert np.array_equal(np.concatenate([bhat[p * N:(p + 1) * N] for p in (0, 1, batch - 1)]), fb)
    rng = np.random.default_rng(bits)
    for lazy, bcast, acc in ((False, False, False), (True, False, True), (False, True, True), (True, True, False)):
        bw = bhat + (rng.integers(0, 4, bhat.size).astype(np.uint64) * np.uint64(q) if (lazy and bits < 52) else np.uint64(0))
        flags = (lib.MUL_LAZY_IN if lazy else 0) | (lib.MUL_B_BROADCAST if bcast else 0) | (lib.MUL_ACCUMULATE if acc else 0)
        res = {}
        for mode in (1, 0):
            plan.set_option(lib.OPT_ONE_PASS, mode)
            da.upload(a), db.upload(bw), dc.upload(c0)
            plan.fwd_mul(dc.ptr, da.ptr, db.ptr, batch, flags)
            res[mode] = dc.download()
            if mode == 1:
                assert np.array_equal(da.download(), a), "one pass: a is left as it was"
        assert np.array_equal(res[0], res[1]), (lazy, bcast, acc)
        for p in (0, 1, batch - 1):
            sl = slice(p * N, (p + 1) * N)
            exp = oracle.pointwise(cx.fwd(a[sl].copy()), bhat[:N].copy() if bcast else bhat[sl].copy(), q)
            if acc:
                exp = (exp + c0[sl]) % np.uint64(q)
            assert np.array_equal(res[1][sl], exp), (lazy, bcast, acc, p)
    # aliasing: c^ on a, c^ on b^
    plan.set_option(lib.OPT_ONE_PASS, 1)
    da.upload(a), db.upload(bhat)
    plan.fwd_mul(da.ptr, da.ptr, db.ptr, batch)
    got = da.download()
    da.upload(a)
    plan.fwd_mul(db.ptr, da.ptr, db.ptr, batch)
    assert np.array_equal(db.download(), got)
    sl = slice((batch - 1) * N, batch * N)
    assert np.array_equal(got[sl], oracle.pointwise(cx.fwd(a[sl].copy()), bhat[sl].copy(), q))
    for x in (da, db, dc):
        x.free()
    plan.destroy()


def test_one_pass_forward_product_over_rns_limbs(lib, oracle):
    """the MULTI variant (blockIdx.y = limb) behind ntt_rns_fwd_mul_batch, both layouts, small per-limb batches in one launch"""
    nl, batch = 3, 7
    qs = [lib.find_prime(50, N, k) for k in range(nl)]
    ws = [lib.min_root(x, N) for x in qs]
    plans = [lib.Plan(N, x, y) for x, y in zip(qs, ws)]
    ctx = [oracle.ctx(N, x, y) for x, y in zip(qs, ws)]
    for p in plans:
        p.set_option(lib.OPT_ONE_PASS, 1)
    a = np.concatenate([oracle.fill_uniform(batch * N, x, 70 + l) for l, x in enumerate(qs)])
    b = np.concatenate([oracle.fill_uniform(batch * N, x, 80 + l) for l, x in enumerate(qs)])
    bh = np.concatenate([ctx[l].fwd(b[(l * batch + p) * N:(l * batch + p + 1) * N].copy()) for l in range(nl) for p in range(batch)])
    exp = np.concatenate([oracle.pointwise(ctx[l].fwd(a[(l * batch + p) * N:(l * batch + p + 1) * N].copy()),
                                           bh[(l * batch + p) * N:(l * batch + p + 1) * N].copy(), qs[l]) for l in range(nl) for p in range(batch)])
    da, db, dc = lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size), lib.DeviceBuffer(a.size)
    for launch in ("0", "1"):
        lib.set_rns_launch(plans, launch)
        da.upload(a), db.upload(bh)
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), exp), launch
        tr = lambda v: v.reshape(nl, batch, N).transpose(1, 0, 2).copy().reshape(-1)
        da.upload(tr(a)), db.upload(tr(bh))
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lib.batch_major(plans))
        assert np.array_equal(dc.download(), tr(exp)), (launch, "batch-major")
    for x in (da, db, dc):
        x.free()
    for p in plans:
        p.destroy()
