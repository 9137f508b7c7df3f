"""GPU (-m gpu): caller-native layouts.  SURVEY 8(d) config 5 keeps RNS operands as [batch][prime][N] (a polynomial's limbs side
by side -- what FHE libraries hold); the plain ntt_rns_* entry points take [limb][batch][N].  The *_strided forms take both
distances in words; every entry point is compared with the oracle polynomial by polynomial in the batch-major layout, in a
padded layout, with one launch over the limbs and limb by limb, at sizes that reach every kernel family (several blocks per
workgroup, persistent block kernels, column passes, the XCD-local one-launch kernels), and with the untouched words checked.
The reference's batching precedent is two caller arrays, fwd_ntt_ref_harvey_lazy_dbl (include/ntt_reference.h:44-49)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GUARD = np.uint64(0xDEADBEEFCAFEF00D)


def _primes(lib, n, nlimbs, bits):
    qs = [lib.find_prime(bits, n, k) for k in range(nlimbs)]
    assert len(set(qs)) == nlimbs
    return qs, [lib.min_root(q, n) for q in qs]


class _Layout:
    """a host image of an RNS operand: polynomial p of limb l at l * limb + p * poly (words), guard words everywhere else"""

    def __init__(self, n, nlimbs, batch, limb, poly):
        self.n, self.nl, self.batch, self.limb, self.poly = n, nlimbs, batch, limb, poly
        self.words = (nlimbs - 1) * limb + (batch - 1) * poly + n

    def scatter(self, dense):
        """dense[l][p][:] -> image"""
        img = np.full(self.words, GUARD, dtype=np.uint64)
        for l in range(self.nl):
            for p in range(self.batch):
                o = l * self.limb + p * self.poly
                img[o:o + self.n] = dense[l, p]
        return img

    def gather(self, img):
        out = np.empty((self.nl, self.batch, self.n), dtype=np.uint64)
        mask = np.ones(self.words, dtype=bool)
        for l in range(self.nl):
            for p in range(self.batch):
                o = l * self.limb + p * self.poly
                out[l, p] = img[o:o + self.n]
                mask[o:o + self.n] = False
        assert (img[mask] == GUARD).all(), "words outside the operand's polynomials were written"
        return out

    @property
    def strides(self):
        return (self.limb, self.poly)


def _layouts(n, nlimbs, batch):
    return {
        "batch_major": _Layout(n, nlimbs, batch, n, nlimbs * n),                       # [batch][prime][N]
        "batch_major_padded": _Layout(n, nlimbs, batch, n + 64, nlimbs * (n + 64) + 128),
        "limb_major_padded": _Layout(n, nlimbs, batch, batch * (n + 32) + 256, n + 32),
    }


def _dense_inputs(oracle, n, qs, batch, seed):
    d = np.empty((len(qs), batch, n), dtype=np.uint64)
    for l, q in enumerate(qs):
        d[l] = oracle.fill_uniform(batch * n, q, seed + l).reshape(batch, n)
        d[l, 0, :4] = q - 1
        d[l, -1, -1] = q - 1
        d[l, 0, 4:8] = 0
    return d


def _expected(oracle, n, qs, roots, a, b):
    ctxs = [oracle.ctx(n, q, w) for q, w in zip(qs, roots)]
    fa = np.stack([cx.fwd(a[l].reshape(-1)).reshape(a[l].shape) for l, cx in enumerate(ctxs)])
    fb = np.stack([cx.fwd(b[l].reshape(-1)).reshape(b[l].shape) for l, cx in enumerate(ctxs)])
    prod = np.stack([cx.inv(oracle.pointwise(fa[l].reshape(-1), fb[l].reshape(-1), q)).reshape(a[l].shape)
                     for l, (cx, q) in enumerate(zip(ctxs, qs))])
    return ctxs, fa, fb, prod


CASES = [
    # logn, nlimbs, batch, bits            what it reaches
    (8, 3, 5, 50),      # several blocks per workgroup (per-lane block addresses), ragged tail
    (12, 4, 3, 50),     # persistent 2^12 kernels, MULTI variants
    (13, 3, 2, 52),     # two blocks per workgroup (forward), the 52-bit policy
    (14, 4, 2, 50),     # the headline kernels
    (14, 3, 2, 57),     # wide integer policy
    (15, 3, 2, 50),     # blocks below a column pass, chunks
    (16, 2, 3, 52),
    (17, 2, 2, 50),
    (5, 2, 3, 45),      # column passes only
]


def _run_entry_points(lib, oracle, monkeypatch, logn, nlimbs, batch, bits, lay_name, loop):
    n = 1 << logn
    qs, roots = _primes(lib, n, nlimbs, bits)
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    lay = _layouts(n, nlimbs, batch)[lay_name]
    a, b = _dense_inputs(oracle, n, qs, batch, 4100), _dense_inputs(oracle, n, qs, batch, 4200)
    ctxs, fa, fb, prod = _expected(oracle, n, qs, roots, a, b)
    lib.set_rns_launch(plans, loop)
    da, db, dc = (lib.DeviceBuffer(lay.words) for _ in range(3))
    try:
        # transforms
        da.upload(lay.scatter(a))
        lib.rns_fwd(plans, da.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(da.download()), fa), "forward"
        lib.rns_inv(plans, da.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(da.download()), a), "inverse"
        # coefficient-domain product, c its own buffer and c aliasing a
        db.upload(lay.scatter(b)), dc.upload(np.full(lay.words, GUARD, dtype=np.uint64))
        lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(dc.download()), prod), "product"
        da.upload(lay.scatter(a)), db.upload(lay.scatter(b))
        lib.rns_negacyclic_mul(plans, da.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(da.download()), prod), "product, c aliasing a"
        # NTT-domain products: c = inv(a^ b^ + b^ a^') with canonical words, then with a broadcast key [limb][N]
        da.upload(lay.scatter(fa)), db.upload(lay.scatter(fb)), dc.upload(np.full(lay.words, GUARD, dtype=np.uint64))
        lib.rns_inv_dot(plans, dc.ptr, [da.ptr, db.ptr], [db.ptr, da.ptr], batch, layout=lay.strides)
        pw = np.stack([oracle.pointwise(fa[l].reshape(-1), fb[l].reshape(-1), q).reshape(a[l].shape) for l, q in enumerate(qs)])
        pw2 = np.stack([(pw[l] + pw[l]) % np.uint64(q) for l, q in enumerate(qs)])                  # (q < 2^61: no overflow)
        exp2 = np.stack([cx.inv(pw2[l].reshape(-1)).reshape(a[l].shape) for l, cx in enumerate(ctxs)])
        assert np.array_equal(lay.gather(dc.download()), exp2), "inner product, k = 2"
        key = np.stack([fb[l, 0] for l in range(nlimbs)])                      # [limb][N]
        dk = lib.DeviceBuffer(key.size).upload(key.reshape(-1))
        lib.rns_inv_dot(plans, dc.ptr, [da.ptr], [dk.ptr], batch, flags=lib.MUL_B_BROADCAST, layout=lay.strides)
        expk = np.stack([cx.inv(oracle.pointwise(fa[l].reshape(-1), np.tile(key[l], batch), q)).reshape(a[l].shape)
                         for l, (cx, q) in enumerate(zip(ctxs, qs))])
        assert np.array_equal(lay.gather(dc.download()), expk), "product with a broadcast key"
        # c = inv(fwd(a) b^)
        da.upload(lay.scatter(a)), db.upload(lay.scatter(fb))
        lib.rns_mul_transformed(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(dc.download()), prod), "fwd(a) times a transformed operand, inverse"
        # c^ = fwd(a) b^, then c^ += fwd(a) b^
        da.upload(lay.scatter(a))
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
        assert np.array_equal(lay.gather(dc.download()), pw), "forward transform times a transformed operand"
        da.upload(lay.scatter(a))
        lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch, flags=lib.MUL_ACCUMULATE, layout=lay.strides)
        assert np.array_equal(lay.gather(dc.download()), pw2), "... accumulated"
        dk.free()
    finally:
        for x in (da, db, dc):
            x.free()
        for p in plans:
            p.destroy()


@pytest.mark.parametrize("loop", ["0", "1"])
@pytest.mark.parametrize("logn,nlimbs,batch,bits", CASES)
def test_rns_entry_points_in_batch_major_layout(lib, oracle, monkeypatch, logn, nlimbs, batch, bits, loop):
    """every ntt_rns_*_strided entry point on [batch][prime][N] operands, one launch over the limbs (loop 0) and limb by limb (1)"""
    _run_entry_points(lib, oracle, monkeypatch, logn, nlimbs, batch, bits, "batch_major", loop)


@pytest.mark.parametrize("lay_name", ["batch_major_padded", "limb_major_padded"])
@pytest.mark.parametrize("logn,nlimbs,batch,bits", [(8, 3, 5, 50), (12, 4, 3, 50), (14, 3, 2, 57), (15, 3, 2, 50)])
def test_rns_entry_points_in_padded_layouts(lib, oracle, monkeypatch, logn, nlimbs, batch, bits, lay_name):
    _run_entry_points(lib, oracle, monkeypatch, logn, nlimbs, batch, bits, lay_name, "0")


@pytest.mark.parametrize("arith", ["u64", "r4", "generic"])
def test_strided_layout_plans_without_the_fused_kernels(lib, oracle, monkeypatch, arith):
    """the reference's butterflies (NTT_ARITH_U64), its radix-4 formulation and column-pass-only plans: per-limb launches and the
    unfused pointwise products honour the layout too"""
    n, nlimbs, batch = 1 << 10, 3, 3
    qs, roots = _primes(lib, n, nlimbs, 50)
    ar = {"u64": lib.ARITH_U64, "r4": lib.ARITH_U64_R4, "generic": lib.ARITH_AUTO}[arith]
    plans = [lib.Plan(n, q, w, arith=ar) for q, w in zip(qs, roots)]
    if arith == "generic":
        for p in plans:
            p.set_generic(True)
    lay = _layouts(n, nlimbs, batch)["batch_major"]
    a, b = _dense_inputs(oracle, n, qs, batch, 5100), _dense_inputs(oracle, n, qs, batch, 5200)
    ctxs, fa, fb, prod = _expected(oracle, n, qs, roots, a, b)
    da, db, dc = (lib.DeviceBuffer(lay.words) for _ in range(3))
    da.upload(lay.scatter(a)), db.upload(lay.scatter(b)), dc.upload(np.full(lay.words, GUARD, dtype=np.uint64))
    lib.rns_fwd(plans, da.ptr, batch, layout=lay.strides)
    assert np.array_equal(lay.gather(da.download()), fa)
    lib.rns_inv(plans, da.ptr, batch, layout=lay.strides)
    assert np.array_equal(lay.gather(da.download()), a)
    lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
    assert np.array_equal(lay.gather(dc.download()), prod)
    da.upload(lay.scatter(fa)), db.upload(lay.scatter(fb))
    lib.rns_inv_dot(plans, dc.ptr, [da.ptr], [db.ptr], batch, layout=lay.strides)
    assert np.array_equal(lay.gather(dc.download()), prod)
    da.upload(lay.scatter(a))
    lib.rns_fwd_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
    pw = np.stack([oracle.pointwise(fa[l].reshape(-1), fb[l].reshape(-1), q).reshape(a[l].shape) for l, q in enumerate(qs)])
    assert np.array_equal(lay.gather(dc.download()), pw)


@pytest.mark.parametrize("m,nl,batch,bits", [(15, 3, 90, 50), (16, 2, 160, 52), (17, 4, 33, 50), (16, 3, 180, 57)])
def test_xcd_local_launches_in_batch_major_layout(lib, oracle, m, nl, batch, bits):
    """large batches of N >= 2^15: the XCD-local one-launch kernels (queue entry = limb and polynomial) on [batch][prime][N]
    operands; every polynomial of the forward transform and of the product against the limb-major call on the same data,
    samples against the oracle"""
    n = 1 << m
    qs, roots = _primes(lib, n, nl, bits)
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    for p in plans:
        p.set_option(lib.OPT_XCD_LOCAL, 1)
    a, b = _dense_inputs(oracle, n, qs, batch, 6100), _dense_inputs(oracle, n, qs, batch, 6200)
    lay = _layouts(n, nl, batch)["batch_major"]
    da, db, dc = (lib.DeviceBuffer(lay.words) for _ in range(3))
    ea, eb, ec = (lib.DeviceBuffer(nl * batch * n) for _ in range(3))
    da.upload(lay.scatter(a)), ea.upload(a.reshape(-1))
    lib.rns_fwd(plans, da.ptr, batch, layout=lay.strides)
    lib.rns_fwd(plans, ea.ptr, batch)
    f_bm, f_lm = lay.gather(da.download()), ea.download().reshape(nl, batch, n)
    assert np.array_equal(f_bm, f_lm)
    for l in (0, nl - 1):
        cx = oracle.ctx(n, qs[l], roots[l])
        for p in (0, batch // 2, batch - 1):
            assert np.array_equal(f_bm[l, p], cx.fwd(a[l, p].copy())), (l, p)
    lib.rns_inv(plans, da.ptr, batch, layout=lay.strides)
    assert np.array_equal(lay.gather(da.download()), a)
    db.upload(lay.scatter(b)), ea.upload(a.reshape(-1)), eb.upload(b.reshape(-1))
    lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, batch, layout=lay.strides)
    lib.rns_negacyclic_mul(plans, ec.ptr, ea.ptr, eb.ptr, batch)
    p_bm, p_lm = lay.gather(dc.download()), ec.download().reshape(nl, batch, n)
    assert np.array_equal(p_bm, p_lm)
    cx = oracle.ctx(n, qs[1], roots[1])
    for p in (0, batch - 1):
        assert np.array_equal(p_bm[1, p], cx.inv(oracle.pointwise(cx.fwd(a[1, p].copy()), cx.fwd(b[1, p].copy()), qs[1]))), p
    for x in (da, db, dc, ea, eb, ec):
        x.free()


@pytest.mark.parametrize("m,bits", [(9, 50), (14, 51), (14, 60), (16, 50)])
def test_single_plan_strided_transform(lib, oracle, m, bits):
    """ntt_transform_batch_strided: one limb of a [batch][3][N] operand, every flag combination"""
    n, batch = 1 << m, 5
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w)
    cx = oracle.ctx(n, q, w)
    stride = 3 * n
    a = oracle.fill_uniform(batch * n, q, 7100).reshape(batch, n)
    img = np.full(batch * stride, GUARD, dtype=np.uint64)
    for p in range(batch):
        img[p * stride + n:p * stride + 2 * n] = a[p]
    d = lib.DeviceBuffer(img.size).upload(img)
    plan.transform_strided(d.ptr + 8 * n, stride, batch)
    got = d.download().reshape(batch, 3, n)
    assert (got[:, 0] == GUARD).all() and (got[:, 2] == GUARD).all()
    assert np.array_equal(got[:, 1], cx.fwd(a.reshape(-1)).reshape(batch, n))
    plan.transform_strided(d.ptr + 8 * n, stride, batch, lib.FLAG_INVERSE)
    assert np.array_equal(d.download().reshape(batch, 3, n)[:, 1], a)
    plan.transform_strided(d.ptr + 8 * n, stride, batch, lib.FLAG_LAZY_OUT)
    lz = d.download().reshape(batch, 3, n)[:, 1]
    assert int(lz.max()) < 4 * q and np.array_equal(lz % np.uint64(q), cx.fwd(a.reshape(-1)).reshape(batch, n))
    plan.transform_strided(d.ptr + 8 * n, stride, batch, lib.FLAG_INVERSE | lib.FLAG_WIDE_IN)
    assert np.array_equal(d.download().reshape(batch, 3, n)[:, 1], a)
    d.free()


def test_layouts_that_overlap_are_refused(lib, oracle):
    n, nl, batch = 1 << 8, 3, 4
    qs, roots = _primes(lib, n, nl, 50)
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    d = lib.DeviceBuffer(nl * batch * n * 2)
    for limb, poly in ((n, n), (n, 2 * n), (n - 1, nl * n), (batch * n - 1, n), (0, 0)):
        with pytest.raises(lib.NttError):
            lib.rns_fwd(plans, d.ptr, batch, layout=(limb, poly))
    with pytest.raises(lib.NttError):
        plans[0].transform_strided(d.ptr, n - 1, batch)
    # the two canonical layouts pass
    lib.rns_fwd(plans, d.ptr, batch, layout=(batch * n, n))
    lib.rns_fwd(plans, d.ptr, batch, layout=(n, nl * n))
    d.free()


@pytest.mark.parametrize("m,bits", [(15, 50), (16, 52), (15, 57)])
def test_xcd_local_launches_on_8_byte_aligned_padded_polynomials(lib, oracle, m, bits):
    """the one-launch kernels (transform, NTT-domain product, forward-side product) on polynomials that start 8 bytes into an
    allocation and lie N + 1 words apart: every 16-byte access of the row and column items is misaligned by half its width; against
    the per-pass / per-chunk launches on the same placement, every word, and samples against the oracle; the pad words stay"""
    n, batch = 1 << m, 72
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    cx = oracle.ctx(n, q, w)
    plan = lib.Plan(n, q, w)
    stride = n + 1
    words = batch * stride + 8
    a = oracle.fill_uniform(batch * n, q, 9900 + m).reshape(batch, n)
    b = oracle.fill_uniform(batch * n, q, 9950 + m).reshape(batch, n)

    def image(x):
        img = np.full(words, GUARD, dtype=np.uint64)
        for p_ in range(batch):
            img[1 + p_ * stride:1 + p_ * stride + n] = x[p_]
        return img

    def rows(img):
        return np.stack([img[1 + p_ * stride:1 + p_ * stride + n] for p_ in range(batch)])

    def pads_intact(img):
        return img[0] == GUARD and all(img[1 + p_ * stride + n] == GUARD for p_ in range(batch))

    da, db, dc = lib.DeviceBuffer(words), lib.DeviceBuffer(words), lib.DeviceBuffer(words)
    ls = (0, stride)                                       # one limb: only the polynomial stride matters
    res = {}
    for form in (0, 1):
        plan.set_option(lib.OPT_XCD_LOCAL, form)
        da.upload(image(a))
        plan.transform_strided(da.ptr + 8, stride, batch)
        f = da.download()
        plan.transform_strided(da.ptr + 8, stride, batch, flags=lib.FLAG_INVERSE)
        back = da.download()
        da.upload(image(f_rows := rows(f))), db.upload(image(b)), dc.upload(image(np.zeros_like(a)))
        lib.rns_inv_dot([plan], dc.ptr + 8, [da.ptr + 8], [db.ptr + 8], batch, layout=(batch * stride, stride))
        dot = dc.download()
        da.upload(image(a)), dc.upload(image(np.zeros_like(a)))
        lib.rns_fwd_mul([plan], dc.ptr + 8, da.ptr + 8, db.ptr + 8, batch, layout=(batch * stride, stride))
        mul = dc.download()
        res[form] = (f, back, dot, mul)
        assert all(pads_intact(x) for x in (f, back, dot, mul)), form
        assert np.array_equal(rows(back), a), form
    for i in range(4):
        assert np.array_equal(res[0][i], res[1][i]), i
    for p_ in (0, batch - 1):
        fa = cx.fwd(a[p_].copy())
        assert np.array_equal(rows(res[1][0])[p_], fa)
        assert np.array_equal(rows(res[1][2])[p_], cx.inv(oracle.pointwise(fa, b[p_], q)))
        assert np.array_equal(rows(res[1][3])[p_], oracle.pointwise(fa, b[p_], q))
    for x in (da, db, dc):
        x.free()
    plan.destroy()


def test_empty_inputs_and_bad_arguments_of_the_round_5_entry_points(lib, oracle):
    """empty batches and counts are no-ops (no launch, NTT_OK); null and misaligned pointers, unknown flags, strides below N and a
    reserve during capture-less misuse are refused with an error, never a crash"""
    n, nl = 1 << 10, 2
    qs, roots = _primes(lib, n, nl, 50)
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    d = lib.DeviceBuffer(nl * 4 * n)
    lib.rns_fwd(plans, d.ptr, 0, layout=(n, nl * n))                       # empty batch in a strided layout
    lib.rns_inv_dot(plans, d.ptr, [d.ptr], [d.ptr], 0, layout=(n, nl * n))
    lib.rns_fwd_mul(plans, d.ptr, d.ptr, d.ptr, 0, layout=(n, nl * n))
    plans[0].transform_strided(d.ptr, 2 * n, 0)
    plans[0].transform_ptrs([])                                            # empty pointer batch
    lib.rns_transform_ptrs(plans, [], n)
    plans[0].reserve(0)                                                    # nothing to reserve
    with pytest.raises(lib.NttError):
        plans[0].transform_ptrs([0])                                       # null pointer
    with pytest.raises(lib.NttError):
        plans[0].transform_ptrs([d.ptr + 4])                               # not 8-byte aligned
    with pytest.raises(lib.NttError):
        plans[0].transform_ptrs([d.ptr], flags=1 << 20)                    # unknown flag
    with pytest.raises(lib.NttError):
        lib.rns_transform_ptrs(plans, [d.ptr], n - 1)                      # limb stride below N
    with pytest.raises(lib.NttError):
        lib.rns_transform_ptrs(plans, [d.ptr], n, flags=lib.FLAG_LAZY_OUT)  # RNS pointer batches take NTT_FLAG_INVERSE only
    with pytest.raises(lib.NttError):
        plans[0].get_option(999)
    with pytest.raises(lib.NttError):
        plans[0].set_option(lib.OPT_CTL_ALLOCATIONS, 1)                    # read-only
    assert plans[0].get_option(lib.OPT_BLOCK_OVERSUB) == 0 and plans[0].get_option(lib.OPT_MAX_BATCH_HINT) == 0
    plans[0].set_option(lib.OPT_BLOCK_OVERSUB, 3)
    assert plans[0].get_option(lib.OPT_BLOCK_OVERSUB) == 3
    d.free()
    for p in plans:
        p.destroy()


@pytest.mark.parametrize("m,bits", [(8, 50), (12, 51), (14, 60), (15, 50)])
def test_pointer_batch_of_polynomials(lib, oracle, m, bits):
    """ntt_transform_ptrs: one device pointer per polynomial (the reference's fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...) form,
    include/ntt_reference.h:44-49, for any number of device-resident polynomials): three equally spaced polynomials of one
    buffer (one strided launch), two at irregular places -- one of them only 8-byte aligned, as the reference's unaligned bench
    passes its arrays (tests/bench.c:160-186) --, two in a second allocation; listed in scrambled order.  Every polynomial
    against the oracle, every other word untouched; forward, lazy forward, inverse."""
    n = 1 << m
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    plan, cx = lib.Plan(n, q, w), oracle.ctx(n, q, w)
    words1, words2 = 12 * n + 64, 3 * n
    offs1 = [0, 3 * n, 6 * n, 8 * n + 16, 10 * n + 33]        # (the last one: odd word offset)
    offs2 = [8, n + 24]
    polys = [oracle.fill_uniform(n, q, 9000 + i) for i in range(7)]
    img1, img2 = np.full(words1, GUARD, dtype=np.uint64), np.full(words2, GUARD, dtype=np.uint64)
    for o, a in zip(offs1, polys[:5]):
        img1[o:o + n] = a
    for o, a in zip(offs2, polys[5:]):
        img2[o:o + n] = a
    d1, d2 = lib.DeviceBuffer(words1).upload(img1), lib.DeviceBuffer(words2).upload(img2)
    ptrs = [d1.ptr + 8 * o for o in offs1] + [d2.ptr + 8 * o for o in offs2]
    order = [3, 6, 0, 2, 5, 1, 4]

    def gather():
        g1, g2 = d1.download(), d2.download()
        m1, m2 = np.ones(words1, dtype=bool), np.ones(words2, dtype=bool)
        out = []
        for o in offs1:
            out.append(g1[o:o + n]); m1[o:o + n] = False
        for o in offs2:
            out.append(g2[o:o + n]); m2[o:o + n] = False
        assert (g1[m1] == GUARD).all() and (g2[m2] == GUARD).all(), "words outside the listed polynomials were written"
        return out

    plan.transform_ptrs([ptrs[i] for i in order])
    for i, got in enumerate(gather()):
        assert np.array_equal(got, cx.fwd(polys[i].copy())), i
    plan.transform_ptrs([ptrs[i] for i in order], lib.FLAG_INVERSE)
    for i, got in enumerate(gather()):
        assert np.array_equal(got, polys[i]), i
    plan.transform_ptrs(ptrs, lib.FLAG_LAZY_OUT)
    for i, got in enumerate(gather()):
        assert int(got.max()) < 4 * q and np.array_equal(got % np.uint64(q), cx.fwd(polys[i].copy())), i
    # overlapping polynomials, a pointer listed twice, a misaligned pointer: refused
    for bad in ([ptrs[0], ptrs[0] + 8 * (n - 1)], [ptrs[1], ptrs[1]], [ptrs[0] + 4]):
        with pytest.raises(lib.NttError):
            plan.transform_ptrs(bad)
    d1.free(), d2.free()


@pytest.mark.parametrize("m,nl,bits", [(12, 3, 50), (14, 4, 57), (16, 2, 50)])
def test_pointer_batch_of_rns_polynomials(lib, oracle, m, nl, bits):
    """ntt_rns_transform_ptrs: separately allocated RNS polynomials ([limb][N] each, the unit an FHE library allocates) as one
    call: five out of a pool at equal spacing (one launch over limbs and polynomials), one by itself"""
    n = 1 << m
    qs, roots = _primes(lib, n, nl, bits)
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, roots)]
    ctxs = [oracle.ctx(n, q, w) for q, w in zip(qs, roots)]
    span, gap = nl * n, nl * n + 128
    pool = lib.DeviceBuffer(5 * gap)
    lone = lib.DeviceBuffer(span)
    data = [np.stack([oracle.fill_uniform(n, q, 9500 + 10 * i + l) for l, q in enumerate(qs)]) for i in range(6)]
    img = np.full(5 * gap, GUARD, dtype=np.uint64)
    for i in range(5):
        img[i * gap:i * gap + span] = data[i].reshape(-1)
    pool.upload(img), lone.upload(data[5].reshape(-1))
    ptrs = [pool.ptr + 8 * i * gap for i in range(5)] + [lone.ptr]
    lib.rns_transform_ptrs(plans, ptrs[::-1], n)
    got = pool.download()
    for i in range(5):
        for l in range(nl):
            assert np.array_equal(got[i * gap + l * n:i * gap + (l + 1) * n], ctxs[l].fwd(data[i][l].copy())), (i, l)
        assert (got[i * gap + span:(i + 1) * gap] == GUARD).all()
    gl = lone.download()
    for l in range(nl):
        assert np.array_equal(gl[l * n:(l + 1) * n], ctxs[l].fwd(data[5][l].copy())), l
    lib.rns_transform_ptrs(plans, ptrs, n, lib.FLAG_INVERSE)
    assert np.array_equal(pool.download(), img) and np.array_equal(lone.download(), data[5].reshape(-1))
    with pytest.raises(lib.NttError):
        lib.rns_transform_ptrs(plans, [ptrs[0], ptrs[0] + 8 * n], n)      # two RNS polynomials that overlap
    pool.free(), lone.free()
    # pointers INTO a [limb][batch][N] slab (limb stride = batch * N): the polynomials interleave without overlapping; all of them
    # (one progression = one limb-major launch), then an irregular subset (polynomial by polynomial), against the slab call
    batch = 7
    slab = np.stack([oracle.fill_uniform(batch * n, q, 9700 + l) for l, q in enumerate(qs)])           # [limb][batch * n]
    d = lib.DeviceBuffer(nl * batch * n).upload(slab.reshape(-1))
    lib.rns_transform_ptrs(plans, [d.ptr + 8 * i * n for i in (3, 0, 6, 1, 5, 2, 4)], batch * n)
    got = d.download().reshape(nl, batch * n)
    for l in (0, nl - 1):
        assert np.array_equal(got[l], ctxs[l].fwd(slab[l].copy())), l
    d.upload(slab.reshape(-1))
    lib.rns_transform_ptrs(plans, [d.ptr + 8 * i * n for i in (0, 1, 3, 6)], batch * n)
    got = d.download().reshape(nl, batch, n)
    for l in range(nl):
        exp = slab[l].reshape(batch, n).copy()
        for i in (0, 1, 3, 6):
            exp[i] = ctxs[l].fwd(exp[i].copy())
        assert np.array_equal(got[l], exp), l
    with pytest.raises(lib.NttError):
        lib.rns_transform_ptrs(plans, [d.ptr, d.ptr + 8 * (n // 2)], batch * n)          # limb images that do overlap
    d.free()
    # a spacing that is neither limb-major nor polynomial-major but still disjoint (two limbs 3N apart, polynomials 2N apart:
    # images {0, 3N}, {2N, 5N}, {4N, 7N}): served polynomial by polynomial
    two = plans[:2]
    words = np.full(8 * n, GUARD, dtype=np.uint64)
    polys = [[oracle.fill_uniform(n, qs[l], 9800 + 10 * i + l) for l in range(2)] for i in range(3)]
    for i in range(3):
        for l in range(2):
            words[(2 * i + 3 * l) * n:(2 * i + 3 * l + 1) * n] = polys[i][l]
    d = lib.DeviceBuffer(8 * n).upload(words)
    lib.rns_transform_ptrs(two, [d.ptr + 8 * 2 * i * n for i in range(3)], 3 * n)
    got = d.download()
    for i in range(3):
        for l in range(2):
            assert np.array_equal(got[(2 * i + 3 * l) * n:(2 * i + 3 * l + 1) * n], ctxs[l].fwd(polys[i][l].copy())), (i, l)
    assert (got[n:2 * n] == GUARD).all() and (got[6 * n:7 * n] == GUARD).all()
    d.free()


@pytest.mark.parametrize("oversub", [1, 3, 64])
def test_block_oversub_settings_are_bit_exact(lib, oracle, oversub):
    """NTT_OPT_BLOCK_OVERSUB (workgroups launched per resident slot of the persistent block kernels; round 6 also the table-sharing
    small-block kernels) changes which workgroup takes which block, never a result: forward, inverse, the product and the NTT-domain
    product at the sizes whose kernels read the option, with batches that are not multiples of any grid (review r05: the option was
    only set and read back by a test, and soaked)"""
    for m, batch in ((10, 1000), (11, 777), (12, 1029), (13, 515), (14, 301)):
        n = 1 << m
        q = lib.find_prime(50, n, 0)
        w = lib.min_root(q, n)
        plan, cx = lib.Plan(n, q, w), oracle.ctx(n, q, w)
        plan.set_option(lib.OPT_BLOCK_OVERSUB, oversub)
        a = oracle.fill_uniform(batch * n, q, 31 + m)
        b = oracle.fill_uniform(batch * n, q, 32 + m)
        da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
        plan.fwd(da.ptr, batch)
        fa = da.download()
        for p in (0, batch // 2, batch - 1):
            assert np.array_equal(fa[p * n:(p + 1) * n], cx.fwd(a[p * n:(p + 1) * n].copy())), (m, p)
        plan.inv(da.ptr, batch)
        assert np.array_equal(da.download(), a), m
        plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch)
        got = dc.download()
        for p in (0, batch - 1):
            sl = slice(p * n, (p + 1) * n)
            assert np.array_equal(got[sl], cx.inv(oracle.pointwise(cx.fwd(a[sl].copy()), cx.fwd(b[sl].copy()), q))), (m, p)
        da.upload(fa)
        plan.fwd(db.upload(b).ptr, batch)
        plan.inv_product(dc.ptr, da.ptr, db.ptr, batch)
        assert np.array_equal(dc.download(), got), m
        for x in (da, db, dc):
            x.free()
        plan.destroy()


def test_layout_check_does_not_wrap(lib):
    """(batch - 1) * stride is formed in 128 bits: an absurd batch with a legal stride is an overlap, not a wrapped 'fits'"""
    n = 1 << 10
    qs = [lib.find_prime(50, n, k) for k in range(2)]
    plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
    d = lib.DeviceBuffer(4 * n)
    with pytest.raises(lib.NttError):
        # limb stride 2^40 words, polynomial stride 2^40 / 2: (batch - 1) * poly wraps 64 bits for batch = 2^25 + 1
        lib.rns_fwd(plans, d.ptr, (1 << 25) + 1, layout=(1 << 40, 1 << 39))
    d.free()
    for p in plans:
        p.destroy()
