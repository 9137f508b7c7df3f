"""Pointer batches placed ANYWHERE, in one launch (round 6; review r05 item 2): the reference's own batch form is one array per
polynomial -- fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...), include/ntt_reference.h:44-49, src/ntt_reference.c:71-91 --; here `count`
device-resident polynomials at irregular, shuffled places go through ntt_transform_ptrs (host array: sorted, overlap-checked,
uploaded) and ntt_transform_dev_ptrs (device array: as given, capturable), the kernels reading every polynomial's address from a
device table (csrc/ntt_core.h poly_offset).  Every polynomial against the oracle, every word between them untouched."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = np.uint64(0xA5A5A5A5A5A5A5A5)


def _scatter(rng, count, n, slack):
    """`count` non-overlapping polynomial starts (word offsets, 8-byte granularity, odd offsets included) in random order, with
    random gaps of 1..slack words between neighbours: no two differences alike, so the sorted pointers hold no progression"""
    gaps = rng.integers(1, slack, size=count)
    gaps = gaps + np.arange(count) % 7          # (make equal neighbours unlikely even for small slack)
    starts = np.cumsum(gaps + n) - n
    order = rng.permutation(count)
    return [int(starts[i]) for i in order], int(starts[-1] + n + 8)


def _place(lib, oracle, n, q, offs, words, seed):
    polys = oracle.fill_uniform(len(offs) * n, q, seed).reshape(len(offs), n)
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    for o, a in zip(offs, polys):
        img[o:o + n] = a
        mask[o:o + n] = False
    return polys, img, mask, lib.DeviceBuffer(words).upload(img)


def _check(d, offs, n, expect, mask, what):
    got = d.download()
    assert (got[mask] == GUARD).all(), "%s: words outside the listed polynomials were written" % what
    for i, o in enumerate(offs):
        assert np.array_equal(got[o:o + n], expect(i)), (what, i)


CASES = [  # (m, bits, arith, count): every block size, the column-only sizes, multi-pass sizes, every policy
    (3, 30, "auto", 40), (6, 30, "auto", 300), (8, 50, "auto", 300), (10, 50, "auto", 200), (11, 50, "auto", 130), (12, 50, "auto", 100),
    (13, 50, "auto", 70), (14, 51, "auto", 40), (15, 50, "auto", 24), (16, 50, "auto", 12), (17, 50, "auto", 6),
    (12, 52, "auto", 50), (14, 52, "auto", 20), (16, 52, "auto", 10),        # FP64, 2^51 < q < 2^52
    (12, 60, "auto", 50), (14, 58, "auto", 20), (16, 60, "auto", 10),        # the wide integer policy
    (12, 50, "u64", 50), (14, 60, "u64", 20), (15, 50, "u64", 10),           # the reference's butterflies
    (12, 50, "r4", 50), (14, 50, "r4", 20), (15, 50, "r4", 10),              # its radix-4 formulation
]


@pytest.mark.parametrize("m,bits,arith,count", CASES)
def test_shuffled_pointer_batch_in_one_launch(lib, oracle, m, bits, arith, count):
    n = 1 << m
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w, arith={"auto": lib.ARITH_AUTO, "u64": lib.ARITH_U64, "r4": lib.ARITH_U64_R4}[arith])
    cx = oracle.ctx(n, q, w)
    rng = np.random.default_rng(1000 * m + bits)
    offs, words = _scatter(rng, count, n, 40)
    polys, img, mask, d = _place(lib, oracle, n, q, offs, words, 31 * m + bits)
    ptrs = [d.ptr + 8 * o for o in offs]
    fwd = [cx.fwd(a.copy()) for a in polys]
    # host array: sorted, checked, uploaded, ONE launch chain
    plan.transform_ptrs(ptrs)
    _check(d, offs, n, lambda i: fwd[i], mask, "forward")
    plan.transform_ptrs(ptrs, lib.FLAG_INVERSE)
    _check(d, offs, n, lambda i: polys[i], mask, "inverse")
    # device array, in the caller's (shuffled) order, no copy
    tab = lib.DeviceBuffer(count).upload(np.array(ptrs, dtype=np.uint64))
    plan.transform_dev_ptrs(tab.ptr, count)
    _check(d, offs, n, lambda i: fwd[i], mask, "forward, device table")
    plan.transform_dev_ptrs(tab.ptr, count, lib.FLAG_INVERSE)
    _check(d, offs, n, lambda i: polys[i], mask, "inverse, device table")
    # lazy outputs and lazy inputs through the same table
    plan.transform_dev_ptrs(tab.ptr, count, lib.FLAG_LAZY_OUT)
    got = d.download()
    bound = (8 if arith == "r4" else 4) * q
    for i, o in enumerate(offs):
        assert int(got[o:o + n].max()) < bound and np.array_equal(got[o:o + n] % np.uint64(q), fwd[i]), i
    plan.transform_dev_ptrs(tab.ptr, count, lib.FLAG_INVERSE | lib.FLAG_WIDE_IN)
    _check(d, offs, n, lambda i: polys[i], mask, "inverse of lazy words")
    tab.free(), d.free(), plan.destroy()


@pytest.mark.parametrize("m,bits,count", [(15, 50, 96), (16, 50, 80), (17, 50, 64), (16, 52, 72), (16, 60, 72)])
def test_pointer_batch_through_the_xcd_local_launch(lib, oracle, m, bits, count):
    """N = 2^15..2^17 with the one-launch XCD-local kernels forced (team_kernel: the queues hand out polynomials by INDEX, the item
    reads the address from the table), forward and inverse, against the per-pass launches over the same table (chunked: the table
    pointer advances with the chunk) and the oracle on sampled polynomials"""
    n = 1 << m
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w)
    cx = oracle.ctx(n, q, w)
    rng = np.random.default_rng(7 * m + bits)
    offs, words = _scatter(rng, count, n, 300)
    polys, img, mask, d = _place(lib, oracle, n, q, offs, words, 5 * m)
    ptrs = [d.ptr + 8 * o for o in offs]
    tab = lib.DeviceBuffer(count).upload(np.array(ptrs, dtype=np.uint64))
    results = {}
    for mode in (1, 0):
        plan.set_option(lib.OPT_XCD_LOCAL, mode)
        plan.set_option(lib.OPT_CHUNK_MIB, 8 if mode == 0 else 256)       # several chunks per call on the per-pass route
        d.upload(img)
        plan.transform_dev_ptrs(tab.ptr, count)
        results[mode] = d.download()
        assert (results[mode][mask] == GUARD).all()
        plan.transform_ptrs(ptrs, lib.FLAG_INVERSE)
        _check(d, offs, n, lambda i: polys[i], mask, "round trip, xcd_local=%d" % mode)
    assert np.array_equal(results[0], results[1])
    for i in (0, count // 2, count - 1):
        assert np.array_equal(results[1][offs[i]:offs[i] + n], cx.fwd(polys[i].copy())), i
    tab.free(), d.free(), plan.destroy()


@pytest.mark.parametrize("m,nl,bits,count,launch", [(12, 3, 50, 40, None), (12, 3, 50, 40, "1"), (14, 4, 50, 9, "0"), (14, 4, 57, 30, None),
                                                   (16, 2, 50, 70, None), (16, 3, 50, 5, "0"), (13, 5, 50, 12, "0")])
def test_shuffled_rns_pointer_batch(lib, oracle, m, nl, bits, count, launch):
    """RNS polynomials held one by one ([limb][N] each: what an FHE library allocates) at shuffled places: ntt_rns_transform_ptrs and
    ntt_rns_transform_dev_ptrs -- one launch over the limbs (the MULTI kernels: the limb's offset is added to the table's address),
    one launch chain per limb, or the XCD-local launch with the limb in the queue entry, whichever the library takes"""
    n = 1 << m
    qs = [lib.find_prime(bits, n, k) for k in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    lib.set_rns_launch(plans, launch)
    ctx = [oracle.ctx(n, q, w) for q, w in zip(qs, ws)]
    rng = np.random.default_rng(11 * m + nl)
    offs, words = _scatter(rng, count, nl * n, 50)
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    polys = []
    for i, o in enumerate(offs):
        limbs = [oracle.fill_uniform(n, q, 900 + 17 * i + l) for l, q in enumerate(qs)]
        polys.append(limbs)
        for l, a in enumerate(limbs):
            img[o + l * n:o + (l + 1) * n] = a
        mask[o:o + nl * n] = False
    d = lib.DeviceBuffer(words).upload(img)
    ptrs = [d.ptr + 8 * o for o in offs]

    def check(expect, what):
        got = d.download()
        assert (got[mask] == GUARD).all(), what
        for i, o in enumerate(offs):
            for l in range(nl):
                assert np.array_equal(got[o + l * n:o + (l + 1) * n], expect(i, l)), (what, i, l)
    lib.rns_transform_ptrs(plans, ptrs, n)
    check(lambda i, l: ctx[l].fwd(polys[i][l].copy()), "forward")
    tab = lib.DeviceBuffer(count).upload(np.array(ptrs, dtype=np.uint64))
    lib.rns_transform_dev_ptrs(plans, tab.ptr, count, n, lib.FLAG_INVERSE)
    check(lambda i, l: polys[i][l], "inverse, device table")
    tab.free(), d.free()
    for p in plans:
        p.destroy()


def test_device_pointer_table_captured_in_a_hip_graph():
    """ntt_transform_dev_ptrs allocates nothing and copies nothing: a forward transform over a shuffled pointer batch captured into
    a HIP graph (torch.cuda.CUDAGraph; a process of its own: torch has to be imported before the library), replayed on fresh data"""
    import subprocess
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, count = 1 << 14, 24
q = lib.find_prime(51, n, 0)
w = lib.min_root(q, n)
plan, cx = lib.Plan(n, q, w), orc.ctx(n, q, w)
rng = np.random.default_rng(5)
gaps = rng.integers(1, 64, size=count) + np.arange(count) %% 7
starts = np.cumsum(gaps + n) - n
offs = [int(starts[i]) for i in rng.permutation(count)]
words = int(starts[-1] + n + 8)
GUARD = np.uint64(0xA5A5A5A5A5A5A5A5)
buf = torch.zeros(words, dtype=torch.int64, device="cuda:0")
ptrs = np.array([buf.data_ptr() + 8 * o for o in offs], dtype=np.uint64)
tab = torch.from_numpy(ptrs.view(np.int64)).to("cuda:0")
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    plan.transform_dev_ptrs(tab.data_ptr(), count, 0, stream=st)
for seed in (1, 2, 3):
    polys = orc.fill_uniform(count * n, q, seed).reshape(count, n)
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    for o, a in zip(offs, polys):
        img[o:o + n] = a
        mask[o:o + n] = False
    buf.copy_(torch.from_numpy(img.view(np.int64)))
    g.replay()
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint64)
    assert (got[mask] == GUARD).all()
    for i in range(count):
        assert np.array_equal(got[offs[i]:offs[i] + n], cx.fwd(polys[i].copy())), (seed, i)
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "graph ok" in out.stdout, out.stderr[-3000:]


def test_products_over_device_tables_captured_in_a_hip_graph():
    """the fused kernels' table forms allocate nothing and copy nothing either: c = a * b, d^ = fwd(c) . key^ and e = inv(a^' . k0^ + b^' . k1^)
    over shuffled device tables, three launches captured into ONE HIP graph and replayed on fresh data (a process of its own: torch
    has to be imported before the library)"""
    import subprocess
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, count, ops = 1 << 13, 20, 5          # operands: a, b, c, d, e
q = lib.find_prime(50, n, 0)
w = lib.min_root(q, n)
plan, cx = lib.Plan(n, q, w), orc.ctx(n, q, w)
rng = np.random.default_rng(11)
total = ops * count
gaps = rng.integers(1, 64, size=total) + np.arange(total) %% 7
starts = np.cumsum(gaps + n) - n
order = rng.permutation(total)
offs = [[int(starts[i]) for i in order[o * count:(o + 1) * count]] for o in range(ops)]
words = int(starts[-1] + n + 8)
GUARD = np.uint64(0xA5A5A5A5A5A5A5A5)
buf = torch.zeros(words, dtype=torch.int64, device="cuda:0")
tabs = [torch.from_numpy(np.array([buf.data_ptr() + 8 * o for o in offs[k]], dtype=np.uint64).view(np.int64)).to("cuda:0") for k in range(ops)]
keys_h = orc.fill_uniform(2 * n, q, 77).reshape(2, n)
keys = torch.from_numpy(keys_h.view(np.int64).copy()).to("cuda:0")
kp = [keys.data_ptr(), keys.data_ptr() + 8 * n]
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
ta, tb, tc, td, te = (t.data_ptr() for t in tabs)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    plan.negacyclic_mul_dev_ptrs(tc, ta, tb, count, stream=st)                                   # c = a * b   (a, b left as they were)
    plan.fwd_mul_dev_ptrs(td, tc, kp[0], count, lib.MUL_B_BROADCAST, stream=st)                   # d^ = fwd(c) . key0^
    plan.inv_dot_dev_ptrs(te, [ta, tb], [kp[0], kp[1]], count, lib.MUL_B_BROADCAST, stream=st)    # e = inv(a . key0^ + b . key1^), a and b read as NTT-domain words
for seed in (1, 2):
    A = orc.fill_uniform(count * n, q, seed).reshape(count, n)
    B = orc.fill_uniform(count * n, q, 100 + seed).reshape(count, n)
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    for k in range(ops):
        for o in offs[k]:
            mask[o:o + n] = False
    for i in range(count):
        img[offs[0][i]:offs[0][i] + n] = A[i]
        img[offs[1][i]:offs[1][i] + n] = B[i]
    buf.copy_(torch.from_numpy(img.view(np.int64)))
    g.replay()
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint64)
    assert (got[mask] == GUARD).all()
    for i in range(count):
        c = cx.inv(orc.pointwise(cx.fwd(A[i].copy()), cx.fwd(B[i].copy()), q))
        assert np.array_equal(got[offs[2][i]:offs[2][i] + n], c), (seed, i, "c")
        assert np.array_equal(got[offs[3][i]:offs[3][i] + n], orc.pointwise(cx.fwd(c.copy()), keys_h[0].copy(), q)), (seed, i, "d")
        e = (orc.pointwise(A[i].copy(), keys_h[0].copy(), q) + orc.pointwise(B[i].copy(), keys_h[1].copy(), q)) %% np.uint64(q)
        assert np.array_equal(got[offs[4][i]:offs[4][i] + n], cx.inv(e)), (seed, i, "e")
        assert np.array_equal(got[offs[0][i]:offs[0][i] + n], A[i]) and np.array_equal(got[offs[1][i]:offs[1][i] + n], B[i]), (seed, i, "operands")
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "graph ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("m", [14, 16])
def test_rns_products_over_device_tables_captured_in_a_hip_graph(m):
    """the RNS twins on a few polynomials x several primes -- one launch (2^14: the fused kernels' MULTI table forms) or one chain (2^16:
    element-wise kernel and transforms over the tables) over the whole run of limbs -- allocate and copy nothing: both operands into the
    NTT domain, e = inv(a^ . b^ + b^ . a^) and c = a' * b' over shuffled device tables captured into ONE HIP graph, replayed on fresh data"""
    import subprocess
    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
m, nl, count, ops = %d, 6, 2, 5          # operands: a, b (transformed in place), e, a', b' (c lands on a')
n = 1 << m
qs = [lib.find_prime(50, n, i) for i in range(nl)]
ws = [lib.min_root(q, n) for q in qs]
plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
ctx = [orc.ctx(n, q, w) for q, w in zip(qs, ws)]
rng = np.random.default_rng(5)
span, total = nl * n, ops * count
gaps = rng.integers(1, 64, size=total) + np.arange(total) %% 7
starts = np.cumsum(gaps + span) - span
order = rng.permutation(total)
offs = [[int(starts[i]) for i in order[o * count:(o + 1) * count]] for o in range(ops)]
words = int(starts[-1] + span + 8)
GUARD = np.uint64(0xA5A5A5A5A5A5A5A5)
buf = torch.zeros(words, dtype=torch.int64, device="cuda:0")
tabs = [torch.from_numpy(np.array([buf.data_ptr() + 8 * o for o in offs[k]], dtype=np.uint64).view(np.int64)).to("cuda:0") for k in range(ops)]
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
ta, tb, te, ta2, tb2 = (t.data_ptr() for t in tabs)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    lib.rns_transform_dev_ptrs(plans, ta, count, n, 0, stream=st)
    lib.rns_transform_dev_ptrs(plans, tb, count, n, 0, stream=st)
    lib.rns_inv_dot_dev_ptrs(plans, te, [ta, tb], [tb, ta], count, n, 0, stream=st)        # e = inv(2 a^ . b^)
    lib.rns_negacyclic_mul_dev_ptrs(plans, ta2, ta2, tb2, count, n, stream=st)              # c = a' * b' on a''s table
for seed in (1, 2):
    A = [[orc.fill_uniform(n, q, 10 * seed + 100 * p + l) for l, q in enumerate(qs)] for p in range(count)]
    B = [[orc.fill_uniform(n, q, 7000 + 10 * seed + 100 * p + l) for l, q in enumerate(qs)] for p in range(count)]
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    for k in range(ops):
        for o in offs[k]:
            mask[o:o + span] = False
    for p in range(count):
        for l in range(nl):
            for k, src in ((0, A), (1, B), (3, A), (4, B)):
                img[offs[k][p] + l * n:offs[k][p] + (l + 1) * n] = src[p][l]
    buf.copy_(torch.from_numpy(img.view(np.int64)))
    g.replay()
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint64)
    assert (got[mask] == GUARD).all()
    for p in range(count):
        for l in range(nl):
            fa, fb = ctx[l].fwd(A[p][l].copy()), ctx[l].fwd(B[p][l].copy())
            prod = orc.pointwise(fa, fb, qs[l])
            assert np.array_equal(got[offs[0][p] + l * n:offs[0][p] + (l + 1) * n], fa), (seed, p, l, "a^")
            assert np.array_equal(got[offs[2][p] + l * n:offs[2][p] + (l + 1) * n], ctx[l].inv((prod + prod) %% np.uint64(qs[l]))), (seed, p, l, "e")
            assert np.array_equal(got[offs[3][p] + l * n:offs[3][p] + (l + 1) * n], ctx[l].inv(prod)), (seed, p, l, "c")
print("graph ok")
""" % (ROOT, os.path.join(ROOT, "tests"), m)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "graph ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_pointer_batch_arguments(lib, oracle):
    n = 1 << 12
    q = lib.find_prime(50, n, 0)
    plan = lib.Plan(n, q, lib.min_root(q, n))
    d = lib.DeviceBuffer(8 * n)
    ptrs = [d.ptr + 8 * o for o in (0, 3 * n + 1, 5 * n)]
    # irregular AND overlapping: refused before anything is launched
    with pytest.raises(lib.NttError):
        plan.transform_ptrs(ptrs + [d.ptr + 8 * (3 * n + 100)])
    with pytest.raises(lib.NttError):
        plan.transform_ptrs(ptrs + [ptrs[1]])
    with pytest.raises(lib.NttError):
        plan.transform_dev_ptrs(0, 3)                                  # null table
    with pytest.raises(lib.NttError):
        plan.transform_dev_ptrs(d.ptr, 3, 1 << 20)                     # unknown flag
    plan.transform_dev_ptrs(0, 0)                                      # empty batch
    # the staging buffer of a stream is reused and regrown: batches of growing and shrinking size, back to back, one stream
    rng = np.random.default_rng(3)
    cx = oracle.ctx(n, q, lib.min_root(q, n))
    for count in (3, 50, 7, 200, 1):
        offs, words = _scatter(rng, count, n, 30)
        polys, img, mask, dd = _place(lib, oracle, n, q, offs, words, count)
        plan.transform_ptrs([dd.ptr + 8 * o for o in offs])
        _check(dd, offs, n, lambda i: cx.fwd(polys[i].copy()), mask, "count %d" % count)
        dd.free()
    d.free(), plan.destroy()


def test_randomly_placed_polynomials_run_at_the_rate_of_a_slab(lib):
    """review r05 item 2, the measured bar: 4096 randomly placed 2^14-point polynomials through ntt_transform_ptrs / _dev_ptrs against
    the same 4096 polynomials as one contiguous slab (profiles/r06/pointer_batches.txt holds the measured figures: within 3 %); here
    a loose bound that separates one launch from 4096"""
    n, count = 1 << 14, 4096
    q = lib.find_prime(51, n, 0)
    plan = lib.Plan(n, q, lib.min_root(q, n))
    rng = np.random.default_rng(1)
    offs, words = _scatter(rng, count, n, 4096)
    d = lib.DeviceBuffer(words)
    lib.fill_uniform(d.ptr, words, q, 1, 0)
    ptrs = [d.ptr + 8 * o for o in offs]
    tab = lib.DeviceBuffer(count).upload(np.array(ptrs, dtype=np.uint64))
    ev0, ev1 = lib.Event(0), lib.Event(0)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        ev0.record(None)
        for _ in range(reps):
            fn()
        ev1.record(None)
        return ev1.elapsed_ms_since(ev0) / reps
    slab = timed(lambda: plan.fwd(d.ptr, count))
    dev = timed(lambda: plan.transform_dev_ptrs(tab.ptr, count))
    host = timed(lambda: plan.transform_ptrs(ptrs))
    assert dev < 1.25 * slab and host < 2.0 * slab, (slab, dev, host)
    tab.free(), d.free(), plan.destroy()


def _tables(lib, oracle, rng, n, q, count, operands, seed, unit=None):
    """`operands` sets of `count` polynomials, every polynomial at its own random place of ONE pool buffer (shuffled, irregular gaps,
    all operands interleaved): returns the pool, the host images, per operand the word offsets and the device table"""
    unit = unit or n
    offs_all, words = _scatter(rng, operands * count, unit, 40)
    img = np.full(words, GUARD, dtype=np.uint64)
    mask = np.ones(words, dtype=bool)
    polys, offs = [], []
    for o in range(operands):
        offs.append(offs_all[o * count:(o + 1) * count])
        polys.append(oracle.fill_uniform(count * n, q, seed + o).reshape(count, n))
    return offs, polys, img, mask, words


@pytest.mark.parametrize("m,bits,arith,count", [(8, 50, "auto", 24), (12, 50, "auto", 24), (14, 52, "auto", 24), (14, 58, "auto", 24), (12, 50, "r4", 24),
                                                 (16, 50, "auto", 24), (4, 30, "auto", 24),
                                                 # the fused kernels' table-reading forms, size by size (up to 2^14: ONE launch per call and limb)
                                                 (6, 40, "auto", 37), (9, 50, "auto", 301), (10, 50, "auto", 77), (11, 50, "auto", 41), (11, 52, "auto", 41),
                                                 (12, 52, "auto", 1100), (13, 50, "auto", 23), (13, 52, "auto", 600), (14, 50, "auto", 300), (13, 57, "auto", 23),
                                                 (12, 59, "auto", 33), (14, 61, "auto", 9), (11, 60, "auto", 19), (15, 50, "auto", 12),
                                                 # 2^15: fwd(a) . b^ in ONE pass with the tables read at its output (batches that give every second CU a polynomial)
                                                 (15, 50, "auto", 160), (15, 52, "auto", 131),
                                                 # 2^15..2^17 from 64 polynomials on: the one-launch (XCD-local) product kernels read the tables
                                                 (15, 50, "auto", 70), (16, 50, "auto", 80), (17, 50, "auto", 66), (16, 52, "auto", 65), (16, 60, "auto", 72)])
def test_products_over_pointer_tables(lib, oracle, m, bits, arith, count):
    """round 6: the NTT-domain products and the product chain over SEPARATELY HELD operands -- every operand a device table of
    pointers into a pool where all polynomials of all operands lie shuffled: c = inv(sum_i a_i^ . b_i^) for k = 1, 3 (canonical, lazy,
    broadcast key), c^ = fwd(a) . b^ and c^ += fwd(a) . key^, c = a * b (c on a table of its own, on a's, squaring); every polynomial
    against the oracle, every word between the polynomials untouched.  Counts above the resident workgroups make the persistent
    loops take several blocks each (the table entries of the NEXT block are read ahead); odd counts leave a workgroup's second
    block idle where two share one."""
    n = 1 << m
    q = lib.find_prime(bits, n, 0)
    w = lib.min_root(q, n)
    plan = lib.Plan(n, q, w, arith={"auto": lib.ARITH_AUTO, "r4": lib.ARITH_U64_R4}[arith])
    cx = oracle.ctx(n, q, w)
    rng = np.random.default_rng(m * 100 + bits)
    K = 3
    if count % 2 == 1:      # a handful of workgroups: every one walks through many blocks
        plan.set_option(lib.OPT_MAX_GRID, 5)
    offs, polys, img, mask, words = _tables(lib, oracle, rng, n, q, count, 2 * K + 1, 500 + m)
    # operands 0..K-1: a_i, K..2K-1: b_i, 2K: c
    A = [polys[i] for i in range(K)]
    B = [polys[K + i] for i in range(K)]
    # (large counts: every polynomial goes through the GPU, a sample of them -- the first and last ones, a workgroup's second and
    # later blocks among them -- through the oracle)
    sample = list(range(count)) if count <= 48 else sorted(set(list(range(6)) + list(range(count - 6, count)) + [int(x) for x in rng.integers(0, count, 12)]))
    if count <= 48:
        Ah = [np.stack([cx.fwd(x.copy()) for x in a]) for a in A]
        Bh = [np.stack([cx.fwd(x.copy()) for x in b]) for b in B]
    else:       # the GPU's own forward transform of the slabs (itself checked against the oracle on the sample)
        def gpu_fwd(x):
            t = lib.DeviceBuffer(count * n).upload(x.reshape(-1))
            plan.fwd(t.ptr, count)
            r = t.download().reshape(count, n)
            t.free()
            for p in sample[:4]:
                assert np.array_equal(r[p], cx.fwd(x[p].copy()))
            return r
        Ah, Bh = [gpu_fwd(a) for a in A], [gpu_fwd(b) for b in B]
    fused = 6 <= m <= 14 and arith == "auto"

    def place(sets):
        im = img.copy()
        mk = mask.copy()
        for o, data in sets.items():
            for off, x in zip(offs[o], data):
                im[off:off + n] = x
                mk[off:off + n] = False
        return im, mk
    d = lib.DeviceBuffer(words)
    tabs = [lib.DeviceBuffer(count).upload(np.array([d.ptr + 8 * o for o in offs[i]], dtype=np.uint64)) for i in range(2 * K + 1)]
    key = lib.DeviceBuffer(n).upload(Bh[0][0])
    for k in (1, K):
        for lazy in (False, True):
            mult = np.uint64(q) * np.uint64(3 if (lazy and q < (1 << 60)) else 0)
            im, mk = place({**{i: Ah[i] + mult for i in range(k)}, **{K + i: Bh[i] + mult for i in range(k)}})
            d.upload(im)
            plan.inv_dot_dev_ptrs(tabs[2 * K].ptr, [tabs[i].ptr for i in range(k)], [tabs[K + i].ptr for i in range(k)], count, lib.MUL_LAZY_IN if lazy else 0)
            got = d.download()
            for p in sample:
                acc = np.zeros(n, dtype=np.uint64)
                for i in range(k):
                    acc = (acc + oracle.pointwise(Ah[i][p].copy(), Bh[i][p].copy(), q)) % np.uint64(q)
                assert np.array_equal(got[offs[2 * K][p]:offs[2 * K][p] + n], cx.inv(acc)), (k, lazy, p)
            mk2 = mk.copy()
            for off in offs[2 * K]:
                mk2[off:off + n] = False
            assert (got[mk2] == GUARD).all(), (k, lazy)
    # broadcast key, k = 1, c on a's own table
    im, mk = place({0: Ah[0]})
    d.upload(im)
    plan.inv_dot_dev_ptrs(tabs[0].ptr, [tabs[0].ptr], [key.ptr], count, lib.MUL_B_BROADCAST)
    got = d.download()
    for p in sample:
        assert np.array_equal(got[offs[0][p]:offs[0][p] + n], cx.inv(oracle.pointwise(Ah[0][p].copy(), Bh[0][0].copy(), q))), p
    assert (got[mk] == GUARD).all()
    # c^ = fwd(a) . b^, then c^ += fwd(a') . key^
    im, mk = place({0: A[0], 1: A[1], K: Bh[0]})
    d.upload(im)
    plan.fwd_mul_dev_ptrs(tabs[2 * K].ptr, tabs[0].ptr, tabs[K].ptr, count)
    plan.fwd_mul_dev_ptrs(tabs[2 * K].ptr, tabs[1].ptr, key.ptr, count, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST)
    got = d.download()
    for p in sample:
        exp = (oracle.pointwise(Ah[0][p].copy(), Bh[0][p].copy(), q) + oracle.pointwise(Ah[1][p].copy(), Bh[0][0].copy(), q)) % np.uint64(q)
        assert np.array_equal(got[offs[2 * K][p]:offs[2 * K][p] + n], exp), p
    mk2 = mk.copy()
    for off in offs[2 * K]:
        mk2[off:off + n] = False
    assert (got[mk2] == GUARD).all()
    if fused or (m == 15 and count >= 128):   # the fused kernel leaves a as it was
        for p in sample:
            assert np.array_equal(got[offs[0][p]:offs[0][p] + n], A[0][p]) and np.array_equal(got[offs[1][p]:offs[1][p] + n], A[1][p]), p
    # c = a * b: own table, a's table, squaring
    for form in ("own", "on_a", "square"):
        im, mk = place({0: A[0], K: B[0]})
        d.upload(im)
        ct = tabs[2 * K] if form == "own" else tabs[0]
        bt = tabs[0] if form == "square" else tabs[K]
        plan.negacyclic_mul_dev_ptrs(ct.ptr, tabs[0].ptr, bt.ptr, count)
        got = d.download()
        co = offs[2 * K] if form == "own" else offs[0]
        for p in sample:
            other = Ah[0][p] if form == "square" else Bh[0][p]
            assert np.array_equal(got[co[p]:co[p] + n], cx.inv(oracle.pointwise(Ah[0][p].copy(), other.copy(), q))), (form, p)
        mk2 = mk.copy()
        for off in co:
            mk2[off:off + n] = False
        assert (got[mk2] == GUARD).all(), form
        if form == "own" and fused and m >= 8 and plan.info()["arith"] == lib.ARITH_F64:   # one launch, both operands left as they were
            for p in sample:
                assert np.array_equal(got[offs[0][p]:offs[0][p] + n], A[0][p]) and np.array_equal(got[offs[K][p]:offs[K][p] + n], B[0][p]), p
    for t in tabs:
        t.free()
    key.free(), d.free(), plan.destroy()


@pytest.mark.parametrize("m,count", [(13, 10), (16, 64)])
def test_rns_products_over_pointer_tables(lib, oracle, m, count):
    """the RNS twins: every table entry points at limb 0 of an RNS polynomial ([limb][N] each), the limbs one after the other
    (2^13: the fused kernels' table forms, one launch per limb; 2^16 x 64: the XCD-local kernels')"""
    nl = 3
    n = 1 << m
    qs = [lib.find_prime(50, n, i) for i in range(nl)]
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    ctx = [oracle.ctx(n, q, w) for q, w in zip(qs, ws)]
    rng = np.random.default_rng(9)
    offs_all, words = _scatter(rng, 3 * count, nl * n, 40)
    offs = [offs_all[o * count:(o + 1) * count] for o in range(3)]
    img = np.full(words, GUARD, dtype=np.uint64)
    a = [[oracle.fill_uniform(n, q, 10 * p + l) for l, q in enumerate(qs)] for p in range(count)]
    b = [[oracle.fill_uniform(n, q, 1000 + 10 * p + l) for l, q in enumerate(qs)] for p in range(count)]
    for p in range(count):
        for l in range(nl):
            img[offs[0][p] + l * n:offs[0][p] + (l + 1) * n] = a[p][l]
            img[offs[1][p] + l * n:offs[1][p] + (l + 1) * n] = b[p][l]
    d = lib.DeviceBuffer(words).upload(img)
    tabs = [lib.DeviceBuffer(count).upload(np.array([d.ptr + 8 * o for o in offs[i]], dtype=np.uint64)) for i in range(3)]
    lib.rns_negacyclic_mul_dev_ptrs(plans, tabs[2].ptr, tabs[0].ptr, tabs[1].ptr, count, n)
    got = d.download()
    for p in range(count):
        for l in range(nl):
            fa, fb = ctx[l].fwd(a[p][l].copy()), ctx[l].fwd(b[p][l].copy())
            assert np.array_equal(got[offs[2][p] + l * n:offs[2][p] + (l + 1) * n], ctx[l].inv(oracle.pointwise(fa, fb, qs[l]))), (p, l)
            if m <= 14:
                assert np.array_equal(got[offs[0][p] + l * n:offs[0][p] + (l + 1) * n], a[p][l]), "the fused kernel leaves a as it was"
                assert np.array_equal(got[offs[1][p] + l * n:offs[1][p] + (l + 1) * n], b[p][l]), "... and b"
    if m > 14:      # (the one-launch form above 2^14 uses a and b as scratch: put them back)
        keep = d.download()
        for p in range(count):
            for l in range(nl):
                keep[offs[0][p] + l * n:offs[0][p] + (l + 1) * n] = a[p][l]
                keep[offs[1][p] + l * n:offs[1][p] + (l + 1) * n] = b[p][l]
        d.upload(keep)
    # the operands into the NTT domain: c = inv(a^ . b^) again through the inner-product form, and c^ += fwd(c) . b^ on top of a^
    lib.rns_transform_dev_ptrs(plans, tabs[0].ptr, count, n)
    lib.rns_transform_dev_ptrs(plans, tabs[1].ptr, count, n)
    lib.rns_inv_dot_dev_ptrs(plans, tabs[2].ptr, [tabs[0].ptr], [tabs[1].ptr], count, n)
    again = d.download()
    for p in range(count):
        assert np.array_equal(again[offs[2][p]:offs[2][p] + nl * n], got[offs[2][p]:offs[2][p] + nl * n]), p
    lib.rns_fwd_mul_dev_ptrs(plans, tabs[0].ptr, tabs[2].ptr, tabs[1].ptr, count, n, lib.MUL_ACCUMULATE)
    acc = d.download()
    for p in (0, count - 1):
        for l in range(nl):
            fa, fb = ctx[l].fwd(a[p][l].copy()), ctx[l].fwd(b[p][l].copy())
            c = ctx[l].inv(oracle.pointwise(fa, fb, qs[l]))
            exp = (fa + oracle.pointwise(ctx[l].fwd(c.copy()), fb, qs[l])) % np.uint64(qs[l])
            assert np.array_equal(acc[offs[0][p] + l * n:offs[0][p] + (l + 1) * n], exp), (p, l)
    for t in tabs:
        t.free()
    d.free()
    for p in plans:
        p.destroy()


RNS_TABLE_CASES = [(8, [50] * 5, 3, 0), (11, [50] * 4, 2, 24), (12, [50] * 17, 1, 0), (13, [50, 50, 49, 48], 3, 8),
                   (14, [50] * 6, 2, 0), (14, [52] * 3, 3, 40), (12, [60, 50, 50, 52, 52, 58, 58], 2, 16),
                   (10, [59] * 5, 4, 0), (14, [57] * 3, 2, 0), (6, [40] * 4, 5, 8),
                   # above 2^14 (fewer than 64 polynomials): the chain -- element-wise launch and transforms over the tables -- per RUN
                   (15, [50] * 3, 2, 0), (16, [50] * 4, 3, 8), (16, [52, 52, 60, 60], 2, 0), (17, [50] * 2, 1, 0), (15, [58] * 3, 5, 24),
                   # ... from 64 polynomials x limbs on: the XCD-local one-launch kernels over the whole run (the limb in the queue entry)
                   (16, [50] * 4, 20, 0), (15, [50] * 3, 70, 8), (17, [50] * 2, 40, 0), (16, [52, 52, 60, 60], 33, 0), (15, [50] * 5, 13, 0)]


@pytest.mark.parametrize("m,bits,count,pad,unfused", [c + (False,) for c in RNS_TABLE_CASES] +
                         # the fused kernels switched off: the chain's launches over the run at sizes the fused kernels serve otherwise
                         [c + (True,) for c in RNS_TABLE_CASES if c[0] in (12, 14) and len(c[1]) <= 7])
def test_rns_products_over_pointer_tables_in_one_launch_over_the_limbs(lib, oracle, m, bits, count, pad, unfused):
    """round 6, DESIGN 9.2: a few separately held RNS polynomials x many primes -- the share of one limb cannot fill the chip, so the
    RNS twins of the products over device tables serve a run of compatible limbs with ONE launch of the fused kernels' table-reading
    MULTI instances (the limb an index of the grid, every polynomial's limbs limb_stride words apart behind its table entry) where
    they looped over the limbs before; above 2^14 the chain's launches (element-wise kernel, transforms over the tables) each over the run.  Forced both ways (NTT_OPT_RNS_LAUNCH 0 / 1) and left to the library: all three equal the
    oracle, limb by limb; mixed chains split into runs (a 60-bit prime in front of 50-bit ones, 52-bit ones, 17 limbs = 16 + 1);
    padded limb strides: the words between the limbs stay untouched; k = 1 and 3, canonical and lazy operands, a broadcast key
    ([limb][N]), accumulate, c on a table of its own and on a's.  unfused: the fused kernels switched off on every plan
    (NTT_OPT_FUSED_PRODUCT 0, NTT_OPT_DOT_FUSED 0) -- the chain's launches over the run at the sizes the fused kernels serve otherwise."""
    nl = len(bits)
    n = 1 << m
    seen = {}
    qs = []
    for b in bits:
        qs.append(lib.find_prime(b, n, seen.get(b, 0)))
        seen[b] = seen.get(b, 0) + 1
    ws = [lib.min_root(q, n) for q in qs]
    plans = [lib.Plan(n, q, w) for q, w in zip(qs, ws)]
    if unfused:
        for p in plans:
            p.set_option(lib.OPT_FUSED_PRODUCT, 0)
            p.set_option(lib.OPT_DOT_FUSED, 0)
    ctx = [oracle.ctx(n, q, w) for q, w in zip(qs, ws)]
    ls = n + pad                       # limb stride (words)
    span = (nl - 1) * ls + n
    K = 3
    rng = np.random.default_rng(m * 1000 + nl)
    offs_all, words = _scatter(rng, (2 * K + 1) * count, span, 40)
    offs = [offs_all[o * count:(o + 1) * count] for o in range(2 * K + 1)]     # 0..K-1: a_i, K..2K-1: b_i, 2K: c
    blank = np.full(words, GUARD, dtype=np.uint64)
    A = [[[oracle.fill_uniform(n, q, 7 + 100 * i + 10 * p + l) for l, q in enumerate(qs)] for p in range(count)] for i in range(K)]
    B = [[[oracle.fill_uniform(n, q, 5000 + 100 * i + 10 * p + l) for l, q in enumerate(qs)] for p in range(count)] for i in range(K)]
    Ah = [[[ctx[l].fwd(A[i][p][l].copy()) for l in range(nl)] for p in range(count)] for i in range(K)]
    Bh = [[[ctx[l].fwd(B[i][p][l].copy()) for l in range(nl)] for p in range(count)] for i in range(K)]

    def image(sets):
        im = blank.copy()
        for o, data in sets.items():
            for p in range(count):
                for l in range(nl):
                    im[offs[o][p] + l * ls:offs[o][p] + l * ls + n] = data[p][l]
        return im

    def limbs_of(got, o, p):
        return [got[offs[o][p] + l * ls:offs[o][p] + l * ls + n] for l in range(nl)]

    def untouched(got, sets):
        mk = np.ones(words, dtype=bool)
        for o in sets:
            for p in range(count):
                for l in range(nl):
                    mk[offs[o][p] + l * ls:offs[o][p] + l * ls + n] = False
        return (got[mk] == GUARD).all()

    d = lib.DeviceBuffer(words)
    tabs = [lib.DeviceBuffer(count).upload(np.array([d.ptr + 8 * o for o in offs[i]], dtype=np.uint64)) for i in range(2 * K + 1)]
    key = lib.DeviceBuffer(nl * n).upload(np.concatenate([Bh[0][0][l] for l in range(nl)]))      # a broadcast operand is [limb][N]
    for mode in ("0", "1", None):
        lib.set_rns_launch(plans, mode)
        # c = inv(sum_i a_i^ . b_i^)
        for k in (1, K):
            for lazy in (False, True):
                mult = [np.uint64(q) * np.uint64(3 if (lazy and q < (1 << 60)) else 0) for q in qs]
                sets = {**{i: [[Ah[i][p][l] + mult[l] for l in range(nl)] for p in range(count)] for i in range(k)},
                        **{K + i: [[Bh[i][p][l] + mult[l] for l in range(nl)] for p in range(count)] for i in range(k)}}
                d.upload(image(sets))
                lib.rns_inv_dot_dev_ptrs(plans, tabs[2 * K].ptr, [tabs[i].ptr for i in range(k)], [tabs[K + i].ptr for i in range(k)], count, ls,
                                         lib.MUL_LAZY_IN if lazy else 0)
                got = d.download()
                for p in range(count):
                    for l, x in enumerate(limbs_of(got, 2 * K, p)):
                        acc = np.zeros(n, dtype=np.uint64)
                        for i in range(k):
                            acc = (acc + oracle.pointwise(Ah[i][p][l].copy(), Bh[i][p][l].copy(), qs[l])) % np.uint64(qs[l])
                        assert np.array_equal(x, ctx[l].inv(acc)), (mode, k, lazy, p, l)
                assert untouched(got, list(sets) + [2 * K]), (mode, k, lazy)
        # a broadcast key, c on a's own table
        d.upload(image({0: Ah[0]}))
        lib.rns_inv_dot_dev_ptrs(plans, tabs[0].ptr, [tabs[0].ptr], [key.ptr], count, ls, lib.MUL_B_BROADCAST)
        got = d.download()
        for p in range(count):
            for l, x in enumerate(limbs_of(got, 0, p)):
                assert np.array_equal(x, ctx[l].inv(oracle.pointwise(Ah[0][p][l].copy(), Bh[0][0][l].copy(), qs[l]))), (mode, p, l)
        assert untouched(got, [0]), mode
        # c^ = fwd(a) . b^, then c^ += fwd(a') . key^
        d.upload(image({0: A[0], 1: A[1], K: Bh[0]}))
        lib.rns_fwd_mul_dev_ptrs(plans, tabs[2 * K].ptr, tabs[0].ptr, tabs[K].ptr, count, ls)
        lib.rns_fwd_mul_dev_ptrs(plans, tabs[2 * K].ptr, tabs[1].ptr, key.ptr, count, ls, lib.MUL_ACCUMULATE | lib.MUL_B_BROADCAST)
        got = d.download()
        for p in range(count):
            for l, x in enumerate(limbs_of(got, 2 * K, p)):
                exp = (oracle.pointwise(Ah[0][p][l].copy(), Bh[0][p][l].copy(), qs[l]) +
                       oracle.pointwise(Ah[1][p][l].copy(), Bh[0][0][l].copy(), qs[l])) % np.uint64(qs[l])
                assert np.array_equal(x, exp), (mode, p, l)
        assert untouched(got, [0, 1, K, 2 * K]), mode
        # c = a * b: a table of its own, then a's
        for form in ("own", "on_a"):
            d.upload(image({0: A[0], K: B[0]}))
            co = 2 * K if form == "own" else 0
            lib.rns_negacyclic_mul_dev_ptrs(plans, tabs[co].ptr, tabs[0].ptr, tabs[K].ptr, count, ls)
            got = d.download()
            for p in range(count):
                for l, x in enumerate(limbs_of(got, co, p)):
                    assert np.array_equal(x, ctx[l].inv(oracle.pointwise(Ah[0][p][l].copy(), Bh[0][p][l].copy(), qs[l]))), (mode, form, p, l)
            assert untouched(got, [0, K, co]), (mode, form)
    for t in tabs:
        t.free()
    key.free(), d.free()
    for p in plans:
        p.destroy()
