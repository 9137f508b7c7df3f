#!/usr/bin/env python3
"""Randomised differential soak of the HIP library against the CPU oracle (GPU box; test infrastructure, like tests/).

Every round draws a transform size 2^1..2^18, a prime of 20..60 bits with 2N | q-1, an arithmetic policy the
library offers for it, a ragged batch (with a bias to the persistent grids' edges: 255, 256, 257, 511, ...), plan
options (chunk size, grid cap, two-phase, XCD-local launch with random lag and residency, column-only engine, fused
product on/off, workgroups per resident slot), now and then an RNS set (one launch over all limbs or the per-prime loop, in
[limb][batch][N] or [batch][limb][N] layout) or a shuffled pointer batch (host array or device table), the one-pass 2^15 kernel forced on / off, and checks, bit for bit against
the oracle: forward, inverse, lazy-input and lazy-output forms, the product chain in all aliasing forms, the products of
operands given in the NTT domain (inner products of k pairs, canonical / lazy / broadcast; one operand transformed beforehand).
usage: python3 tools/soak.py [--seconds 300] [--seed 1] [--max-coeffs 2^22]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ontt
from oracle_binding import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-coeffs", type=int, default=1 << 22)
args = ap.parse_args()
lib, orc = ontt.load(), Oracle()
rng = np.random.default_rng(args.seed)
EDGE = [1, 2, 3, 7, 8, 15, 16, 17, 31, 63, 64, 65, 127, 255, 256, 257, 511, 512, 513, 1023, 1025]
t_end, rounds, checks = time.time() + args.seconds, 0, 0
stats = {}


def scatter(rng, n, operands):
    """the polynomials of several operands (count * n words each) shuffled over ONE pool with irregular gaps: the pool, one device
    table of polynomial addresses per operand, and per operand the word offsets (products over separately held operands)"""
    count = operands[0].size // n
    total = len(operands) * count
    gaps = rng.integers(0, 3, size=total) * int(rng.choice([0, 1, 8, n // 2 + 9]))      # (odd word offsets among them)
    starts = np.cumsum(gaps + n) - n
    order = rng.permutation(total)
    img = np.zeros(int(starts[-1]) + n, dtype=np.uint64)
    offs = []
    for o, x in enumerate(operands):
        st = starts[order[o * count:(o + 1) * count]]
        offs.append(st)
        for i, s0 in enumerate(st):
            img[int(s0):int(s0) + n] = x[i * n:(i + 1) * n]
    pool = lib.DeviceBuffer(img.size).upload(img)
    tabs = [lib.DeviceBuffer(count).upload(np.array([pool.ptr + 8 * int(s0) for s0 in st], dtype=np.uint64)) for st in offs]
    return pool, tabs, offs


def gathered(img, st, n):
    return np.concatenate([img[int(s0):int(s0) + n] for s0 in st])


def fail(what, **kw):
    print("SOAK FAILURE:", what, kw, flush=True)
    sys.exit(1)


while time.time() < t_end:
    m = int(rng.integers(1, 19))
    n = 1 << m
    bits = int(rng.choice([int(rng.integers(max(m + 2, 20), 61)), 50, 51, 52, 60, 58, 57]))
    q = lib.find_prime(bits, n, int(rng.integers(0, 4)))
    if q == 0:
        continue
    w = lib.min_root(q, n)
    if w == 0:
        fail("min_root", q=hex(q), n=n)
    cap = max(1, args.max_coeffs // n)
    batch = int(rng.choice(EDGE)) if rng.random() < 0.5 else int(rng.integers(1, 600))
    batch = max(1, min(batch, cap))
    policies = [lib.ARITH_AUTO, lib.ARITH_U64]
    if q < (1 << 52):
        policies.append(lib.ARITH_F64)
    if 6 <= m <= 18 and q < (1 << 60):
        policies.append(lib.ARITH_U64_R4)      # (above 2^14: two passes)
    arith = int(rng.choice(policies))
    try:
        plan = lib.Plan(n, q, w, arith=arith)
    except lib.NttError as e:
        fail("plan", q=hex(q), n=n, arith=arith, err=str(e))
    info = plan.info()
    opts = {}
    if rng.random() < 0.3:
        opts["chunk"] = int(rng.choice([1, 2, 8, 64, 256]))
        plan.set_option(lib.OPT_CHUNK_MIB, opts["chunk"])
    if rng.random() < 0.3:
        opts["grid"] = int(rng.choice([1, 3, 64, 255, 256, 1000]))
        plan.set_option(lib.OPT_MAX_GRID, opts["grid"])
    if rng.random() < 0.3:
        opts["two_phase"] = 1
        plan.set_option(lib.OPT_TWO_PHASE, 1)
    if info["arith"] == lib.ARITH_U64 and (1 << 40) <= q and rng.random() < 0.5:
        # the wide integer policy: on for plans that did not choose it, off for those that did, or a narrower class
        best = 3 if q < (1 << 58) else (1 if q < (1 << 60) else 0)
        opts["int_wide"] = int(rng.choice([0, 1] + [10 + c for c in (0, 1, 3) if c <= best]))
        plan.set_option(lib.OPT_INT_WIDE, opts["int_wide"])
        info = plan.info()
    if rng.random() < 0.2 and arith != lib.ARITH_U64_R4:
        opts["generic"] = 1
        plan.set_generic(1)
    if rng.random() < 0.3 and m in (15, 16):
        opts["block"] = int(rng.choice([12, 14]))
        plan.set_option(lib.OPT_BLOCK_LOG, opts["block"])
    if rng.random() < 0.3:
        opts["oversub"] = int(rng.choice([1, 2, 3, 8, 16, 64]))      # workgroups per resident slot of the persistent block kernels
        plan.set_option(lib.OPT_BLOCK_OVERSUB, opts["oversub"])
    if rng.random() < 0.3:
        opts["fused_product"] = int(rng.choice([0, 0, 2]))      # 2: a's forward transform as a launch of its own
        plan.set_option(lib.OPT_FUSED_PRODUCT, opts["fused_product"])
    if m == 15 and rng.random() < 0.7:
        # round 6: N = 2^15 in one pass (FP64 policies): forced on / off; the automatic choice wants a polynomial per CU
        opts["one_pass"] = int(rng.choice([1, 1, 0]))
        plan.set_option(lib.OPT_ONE_PASS, opts["one_pass"])
        if rng.random() < 0.3:
            batch = max(1, min(int(rng.choice([255, 256, 257, 300, 513])), cap))
    if m in (15, 16, 17) and rng.random() < 0.6:
        # both passes as items of one launch (needs batch >= 64; forced on for both directions, random lag / residency)
        opts["xcd_local"] = int(rng.choice([1, 1, 0]))
        plan.set_option(lib.OPT_XCD_LOCAL, opts["xcd_local"])
        opts["lag"], opts["wpc"] = int(rng.choice(list(range(13)) + [16, 24, 32, 48])), int(rng.integers(0, 5))
        plan.set_option(lib.OPT_XCD_LOCAL_LAG, opts["lag"])
        plan.set_option(lib.OPT_XCD_LOCAL_WGS_PER_CU, opts["wpc"])
        if opts["xcd_local"] == 1:
            batch = max(batch, int(rng.choice([64, 65, 71, 96, 129])))
            if rng.random() < 0.25:
                batch = int(rng.choice([512, 513, 519, 530]))      # large enough for the three-pass product launch
    cx = orc.ctx(n, q, w)
    a = orc.fill_uniform(batch * n, q, int(rng.integers(1, 1 << 40)))
    k = int(rng.integers(0, 5))
    if k == 0:
        a[:] = q - 1
    elif k == 1:
        a[: a.size // 2] = 0
    key = (m, info["arith"], info["f64_class"])
    stats[key] = stats.get(key, 0) + 1
    ctxt = dict(m=m, q=hex(q), arith=arith, resolved=info["arith"], cls=info["f64_class"], batch=batch, opts=opts)
    want = cx.fwd(a)
    got = plan.fwd_host(a)
    if not np.array_equal(got, want):
        fail("fwd", **ctxt)
    if info["arith"] == lib.ARITH_U64_R4:
        if not np.array_equal(plan.fwd_host(a, lazy=True), cx.fwd_r4_lazy(a)):      # the reference's lazy values, any size
            fail("fwd lazy == fwd_ntt_radix4_lazy", **ctxt)
        checks += 1
    if not np.array_equal(plan.inv_host(want), a):
        fail("inv", **ctxt)
    lazy_mult = 8 if q < (1 << 60) else 4
    lz = a + np.uint64(q) * rng.integers(0, lazy_mult, size=a.shape, dtype=np.uint64)
    if not np.array_equal(plan.fwd_host(lz, wide=True), want):
        fail("fwd wide", **ctxt)
    lz = want + np.uint64(q) * rng.integers(0, lazy_mult, size=a.shape, dtype=np.uint64)
    if not np.array_equal(plan.inv_host(lz, wide=True), a):
        fail("inv wide", **ctxt)
    out = plan.fwd_host(a, lazy=True)
    bound = 8 if info["arith"] == lib.ARITH_U64_R4 else 4
    if int(out.max()) >= bound * q or not np.array_equal(out % np.uint64(q), want):
        fail("fwd lazy", **ctxt)
    out = plan.inv_host(want, lazy=True)
    if int(out.max()) >= 2 * q or not np.array_equal(out % np.uint64(q), a):
        fail("inv lazy", **ctxt)
    checks += 6
    if rng.random() < 0.25 and batch >= 2:
        # the same batch as separately placed polynomials: a pointer per polynomial, shuffled, over a padded pool (ntt_transform_ptrs)
        gap = int(rng.choice([0, 0, 8, n, 3 * n + 2]))
        pool = lib.DeviceBuffer(batch * (n + gap))
        order = rng.permutation(batch)
        host = np.zeros(batch * (n + gap), dtype=np.uint64)
        for i, slot in enumerate(order):
            host[slot * (n + gap): slot * (n + gap) + n] = a[i * n:(i + 1) * n]
        pool.upload(host)
        # irregular gaps every other time (no progression: the device table), and the device-array form (as given, no upload)
        ptrs = [pool.ptr + 8 * int(slot) * (n + gap) for slot in order]
        if rng.random() < 0.5:
            tab = lib.DeviceBuffer(batch).upload(np.array(ptrs, dtype=np.uint64))
            plan.transform_dev_ptrs(tab.ptr, batch)
            tab.free()
        else:
            plan.transform_ptrs(ptrs)
        res = pool.download()
        if batch >= 4 and rng.random() < 0.5:
            # ... and back, leaving one polynomial of the pool out: the sorted pointers are then no single progression (table path of
            # the host-array form); the polynomial left out must come back transformed, all others as they went in
            skip = batch // 3
            plan.transform_ptrs([p_ for i_, p_ in enumerate(ptrs) if i_ != skip], lib.FLAG_INVERSE)
            back = pool.download()
            for i in (0, skip, batch - 1):
                slot = int(order[i])
                exp = want[i * n:(i + 1) * n] if i == skip else a[i * n:(i + 1) * n]
                if not np.array_equal(back[slot * (n + gap): slot * (n + gap) + n], exp):
                    fail("transform_ptrs, irregular subset", gap=gap, poly=i, **ctxt)
        for i in (0, batch // 2, batch - 1):
            slot = int(order[i])
            if not np.array_equal(res[slot * (n + gap): slot * (n + gap) + n], want[i * n:(i + 1) * n]):
                fail("transform_ptrs", gap=gap, poly=i, **ctxt)
        pool.free()
        checks += 1
    if True:   # every policy has a product chain (radix-4: its own transforms around the pointwise product, N <= 2^14)
        b = orc.fill_uniform(batch * n, q, int(rng.integers(1, 1 << 40)))
        prod = cx.inv(orc.pointwise(want, cx.fwd(b), q))
        form = int(rng.integers(0, 4))
        da, db = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b)
        dc = lib.DeviceBuffer(a.size)
        if form == 0:
            plan.negacyclic_mul(dc.ptr, da.ptr, db.ptr, batch); res = dc.download()
        elif form == 1:
            plan.negacyclic_mul(da.ptr, da.ptr, db.ptr, batch); res = da.download()
        elif form == 2:
            plan.negacyclic_mul(db.ptr, da.ptr, db.ptr, batch); res = db.download()
        else:
            plan.negacyclic_mul(dc.ptr, da.ptr, da.ptr, batch); res = dc.download()
            prod = cx.inv(orc.pointwise(want, want, q))
        if not np.array_equal(res, prod):
            fail("product form %d" % form, **ctxt)
        for d in (da, db, dc):
            d.free()
        checks += 1
        if rng.random() < 0.35:
            # the same product over SEPARATELY HELD operands (device tables; up to 2^14 inside the fused product kernels)
            pool, tabs, offs = scatter(rng, n, [a, b, np.zeros_like(a)])
            ci = {0: 2, 1: 0, 2: 1, 3: 2}[form]
            plan.negacyclic_mul_dev_ptrs(tabs[ci].ptr, tabs[0].ptr, tabs[0 if form == 3 else 1].ptr, batch)
            if not np.array_equal(gathered(pool.download(), offs[ci], n), prod):
                fail("product over tables, form %d" % form, **ctxt)
            for d in tabs + [pool]:
                d.free()
            checks += 1
    # operands in the NTT domain: c = inv(sum_i a_i^ (.) b_i^) with k pairs, canonical or lazy words, per-polynomial or
    # broadcast b, c aliasing an operand (k = 1); c = inv(fwd(a) (.) b^) in a random aliasing form
    if rng.random() < 0.6:
        kk = int(rng.choice([1, 1, 2, 3, 5, 8]))
        if kk * batch * n > 4 * args.max_coeffs:
            kk = 1
        lz_in, bc = bool(rng.random() < 0.4), bool(rng.random() < 0.35)
        mult = 4 if lz_in else 1
        ah = [orc.fill_uniform(batch * n, q, int(rng.integers(1, 1 << 40))) for _ in range(kk)]
        bh = [orc.fill_uniform((1 if bc else batch) * n, q, int(rng.integers(1, 1 << 40))) for _ in range(kk)]
        if lz_in:
            ah = [x + np.uint64(q) * rng.integers(0, mult, size=x.shape, dtype=np.uint64) for x in ah]
            bh = [x + np.uint64(q) * rng.integers(0, mult, size=x.shape, dtype=np.uint64) for x in bh]
        expd = cx.inv(orc.dot(ah, bh, q, n, bc))
        dah = [lib.DeviceBuffer(x.size).upload(x) for x in ah]
        dbh = [lib.DeviceBuffer(x.size).upload(x) for x in bh]
        flags = (lib.MUL_LAZY_IN if lz_in else 0) | (lib.MUL_B_BROADCAST if bc else 0)
        alias = int(rng.integers(0, 3)) if kk == 1 else 0
        if alias == 2 and bc:
            alias = 0
        dst = lib.DeviceBuffer(batch * n) if alias == 0 else (dah[0] if alias == 1 else dbh[0])
        plan.inv_dot(dst.ptr, [x.ptr for x in dah], [x.ptr for x in dbh], batch, flags)
        if not np.array_equal(dst.download(), expd):
            fail("inv_dot", k=kk, lazy=lz_in, bcast=bc, alias=alias, **ctxt)
        for d in dah + dbh + ([dst] if alias == 0 else []):
            d.free()
        checks += 1
        if rng.random() < 0.4:
            # ... over device tables: every a_i^ (and every b_i^ that is not a shared key) a table of its own, c on a table of its own
            # or (k = 1) on an operand's
            ops = list(ah) + ([] if bc else list(bh)) + [np.zeros(batch * n, dtype=np.uint64)]
            pool, tabs, offs = scatter(rng, n, ops)
            keys = [lib.DeviceBuffer(x.size).upload(x) for x in bh] if bc else []
            ci = len(ops) - 1 if alias == 0 else (0 if alias == 1 else kk)
            plan.inv_dot_dev_ptrs(tabs[ci].ptr, [t.ptr for t in tabs[:kk]], [x.ptr for x in keys] if bc else [t.ptr for t in tabs[kk:2 * kk]], batch, flags)
            if not np.array_equal(gathered(pool.download(), offs[ci], n), expd):
                fail("inv_dot over tables", k=kk, lazy=lz_in, bcast=bc, alias=alias, **ctxt)
            for d in tabs + keys + [pool]:
                d.free()
            checks += 1
        # (lazy words for the product kernels must stay below 2^53; the radix-4 policy takes canonical words)
        bw = want if (4 * q > (1 << 53) or info["arith"] == lib.ARITH_U64_R4 or rng.random() < 0.5) else \
            want + np.uint64(q) * rng.integers(0, 3, size=want.shape, dtype=np.uint64)
        b2 = orc.fill_uniform(batch * n, q, int(rng.integers(1, 1 << 40)))
        expm = cx.inv(orc.pointwise(cx.fwd(b2), want, q))
        dco, dbw = lib.DeviceBuffer(b2.size).upload(b2), lib.DeviceBuffer(bw.size).upload(bw)
        alias = int(rng.integers(0, 3))
        dst = lib.DeviceBuffer(batch * n) if alias == 0 else (dco if alias == 1 else dbw)
        plan.mul_transformed(dst.ptr, dco.ptr, dbw.ptr, batch, lib.MUL_LAZY_IN if bw is not want else 0)
        if not np.array_equal(dst.download(), expm):
            fail("mul_transformed", alias=alias, lazy=bw is not want, **ctxt)
        for d in [dco, dbw] + ([dst] if alias == 0 else []):
            d.free()
        checks += 1
        # c^ = fwd(a) (.) b^ (+ c^): the result stays in the NTT domain; lazy / broadcast b^, accumulator, aliasing
        lz_in, bc, ac = bool(rng.random() < 0.4), bool(rng.random() < 0.4), bool(rng.random() < 0.5)
        bh = orc.fill_uniform((1 if bc else batch) * n, q, int(rng.integers(1, 1 << 40)))
        bwv = bh + (np.uint64(q) * rng.integers(0, 4, size=bh.shape, dtype=np.uint64) if lz_in else np.uint64(0))
        c0 = orc.fill_uniform(batch * n, q, int(rng.integers(1, 1 << 40)))
        expf = orc.pointwise(want, np.tile(bh, batch) if bc else bh, q)
        if ac:
            expf = (expf + c0) % np.uint64(q)
        dco, dbw, dcc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(bwv.size).upload(bwv), lib.DeviceBuffer(a.size).upload(c0)
        alias = 0 if ac else int(rng.integers(0, 3))
        if alias == 2 and bc:
            alias = 0
        dst = dcc if alias == 0 else (dco if alias == 1 else dbw)
        plan.fwd_mul(dst.ptr, dco.ptr, dbw.ptr, batch, (lib.MUL_LAZY_IN if lz_in else 0) | (lib.MUL_B_BROADCAST if bc else 0) |
                     (lib.MUL_ACCUMULATE if ac else 0))
        if not np.array_equal(dst.download(batch * n), expf):
            fail("fwd_mul", lazy=lz_in, bcast=bc, acc=ac, alias=alias, **ctxt)
        for d in (dco, dbw, dcc):
            d.free()
        checks += 1
        if rng.random() < 0.4:
            ops = [a, c0] + ([] if bc else [bwv])
            pool, tabs, offs = scatter(rng, n, ops)
            keyd = lib.DeviceBuffer(bwv.size).upload(bwv) if bc else None
            ci = 1 if alias == 0 else (0 if alias == 1 else 2)
            plan.fwd_mul_dev_ptrs(tabs[ci].ptr, tabs[0].ptr, keyd.ptr if bc else tabs[2].ptr, batch,
                                  (lib.MUL_LAZY_IN if lz_in else 0) | (lib.MUL_B_BROADCAST if bc else 0) | (lib.MUL_ACCUMULATE if ac else 0))
            if not np.array_equal(gathered(pool.download(), offs[ci], n), expf):
                fail("fwd_mul over tables", lazy=lz_in, bcast=bc, acc=ac, alias=alias, **ctxt)
            for d in tabs + ([keyd] if bc else []) + [pool]:
                d.free()
            checks += 1
    plan.destroy()
    # reference-signature entry points on the caller's own tables (one polynomial, host pointers): bit-exact lazy values
    if rng.random() < 0.15 and m >= 2 and q < (1 << 60):
        U64P = lib.U64P
        tw, twc, e, ec = cx.table("w"), cx.table("wcon"), cx.table("e"), cx.table("econ")
        one = a[:n].copy()
        x = one.copy()
        lib._lib.fwd_ntt_ref_harvey_lazy(x.ctypes.data_as(U64P), n, q, tw.ctypes.data_as(U64P), twc.ctypes.data_as(U64P))
        if not np.array_equal(x, cx.fwd_lazy(one)):
            fail("shim fwd_ntt_ref_harvey_lazy", **ctxt)
        x = one.copy()
        lib._lib.fwd_ntt_radix4_lazy(x.ctypes.data_as(U64P), n, q, e.ctypes.data_as(U64P), ec.ctypes.data_as(U64P))
        ok = np.array_equal(x, cx.fwd_r4_lazy(one)) if 6 <= m <= 18 else (int(x.max()) < 8 * q and np.array_equal(x % np.uint64(q), want[:n]))
        if not ok:
            fail("shim fwd_ntt_radix4_lazy", **ctxt)
        checks += 2
        if rng.random() < 0.3:
            lib.compat_release()
    # RNS sets: one launch over all limbs against the per-prime loop's semantics (every limb against the oracle)
    if rng.random() < 0.15 and 8 <= m <= 17:
        nl = int(rng.choice([2, 3, 4, 7, 16, 17]))
        rb = int(rng.choice([1, 2, 3, 8]))
        if m >= 15 and rng.random() < 0.5:
            nl, rb = int(rng.choice([2, 3, 4])), int(rng.choice([22, 33, 64, 70]))   # one XCD-local launch over the limbs
        if nl * rb * n <= max(args.max_coeffs, (1 << 25) if m >= 15 else 0):
            rbits = int(rng.choice([45, 49, 50, 52, 57, 60]))    # (52: the reduce-as-scheduled FP64 policy; 57, 60: the wide integer policy)
            if rng.random() < 0.3:
                # a modulus chain with primes of several sizes: served as runs of consecutive compatible limbs
                sizes = [int(x) for x in rng.choice([45, 50, 52, 57, 60], size=nl)]
                seen = {}
                qs = []
                for bts in sizes:
                    qs.append(lib.find_prime(bts, n, seen.get(bts, 0)))
                    seen[bts] = seen.get(bts, 0) + 1
            else:
                qs = [lib.find_prime(rbits, n, i) for i in range(nl)]
            if all(qs) and len(set(qs)) == nl:
                ws = [lib.min_root(x, n) for x in qs]
                plans = [lib.Plan(n, x, y) for x, y in zip(qs, ws)]
                ra = np.concatenate([orc.fill_uniform(rb * n, x, int(rng.integers(1, 1 << 40))) for x in qs])
                rbv = np.concatenate([orc.fill_uniform(rb * n, x, int(rng.integers(1, 1 << 40))) for x in qs])
                rloop = int(rng.integers(0, 2))
                lib.set_rns_launch(plans, rloop)
                # half of the rounds in the caller-native layout [batch][limb][N] (the *_strided entry points)
                bm = bool(rng.random() < 0.5)
                lay = (n, nl * n) if bm else None

                def place(x):      # [limb][batch][N] -> the layout's image, and back
                    return np.ascontiguousarray(x.reshape(nl, rb, n).transpose(1, 0, 2)).reshape(-1) if bm else x

                def gather(x):
                    return np.ascontiguousarray(x.reshape(rb, nl, n).transpose(1, 0, 2)).reshape(-1) if bm else x
                da, db, dc = lib.DeviceBuffer(ra.size).upload(place(ra)), lib.DeviceBuffer(ra.size).upload(place(rbv)), lib.DeviceBuffer(ra.size)
                lib.rns_fwd(plans, da.ptr, rb, layout=lay)
                f = gather(da.download())
                lib.rns_inv(plans, da.ptr, rb, layout=lay)
                back = gather(da.download())
                lib.rns_negacyclic_mul(plans, dc.ptr, da.ptr, db.ptr, rb, layout=lay)
                pr = gather(dc.download())
                da.upload(place(ra)), db.upload(place(f))
                lib.rns_mul_transformed(plans, dc.ptr, da.ptr, db.ptr, rb, layout=lay)        # inv(fwd(a) (.) fwd(a)): the square
                sq = gather(dc.download())
                da.upload(place(f)), db.upload(place(f))
                lib.rns_inv_dot(plans, dc.ptr, [da.ptr, db.ptr], [db.ptr, da.ptr], rb, layout=lay)   # inv(2 a^ (.) a^)
                dt = gather(dc.download())
                # the same set through the twins over DEVICE POINTER TABLES: a [batch][limb][N] image is `rb` separately held RNS polynomials
                # whose limbs are N words apart -- shuffled tables into the three buffers; small batches: one launch over a run's limbs
                tp = tp_dt = tp_fm = None
                if bm:
                    perm = rng.permutation(rb)
                    tabs = [lib.DeviceBuffer(rb).upload(np.array([buf.ptr + 8 * int(pp) * nl * n for pp in perm], dtype=np.uint64)) for buf in (da, db, dc)]
                    lib.rns_inv_dot_dev_ptrs(plans, tabs[2].ptr, [tabs[0].ptr, tabs[1].ptr], [tabs[1].ptr, tabs[0].ptr], rb, n)
                    tp_dt = gather(dc.download())
                    da.upload(place(ra))
                    lib.rns_fwd_mul_dev_ptrs(plans, tabs[2].ptr, tabs[0].ptr, tabs[1].ptr, rb, n)        # fwd(a) (.) a^
                    tp_fm = gather(dc.download())
                    da.upload(place(ra)), db.upload(place(rbv))
                    lib.rns_negacyclic_mul_dev_ptrs(plans, tabs[2].ptr, tabs[0].ptr, tabs[1].ptr, rb, n)
                    tp = gather(dc.download())
                    for t_ in tabs:
                        t_.free()
                for l, (x, y) in enumerate(zip(qs, ws)):
                    c2 = orc.ctx(n, x, y)
                    sl = slice(l * rb * n, (l + 1) * rb * n)
                    fa = c2.fwd(ra[sl])
                    if not np.array_equal(f[sl], fa) or not np.array_equal(back[sl], ra[sl]) or \
                       not np.array_equal(pr[sl], c2.inv(orc.pointwise(fa, c2.fwd(rbv[sl]), x))) or \
                       not np.array_equal(sq[sl], c2.inv(orc.pointwise(fa, fa, x))) or \
                       not np.array_equal(dt[sl], c2.inv(orc.dot([fa, fa], [fa, fa], x))):
                        fail("rns", m=m, limbs=nl, batch=rb, limb=l, loop=rloop, batch_major=bm, q=hex(x))
                    if bm and (not np.array_equal(tp[sl], pr[sl]) or not np.array_equal(tp_dt[sl], dt[sl]) or
                               not np.array_equal(tp_fm[sl], orc.pointwise(fa, fa, x))):
                        fail("rns over pointer tables", m=m, limbs=nl, batch=rb, limb=l, loop=rloop, q=hex(x))
                for d in (da, db, dc):
                    d.free()
                for pl in plans:
                    pl.destroy()
                checks += (8 if bm else 5) * nl
                stats_rns = stats.setdefault(("rns", 0, 0), 0)
                stats[("rns", 0, 0)] = stats_rns + 1
    rounds += 1

print("soak ok: %d rounds, %d checks in %.0f s (seed %d)" % (rounds, checks, args.seconds, args.seed))
by_arith = {}
rns_rounds = stats.pop(("rns", 0, 0), 0)
for (m, ar, cls), c in stats.items():
    by_arith[(ar, cls)] = by_arith.get((ar, cls), 0) + c
print("RNS rounds:", rns_rounds)
print("rounds per (resolved policy, FP64 class):", dict(sorted(by_arith.items())))
print("sizes seen:", sorted({m for (m, _, _) in stats}))
