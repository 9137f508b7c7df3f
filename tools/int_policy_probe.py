#!/usr/bin/env python3
"""tools/int_policy_probe.py [--bits 57] [--logn 14 12] [--ops fwd inv]: integer-policy transforms of one plan checked against the
oracle (first, middle and last polynomial of the slab) and timed with HIP events -- the A/B harness of the integer butterflies
(NTT_LIB selects the build, tools/build_tu_variant.sh)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ontt  # noqa: E402
lib = ontt.load()
from oracle_binding import Oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=int, nargs="+", default=[57])
ap.add_argument("--logn", type=int, nargs="+", default=[14])
ap.add_argument("--ops", nargs="+", default=["fwd", "inv"])
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--arith", default="u64")
a = ap.parse_args()
orc = Oracle()
ARITH = {"u64": lib.ARITH_U64, "auto": lib.ARITH_AUTO, "f64": lib.ARITH_F64}[a.arith]
print("lib", os.environ.get("NTT_LIB", "tree"))
for m in a.logn:
    n = 1 << m
    batch = int(a.gib * 2**30 / (8 * n))
    for bits in a.bits:
        q = orc.find_prime(bits, n)
        w = orc.min_root(q, n)
        cx = orc.ctx(n, q, w)
        plan = lib.Plan(n, q, w, arith=ARITH)
        if os.environ.get("NTT_INT_WIDE") == "0" and plan.info()["arith"] == lib.ARITH_U64:
            plan.set_option(lib.OPT_INT_WIDE, 0)   # (tools/ab_int_wide.sh: the reference's butterflies; the library reads no environment)
        host = orc.fill_uniform(batch * n, q, 31 + m)
        d = torch.from_numpy(host.view(np.int64)).cuda()
        picks = [0, batch // 2, batch - 1]
        for op in a.ops:
            d.copy_(torch.from_numpy(host.view(np.int64)))
            (plan.fwd if op == "fwd" else plan.inv)(d.data_ptr(), batch)
            torch.cuda.synchronize()
            out = d.cpu().numpy().view(np.uint64)
            ok = all(np.array_equal(out[p * n:(p + 1) * n], (cx.fwd if op == "fwd" else cx.inv)(host[p * n:(p + 1) * n])) for p in picks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                (plan.fwd if op == "fwd" else plan.inv)(d.data_ptr(), batch)
            e0.record()
            for _ in range(a.steps):
                (plan.fwd if op == "fwd" else plan.inv)(d.data_ptr(), batch)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            print("2^%-2d q=%#x (%d bits) %s  %s  %8.3f M NTT/s  frac %.3f" % (m, q, bits, op, "ok" if ok else "MISMATCH", batch / ms / 1e3,
                                                                            16 * n * batch / ms / 1e6 / 8000))
        plan.destroy()
