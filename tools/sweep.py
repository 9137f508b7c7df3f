#!/usr/bin/env python3
"""Throughput sweep over sizes / directions / arithmetic (GPU box): one line per configuration.
usage: python3 tools/sweep.py [--logn 10 12 14 16 17] [--ops fwd inv mul] [--bytes 8e9]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, nargs="+", default=[10, 12, 13, 14, 15, 16, 17])
ap.add_argument("--ops", nargs="+", default=["fwd", "inv"])
ap.add_argument("--bytes", type=float, default=8e9)
ap.add_argument("--qs", nargs="+", default=["0x7fffffffe0001"])
ap.add_argument("--arith", nargs="+", default=["auto"])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--two-phase", type=int, default=-1, help="N=2^16, 2^17: 1 = one launch per transform, 0 = one launch per pass, -1 = the library's choice (default)")
ap.add_argument("--fused-product", type=int, default=1, help="ntt_negacyclic_mul_batch: 1 = fused product kernel where built (default), 0 = four transforms")
ap.add_argument("--xcd-local", type=int, default=-1, help="N=2^15..2^17: 1 = both passes as items of one launch, intermediate kept in the XCD's L2; 0 = per-pass launches")
ap.add_argument("--lag", type=int, default=0, help="--xcd-local 1: polynomials between the two passes (0 = default)")
ap.add_argument("--wpc", type=int, default=0, help="--xcd-local 1: workgroups per CU (0 = default)")
ap.add_argument("--max-grid", type=int, default=0, help="cap on workgroups per launch (0 = the kernels' own choice)")
ap.add_argument("--oversub", type=int, default=0, help="persistent block kernels: workgroups per resident slot (0 = the library's choice)")
ap.add_argument("--one-pass", type=int, default=-1, help="N=2^15, FP64: 1 = the one-pass kernel, 0 = the two-pass forms, -1 = the library's choice")
ap.add_argument("--block-log", type=int, default=0, help="N=2^15, 2^16: block size below the column pass (12, 14, 0 = library's choice)")
a = ap.parse_args()
ap2 = None
AR = {"auto": lib.ARITH_AUTO, "u64": lib.ARITH_U64, "f64": lib.ARITH_F64, "r4": lib.ARITH_U64_R4}
print("%-6s %-18s %-5s %-5s %10s %12s %9s %6s" % ("logn", "q", "arith", "op", "batch", "M NTT/s", "GB/s", "frac"))
for qs in a.qs:
    q = int(qs, 0)
    for ln in a.logn:
        n = 1 << ln
        if (q - 1) % (2 * n): continue
        w = lib.min_root(q, n)
        batch = max(1, int(a.bytes / (8 * n)))
        for ar in a.arith:
            try:
                plan = lib.Plan(n, q, w, arith=AR[ar])
            except lib.NttError as e:
                print(ln, qs, ar, "unsupported:", e); continue
            plan.set_option(lib.OPT_TWO_PHASE, a.two_phase)
            plan.set_option(lib.OPT_FUSED_PRODUCT, a.fused_product)
            plan.set_option(lib.OPT_XCD_LOCAL, a.xcd_local)
            plan.set_option(lib.OPT_XCD_LOCAL_LAG, a.lag)
            plan.set_option(lib.OPT_XCD_LOCAL_WGS_PER_CU, a.wpc)
            if a.one_pass >= 0 and hasattr(lib, "OPT_ONE_PASS"):
                try: plan.set_option(lib.OPT_ONE_PASS, a.one_pass)
                except lib.NttError: pass
            if a.max_grid: plan.set_option(lib.OPT_MAX_GRID, a.max_grid)
            if os.environ.get("NTT_DOT_UNFUSED") == "1": plan.set_option(lib.OPT_DOT_FUSED, 0)   # (tools/ab_product_tail.sh)
            if a.oversub and hasattr(lib, "OPT_BLOCK_OVERSUB"):
                try: plan.set_option(lib.OPT_BLOCK_OVERSUB, a.oversub)
                except lib.NttError: pass     # (an older build of the library under NTT_LIB)
            if a.block_log and ln in (15, 16): plan.set_option(lib.OPT_BLOCK_LOG, a.block_log)
            nb = 3 if "mul" in a.ops else 1
            bufs = [lib.DeviceBuffer(batch * n) for _ in range(nb)]
            for i, b in enumerate(bufs): lib.fill_uniform(b.ptr, batch * n, q, 77 + i)
            for op in a.ops:
                if op == "mul" and ar == "r4": continue      # the radix-4 policy has no product chain
                def run():
                    if op == "fwd": plan.fwd(bufs[0].ptr, batch)
                    elif op == "inv": plan.inv(bufs[0].ptr, batch)
                    elif op == "fwdlazy": plan.fwd(bufs[0].ptr, batch, lazy=True, wide=True)   # lazy in, lazy out: chained
                    elif op == "invlazy": plan.inv(bufs[0].ptr, batch, lazy=True, wide=True)
                    else: plan.negacyclic_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch)
                for _ in range(2): run()
                e0, e1 = lib.Event(), lib.Event()
                lib.stream_sync(); e0.record()
                for _ in range(a.steps): run()
                e1.record(); ms = e1.elapsed_ms_since(e0) / a.steps
                # algorithmic bytes per polynomial: a transform reads and writes every coefficient once; a product
                # reads a and b and writes c -- 24N, what the one-launch product kernels actually move up to N = 2^14 (SURVEY's
                # 56N counted three separate transforms; with it the small sizes would show fractions above 1)
                per = {"fwd": 16, "inv": 16, "fwdlazy": 16, "invlazy": 16, "mul": 24}[op] * n
                gbs = batch * per / ms / 1e6
                print("%-6d %-18s %-5s %-7s %10d %12.3f %9.0f %6.3f" % (ln, qs, (["auto","u64","f64","r4"][plan.info()["arith"]] + ("w" if plan.info()["f64_class"] == 52 else "")), op, batch, batch / ms / 1e3, gbs, gbs / 8000))
            for b in bufs: b.free()
            plan.destroy()
