#!/usr/bin/env python3
"""BASELINE config 5, one GPU's share (GPU box): FHE-style fwd -> pointwise -> inv pipeline at N=2^17 over a
4-prime RNS basis, 512 polynomial pairs per GPU; prints products/s and the bytes moved per second.
usage: python3 tools/pipeline_bench.py [--logn 17] [--batch 512] [--limbs 4] [--bits 50] [--steps 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ontt
lib = ontt.load()
ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=17)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--limbs", type=int, default=4)
ap.add_argument("--bits", type=int, default=50)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--lag", type=int, default=0, help="XCD-local launches: polynomials between passes (0 = default)")
ap.add_argument("--wpc", type=int, default=0, help="XCD-local launches: workgroups per CU (0 = default)")
ap.add_argument("--xcd-local", type=int, default=-1)
ap.add_argument("--batch-major", action="store_true", help="operands laid out [batch][limb][N] (SURVEY 8d) instead of [limb][batch][N]")
ap.add_argument("--alias", choices=["c", "a", "b"], default="c", help="where the product goes: its own buffer, over a, or over b")
a = ap.parse_args()
n = 1 << a.logn
qs = [lib.find_prime(a.bits, n, k) for k in range(a.limbs)]
plans = [lib.Plan(n, q, lib.min_root(q, n)) for q in qs]
for p in plans:
    p.set_option(lib.OPT_XCD_LOCAL, a.xcd_local); p.set_option(lib.OPT_XCD_LOCAL_LAG, a.lag); p.set_option(lib.OPT_XCD_LOCAL_WGS_PER_CU, a.wpc)
# (the A/B scripts pass the launch form as NTT_RNS_LOOP=0|1: it is this tool that reads it -- the library reads no environment -- and
# sets NTT_OPT_RNS_LAUNCH on the plans)
lib.set_rns_launch(plans, os.environ.get("NTT_RNS_LOOP"))
per = a.batch * n
bufs = [lib.DeviceBuffer(a.limbs * per) for _ in range(3)]
qmin = min(qs)
def fill():
    for i, b in enumerate(bufs[:2]):
        if a.batch_major:
            lib.fill_uniform(b.ptr, a.limbs * per, qmin, 1000 + i, 0)     # (values below every limb's modulus: one fill for the interleaved slab)
        else:
            for l, q in enumerate(qs):
                lib.fill_uniform(b.ptr + 8 * l * per, per, q, 1000 + i, l * per)
layout = lib.batch_major(plans) if a.batch_major else None
def step():
    lib.rns_negacyclic_mul(plans, bufs[{"c": 2, "a": 0, "b": 1}[a.alias]].ptr, bufs[0].ptr, bufs[1].ptr, a.batch, layout=layout)
fill(); step(); lib.stream_sync()
e0, e1 = lib.Event(), lib.Event()
tot = 0.0
for _ in range(a.steps):
    fill(); lib.stream_sync()           # operands are overwritten by the pipeline: regenerate outside the timed region
    e0.record(); step(); e1.record()
    tot += e1.elapsed_ms_since(e0)
ms = tot / a.steps
prods = a.batch / (ms * 1e-3)
# algorithmic bytes of a limb-product: SURVEY 8d's fused figure, 56N (fwd a 16N + fwd b 16N + product fused into the
# inverse 24N); what the launches actually move: see below
byts = 56 * n * a.limbs * a.batch / (ms * 1e-3)
# (one-launch products: 24N up to 2^14; 48N across the fabric from 2^23 coefficients per operand on, else 88N)
moved = (24 if a.logn <= 14 else (48 if a.batch * n >= (1 << 23) and a.batch >= 64 else 88)) * n * a.limbs * a.batch / (ms * 1e-3)
print("N=2^%d limbs=%d batch=%d %s: %.3f ms/step  %.0f RNS products/s  %.3f M limb-products/s  %.0f GB/s of 56N algorithmic bytes "
      "per limb-product = %.3f of 8 TB/s (bytes moved by the launches: %.0f GB/s)"
      % (a.logn, a.limbs, a.batch, ("[batch][limb][N] " if a.batch_major else "") + "NTT_RNS_LOOP=" + os.environ.get("NTT_RNS_LOOP", "auto"), ms, prods, prods * a.limbs / 1e6,
         byts / 1e9, byts / 8e12, moved / 1e9))
