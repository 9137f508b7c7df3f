#!/usr/bin/env python3
"""development aid: one small XCD-local transform with the queue state printed (NTT_TEAM_DEBUG=1, watchdog build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ontt
from oracle_binding import Oracle
lib = ontt.load(); orc = Oracle()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
inv = len(sys.argv) > 3 and sys.argv[3] == "inv"
n = 1 << m
q = lib.find_prime(50, n, 1); w = lib.min_root(q, n)
print("plan...", flush=True)
plan = lib.Plan(n, q, w); plan.set_option(lib.OPT_XCD_LOCAL, 1)
print("plan ok", flush=True)
a = orc.fill_uniform(batch * n, q, 5)
cx = orc.ctx(n, q, w)
buf = lib.DeviceBuffer(a.size).upload(a)
print("launch...", flush=True)
(plan.inv if inv else plan.fwd)(buf.ptr, batch)
print("launched", flush=True)
lib.stream_sync()
got = buf.download()
exp = (cx.inv if inv else cx.fwd)(a)
bad = [p for p in range(batch) if not np.array_equal(got[p*n:(p+1)*n], exp[p*n:(p+1)*n])]
print("m", m, "batch", batch, "inv" if inv else "fwd", "bad polys:", len(bad), bad[:16])
