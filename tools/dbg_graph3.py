"""tools/dbg_graph3.py [between]: the XCD-local TRANSFORM and PRODUCT launches (team_kernel, team_product_kernel: rounds 3-4) captured in a
HIP graph and replayed with other work in between -- does the control block's clearing hold?  (run with NTT_LIB=build/libntt_prev.so
for the library of round 4, whose launches cleared the block with hipMemsetAsync)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
torch.cuda.set_device(0)
import numpy as np
import ontt
from oracle_binding import Oracle
lib, orc = ontt.load(), Oracle()
n, batch, q = 1 << 16, 96, 0x7fffffffe0001
w = lib.min_root(q, n)
cx = orc.ctx(n, q, w)
between = sys.argv[1] if len(sys.argv) > 1 else "flush"
plan = lib.Plan(n, q, w, device=0)
plan.set_option(lib.OPT_XCD_LOCAL, 1)
z = lambda: torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
sa, sb, ta, tb, tc = z(), z(), z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):        # the (plan, stream) blocks come from a direct call before the capture (the rule of rounds 3-4)
    plan.fwd(ta.data_ptr(), batch, stream=s.cuda_stream)
    plan.negacyclic_mul(tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=s.cuda_stream)
s.synchronize()
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    if between != "flush_inside": pass
    plan.fwd(ta.data_ptr(), batch, stream=st)
    plan.negacyclic_mul(tc.data_ptr(), tb.data_ptr(), tb.data_ptr(), batch, stream=st)      # tc = b * b
for seed in (1, 2, 3, 1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    ta.copy_(torch.from_numpy(a.view(np.int64))); tb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay(); torch.cuda.synchronize()
    got_f, got_p = ta.cpu().numpy().view(np.uint64), tc.cpu().numpy().view(np.uint64)
    badf = [j for j in range(batch) if not np.array_equal(got_f[j*n:(j+1)*n], cx.fwd(a[j*n:(j+1)*n].copy()))]
    badp = [j for j in range(batch) if not np.array_equal(got_p[j*n:(j+1)*n], cx.inv(orc.pointwise(cx.fwd(b[j*n:(j+1)*n].copy()), cx.fwd(b[j*n:(j+1)*n].copy()), q)))]
    print("seed %d: forward transform wrong polynomials %d, product wrong polynomials %d" % (seed, len(badf), len(badp)))
    if "flush" in between:
        big = torch.empty(1 << 28, dtype=torch.int64, device="cuda:0"); big.fill_(1); torch.cuda.synchronize(); del big
