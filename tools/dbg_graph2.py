"""tools/dbg_graph2.py MODE FORM BETWEEN: a HIP graph of the NTT-domain products (MODE dot | mul | both; FORM 1 = one launch, 0 = per chunk) replayed six
times on fresh inputs with something between the replays (BETWEEN none | alloc | direct[_other|_fwd|_same|_pw|_nodl] | flush, suffix _outside = the
input copies outside the graph); every word against the oracle.  What found the unreliable captured memset
(profiles/r05/graph_replay_control_block_clear.txt)."""
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
mode = sys.argv[1]            # which calls the graph holds: dot | mul | both ; form: 1 = one launch, 0 = per chunk
form = int(sys.argv[2])
between = sys.argv[3]         # what happens between replays: none | alloc | direct
plan, ref = lib.Plan(n, q, w, device=0), lib.Plan(n, q, w, device=0)
plan.set_option(lib.OPT_XCD_LOCAL, form); ref.set_option(lib.OPT_XCD_LOCAL, 0)
z = lambda: torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
sa, sb, ta, tb, tc, td = z(), z(), z(), z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
plan.reserve(batch, stream=s.cuda_stream)
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    if "outside" not in between: ta.copy_(sa); tb.copy_(sb)
    if mode in ("both", "dot"): plan.inv_product(tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)
    if mode in ("both", "mul"): plan.fwd_mul(td.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)
exp = {}
for seed in (1, 2, 3, 1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    if seed not in exp:
        exp[seed] = (np.concatenate([cx.inv(orc.pointwise(a[j*n:(j+1)*n], b[j*n:(j+1)*n], q)) for j in range(batch)]),
                     np.concatenate([orc.pointwise(cx.fwd(a[j*n:(j+1)*n]), b[j*n:(j+1)*n], q) for j in range(batch)]))
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    if "outside" in between:
        ta.copy_(sa); tb.copy_(sb)
    g.replay(); torch.cuda.synchronize()
    got_c, got_d = tc.cpu().numpy().view(np.uint64).copy(), td.cpu().numpy().view(np.uint64).copy()
    msg = "seed %d:" % seed
    if mode != "mul":
        bad = np.nonzero(got_c != exp[seed][0])[0]
        msg += " inv_product wrong words %d (polys %s, first %s)" % (bad.size, sorted(set((bad // n).tolist()))[:6], bad[:6].tolist())
    if mode != "dot":
        bad = np.nonzero(got_d != exp[seed][1])[0]
        msg += " fwd_mul wrong words %d (polys %s, first %s)" % (bad.size, sorted(set((bad // n).tolist()))[:6], bad[:6].tolist())
    print(msg)
    if between == "alloc":
        x = lib.DeviceBuffer(a.size).upload(a); x.free()
    elif between.startswith("direct"):
        a9 = orc.fill_uniform(batch * n, q, 999) if "other" in between else a
        da, db, dc = lib.DeviceBuffer(a.size).upload(a9), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
        if "fwd" in between: ref.fwd(da.ptr, batch)
        elif "same" in between: plan.inv_product(dc.ptr, da.ptr, db.ptr, batch)
        elif "pw" in between: ref.pointwise_mul(dc.ptr, da.ptr, db.ptr, batch)
        else: ref.inv_product(dc.ptr, da.ptr, db.ptr, batch)
        if "nodl" not in between: dc.download()
        for x in (da, db, dc): x.free()
    elif "flush" in between:
        big = torch.empty(1 << 28, dtype=torch.int64, device="cuda:0"); big.fill_(1); torch.cuda.synchronize(); del big
