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
plan, ref = lib.Plan(n, q, w, device=0), lib.Plan(n, q, w, device=0)
plan.set_option(lib.OPT_XCD_LOCAL, 1); ref.set_option(lib.OPT_XCD_LOCAL, 0)
z = lambda: torch.zeros(batch * n, dtype=torch.int64, device="cuda:0")
sa, sb, ta, tb, tc, td = z(), z(), z(), z(), z(), z()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=0)
s.wait_stream(torch.cuda.current_stream())
plan.reserve(batch, stream=s.cuda_stream)
only = sys.argv[1] if len(sys.argv) > 1 else "both"
with torch.cuda.graph(g, stream=s):
    st = torch.cuda.current_stream().cuda_stream
    ta.copy_(sa); tb.copy_(sb)
    if only in ("both", "dot"): plan.inv_product(tc.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)
    if only in ("both", "mul"): plan.fwd_mul(td.data_ptr(), ta.data_ptr(), tb.data_ptr(), batch, stream=st)
for seed in (1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    got_c, got_d = tc.cpu().numpy().view(np.uint64).copy(), td.cpu().numpy().view(np.uint64).copy()
    bad_c = [j for j in range(batch) if not np.array_equal(got_c[j*n:(j+1)*n], cx.inv(orc.pointwise(a[j*n:(j+1)*n], b[j*n:(j+1)*n], q)))] if only != "mul" else []
    bad_d = [j for j in range(batch) if not np.array_equal(got_d[j*n:(j+1)*n], orc.pointwise(cx.fwd(a[j*n:(j+1)*n]), b[j*n:(j+1)*n], q))] if only != "dot" else []
    print("seed", seed, "graph vs oracle: wrong inv_product polys", bad_c[:20], len(bad_c), "wrong fwd_mul polys", bad_d[:20], len(bad_d))
print("== direct per-chunk calls (ref plan), fresh buffers per seed, inv_product then fwd_mul")
for seed in (1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
    ref.inv_product(dc.ptr, da.ptr, db.ptr, batch)
    r1 = dc.download()
    bad1 = [j for j in range(batch) if not np.array_equal(r1[j*n:(j+1)*n], cx.inv(orc.pointwise(a[j*n:(j+1)*n], b[j*n:(j+1)*n], q)))]
    ref.fwd_mul(dc.ptr, da.ptr, db.ptr, batch)
    r2 = dc.download()
    bad2 = [j for j in range(batch) if not np.array_equal(r2[j*n:(j+1)*n], orc.pointwise(cx.fwd(a[j*n:(j+1)*n]), b[j*n:(j+1)*n], q))]
    print("seed", seed, "ref inv_product wrong polys", bad1[:10], len(bad1), "ref fwd_mul wrong polys", bad2[:10], len(bad2), "ptrs", hex(da.ptr), hex(db.ptr), hex(dc.ptr))
    for x in (da, db, dc): x.free()
print("== interleaved exactly as the test does")
for seed in (1, 2, 3):
    a = orc.fill_uniform(batch * n, q, 20 * seed); b = orc.fill_uniform(batch * n, q, 20 * seed + 1)
    sa.copy_(torch.from_numpy(a.view(np.int64))); sb.copy_(torch.from_numpy(b.view(np.int64)))
    g.replay(); torch.cuda.synchronize()
    got_c, got_d = tc.cpu().numpy().view(np.uint64).copy(), td.cpu().numpy().view(np.uint64).copy()
    da, db, dc = lib.DeviceBuffer(a.size).upload(a), lib.DeviceBuffer(b.size).upload(b), lib.DeviceBuffer(a.size)
    ref.inv_product(dc.ptr, da.ptr, db.ptr, batch)
    r1 = dc.download()
    exp_c = np.concatenate([cx.inv(orc.pointwise(a[j*n:(j+1)*n], b[j*n:(j+1)*n], q)) for j in range(batch)])
    print("seed", seed, "graph inv_product == oracle", np.array_equal(got_c, exp_c), " ref == oracle", np.array_equal(r1, exp_c),
          " first bad word graph", int(np.argmax(got_c != exp_c)) if not np.array_equal(got_c, exp_c) else -1,
          " first bad word ref", int(np.argmax(r1 != exp_c)) if not np.array_equal(r1, exp_c) else -1)
    ref.fwd_mul(dc.ptr, da.ptr, db.ptr, batch)
    for x in (da, db, dc): x.free()
