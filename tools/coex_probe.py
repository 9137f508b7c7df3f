import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l})
if order == "torch_first":
    import torch
    torch.cuda.set_device(0); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
    import ontt; lib = ontt.load()
else:
    import ontt; lib = ontt.load()
    if order == "ours_first_init": print("devices (ours):", lib.device_count())
    import torch
    x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print(order, maps())
try:
    print("device_count", lib.device_count())
    import numpy as np
    from oracle_binding import Oracle
    o = Oracle(); n, q = 256, 0x1e01; w = o.min_root(q, n); a = o.fill_uniform(n, q, 1)
    p = lib.Plan(n, q, w); print("ok", np.array_equal(p.fwd_host(a), o.ctx(n, q, w).fwd(a)))
    y = torch.arange(8, device="cuda").sum().item(); print("torch still works", y)
except Exception as e:
    print("FAILED:", e)
