import sys, os
sys.path.insert(0, os.getcwd())
import ontt
lib = ontt.load()
N, Q = 1 << 14, 0x7fffffffe0001
plan = lib.Plan(N, Q, lib.min_root(Q, N))
batch = 32768
bufs = [lib.DeviceBuffer(batch * N) for _ in range(3)]
for i, b in enumerate(bufs[:2]): lib.fill_uniform(b.ptr, batch * N, Q, 5 + i)
for g in range(3):
    plan.pointwise_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch)
lib.stream_sync()
e0, e1 = lib.Event(), lib.Event()
e0.record()
for _ in range(10): plan.pointwise_mul(bufs[2].ptr, bufs[0].ptr, bufs[1].ptr, batch)
e1.record()
ms = e1.elapsed_ms_since(e0) / 10
print("pointwise %.3f ms  %.0f GB/s (24 B per coefficient)" % (ms, 24 * batch * N / ms / 1e6))
