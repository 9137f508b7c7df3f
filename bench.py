#!/usr/bin/env python3
"""bench.py -- batched forward NTTs/s at N=2^14, ~50-bit q, on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one in-place forward negacyclic NTT (reduced output, the semantics of
the reference's fwd_ntt_radix4 / fwd_ntt_ref_harvey) over this rank's shard of
independent polynomials, already resident in HBM.  Workload = BASELINE.json
config 4 ("N=16384, 50-bit q, batch=2^20 sharded across 8 GPUs"): every rank holds
2^20/8 = 131072 polynomials (16 GiB) of N = 2^14 coefficients modulo
q = 0x7fffffffe0001 (reference tests/test_cases.h case 12, the 51-bit prime SURVEY
8d maps this config to) -- weak scaling, no collective on the data path.

Rank 0 prints ONE JSON line: metric/value (whole-job NTT/s), roofline of the
dominant kernel (algorithmic bytes 16*N per NTT / measured launch time, HIP events
on the launch stream) and, at N=1, the CPU baseline timed on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LOGN = 14
N = 1 << LOGN
Q = 0x7fffffffe0001          # reference tests/test_cases.h case 12 (51 bits)
ROOT_W = 83051296654         # its minimum primitive 2N-th root
TOTAL_BATCH = 1 << 20        # config 4
SHARDS = 8                   # ... across 8 GPUs -> 131072 per GPU
SEED = 0x5EED5EED
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_NTT = 16 * N       # one 8-byte read + one 8-byte write per coefficient


def cpu_baseline(budget_s=12.0):
    """Reference radix-4 CPU path on all host cores (bounded sample).

    kind "reference": oracle/_ref/libntt_ref.so = the reference's own
    src/ntt_radix4.c compiled in the build container; falls back to the oracle's
    restatement ("port") if that file did not travel."""
    import numpy as np
    from oracle_binding import Oracle, ptr
    orc = Oracle()
    cx = orc.ctx(N, Q, ROOT_W)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libntt_ref.so")
    kind = "port"
    U64P = C.POINTER(C.c_uint64)
    if os.path.exists(ref_path):
        ref = C.CDLL(ref_path)
        ref.ref_fwd_r4_batch.argtypes = [U64P, C.c_uint64, C.c_uint64, C.c_uint64, U64P, U64P]
        e, econ = cx.table("e"), cx.table("econ")
        kind = "reference"

        def run(buf, nb):
            ref.ref_fwd_r4_batch(ptr(buf), nb, N, Q, ptr(e), ptr(econ))
    else:
        def run(buf, nb):
            orc.lib.orc_fwd_r4_batch(ptr(buf), nb, cx.h)
    cores = os.cpu_count() or 1
    per_thread = 16                       # polynomials per call (2 MiB, stays in L2)
    bufs = [orc.fill_uniform(per_thread * N, Q, SEED, t * per_thread * N) for t in range(cores)]
    # parity of the baseline itself against the oracle (first polynomial)
    chk = bufs[0][:N].copy()
    run(chk, 1)
    assert np.array_equal(chk, cx.fwd(bufs[0][:N])), "CPU baseline disagrees with the oracle"
    counts = [0] * cores
    stop = time.perf_counter() + budget_s

    def worker(t):
        while time.perf_counter() < stop:
            run(bufs[t], per_thread)      # ctypes releases the GIL
            counts[t] += per_thread

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    total = sum(counts)
    # single-core figure with the reference's own methodology (min of means, tests/measurements.h:57-75)
    one = bufs[0][:N].copy()
    best = 1e9
    for _ in range(5):
        s = time.perf_counter()
        for _ in range(20):
            run(one, 1)
        best = min(best, (time.perf_counter() - s) / 20)
    return {"value": total / dt, "unit": "NTT/s", "cores": cores, "kind": kind,
            "sample": "%d forward radix-4 NTTs (N=2^14, q=0x7fffffffe0001, reduced output) in %.1f s on %d threads; "
                      "single core %.1f us/NTT" % (total, dt, cores, best * 1e6)}


def measured_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01/pmc_traffic.json:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), valid for the default workload only."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t["batch"] == batch and t["N"] == N:
            return t["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def shard_of_rank(rank, per_gpu_batch, n):
    """Weak-scaling shard of the global batch: rank r owns polynomials
    [r*per_gpu_batch, (r+1)*per_gpu_batch); returns (first polynomial, first coefficient
    index) -- the latter seeds the device-side generator so shards are distinct and any
    polynomial can be regenerated on a host.  No data moves between ranks."""
    first = rank * per_gpu_batch
    return first, first * n


def allreduce_max(dist, value, device=None):
    """max over ranks of a python float (the bench contract's elapsed time); identity without dist"""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=TOTAL_BATCH // SHARDS, help="polynomials per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--logn", type=int, default=LOGN, help="(experiments) other transform sizes; the metric is quoted on 14")
    args = ap.parse_args()
    global N, BYTES_PER_NTT, ROOT_W
    if args.logn != LOGN:
        import ontt as _o
        N = 1 << args.logn
        BYTES_PER_NTT = 16 * N
        ROOT_W = _o.load().min_root(Q, N)
        if args.batch == TOTAL_BATCH // SHARDS:
            args.batch = (TOTAL_BATCH // SHARDS) * (1 << LOGN) // N   # same bytes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev_mod = int(os.environ.get("NTT_BENCH_DEVICE_MOD", "0"))   # test hook: fold ranks onto fewer GPUs
    if ndev_mod:
        local_rank %= ndev_mod
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")      # "gloo" only for the folded test above
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    import numpy as np
    import ontt
    lib = ontt.load()
    dev = local_rank if world > 1 else 0
    batch = args.batch

    plan = lib.Plan(N, Q, ROOT_W, device=dev)
    buf = lib.DeviceBuffer(batch * N, device=dev)
    stream_h = C.c_void_p()
    lib._check(lib._lib.ntt_stream_create(dev, C.byref(stream_h)))
    stream = stream_h.value
    # synthetic, device-generated, rank-distinct inputs: a[p][i] = splitmix64(seed ^ (offset+i)) mod q
    _, offset = shard_of_rank(rank, batch, N)
    lib.fill_uniform(buf.ptr, batch * N, Q, SEED, offset, device=dev, stream=stream)
    lib.stream_sync(dev, stream)

    def barrier():
        lib.stream_sync(dev, stream)
        if dist is not None:
            dist.barrier()

    # parity spot check on the benchmarked launch itself: the first warm-up step transforms the whole
    # shard; polynomial 0 and the last one are compared with the oracle (no separate probe launch, so
    # the profiler's per-kernel statistics contain full-size launches only)
    check = rank == 0 and not os.environ.get("NTT_BENCH_NOCHECK")   # (ablation builds compute garbage on purpose)
    if check:
        a0 = buf.download(N, 0)
        a1 = buf.download(N, (batch - 1) * N)
    for i in range(max(args.warmup, 1 if check else 0)):
        plan.fwd(buf.ptr, batch, stream=stream)
        if check and i == 0:
            lib.stream_sync(dev, stream)
            from oracle_binding import Oracle
            cx = Oracle().ctx(N, Q, ROOT_W)
            got = np.concatenate([buf.download(N, 0), buf.download(N, (batch - 1) * N)])
            assert np.array_equal(got, cx.fwd(np.concatenate([a0, a1]))), "GPU forward NTT differs from the oracle"
    ev0, ev1 = lib.Event(dev), lib.Event(dev)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        plan.fwd(buf.ptr, batch, stream=stream)
    ev1.record(stream)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev1.elapsed_ms_since(ev0) / max(args.steps, 1)   # avg launch duration on the launch stream
    elapsed = allreduce_max(dist, elapsed, device="cuda" if dist is not None and dist.get_backend() == "nccl" else None)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * batch / (elapsed / args.steps)
        achieved = batch * BYTES_PER_NTT / (kernel_ms * 1e-3) / 1e9
        info = plan.info()
        out = {
            "metric": "batched forward NTTs/sec at N=2^14, 50-bit q; achieved HBM GB/s vs peak",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if info["arith"] == 2 else "u64", "data": "synthetic",
            "config": {"workload": "config4: forward negacyclic NTT, N=16384, q=0x7fffffffe0001 (51-bit, reference "
                                   "test case 12), batch 2^20 sharded over 8 GPUs = 131072 polynomials (16 GiB) per GPU, "
                                   "in place, reduced output",
                       "N": N, "q": hex(Q), "batch_per_gpu": batch, "global_batch": world * batch,
                       "parallelism": "batch-sharded x%d, no collective" % world,
                       "arith": "f64-balanced" if info["arith"] == 2 else "u64-shoup", "hbm_passes": info["hbm_passes"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(batch),
                         "kernel": "fused_kernel<%s,14,fwd>" % ("ArithF64" if info["arith"] == 2 else "ArithU64"),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": batch * BYTES_PER_NTT},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    barrier()
    buf.free()
    plan.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
