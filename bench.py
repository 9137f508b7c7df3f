#!/usr/bin/env python3
"""bench.py -- batched forward NTTs/s at N=2^14, ~50-bit q, on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One "step" = one in-place forward negacyclic NTT (reduced output, the semantics of
the reference's fwd_ntt_radix4 / fwd_ntt_ref_harvey) over every GPU's shard of
independent polynomials, already resident in HBM.  Workload = BASELINE.json
config 4 ("N=16384, 50-bit q, batch=2^20 sharded across 8 GPUs"): N = 2^14
coefficients modulo q = 0x7fffffffe0001 (reference tests/test_cases.h case 12, the
51-bit prime SURVEY 8d maps this config to).

  --scaling weak   (default) every GPU holds 2^20/8 = 131072 polynomials (16 GiB)
  --scaling strong the 2^20 polynomials (128 GiB) are split over the N GPUs

The path shards by independent polynomials: no collective on the data path
(SURVEY 8e).  Two ways to run N > 1, same shard code either way:

  * `python bench.py --gpus N` alone: ONE process drives the N devices, one HIP
    stream and one pair of HIP events per device, one host wall clock around all
    of them (north_star: "per-GPU HIP streams only, no RCCL");
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (WORLD_SIZE set): one rank per GPU; torch.distributed (RCCL) is used for the
    barrier and the MAX-reduction of the elapsed time only.

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
METRIC = "batched forward NTTs/sec at N=2^14, 50-bit q; achieved HBM GB/s vs peak"
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r02", "pmc_traffic.json")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    return {"value": total / dt, "unit": "NTT/s", "cores": cores, "cpu_model": cpu_model(), "kind": kind,
            "sample": "%d forward radix-4 NTTs (N=2^14, q=0x7fffffffe0001, reduced output) in %.1f s on %d threads "
                      "of %s; single core %.1f us/NTT" % (total, dt, cores, cpu_model(), best * 1e6)}


def kernel_name(arith):
    return "fused_kernel<%s,14,fwd>" % ("ArithF64" if arith == 2 else "ArithU64")


def measured_traffic(batch, kernel, path=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, MI355X_MICROARCH.md section HBM).  Valid only for the kernel and batch the passes were taken
    on: anything else (another kernel symbol, another shard size) reports null instead of a stale number.
    tests/test_abi.py::test_traffic_json_matches_bench ties the file to the default workload."""
    try:
        with open(path or TRAFFIC_JSON) as f:
            t = json.load(f)
        if t["batch"] == batch and t["N"] == N and t["kernel"] == kernel:
            return t["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def shard_of_rank(rank, per_gpu_batch, n):
    """Shard of the global batch: GPU r owns polynomials [r*per_gpu_batch, (r+1)*per_gpu_batch);
    returns (first polynomial, first coefficient index) -- the latter seeds the device-side generator so
    shards are distinct and any polynomial can be regenerated on a host.  No data moves
    between GPUs."""
    first = rank * per_gpu_batch
    return first, first * n


def per_gpu_batch(scaling, n_gpus, n=N):
    """polynomials per GPU: weak = config 4's fixed 2^20/8 share, strong = 2^20 split over the GPUs in use
    (both scaled to the same bytes when --logn changes the transform size)"""
    scale = (1 << LOGN) / n
    if scaling == "strong":
        return max(1, int(TOTAL_BATCH * scale) // n_gpus)
    return max(1, int(TOTAL_BATCH // SHARDS * scale))


def allreduce_max(dist, value, device=None):
    """max over ranks of a python float (the bench contract's elapsed time); identity without dist"""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GpuShard:
    """One GPU's shard: plan (replicated tables), resident coefficients, its own stream and events."""

    def __init__(self, lib, device, index, batch, n=None, q=Q, root=None):
        self.lib, self.device, self.index, self.batch = lib, device, index, batch
        self.n = n or N
        self.plan = lib.Plan(self.n, q, root or ROOT_W, device=device)
        self.buf = lib.DeviceBuffer(batch * self.n, device=device)
        h = C.c_void_p()
        lib._check(lib._lib.ntt_stream_create(device, C.byref(h)))
        self.stream = h.value
        self.ev0, self.ev1 = lib.Event(device), lib.Event(device)
        self.q = q

    def fill(self):
        # synthetic, device-generated, shard-distinct inputs: a[p][i] = splitmix64(seed ^ (offset+i)) mod q
        _, offset = shard_of_rank(self.index, self.batch, self.n)
        self.lib.fill_uniform(self.buf.ptr, self.batch * self.n, self.q, SEED, offset, device=self.device,
                              stream=self.stream)

    def launch(self):
        self.plan.fwd(self.buf.ptr, self.batch, stream=self.stream)

    def sync(self):
        self.lib.stream_sync(self.device, self.stream)

    def mark_start(self):
        self.ev0.record(self.stream)

    def mark_stop(self):
        self.ev1.record(self.stream)

    def kernel_ms(self, steps):
        return self.ev1.elapsed_ms_since(self.ev0) / max(steps, 1)

    def copy_ceiling_gbs(self, reps=6):
        """GB/s of a plain in-place read-modify-write (16 B per lane, no arithmetic) over this shard's buffer,
        HIP events on the shard's stream, best of `reps` after one warm-up: the measured ceiling SURVEY 8d asks
        for next to the 8 TB/s specification figure.  Runs after the timed region; mask 0 leaves the data alone."""
        words, best = self.batch * self.n, None
        for r in range(reps + 1):
            self.ev0.record(self.stream)
            self.lib.rmw_probe(self.buf.ptr, words, 0, device=self.device, stream=self.stream)
            self.ev1.record(self.stream)
            ms = self.ev1.elapsed_ms_since(self.ev0)
            if r and (best is None or ms < best):
                best = ms
        return words * 16 / (best * 1e-3) / 1e9

    def polys(self, which):
        import numpy as np
        return np.concatenate([self.buf.download(self.n, p * self.n) for p in which])

    def arith(self):
        return self.plan.info()["arith"]

    def hbm_passes(self):
        return self.plan.info()["hbm_passes"]

    def close(self):
        self.buf.free()
        self.plan.destroy()
        self.lib._lib.ntt_stream_destroy(self.device, self.stream)


def run_steps(shards, steps, warmup, barrier, after_first_warmup=None):
    """W untimed steps, then exactly K timed steps on every shard, bracketed by barrier() on both sides.
    Returns (host seconds for the K steps, [average launch ms per shard])."""
    for s in shards:
        s.fill()
    for i in range(warmup):
        for s in shards:
            s.launch()
        if i == 0 and after_first_warmup:
            after_first_warmup()
    barrier()
    t0 = time.perf_counter()
    for s in shards:
        s.mark_start()
    for _ in range(steps):
        for s in shards:
            s.launch()
    for s in shards:
        s.mark_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [s.kernel_ms(steps) for s in shards]


def literal_50_bit(lib, shard, steps):
    """BASELINE.json's metric says "50-bit q"; SURVEY 8d maps it to the reference's case 12, a 51-bit prime, which is
    what `value` is measured on.  For the record, the same launch on the largest prime BELOW 2^50 with 2N | q-1
    (one more bit of FP64 headroom: a reduction schedule with fewer reducing stages), same buffer, same batch,
    HIP events on the same stream, after the timed region.  Not the headline."""
    q50 = lib.find_prime(50, shard.n)
    plan = lib.Plan(shard.n, q50, lib.min_root(q50, shard.n), device=shard.device)
    lib.fill_uniform(shard.buf.ptr, shard.batch * shard.n, q50, SEED, 0, device=shard.device, stream=shard.stream)
    for _ in range(2):
        plan.fwd(shard.buf.ptr, shard.batch, stream=shard.stream)
    shard.ev0.record(shard.stream)
    for _ in range(steps):
        plan.fwd(shard.buf.ptr, shard.batch, stream=shard.stream)
    shard.ev1.record(shard.stream)
    ms = shard.ev1.elapsed_ms_since(shard.ev0) / steps
    plan.destroy()
    gbs = shard.batch * 16 * shard.n / (ms * 1e-3) / 1e9
    return {"q": hex(q50), "value": shard.batch / (ms * 1e-3), "unit": "NTT/s", "kernel_ms": ms,
            "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}


def make_report(args, n_gpus, batch, elapsed, kernel_ms, arith, hbm_passes, n=None, copy_gbs=None):
    n = n or N
    bytes_per_ntt = 16 * n
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * batch / (elapsed / args.steps)
    slowest = max(kernel_ms)
    achieved = batch * bytes_per_ntt / (slowest * 1e-3) / 1e9
    kname = kernel_name(arith)
    if args.scaling == "strong":
        share = "batch 2^20 (128 GiB) split over %d GPU%s = %d polynomials (%.0f GiB) per GPU" % (
            n_gpus, "" if n_gpus == 1 else "s", batch, batch * n * 8 / 2**30)
    else:
        share = "batch 2^20 sharded over 8 GPUs = %d polynomials (%.0f GiB) per GPU" % (batch, batch * n * 8 / 2**30)
    return {
        "metric": METRIC, "value": value, "unit": "NTT/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64" if arith == 2 else "u64", "data": "synthetic",
        "config": {"workload": "config4: forward negacyclic NTT, N=%d, q=0x7fffffffe0001 (51-bit, reference test "
                               "case 12), %s, in place, reduced output" % (n, share),
                   "N": n, "q": hex(Q), "batch_per_gpu": batch, "global_batch": n_gpus * batch,
                   "parallelism": "batch-sharded x%d, no collective" % n_gpus,
                   "arith": "f64-balanced" if arith == 2 else "u64-shoup", "hbm_passes": hbm_passes},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": measured_traffic(batch, kname) if n == N else None,
                     "kernel": kname, "kernel_ms": slowest, "kernel_ms_per_gpu": kernel_ms,
                     "algorithmic_bytes_per_launch": batch * bytes_per_ntt,
                     # measured in this run: in-place read-modify-write of the same buffer without arithmetic
                     "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--batch", type=int, default=0, help="polynomials per GPU (default: from --scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the side measurements (copy ceiling, literal 50-bit prime): profiler runs then see one kernel")
    ap.add_argument("--logn", type=int, default=LOGN, help="(experiments) other transform sizes; the metric is quoted on 14")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    n_gpus = max(args.gpus, 1)
    n_local = 1 if world > 1 else n_gpus          # devices this process drives
    ndev_mod = int(os.environ.get("NTT_BENCH_DEVICE_MOD", "0"))   # test hook: fold shards onto fewer GPUs

    dist = None
    # (test hook NTT_BENCH_FORCE_DIST: a single rank still goes through RCCL init, barrier, MAX-reduction and gather)
    if world > 1 or os.environ.get("NTT_BENCH_FORCE_DIST"):
        # torch BEFORE the library: the torch wheel bundles its own HIP runtime under the same SONAME, the library
        # then binds to the copy torch loaded (profiles/r02/torch_runtime_coexistence.txt)
        import torch
        import torch.distributed as dist
        backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")      # "gloo" only for the folded test hook
        dev = local_rank % ndev_mod if ndev_mod else local_rank
        if backend == "nccl":
            torch.cuda.set_device(dev)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    import numpy as np
    import ontt
    lib = ontt.load()

    n, root = N, ROOT_W
    if args.logn != LOGN:
        n = 1 << args.logn
        root = lib.min_root(Q, n)
    batch = args.batch or per_gpu_batch(args.scaling, n_gpus, n)

    have = lib.device_count()
    shards = []
    for i in range(n_local):
        index = rank if world > 1 else i
        device = local_rank if world > 1 else i
        if ndev_mod:
            device %= ndev_mod
        if device >= have:
            sys.exit("bench.py: --gpus %d but only %d HIP device(s) visible" % (n_gpus, have))
        shards.append(GpuShard(lib, device, index, batch, n=n, root=root))

    def barrier():
        for s in shards:
            s.sync()
        if dist is not None:
            dist.barrier()

    # parity spot check on the benchmarked launch itself: the first warm-up step transforms the whole shard;
    # polynomial 0 and the last one of shard 0 are compared with the oracle (no separate probe launch, so the
    # profiler's per-kernel statistics contain full-size launches only)
    check = rank == 0 and not os.environ.get("NTT_BENCH_NOCHECK")   # (ablation builds compute garbage on purpose)
    before = {}

    def parity():
        s = shards[0]
        s.sync()
        from oracle_binding import Oracle
        cx = Oracle().ctx(n, Q, root)
        got = s.polys([0, batch - 1])
        assert np.array_equal(got, cx.fwd(before["a"])), "GPU forward NTT differs from the oracle"

    if check:
        shards[0].fill()
        shards[0].sync()
        before["a"] = shards[0].polys([0, batch - 1])
    elapsed, kernel_ms = run_steps(shards, args.steps, max(args.warmup, 1 if check else 0), barrier,
                                   parity if check else None)
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    elapsed = allreduce_max(dist, elapsed, device="cuda" if on_gpu else None)
    if dist is not None:
        # per-GPU launch times of the other ranks, for the report only
        import torch
        t = torch.tensor(kernel_ms, dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        allk = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allk, t)
        kernel_ms = [float(x.item()) for x in allk]

    if rank == 0:
        # after the timed region, on shard 0's resident buffer
        copy_gbs = None if args.headline_only else shards[0].copy_ceiling_gbs()
        out = make_report(args, n_gpus, batch, elapsed, kernel_ms, shards[0].arith(), shards[0].hbm_passes(), n=n,
                          copy_gbs=copy_gbs)
        if n_gpus == 1 and n == N and not args.headline_only:
            out["also_literal_50_bit_q"] = literal_50_bit(lib, shards[0], args.steps)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    barrier()
    for s in shards:
        s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
