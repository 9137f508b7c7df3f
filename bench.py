#!/usr/bin/env python3
"""bench.py -- batched forward NTTs/s at N=2^14, ~50-bit q, on MI355X (and BASELINE's other GPU configs).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config 2|3|4|5]

Default (= --config 4, the configuration BASELINE.json's metric is quoted on): one "step" = one in-place forward
negacyclic NTT (reduced output, the semantics of the reference's fwd_ntt_radix4 / fwd_ntt_ref_harvey) over every
GPU's shard of independent polynomials, already resident in HBM: N = 2^14 coefficients modulo q = 0x7fffffffe0001
(reference tests/test_cases.h case 12, the 51-bit prime SURVEY 8d maps "50-bit q" to), batch 2^20 sharded over 8 GPUs.

  --scaling weak   (default) every GPU holds the config's per-GPU share (config 4: 2^20/8 = 131072 polynomials, 16 GiB)
  --scaling strong the config's total batch is split over the N GPUs in use

  --config 2   N=4096, 50-bit q, batch 65536, forward                      unit NTT/s,            16N bytes per unit
  --config 3   N=65536, 52-bit q, batch 8192, forward + inverse round trip unit round trips/s,    32N bytes per unit
  --config 4   (default, above)                                             unit NTT/s,            16N bytes per unit
  --config 5   N=2^17, 4-prime RNS, batch 4096 over 8 GPUs: per limb fwd(a), fwd(b), pointwise, inv
                                                                            unit RNS products/s,   4 x 56N bytes per unit
  (config 1 is the reference's own CPU case: it is a parity test, tests/test_oracle_golden.py, not a bench line)

The path shards by independent polynomials: no collective on the data path.  Two ways to run N > 1, same
shard code either way:

  * `python bench.py --gpus N` alone: ONE process drives the N devices, one HIP stream and one set of HIP events per
    device, one host wall clock around all of them (north_star: "per-GPU HIP streams only, no RCCL");
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE set): one rank per
    GPU; torch.distributed (RCCL) is used for the barrier and the MAX-reduction of the elapsed time only.

Rank 0 prints ONE JSON line: metric/value (whole-job units/s), roofline (algorithmic bytes per step / device time of a
step, HIP events on the launch stream, with mean, min and median over the K steps) and, at N=1, the CPU baseline:
the reference's own radix-4 functions (oracle/_ref) timed on this box's host cores by a pthread harness.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
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
TRAFFIC_DIR = os.path.join(ROOT, "profiles", "r06")
TRAFFIC_JSON = os.path.join(TRAFFIC_DIR, "pmc_traffic.json")      # the headline launch (config 4); pmc_traffic_config{2,3,5}.json beside it


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def traffic_lib_sha256(path=None):
    """sha256 of the libntt_mi355x.so the committed counter passes were taken on (written by tools/pmc_traffic_json.py), or None"""
    try:
        with open(path or TRAFFIC_JSON) as f:
            return json.load(f).get("lib_sha256")
    except Exception:
        return None


# --------------------------------------------------------------------------------------------------------------
# workloads = BASELINE.json configs 2..5 on one GPU's shard
# --------------------------------------------------------------------------------------------------------------
class Workload:
    """What one step does, in which unit it is counted and how many algorithmic HBM bytes a unit moves."""

    def __init__(self, config, logn, kind, total_batch, shards, unit, metric, bytes_per_unit, cpu_op, limbs=1, qbits=None,
                 q=None, root=None, note=""):
        self.config, self.logn, self.n, self.kind = config, logn, 1 << logn, kind
        self.total_batch, self.shards, self.unit, self.metric = total_batch, shards, unit, metric
        self.bytes_per_unit, self.cpu_op, self.limbs = bytes_per_unit, cpu_op, limbs
        self.qbits, self.q, self.root, self.note = qbits, q, root, note
        self.qs, self.roots = None, None

    def resolve(self, lib):
        """moduli and roots (generated ones: the library's prime finder / minimum-root rule)"""
        if self.qs is None:
            if self.q is not None:
                self.qs = [self.q]
                self.roots = [self.root if self.root and self.n == N else lib.min_root(self.q, self.n)]
            else:
                self.qs = [lib.find_prime(self.qbits, self.n, k) for k in range(self.limbs)]
                self.roots = [lib.min_root(q, self.n) for q in self.qs]
        return self

    def per_gpu_batch(self, scaling, n_gpus):
        if scaling == "strong":
            return max(1, self.total_batch // n_gpus)
        return max(1, self.total_batch // self.shards)


def workload_for(config, logn=None):
    if config == 4:
        w = Workload(4, LOGN, "fwd", TOTAL_BATCH, SHARDS, "NTT/s", METRIC, 16 * N, 0, q=Q, root=ROOT_W)
        if logn and logn != LOGN:   # (experiments: same bytes at another size, same prime)
            scale = (1 << LOGN) / (1 << logn)
            w = Workload(4, logn, "fwd", int(TOTAL_BATCH * scale), SHARDS, "NTT/s",
                         METRIC.replace("2^14", "2^%d" % logn), 16 << logn, 0, q=Q)
        return w
    if config == 2:
        return Workload(2, 12, "fwd", 65536, 1, "NTT/s",
                        "batched forward NTTs/sec at N=2^12, 50-bit q, batch 65536; achieved HBM GB/s vs peak", 16 << 12, 0,
                        qbits=50)
    if config == 3:
        return Workload(3, 16, "roundtrip", 8192, 1, "round trips/s",
                        "forward+inverse NTT round trips/sec at N=2^16, 52-bit q, batch 8192; achieved HBM GB/s vs peak",
                        32 << 16, 1, qbits=52)
    if config == 5:
        return Workload(5, 17, "rns_product", 4096, 8, "RNS products/s",
                        "FHE-style RNS negacyclic products/sec (per limb: fwd a, fwd b, pointwise, inv) at N=2^17, 4 primes, "
                        "batch 4096 over 8 GPUs; achieved HBM GB/s vs peak", 4 * (56 << 17), 2, limbs=4, qbits=50)
    if config == 60:
        # not a BASELINE configuration: the shape of a bootstrappable CKKS parameter set (N = 2^16, 60-bit primes), which the
        # FP64 policies cannot serve -- the wide integer policy's XCD-local launch, reported beside the BASELINE configs
        return Workload(60, 16, "fwd", 8192, 1, "NTT/s",
                        "batched forward NTTs/sec at N=2^16, 60-bit q (integer arithmetic), batch 8192; achieved HBM GB/s vs peak",
                        16 << 16, 0, qbits=60)
    raise SystemExit("bench.py: --config must be 2, 3, 4 or 5")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None when
    unlimited or unreadable.  sched_getaffinity can list every CPU of the machine while this quota is a handful."""
    def own_cgroup(controller):
        try:
            with open("/proc/self/cgroup") as f:
                for line in f:
                    parts = line.strip().split(":", 2)
                    if len(parts) == 3 and (parts[1] == controller or (controller == "" and parts[0] == "0")):
                        return parts[2]
        except OSError:
            pass
        return "/"
    cands = []
    for base in ("/sys/fs/cgroup" + own_cgroup(""), "/sys/fs/cgroup"):
        cands.append((os.path.join(base, "cpu.max"), None))
    for ctl in ("cpu,cpuacct", "cpu"):
        for base in ("/sys/fs/cgroup/%s%s" % (ctl, own_cgroup(ctl)), "/sys/fs/cgroup/%s" % ctl):
            cands.append((os.path.join(base, "cpu.cfs_quota_us"), os.path.join(base, "cpu.cfs_period_us")))
    best = None
    for qf, pf in cands:
        try:
            with open(qf) as f:
                txt = f.read().split()
            if pf is None:
                if txt[0] == "max":
                    continue
                quota, period = float(txt[0]), float(txt[1])
            else:
                quota = float(txt[0])
                with open(pf) as f:
                    period = float(f.read().split()[0])
            if quota > 0 and period > 0:
                q = quota / period
                best = q if best is None else min(best, q)
        except (OSError, ValueError, IndexError):
            continue
    return best


# --------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's own functions under a pthread harness (oracle/cpu_bench.inc)
# --------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload=None, lib=None, budget_s=10.0):
    """The reference's radix-4 CPU path on this box's host cores (bounded sample).

    kind "reference": oracle/_ref/libntt_ref.so = the reference's own src/ntt_radix4.c compiled in the build
    container; kind "port" (oracle/libntt_oracle.so, the restatement) only if that file did not travel -- the JSON
    then says so in `kind` AND in `warning`.  Both legs run in C (oracle/cpu_bench.inc): the all-core leg is one
    pthread per CPU of the process's affinity mask, pinned, each on its own slab; the single-core leg is the
    reference's MEASURE() (tests/measurements.h:38-75: 10 warm-ups, 10 x 200 calls, minimum of the means)."""
    import numpy as np
    from oracle_binding import Oracle, ptr
    w = workload or workload_for(4)
    if w.qs is None:
        if lib is None:
            import ontt
            lib = ontt.load()
        w.resolve(lib)
    n, q, root = w.n, w.qs[0], w.roots[0]      # RNS: every limb costs the same on the CPU; limb 0 is timed
    orc = Oracle()
    cx = orc.ctx(n, q, root)
    U64P = C.POINTER(C.c_uint64)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libntt_ref.so")
    if os.path.exists(ref_path):
        dll, pre, kind = C.CDLL(ref_path), "ref_", "reference"
    else:
        dll, pre, kind = orc.lib, "orc_", "port"
    single = getattr(dll, pre + "bench_single")
    single.restype = C.c_double
    single.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, U64P, U64P, U64P, U64P, C.c_int, C.c_int, C.c_int]
    threads = getattr(dll, pre + "bench_threads")
    threads.restype = C.c_int
    threads.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, U64P, U64P, U64P, U64P, C.c_int, C.c_double, C.c_uint64,
                        C.POINTER(C.c_double)]
    e, econ, einv, einvcon = [cx.table(t) for t in ("e", "econ", "einv", "einv_con")]
    tabs = (ptr(e), ptr(econ), ptr(einv), ptr(einvcon))
    # parity of the baseline itself against the oracle (one polynomial through the same entry point the harness calls)
    if kind == "reference":
        dll.ref_fwd_r4_generic.argtypes = [U64P, C.c_uint64, C.c_uint64, U64P, U64P]
        chk = orc.fill_uniform(n, q, SEED, 0)
        exp = cx.fwd(chk)
        dll.ref_fwd_r4_generic(ptr(chk), n, q, tabs[0], tabs[1])
        assert np.array_equal(chk, exp), "CPU baseline disagrees with the oracle"
    # all-core leg
    slab_ops = max(1, (2 << 20) // (8 * n * (3 if w.cpu_op == 2 else 1)))     # about 2 MiB per thread: L2-sized
    out = (C.c_double * 5)()
    # threads: one per CPU the process may run on -- the affinity mask, capped by the container's CPU-time quota (a box
    # can expose 256 CPUs in the mask and grant 8 CPUs' worth of time: 256 spinning threads then only throttle each other)
    quota = cpu_quota()
    want = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if quota is not None:
        want = max(1, min(want, int(quota + 0.999)))
    # ... and by what actually scales: a one-second probe per thread count (whatever limits the box -- quota, SMT
    # siblings, a hypervisor -- the count with the highest throughput is the one the baseline is quoted on)
    probe = {}
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128, want) if c <= want} | {want})
    if len(cands) > 1:
        for c in cands:
            if threads(w.cpu_op, n, q, cx.c.ninv, *tabs, c, min(1.0, budget_s / 8), slab_ops, out) == 0 and out[1] > 0:
                probe[c] = out[0] / out[1]
        if probe:
            want = max(probe, key=lambda c: probe[c])
    rc = threads(w.cpu_op, n, q, cx.c.ninv, *tabs, want, budget_s, slab_ops, out)
    if rc != 0:
        raise RuntimeError("cpu baseline harness failed (%d)" % rc)
    ops, secs, nthr, allowed, online = out[0], out[1], int(out[2]), int(out[3]), int(out[4])
    # single-core leg: the reference's repetition counts where they fit the time budget (about 6 s), fewer inner
    # calls for the long transforms -- the counts used are stated in `sample`
    est_us = 6e-3 * n * w.logn / 14 * (1, 3, 5)[w.cpu_op]        # rough: 90 us per 2^14-point transform
    inner = int(max(10, min(200, 6e6 / (10 * est_us))))
    ns = single(w.cpu_op, n, q, cx.c.ninv, *tabs, 10, 10, inner)
    per_limb = ops / secs
    value = per_limb / w.limbs                                    # RNS product = `limbs` limb-products
    what = {0: "forward radix-4 NTTs (fwd_ntt_radix4, reduced output)",
            1: "forward+inverse radix-4 round trips (fwd_ntt_radix4 + inv_ntt_radix4)",
            2: "limb-products (fwd_ntt_radix4 x2, 128-bit pointwise %, inv_ntt_radix4)"}[w.cpu_op]
    res = {"value": value, "unit": w.unit, "cores": nthr, "threads": nthr, "cpus_allowed": allowed, "cpus_online": online,
           "cpu_quota": quota, "thread_count_probe_ops_per_s": {str(c): v / w.limbs for c, v in probe.items()},
           "cpu_model": cpu_model(), "kind": kind, "single_core_us": ns / 1e3 * w.limbs,
           "all_core_over_single_core": per_limb * ns * 1e-9,
           "sample": "%d %s at N=2^%d, q=%s in %.1f s on %d pthreads (sched_getaffinity: %d of %d online CPUs; cgroup CPU quota: %s), "
                     "%d-op slabs per thread; single core: %d warm-ups, 10 x %d calls, min of means = %.1f us per call"
                     % (int(ops), what, w.logn, hex(q), secs, nthr, allowed, online,
                        "none" if quota is None else "%.1f CPUs" % quota, slab_ops, 10, inner, ns / 1e3)}
    res["sample"] += ("; the %s is compiled gcc -O3 -march=x86-64-v3 in the build container (oracle/Makefile) and travels as a "
                      "binary -- built off-box, not -march=native on this host (scalar 64-bit code: the difference is small)"
                      % ("reference" if kind == "reference" else "restatement"))
    if w.limbs > 1:
        res["sample"] += "; an RNS product = %d limb-products" % w.limbs
    if kind != "reference":
        res["warning"] = "oracle/_ref/libntt_ref.so did not travel: this is the oracle RESTATEMENT, not the compiled reference"
    return res


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


def measured_traffic_config(config, batch, n, layout=None):
    """the same for one step of BASELINE config 2, 3 or 5 (also_configN blocks): bytes of ALL kernels of a step (FETCH_SIZE x2 +
    WRITE_SIZE summed over the step's launches, tools/pmc_traffic_json.py), valid for the batch and size the passes were taken on"""
    try:
        with open(os.path.join(TRAFFIC_DIR, "pmc_traffic_config%d%s.json" % (config, "_batch_major" if layout else ""))) as f:
            t = json.load(f)
        if t["batch"] == batch and t["N"] == n:
            return t["hbm_bytes_per_step"]
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
    """config 4: polynomials per GPU: weak = the fixed 2^20/8 share, strong = 2^20 split over the GPUs in use
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


# --------------------------------------------------------------------------------------------------------------
# one GPU's shard
# --------------------------------------------------------------------------------------------------------------
class GpuShard:
    """One GPU's shard: plans (replicated tables), resident coefficients, its own stream and events.

    kind "fwd": one in-place forward transform per step (configs 2, 4); "roundtrip": forward then inverse in place
    (config 3); "rns_product": c = a * b per limb over [limb][batch][N] slabs (config 5) -- the product overwrites
    its operands, so every step (warm-ups included) gets its own operand pair, generated before the timed region."""

    def __init__(self, lib, device, index, batch, n=None, q=Q, root=None, kind="fwd", qs=None, roots=None, steps_total=1, layout=None,
                 warm_steps=None):
        self.lib, self.device, self.index, self.batch, self.kind = lib, device, index, batch, kind
        self.layout = layout                          # None: [limb][batch][N]; "batch_major": SURVEY 8(d)'s [batch][prime][N]
        self.n = n or N
        self.qs = qs or [q]
        self.roots = roots or [root or ROOT_W]
        self.plans = [lib.Plan(self.n, qq, rr, device=device) for qq, rr in zip(self.qs, self.roots)]
        self.plan = self.plans[0]
        self.q = self.qs[0]
        self.limbs = len(self.qs)
        self.slab = batch * self.n                     # words of one limb
        # (limb stride, polynomial stride) in words, for the *_strided entry points
        self.strides = (self.n, self.limbs * self.n) if layout == "batch_major" else None
        words = self.limbs * self.slab
        # rns_product: every TIMED step and the first (parity-checked) warm-up get an operand pair of their own; the other warm-ups
        # share one (what a product leaves in its operands are canonical residues again: valid inputs of unknown meaning)
        self.warm_steps = steps_total if warm_steps is None else warm_steps
        self.warm_sets = min(self.warm_steps, 2)
        self.sets = (steps_total - self.warm_steps + self.warm_sets) if kind == "rns_product" else 1
        need = words * 8 * (2 * self.sets + 1 if kind == "rns_product" else 1)
        if need > 230 * 2**30:
            raise SystemExit("bench.py: config needs %.0f GiB of operands on one GPU (reduce --steps)" % (need / 2**30))
        self.buf = lib.DeviceBuffer(words * (2 * self.sets if kind == "rns_product" else 1), device=device)
        self.out = lib.DeviceBuffer(words, device=device) if kind == "rns_product" else None
        h = C.c_void_p()
        lib._check(lib._lib.ntt_stream_create(device, C.byref(h)))
        self.stream = h.value
        self.ev0, self.ev1 = lib.Event(device), lib.Event(device)
        self.step_events = []
        self.step_no = 0

    # ---- operands -----------------------------------------------------------------------------------------
    def operand_ptr(self, s, which):
        """rns_product: operand `which` (0 = a, 1 = b) of step s"""
        return self.buf.ptr + 8 * (2 * s + which) * self.limbs * self.slab

    def fill(self):
        # synthetic, device-generated, shard-distinct inputs: a[p][i] = splitmix64(seed ^ (offset+i)) mod q
        _, offset = shard_of_rank(self.index, self.batch, self.n)
        if self.kind != "rns_product":
            self.lib.fill_uniform(self.buf.ptr, self.slab, self.q, SEED, offset, device=self.device, stream=self.stream)
            return
        for s in range(self.sets):
            for which in (0, 1):
                if self.layout == "batch_major":
                    # one fill for the interleaved slab: values below every limb's modulus
                    self.lib.fill_uniform(self.operand_ptr(s, which), self.limbs * self.slab, min(self.qs), SEED + 2 * s + which, offset,
                                          device=self.device, stream=self.stream)
                    continue
                for l, q in enumerate(self.qs):
                    self.lib.fill_uniform(self.operand_ptr(s, which) + 8 * l * self.slab, self.slab, q, SEED + 2 * s + which,
                                          offset + l * self.slab, device=self.device, stream=self.stream)
        self.step_no = 0

    # ---- one step -------------------------------------------------------------------------------------------
    def launch(self):
        if self.kind == "fwd":
            self.plan.fwd(self.buf.ptr, self.batch, stream=self.stream)
        elif self.kind == "roundtrip":
            self.plan.fwd(self.buf.ptr, self.batch, stream=self.stream)
            self.plan.inv(self.buf.ptr, self.batch, stream=self.stream)
        else:
            i = self.step_no
            s = (min(i, self.warm_sets - 1) if i < self.warm_steps else i - self.warm_steps + self.warm_sets) % self.sets
            self.lib.rns_negacyclic_mul(self.plans, self.out.ptr, self.operand_ptr(s, 0), self.operand_ptr(s, 1), self.batch,
                                        stream=self.stream, layout=self.strides)
            self.step_no += 1

    def sync(self):
        self.lib.stream_sync(self.device, self.stream)

    def mark_start(self):
        self.step_events = [self.lib.Event(self.device)]
        self.step_events[0].record(self.stream)

    def mark_step(self):
        e = self.lib.Event(self.device)
        e.record(self.stream)
        self.step_events.append(e)

    def mark_stop(self):
        pass                                    # the last mark_step() is the stop event

    def kernel_ms(self, steps):
        return self.step_events[-1].elapsed_ms_since(self.step_events[0]) / max(steps, 1)

    def step_ms(self):
        ev = self.step_events
        return [ev[i + 1].elapsed_ms_since(ev[i]) for i in range(len(ev) - 1)]

    def copy_ceiling_gbs(self, reps=6):
        """GB/s of a plain in-place read-modify-write (16 B per lane, no arithmetic) over this shard's buffer,
        HIP events on the shard's stream, best of `reps` after one warm-up: a measured reference point next to the
        8 TB/s specification figure.  Runs after the timed region; mask 0 leaves the data alone."""
        words, best = self.limbs * self.slab, None
        for r in range(reps + 1):
            self.ev0.record(self.stream)
            self.lib.rmw_probe(self.buf.ptr, words, 0, device=self.device, stream=self.stream)
            self.ev1.record(self.stream)
            ms = self.ev1.elapsed_ms_since(self.ev0)
            if r and (best is None or ms < best):
                best = ms
        return words * 16 / (best * 1e-3) / 1e9

    def out_of_place_copy_gbs(self, reps=6):
        """GB/s (read + written bytes) of an out-of-place 16-byte-per-lane copy of half the shard's buffer onto the other half: the
        copy shape MI355X_MICROARCH.md quotes the achievable HBM rate for (about 6.3 TB/s).  Runs after the timed region and
        overwrites the upper half of the buffer."""
        half = ((self.limbs * self.slab) // 2) & ~1
        if half == 0 or not hasattr(self.lib, "copy_probe"):
            return None
        best = None
        for r in range(reps + 1):
            self.ev0.record(self.stream)
            self.lib.copy_probe(self.buf.ptr + 8 * half, self.buf.ptr, half, device=self.device, stream=self.stream)
            self.ev1.record(self.stream)
            ms = self.ev1.elapsed_ms_since(self.ev0)
            if r and (best is None or ms < best):
                best = ms
        return half * 16 / (best * 1e-3) / 1e9

    def shape_ceiling_gbs(self, reps=6):
        """the same with the memory shape of the 2^14 block kernels themselves (ntt_shape_probe: persistent workgroups,
        next block prefetched in registers, 16-byte accesses): the best memory-only skeleton, measured in this run"""
        words = (self.limbs * self.slab) & ~((1 << 14) - 1)
        if words == 0:
            return None
        best = None
        for r in range(reps + 1):
            self.ev0.record(self.stream)
            self.lib.shape_probe(self.buf.ptr, words, 0, device=self.device, stream=self.stream)
            self.ev1.record(self.stream)
            ms = self.ev1.elapsed_ms_since(self.ev0)
            if r and (best is None or ms < best):
                best = ms
        return words * 16 / (best * 1e-3) / 1e9

    def polys(self, which, base=None, limb=0):
        import numpy as np
        buf = base if base is not None else self.buf
        ls, ps = self.strides if self.strides else (self.slab, self.n)
        return np.concatenate([buf.download(self.n, limb * ls + p * ps) for p in which])

    def arith(self):
        return self.plan.info()["arith"]

    def f64_class(self):
        return self.plan.info()["f64_class"]

    def hbm_passes(self):
        return self.plan.info()["hbm_passes"]

    def close(self):
        self.buf.free()
        if self.out:
            self.out.free()
        for p in self.plans:
            p.destroy()
        self.lib._lib.ntt_stream_destroy(self.device, self.stream)


def run_steps(shards, steps, warmup, barrier, after_first_warmup=None):
    """W untimed steps, then exactly K timed steps on every shard, bracketed by barrier() on both sides.
    Returns (host seconds for the K steps, [average device ms per step and shard])."""
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
            if hasattr(s, "mark_step"):
                s.mark_step()
    for s in shards:
        s.mark_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [s.kernel_ms(steps) for s in shards]


def side_forward(lib, shard, q, steps, label):
    """the same forward launch on another prime, same buffer, same batch, HIP events on the same stream, after the
    timed region.  Not the headline."""
    plan = lib.Plan(shard.n, q, lib.min_root(q, shard.n), device=shard.device)
    lib.fill_uniform(shard.buf.ptr, shard.batch * shard.n, q, SEED, 0, device=shard.device, stream=shard.stream)
    for _ in range(2):
        plan.fwd(shard.buf.ptr, shard.batch, stream=shard.stream)
    shard.ev0.record(shard.stream)
    for _ in range(steps):
        plan.fwd(shard.buf.ptr, shard.batch, stream=shard.stream)
    shard.ev1.record(shard.stream)
    ms = shard.ev1.elapsed_ms_since(shard.ev0) / steps
    plan.destroy()
    gbs = shard.batch * 16 * shard.n / (ms * 1e-3) / 1e9
    return {"q": hex(q), "what": label, "value": shard.batch / (ms * 1e-3), "unit": "NTT/s", "kernel_ms": ms,
            "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}


def literal_50_bit(lib, shard, steps):
    """BASELINE.json's metric says "50-bit q"; SURVEY 8d maps it to the reference's case 12, a 51-bit prime, which is
    what `value` is measured on.  For the record, the same launch on the largest prime BELOW 2^50 with 2N | q-1
    (one more bit of FP64 headroom: a reduction schedule with fewer reducing stages)."""
    return side_forward(lib, shard, lib.find_prime(50, shard.n), steps, "largest prime below 2^50")


def kernel_chain(w, arith, f64_class, hbm_passes=1, batch=None):
    """the device work of one step, for the report.  hbm_passes: ntt_plan_info()[5] -- 1 when one launch carries both passes of a
    transform above 2^14 (the XCD-local kernel; its automatic choice needs 512 polynomials), else the pass list's length"""
    pol = "ArithU64" if arith != 2 else ("ArithF64W" if f64_class == 52 else "ArithF64")
    if w is None or (w.kind == "fwd" and w.logn == LOGN):
        return kernel_name(arith), 1
    one_launch = hbm_passes == 1 and (batch is None or batch >= 512)
    if w.kind == "fwd" and w.logn > LOGN and not one_launch:
        return "per 256 MiB chunk column_kernel<%s,%d,fwd> + fused_kernel<%s,%d,fwd>" % (pol, w.logn - (12 if w.logn <= 16 else 14), pol,
                                                                                         12 if w.logn <= 16 else 14), 2 * max(1, (batch or 0) * 8 * w.n // (256 << 20))
    if w.kind == "fwd" and w.logn > LOGN:
        return "team_kernel<%s,%d,fwd> (both forward passes as items of one launch)" % (
            pol if arith == 2 else "ArithU64X", w.logn - 12), 1
    if w.kind == "fwd":
        return "fused_kernel<%s,%d,fwd>" % (pol, w.logn), 1
    if w.kind == "roundtrip":
        return ("team_kernel<%s,4,fwd> (both forward passes as items of one launch) + per 256 MiB chunk "
                "fused_kernel<%s,12,inv> + column_kernel<%s,4,inv>" % (pol, pol, pol)), 33
    return ("ONE launch over all limbs: team_product_kernel<%s,5,four,multi> (column stages of b and a, block products -- both "
            "blocks through their twelve stages, product, inverse stages --, inverse column stages of c as items of one launch, "
            "the limb in the queue entry; a^ never exists in memory)" % pol), 1


def make_report(args, n_gpus, batch, elapsed, kernel_ms, arith, hbm_passes, n=None, copy_gbs=None, workload=None,
                step_ms=None, f64_class=0, shape_gbs=None, oop_gbs=None, layout=None, shards_checked=None):
    n = n or N
    w = workload
    bytes_per_unit = w.bytes_per_unit if w else 16 * n
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * batch / (elapsed / args.steps)
    slowest = max(kernel_ms)
    achieved = batch * bytes_per_unit / (slowest * 1e-3) / 1e9
    kname, launches = kernel_chain(w, arith, f64_class, hbm_passes, batch)
    scaling = getattr(args, "scaling", "weak")
    total = w.total_batch if w else TOTAL_BATCH
    shards = w.shards if w else SHARDS
    gib = batch * n * 8 * (w.limbs if w else 1) / 2**30
    if scaling == "strong":
        share = "batch %d (%.0f GiB) split over %d GPU%s = %d per GPU (%.1f GiB)" % (
            total, total * n * 8 * (w.limbs if w else 1) / 2**30, n_gpus, "" if n_gpus == 1 else "s", batch, gib)
    elif shards > 1:
        share = "batch %d sharded over %d GPUs = %d per GPU (%.1f GiB)" % (total, shards, batch, gib)
    else:
        share = "batch %d per GPU (%.1f GiB)" % (batch, gib)
    qs = w.qs if (w and w.qs) else [Q]
    cfg = w.config if w else 4
    what = {"fwd": "forward negacyclic NTT, in place, reduced output",
            "roundtrip": "forward then inverse negacyclic NTT, in place (bit-exact round trip)",
            "rns_product": "RNS negacyclic product c = a*b over %d primes: per limb fwd(a), fwd(b), pointwise, inv"
                           % (w.limbs if w else 1)}[w.kind if w else "fwd"]
    qdesc = ", ".join(hex(q) for q in qs)
    if cfg == 4:
        qdesc += " (51-bit, reference test case 12)"
    traffic = measured_traffic(batch, kname) if (cfg == 4 and n == N) else measured_traffic_config(cfg, batch, n, layout)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "committed rocprofv3 --pmc passes of the same launch (profiles/r06/pmc_traffic*.json), "
                              "not collected in this run" if traffic is not None else None,
            "kernel": kname, "launches_per_step": launches, "kernel_ms": slowest, "kernel_ms_per_gpu": kernel_ms,
            "algorithmic_bytes_per_step": batch * bytes_per_unit,
            "algorithmic_bytes_per_unit": bytes_per_unit,
            # measured in this run, three reference points next to the 8 TB/s specification figure, slowest first:
            #   in-place read-modify-write of the same buffer by a plain grid-stride kernel (NOT a ceiling: the transform
            #   kernels' own memory shape is faster, see shape_probe);
            #   out-of-place 16-byte-per-lane copy (the shape MI355X_MICROARCH.md quotes about 6.3 TB/s for);
            #   shape_probe below
            "inplace_rmw_probe": copy_gbs, "frac_of_inplace_rmw_probe": (achieved / copy_gbs) if copy_gbs else None,
            "out_of_place_copy_probe": oop_gbs, "frac_of_out_of_place_copy_probe": (achieved / oop_gbs) if oop_gbs else None,
            # measured in this run as well (ntt_shape_probe): the memory shape of the 2^14 block kernels without their
            # arithmetic -- persistent 1024-thread workgroups, register prefetch, 16-byte accesses; null when not measured
            "shape_probe": shape_gbs,
            "best_memory_only_skeleton_frac": (shape_gbs / HBM_PEAK_GBS) if shape_gbs else None,
            "best_memory_only_skeleton_source": "ntt_shape_probe, measured in this run after the timed region" if shape_gbs else None,
            "frac_of_best_memory_only_skeleton": (achieved / shape_gbs) if shape_gbs else None}
    if traffic is not None:
        roof["traffic_over_algorithmic"] = traffic / (batch * bytes_per_unit)
    if step_ms:
        roof["step_ms_min"] = min(step_ms)
        roof["step_ms_median"] = statistics.median(step_ms)
        roof["step_ms_mean"] = sum(step_ms) / len(step_ms)
        roof["frac_at_min"] = batch * bytes_per_unit / (min(step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["frac_at_median"] = batch * bytes_per_unit / (statistics.median(step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {
        "metric": w.metric if w else METRIC, "value": value, "unit": w.unit if w else "NTT/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64" if arith == 2 else "u64", "data": "synthetic",
        # dtype names the unit the arithmetic is CARRIED in, not its precision: every coefficient is an integer below 2^53 held in a
        # double, every product is formed without rounding (error-free product + exact quotient step), and the outputs
        # are compared bit for bit with the u64 oracle -- exact modular arithmetic, the same words the reference's uint64_t code writes
        "arith_exact": True,
        "dtype_note": "u64 results, bit-exact; carried in f64 (integer-valued doubles below 2^53, no rounding anywhere)" if arith == 2
                      else "u64 Harvey/Shoup arithmetic, bit-exact",
        "parity": {"shards_checked": shards_checked, "of": n_gpus,
                   "what": "first and last polynomial of every shard against the oracle, on the first warm-up step of the benchmarked launches"},
        "config": {"workload": "config%d: %s, N=%d, q=%s, %s" % (cfg, what, n, qdesc, share),
                   "N": n, "q": hex(qs[0]) if len(qs) == 1 else [hex(q) for q in qs], "batch_per_gpu": batch,
                   "global_batch": n_gpus * batch,
                   "parallelism": "batch-sharded x%d, no collective" % n_gpus,
                   "arith": ("f64-balanced" if f64_class != 52 else "f64-balanced (both operands reduced: 2^51 < q < 2^52)")
                   if arith == 2 else "u64-shoup", "hbm_passes": hbm_passes},
        "roofline": roof,
    }


class Parity:
    """Parity spot check on the benchmarked launches themselves: the first warm-up step runs on the whole shard; the first
    and the last polynomial of the shard are compared with the oracle (no separate probe launch, so a profiler's per-kernel
    statistics contain full-size launches only).  capture() before the steps, check() after the first warm-up."""

    def __init__(self, w, shard, batch):
        self.w, self.s, self.batch, self.n = w, shard, batch, w.n
        self.before, self.mid = {}, {}

    def capture(self):
        s, w = self.s, self.w
        s.fill()
        s.sync()
        which = [0, self.batch - 1]
        if w.kind == "rns_product":
            words = s.limbs * s.slab

            class View:      # operand set 0 inside the big buffer
                def __init__(self, off):
                    self.off = off

                def download(self, cnt, offset):
                    return s.buf.download(cnt, self.off + offset)
            for l in (0, w.limbs - 1):
                self.before["a%d" % l] = s.polys(which, base=View(0), limb=l)
                self.before["b%d" % l] = s.polys(which, base=View(words), limb=l)
        else:
            self.before["a"] = s.polys(which)

    def check(self):
        import numpy as np
        from oracle_binding import Oracle
        s, w, n = self.s, self.w, self.n
        s.sync()
        orc = Oracle()
        which = [0, self.batch - 1]
        if w.kind == "fwd":
            cx = orc.ctx(n, w.qs[0], w.roots[0])
            assert np.array_equal(s.polys(which), cx.fwd(self.before["a"])), "shard %d: GPU forward NTT differs from the oracle" % s.index
        elif w.kind == "roundtrip":
            assert np.array_equal(s.polys(which), self.before["a"]), "shard %d: forward+inverse round trip is not the identity" % s.index
        else:
            for l in (0, w.limbs - 1):
                cx = orc.ctx(n, w.qs[l], w.roots[l])
                exp = cx.inv(orc.pointwise(cx.fwd(self.before["a%d" % l]), cx.fwd(self.before["b%d" % l]), w.qs[l]))
                assert np.array_equal(s.polys(which, base=s.out, limb=l), exp), "shard %d: GPU RNS product differs from the oracle" % s.index

    def check_roundtrip_forward(self):
        """config 3: the forward half against the oracle, on a separate small launch after the timed region (the timed
        steps are whole round trips)"""
        import numpy as np
        from oracle_binding import Oracle
        cx = Oracle().ctx(self.n, self.w.qs[0], self.w.roots[0])
        a = self.before["a"]
        assert np.array_equal(self.s.plan.fwd_host(a), cx.fwd(a)), "GPU forward NTT differs from the oracle"
        self.mid["fwd_checked"] = True


def parity_all(pars):
    """every shard this process drives, not shard 0 only: a wrong shard anywhere stops the run before a line is printed"""
    for p in pars:
        p.check()
    return len(pars)


def also_config(lib, config, steps=8, warmup=3, check=True, layout=None):
    """BASELINE configs 2, 3 and 5 beside the headline (only the default run is driver-timed): the
    config's one-GPU share through the same GpuShard / run_steps / Parity code as `--config N`, after the headline's timed
    region, a few steps each, parity spot-checked on the first warm-up step like the headline.  Compact block: value, unit,
    frac of 8 TB/s at the config's algorithmic bytes per unit, mean / min step time."""
    w = workload_for(config).resolve(lib)
    batch = w.per_gpu_batch("weak", 1)
    shard = GpuShard(lib, 0, 0, batch, n=w.n, kind=w.kind, qs=w.qs, roots=w.roots, steps_total=steps + max(warmup, 1), layout=layout,
                     warm_steps=max(warmup, 1))
    par = Parity(w, shard, batch)
    if check:
        par.capture()
    elapsed, kernel_ms = run_steps([shard], steps, max(warmup, 1 if check else 0), shard.sync, par.check if check else None)
    if check and w.kind == "roundtrip":
        par.check_roundtrip_forward()
    step_ms = shard.step_ms()
    gbs = batch * w.bytes_per_unit / (kernel_ms[0] * 1e-3) / 1e9
    kname, launches = kernel_chain(w, shard.arith(), shard.f64_class(), shard.hbm_passes(), batch)
    traffic = measured_traffic_config(config, batch, w.n, layout)
    out = {"value": batch / (elapsed / steps), "unit": w.unit, "frac": gbs / HBM_PEAK_GBS, "achieved_GBs": gbs,
           # HBM-side bytes of one step from the committed counter passes (FETCH_SIZE x2 + WRITE_SIZE over the step's launches;
           # profiles/r06/pmc_traffic_config*.json), and their ratio to the algorithmic bytes: > 1 = bytes moved twice
           "traffic": traffic, "traffic_over_algorithmic": (traffic / (batch * w.bytes_per_unit)) if traffic else None,
           "layout": "[batch][limb][N]" if layout else ("[limb][batch][N]" if w.kind == "rns_product" else "[batch][N]"),
           "kernel_ms": kernel_ms[0], "ms_per_step": elapsed * 1e3 / steps, "steps": steps, "warmup": max(warmup, 1 if check else 0),
           # (what the untimed steps amounted to: the first carries the parity check, the rest must outlast the ~20 ms the clocks
           # need after that idle gap -- profiles/r05/clock_settling_after_idle.txt, bench_warmup_length.txt)
           "warmup_device_ms": kernel_ms[0] * max(warmup - 1, 0),
           "step_ms_min": min(step_ms), "frac_at_min": batch * w.bytes_per_unit / (min(step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "algorithmic_bytes_per_unit": w.bytes_per_unit, "batch_per_gpu": batch, "N": w.n,
           "q": [hex(q) for q in w.qs], "parity_checked": bool(check), "kernel": kname, "launches_per_step": launches,
           "workload": w.metric}
    if config == 2:
        # what this block is NOT: a cold 20-step figure.  The same launch timed over 20 steps behind 3 warm-ups (the shape the
        # driver asks the headline for) reads 0.57-0.59 -- the clocks are still settling behind the idle gap of the parity check
        # (profiles/r05/clock_settling_after_idle.txt, bench_warmup_length.txt); `--config 2 --steps 20 --warmup 3` reproduces it
        out["steady_state"] = True
        out["cold_20_steps_3_warmups_frac"] = "0.57-0.59 (rounds 1-4, same kernel; one-millisecond steps measured inside the clock-settling window)"
    shard.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8, help="untimed steps (the first carries the parity check; eight = 50 ms of work behind it, see also_config's note on settling clocks)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config: 2, 3, 4 (default, the metric) or 5 (60: N=2^16 with a 60-bit modulus, not a BASELINE configuration)")
    ap.add_argument("--batch", type=int, default=0, help="units per GPU (default: from --config and --scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0, help="seconds of CPU work of the all-core leg of the CPU baseline (tests: less)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the side measurements (copy probe, other primes): profiler runs then see the step's kernels only")
    ap.add_argument("--logn", type=int, default=0, help="(experiments, config 4 only) other transform sizes")
    ap.add_argument("--no-also", action="store_true", help="default run: skip the also_config2/3/5 blocks")
    ap.add_argument("--layout", choices=("limb-major", "batch-major"), default="limb-major",
                    help="config 5: RNS operands as [limb][batch][N] (default) or as SURVEY 8(d) lays them out, [batch][prime][N]")
    ap.add_argument("--q", default=None,
                    help="another modulus for the chosen configuration's transforms (hex or decimal, prime with 2N | q - 1; single-prime "
                         "configs 2, 3, 4: the reference's bench walks its (q, N) cases the same way, tests/bench.c:58-140); the metric string says so")
    ap.add_argument("--also-steps", type=int, default=8, help="timed steps of each also_configN block (warm-ups covering >= 60 ms of work in front)")
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

    w = workload_for(args.config, args.logn or None)
    if args.q:
        if w.limbs != 1:
            sys.exit("bench.py: --q applies to the single-prime configurations (2, 3, 4)")
        w.q, w.root = int(args.q, 0), None
        w.metric += " [--q %s: not BASELINE's modulus]" % hex(w.q)
    w = w.resolve(lib)
    n = w.n
    batch = args.batch or w.per_gpu_batch(args.scaling, n_gpus)
    if w.kind == "rns_product" and args.steps + min(args.warmup, 2) > 24 and not args.batch:
        sys.exit("bench.py: config 5 keeps one operand pair per timed step resident: use --steps <= 22")

    have = lib.device_count()
    shards = []
    for i in range(n_local):
        index = rank if world > 1 else i
        device = local_rank if world > 1 else i
        if ndev_mod:
            device %= ndev_mod
        if device >= have:
            sys.exit("bench.py: --gpus %d but only %d HIP device(s) visible" % (n_gpus, have))
        shards.append(GpuShard(lib, device, index, batch, n=n, kind=w.kind, qs=w.qs, roots=w.roots,
                               steps_total=args.steps + max(args.warmup, 1), warm_steps=max(args.warmup, 1),
                               layout="batch_major" if (args.layout == "batch-major" and w.kind == "rns_product") else None))

    def barrier():
        for s in shards:
            s.sync()
        if dist is not None:
            dist.barrier()

    # parity spot check on the benchmarked launches themselves (class Parity): first and last polynomial of EVERY shard --
    # every rank checks the shards it drives -- against the oracle after the first warm-up step.  A rank whose shard is wrong
    # raises: under torch.distributed.run that ends the job non-zero, and no JSON line is printed.
    check = not os.environ.get("NTT_BENCH_NOCHECK")   # (ablation builds compute garbage on purpose)
    pars = [Parity(w, s, batch) for s in shards]
    par = pars[0]

    def check_all():
        parity_all(pars)

    if check:
        for p in pars:
            p.capture()
    elapsed, kernel_ms = run_steps(shards, args.steps, max(args.warmup, 1 if check else 0), barrier,
                                   check_all if check else None)
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    elapsed = allreduce_max(dist, elapsed, device="cuda" if on_gpu else None)
    # shards whose first and last polynomial were compared with the oracle, summed over the ranks (= n_gpus when all did)
    shards_checked = len(pars) if check else 0
    if dist is not None:
        import torch
        t = torch.tensor([float(shards_checked)], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        shards_checked = int(t.item())
    if dist is not None:
        # per-GPU launch times of the other ranks, for the report only
        import torch
        t = torch.tensor(kernel_ms, dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        allk = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allk, t)
        kernel_ms = [float(x.item()) for x in allk]

    also_failed = False
    if rank == 0:
        s0 = shards[0]
        if check and w.kind == "roundtrip":
            par.check_roundtrip_forward()
        # after the timed region, on shard 0's resident buffer
        copy_gbs = None if args.headline_only else s0.copy_ceiling_gbs()
        shape_gbs = None if args.headline_only else s0.shape_ceiling_gbs()
        oop_gbs = None if args.headline_only else s0.out_of_place_copy_gbs()   # (last: it overwrites half of the buffer)
        slow = max(range(len(shards)), key=lambda i: kernel_ms[i] if i < len(kernel_ms) else 0) if world == 1 else 0
        out = make_report(args, n_gpus, batch, elapsed, kernel_ms, s0.arith(), s0.hbm_passes(), n=n, copy_gbs=copy_gbs,
                          workload=w, step_ms=shards[slow].step_ms(), f64_class=s0.f64_class(), shape_gbs=shape_gbs, oop_gbs=oop_gbs,
                          layout=s0.layout, shards_checked=shards_checked)
        if s0.layout:
            out["config"]["layout"] = "[batch][limb][N]"
        # the binary this line was measured on, and whether the committed counter passes (roofline.traffic) were taken on the same one
        out["lib_sha256"] = file_sha256(lib.LIB_PATH)
        out["roofline"]["traffic_lib_sha256"] = traffic_lib_sha256()
        out["roofline"]["traffic_from_this_binary"] = out["roofline"]["traffic_lib_sha256"] == out["lib_sha256"]
        if not args.headline_only:
            if w.config == 4 and n == N:
                out["also_literal_50_bit_q"] = literal_50_bit(lib, s0, args.steps)
            if w.config == 3:
                out["also_reference_case_17"] = side_forward(lib, s0, Q, max(args.steps // 2, 2),
                                                             "forward only, 51-bit q of reference test case 17")
            if w.config == 4 and n == N and not args.no_also and not args.batch and args.scaling == "weak" and not args.q:
                # BASELINE's other GPU configurations, one GPU's share each, after the headline's timed region (untouched
                # above): the driver only runs this default command, so their numbers ride on its line.  At N > 1 they run on
                # rank 0's GPU (shard 0's device) while the other ranks wait at the closing barrier: one GPU's share is what they
                # time at every N
                if n_gpus > 1:
                    out["also_scope"] = "the also_* blocks time ONE GPU's share on rank 0's GPU (device %d), after the timed region" % s0.device
                for cfg, lay in ((2, None), (3, None), (5, None), (5, "batch_major"), (60, None)):
                    key = "also_config%d" % cfg if cfg != 60 else "also_60_bit_q_n65536"
                    if lay:
                        key += "_" + lay      # config 5 with the operands as SURVEY 8(d) lays them out: [batch][prime][N]
                    # Warm-ups cover >= 60 ms of device work in every block: the first one carries the parity check (the GPU idles
                    # while the oracle runs on the host), and for about 20 ms after an idle gap the clocks are still settling --
                    # the operation measured first in a process reads 10 % low over 20 ms, 1.5 % low over 160 ms, whichever
                    # operation it is (profiles/r05/clock_settling_after_idle.txt).  Config 2's step is one millisecond: a hundred
                    # of them, so that its mean is as steady as the others'.
                    st, wu = {2: (max(args.also_steps, 100), 80), 3: (args.also_steps, 9), 5: (args.also_steps, 18),
                              60: (args.also_steps, 12)}[cfg]
                    try:
                        out[key] = also_config(lib, cfg, steps=st, warmup=wu, check=check, layout=lay)
                    except Exception as e:      # a side block must never cost the headline line ...
                        out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                        # ... but a WRONG RESULT on a BASELINE configuration must not pass unnoticed either: flagged at the top
                        # level, and the process exits non-zero after the line is printed
                        if isinstance(e, AssertionError):
                            out["also_failed"] = True
                            also_failed = True
        # north_star: the CPU baseline "timed on the host cores of the same box in the same run" -- on every line, N > 1 included
        # (rank 0, after the timed region; the other ranks wait at the closing barrier)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, lib, budget_s=args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    barrier()
    for s in shards:
        s.close()
    if dist is not None:
        dist.destroy_process_group()
    if also_failed:
        sys.exit(4)      # a side block's parity check failed: the line above says which


if __name__ == "__main__":
    main()
