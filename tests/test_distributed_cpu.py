"""CPU, world_size 2, gloo: the N>1 path of bench.py -- batch sharding by rank (no data-path
collective), barrier, MAX-reduction of the elapsed time -- with the CPU emulation of the kernel
templates standing in for the GPU on each rank.  The union of the ranks' outputs must equal the
oracle's transform of the whole global batch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

M, Q, W = 10, 0x7ffe0001, None      # BASELINE config 1 size: N=1024, 31-bit q
PER_RANK = 6
SEED = 0x5EED5EED


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import bench
    from emu_binding import Emu
    from oracle_binding import Oracle
    orc, emu = Oracle(), Emu()
    n = 1 << M
    w = orc.min_root(Q, n)
    first, offset = bench.shard_of_rank(rank, PER_RANK, n)
    assert first == rank * PER_RANK and offset == first * n
    a = orc.fill_uniform(PER_RANK * n, Q, SEED, offset)          # same generator as ntt_fill_uniform
    dist.barrier()
    rc, out = emu.transform(a, M, Q, w, 1)                        # FP64 policy, fused pass
    assert rc == 0
    rc, back = emu.transform(out, M, Q, w, 1, inverse=True)
    assert rc == 0 and np.array_equal(back, a)
    slowest = bench.allreduce_max(dist, 1.0 + rank)               # rank-dependent "elapsed time"
    assert slowest == float(world)
    # the only cross-rank traffic is test plumbing: gather results on rank 0 to compare with the oracle
    bufs = [torch.zeros(PER_RANK * n, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(out.view(np.int64)), bufs, dst=0)
    if rank == 0:
        got = np.concatenate([b.numpy().view(np.uint64) for b in bufs])
        whole = orc.fill_uniform(world * PER_RANK * n, Q, SEED, 0)
        expect = orc.ctx(n, Q, w).fwd(whole)
        assert np.array_equal(got, expect)
        open(os.path.join(outdir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


class _EmuShard:
    """test double for bench.GpuShard: the CPU emulation of the kernel templates stands in for one GPU"""

    def __init__(self, index, batch):
        from emu_binding import Emu
        from oracle_binding import Oracle
        self.orc, self.emu, self.index, self.batch = Oracle(), Emu(), index, batch
        self.n = 1 << M
        self.w = self.orc.min_root(Q, self.n)
        self.t0 = self.t1 = 0.0
        self.launches = 0

    def fill(self):
        import bench
        _, offset = bench.shard_of_rank(self.index, self.batch, self.n)
        self.a = self.orc.fill_uniform(self.batch * self.n, Q, SEED, offset)

    def launch(self):
        rc, self.a = self.emu.transform(self.a, M, Q, self.w, 1)
        assert rc == 0
        self.launches += 1

    def sync(self):
        pass

    def mark_start(self):
        import time
        self.t0 = time.perf_counter()

    def mark_stop(self):
        import time
        self.t1 = time.perf_counter()

    def kernel_ms(self, steps):
        return (self.t1 - self.t0) * 1e3 / steps


def test_single_process_multi_shard_path():
    """`bench.py --gpus 2` without a launcher: run_steps() drives two shards from one process and
    make_report() states n_gpus == 2; the union of the shards' outputs is the oracle's transform of the
    global batch.  Same functions as the GPU run, with emulator shards in place of GpuShard."""
    import argparse
    import bench
    from oracle_binding import Oracle
    shards = [_EmuShard(i, PER_RANK) for i in range(2)]
    syncs = []
    elapsed, kms = bench.run_steps(shards, steps=1, warmup=0, barrier=lambda: syncs.append(1))
    assert len(syncs) == 2 and [s.launches for s in shards] == [1, 1] and len(kms) == 2
    args = argparse.Namespace(steps=1, warmup=0, scaling="weak")
    n = 1 << M
    rep = bench.make_report(args, 2, PER_RANK, elapsed, kms, 2, 1, n=n)
    assert rep["n_gpus"] == 2 and rep["config"]["global_batch"] == 2 * PER_RANK and rep["scaling"] == "weak"
    assert rep["value"] == pytest.approx(2 * PER_RANK / elapsed)
    assert rep["roofline"]["traffic"] is None          # not the profiled workload
    orc = Oracle()
    whole = orc.fill_uniform(2 * PER_RANK * n, Q, SEED, 0)
    expect = orc.ctx(n, Q, shards[0].w).fwd(whole)
    assert np.array_equal(np.concatenate([s.a for s in shards]), expect)


def test_eight_shards_every_one_parity_checked():
    """the driver's 8-GPU shape on the CPU: run_steps() over eight shards with the after-first-warm-up hook bench.main() installs
    (parity_all: every shard, not shard 0), make_report() carrying the count; a corrupted LAST shard must stop the run"""
    import argparse
    import bench
    from oracle_binding import Oracle
    orc = Oracle()
    n = 1 << M

    class _Par:
        def __init__(self, s):
            self.s, self.checked = s, 0

        def capture(self):
            self.s.fill()
            self.before = self.s.a.copy()

        def check(self):
            exp = orc.ctx(n, Q, self.s.w).fwd(self.before)
            assert np.array_equal(self.s.a, exp), "shard %d differs from the oracle" % self.s.index
            self.checked += 1

    shards = [_EmuShard(i, 2) for i in range(8)]
    pars = [_Par(s) for s in shards]
    for p in pars:
        p.capture()
    for s in shards:               # run_steps() fills again: same generator, same data
        s.fill = lambda: None
    elapsed, kms = bench.run_steps(shards, steps=1, warmup=1, barrier=lambda: None, after_first_warmup=lambda: bench.parity_all(pars))
    assert [p.checked for p in pars] == [1] * 8 and len(kms) == 8
    args = argparse.Namespace(steps=1, warmup=1, scaling="weak")
    rep = bench.make_report(args, 8, 2, elapsed, kms, 2, 1, n=n, shards_checked=8)
    assert rep["parity"]["shards_checked"] == 8 and rep["parity"]["of"] == 8 and rep["arith_exact"] is True and rep["dtype"] == "f64"
    # distinct shards: the generator offsets differ, so do the inputs
    assert len({p.before.tobytes() for p in pars}) == 8
    # a wrong word in the last shard is caught
    shards[7].a[5] ^= np.uint64(1)
    pars[7].before = shards[7].a.copy()
    shards[7].a[9] ^= np.uint64(1)
    with pytest.raises(AssertionError, match="shard 7"):
        bench.parity_all([_Par7(pars[7])])


class _Par7:
    """parity of an already transformed shard against a tampered input: must fail"""

    def __init__(self, p):
        self.p = p

    def check(self):
        self.p.check()


def test_scaling_modes():
    import bench
    assert bench.per_gpu_batch("weak", 1) == bench.per_gpu_batch("weak", 8) == 131072
    assert [bench.per_gpu_batch("strong", g) for g in (1, 2, 4, 8)] == [1 << 20, 1 << 19, 1 << 18, 1 << 17]
    assert bench.per_gpu_batch("weak", 4, 1 << 12) == 4 * 131072   # same bytes at another size


def test_shard_helpers():
    import bench
    assert bench.shard_of_rank(0, 131072, 1 << 14) == (0, 0)
    assert bench.shard_of_rank(7, 131072, 1 << 14) == (7 * 131072, 7 * 131072 << 14)
    assert bench.allreduce_max(None, 3.5) == 3.5


def test_bench_config_workloads():
    """bench.py --config 2|3|4|5: units, algorithmic bytes per unit and per-GPU shares of every BASELINE config"""
    import bench
    w4 = bench.workload_for(4)
    assert (w4.n, w4.kind, w4.unit, w4.bytes_per_unit, w4.metric) == (1 << 14, "fwd", "NTT/s", 16 << 14, bench.METRIC)
    assert w4.per_gpu_batch("weak", 1) == w4.per_gpu_batch("weak", 8) == 131072 and w4.per_gpu_batch("strong", 4) == 1 << 18
    w2 = bench.workload_for(2)
    assert (w2.n, w2.kind, w2.bytes_per_unit, w2.per_gpu_batch("weak", 1)) == (4096, "fwd", 16 * 4096, 65536)
    w3 = bench.workload_for(3)
    assert (w3.n, w3.kind, w3.unit, w3.bytes_per_unit, w3.per_gpu_batch("weak", 1)) == (65536, "roundtrip", "round trips/s", 32 * 65536, 8192)
    w5 = bench.workload_for(5)
    assert (w5.n, w5.kind, w5.limbs, w5.unit) == (1 << 17, "rns_product", 4, "RNS products/s")
    assert w5.bytes_per_unit == 4 * 56 * (1 << 17)                     # 56N per limb-product, four limbs
    assert w5.per_gpu_batch("weak", 8) == 512 and w5.per_gpu_batch("strong", 2) == 2048
    with pytest.raises(SystemExit):
        bench.workload_for(1)                                          # config 1 is the CPU plumbing case: a parity test, not a bench line


def test_bench_report_for_other_configs():
    """make_report on config 3 / 5 workloads: metric, unit, roofline bytes, min/median step times, no stale traffic figure"""
    import argparse
    import bench
    args = argparse.Namespace(steps=4, warmup=1, scaling="weak")
    w3 = bench.workload_for(3)
    w3.qs, w3.roots = [0xffffffff00001], [3]
    rep = bench.make_report(args, 1, 8192, 0.02, [5.0], 2, 2, n=w3.n, workload=w3, step_ms=[5.2, 4.9, 5.0, 5.1], f64_class=52)
    assert rep["unit"] == "round trips/s" and rep["metric"] == w3.metric and rep["config"]["workload"].startswith("config3")
    assert rep["value"] == pytest.approx(8192 / (0.02 / 4))
    r = rep["roofline"]
    # the traffic figure is the committed counter pass of THIS configuration and batch (profiles/r05/pmc_traffic_config3.json:
    # 2.00 x the algorithmic bytes at 2^16) -- and absent for any other batch: no stale figure
    assert r["algorithmic_bytes_per_step"] == 8192 * 32 * 65536
    assert r["traffic"] == bench.measured_traffic_config(3, 8192, 65536) and 1.9 < r["traffic"] / r["algorithmic_bytes_per_step"] < 2.1
    assert "pmc_traffic" in r["traffic_source"]
    other = bench.make_report(args, 1, 4096, 0.02, [5.0], 2, 2, n=w3.n, workload=w3, step_ms=[5.2, 4.9, 5.0, 5.1], f64_class=52)["roofline"]
    assert other["traffic"] is None and other["traffic_source"] is None
    assert r["step_ms_min"] == 4.9 and r["step_ms_median"] == pytest.approx(5.05) and r["launches_per_step"] == 33
    assert r["frac"] == pytest.approx(8192 * 32 * 65536 / 5.0e-3 / 1e9 / 8000.0)
    assert "reduced" in rep["config"]["arith"]
    w5 = bench.workload_for(5)
    w5.qs, w5.roots = [11, 13, 17, 19], [1, 1, 1, 1]
    rep = bench.make_report(args, 8, 512, 0.04, [9.0] * 8, 2, 2, n=w5.n, workload=w5, step_ms=[9.0] * 4)
    assert rep["n_gpus"] == 8 and rep["unit"] == "RNS products/s" and rep["config"]["global_batch"] == 4096
    assert rep["roofline"]["algorithmic_bytes_per_unit"] == 4 * 56 * (1 << 17) and len(rep["config"]["q"]) == 4


def test_cpu_quota_parsing(tmp_path, monkeypatch):
    """cpu_baseline sizes its thread count by the container's CPU-time quota when there is one"""
    import bench
    q = bench.cpu_quota()
    assert q is None or q > 0


def test_cpu_baseline_harness_runs_on_the_reference():
    """bench.py's CPU leg: the pthread harness around the compiled reference (oracle/_ref; the oracle restatement where
    that file did not travel): plausible numbers, the fields the judge reads, and a thread count the box can serve"""
    import os
    import bench
    r = bench.cpu_baseline(budget_s=0.4)
    assert r["kind"] in ("reference", "port") and (r["kind"] == "reference") == os.path.exists(
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libntt_ref.so"))
    assert r["unit"] == "NTT/s" and r["value"] > 1000 and r["single_core_us"] > 10
    assert 1 <= r["threads"] == r["cores"] <= r["cpus_allowed"] <= max(r["cpus_online"], r["cpus_allowed"])
    assert 0.3 < r["all_core_over_single_core"] < 1.5 * r["threads"]
    assert "pthreads" in r["sample"] and "min of means" in r["sample"]
    if r["kind"] != "reference":
        assert "warning" in r
