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
