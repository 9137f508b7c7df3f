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


def test_shard_helpers():
    import bench
    assert bench.shard_of_rank(0, 131072, 1 << 14) == (0, 0)
    assert bench.shard_of_rank(7, 131072, 1 << 14) == (7 * 131072, 7 * 131072 << 14)
    assert bench.allreduce_max(None, 3.5) == 3.5
