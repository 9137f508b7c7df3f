"""CPU model of the in-launch queue protocol of the XCD-local kernels (team_kernel, team_product_kernel; csrc/ntt_kernels_team.h, ntt_kernels_products.h).

The device code that turns a queue entry into an item -- csrc/ntt_core.h team_decode -- is compiled for the host
(tests/emu) and driven here by a simulation of what the kernels do around it: workgroups resident on 1..8 XCDs claim queues
(compare-and-swap on first touch, own queue first, then whatever nobody claimed), pull entries from an atomic counter,
wait for the hand-off counter of their polynomial when the item belongs to a later pass, and signal completion of an item
at the TOP of their next iteration (as the kernels do).  A random scheduler interleaves the workgroups arbitrarily.
Checked: every (polynomial, pass, item) is executed exactly once; an item of pass P > 0 never starts before all items of
pass P - 1 of its polynomial have signalled; nobody ever waits for an item that has not been handed out (so the protocol
cannot deadlock whatever the residency: a single resident workgroup completes the launch); every queue is processed by
exactly one XCD, also when fewer than eight XCDs answer; ragged batches and batches smaller than eight."""
import random

import pytest

from emu_binding import Emu


@pytest.fixture(scope="module")
def emu():
    return Emu()


class Launch:
    """state of one launch: the control block of the kernels (next[], owner[], done[]) and the execution log"""

    def __init__(self, emu, total, lag, n):
        self.emu, self.total, self.lag, self.n = emu, total, lag, n          # n = (n0, n1, n2), n2 = 0: two passes
        self.npass = 3 if n[2] else 2
        self.next = [0] * 8
        self.owner = [0] * 8
        self.done = [0] * (self.npass * total)                             # done[(pass) * total + v]: signalled items
        self.handed = set()                                                  # (v, pass, item) handed out so far
        self.executed = []
        self.queue_xcd = {}

    def decode(self, k, q):
        return self.emu.team_decode(k, q, self.total, self.lag, *self.n)


class Workgroup:
    """one workgroup as a state machine; step() advances it to its next blocking point or by one item"""

    def __init__(self, launch, xcd):
        self.L, self.xcd = launch, xcd
        self.qq, self.q, self.sig, self.state, self.cur = 0, None, None, "claim", None
        self.finished = False

    def runnable(self):
        if self.finished:
            return False
        if self.state == "wait":
            v, ps, item = self.cur
            return self.L.done[(ps - 1) * self.L.total + v] >= self.L.n[ps - 1]
        return True

    def step(self):
        L = self.L
        if self.state == "claim":
            if self.qq == 8:
                self.finished = True
                return
            q = (self.xcd + self.qq) & 7
            self.qq += 1
            if L.owner[q] in (0, self.xcd + 1):                       # atomicCAS(owner, 0, my + 1): unclaimed or mine
                L.owner[q] = self.xcd + 1
                L.queue_xcd.setdefault(q, set()).add(self.xcd)
                self.q, self.sig, self.state = q, None, "fetch"
            return
        if self.state == "fetch":
            if self.sig is not None:                                      # the finished item's signal rides at the top
                L.done[self.sig] += 1
                self.sig = None
            k = L.next[self.q]
            L.next[self.q] += 1
            stop, valid, ps, item, v = L.decode(k, self.q)
            if stop:
                self.state = "claim"
                return
            if not valid:
                return                                                    # a hole: fetch again
            assert v % 8 == self.q and v < L.total, "a queue only ever sees its own polynomials"
            assert (v, ps, item) not in L.handed, "item handed out twice"
            L.handed.add((v, ps, item))
            self.cur = (v, ps, item)
            if ps > 0:
                # whoever waits, waits for items that are already in somebody's hands
                for i in range(L.n[ps - 1]):
                    assert (v, ps - 1, i) in L.handed, "an item waits for one that was never handed out: deadlock possible"
                self.state = "wait"
            else:
                self.state = "run"
            return
        if self.state == "wait":
            self.state = "run"
            return
        if self.state == "run":
            v, ps, item = self.cur
            if ps > 0:
                assert L.done[(ps - 1) * L.total + v] == L.n[ps - 1], "a later pass started before the earlier one finished"
            L.executed.append(self.cur)
            if ps < L.npass - 1:
                self.sig = ps * L.total + v
            self.state = "fetch"


def simulate(emu, total, lag, n, xcds, wgs_per_xcd, seed):
    L = Launch(emu, total, lag, n)
    rng = random.Random(seed)
    wgs = [Workgroup(L, x) for x in xcds for _ in range(wgs_per_xcd)]
    while True:
        live = [w for w in wgs if not w.finished]
        if not live:
            break
        ready = [w for w in live if w.runnable()]
        assert ready, "deadlock: every resident workgroup waits"
        rng.choice(ready).step()
    return L


SHAPES = {"transform 2^15 fwd": (16, 8, 0), "transform 2^17 inv": (32, 16, 0), "product 2^16 (both operands)": (32, 16, 16),
          "product 2^15 (a^ given)": (16, 8, 16)}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("total,lag", [(1, 1), (5, 2), (8, 8), (9, 3), (37, 10), (64, 6), (24, 0 + 20)])
def test_every_item_once_and_in_order(emu, shape, total, lag):
    n = SHAPES[shape]
    npass = 3 if n[2] else 2
    for xcds, per, seed in (([0, 1, 2, 3, 4, 5, 6, 7], 3, 1), ([0, 1, 2, 3, 4, 5, 6, 7], 1, 2), ([3], 1, 3), ([0, 5], 2, 4),
                            ([1, 2, 3, 4, 6], 4, 5)):
        L = simulate(emu, total, lag, n, xcds, per, seed)
        want = {(v, ps, i) for v in range(total) for ps in range(npass) for i in range(n[ps])}
        assert len(L.executed) == len(want) and set(L.executed) == want, (shape, total, lag, xcds)
        assert all(len(s) == 1 for s in L.queue_xcd.values()), "a queue was processed by two XCDs"
        assert set(L.queue_xcd) == set(range(8)), "a queue was never claimed"


def test_single_workgroup_never_waits_forever(emu):
    """the extreme residency: ONE workgroup on one XCD runs the whole launch alone -- every item it would wait for was
    handed out to itself and finished earlier"""
    for n in SHAPES.values():
        L = simulate(emu, 19, 7, n, [6], 1, 11)
        assert len(L.executed) == 19 * sum(n)


def test_decode_holes_and_stop(emu):
    """the head and the tail of a queue: entries whose polynomial index falls outside [0, J) are holes, the queue stops after
    J + (passes - 1) * lag steps"""
    n0, n1, n2, lag, total = 4, 2, 3, 5, 20          # queue 3 owns polynomials 3, 11, 19: J = 3
    per, J = n0 + n1 + n2, 3
    for k in range((J + 2 * lag + 2) * per):
        stop, valid, ps, item, v = emu.team_decode(k, 3, total, lag, n0, n1, n2)
        step, r = divmod(k, per)
        assert stop == (step >= J + 2 * lag)
        exp_pass = 0 if r < n0 else (1 if r < n0 + n1 else 2)
        j = step - exp_pass * lag
        assert valid == (not stop and 0 <= j < J)
        if valid:
            assert (ps, item, v) == (exp_pass, r - (0, n0, n0 + n1)[exp_pass], 3 + 8 * j)
