"""Multi-GPU sharding of the MSM path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Two partitions of one MSM, both ending in the same exchange:
  * POINT RANGE: rank g owns bases / scalars [lo_g, hi_g) and computes their partial sum with the full single-GPU
    pipeline (`shard_range`);
  * WINDOWS: rank g holds, for ALL points, the window tables {w : w mod world == g} (`shard_windows`; the library
    builds exactly those when the options msm_shard_rank / msm_shard_world are set at upload) and sums those
    windows only -- the bucket-window shard BASELINE.json's north_star names.
The only exchange is one all-gather of the Jacobian partial results (3 coordinates x 48 B for BLS12-381 G1).  RCCL has no elliptic-curve reduction operator, so the "reduce" is
all-gather + a fold of `world` points on every rank (zkhip_jacobian_sum_dev).  The message is latency-bound
(about a microsecond class transfer over xGMI), nothing here is bandwidth-bound.

Independent NTT batches (KZG columns, the three witness vectors of Groth16) shard by polynomial with no
collective at all.
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of [0, n): the first n % world ranks get one extra element."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_windows(num_windows: int, rank: int, world: int) -> List[int]:
    """Pippenger windows owned by `rank` under the window partition: w = rank, rank + world, ... (the tables are
    equal-weight, so any assignment is balanced up to one window; ranks >= num_windows stay idle)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, num_windows, world))


def shard_polys(batch: int, rank: int, world: int) -> List[int]:
    """Polynomial indices of an NTT batch owned by `rank` (round-robin: equal work, no collective)."""
    return list(range(rank, batch, world))


def allgather_fold(partial, world: int, all_gather: Callable, fold: Callable, always: bool = False):
    """partial: this rank's Jacobian result as a flat tensor (device or host).
    all_gather(out, inp): the collective (torch.distributed.all_gather_into_tensor or a test double);
    fold(gathered, world): sums `world` Jacobian points (zkhip_jacobian_sum_dev on the GPU).
    always: run the collective and the fold at world == 1 too (exercises the N > 1 code path on one GPU)."""
    if world == 1 and not always:
        return partial
    gathered = partial.new_zeros(world * partial.numel())
    all_gather(gathered, partial)
    return fold(gathered, world)


class AllgatherFoldPipeline:
    """allgather_fold with the exchange of step i UNDER the multiexp of step i + 1.

    The 144-byte all-gather is latency, not bandwidth (a kernel launch on RCCL's stream + the slowest rank's skew): issued synchronously it
    sits between two multiexps on every rank.  Here step i's collective is started asynchronously (`all_gather(out, inp)` returns a handle
    with .wait(), e.g. torch.distributed.all_gather_into_tensor(..., async_op=True): RCCL's stream waits for the partial sum, the compute
    stream goes on), and its wait + fold are enqueued one step later, after the next multiexp is in the queue.  Two partial-sum buffers and
    two receive buffers alternate, so nothing a collective still reads is overwritten.  Results come out in order, one step late; `flush`
    returns the last one.  `all_gather` may also be synchronous (returns None: gloo through the host, test doubles): then this is
    allgather_fold with preallocated buffers.
    """

    def __init__(self, world: int, all_gather: Callable, fold: Callable, new_buffer: Callable):
        """new_buffer(n): a zeroed flat buffer of n elements where the partial sums live (two receive buffers are made up front)"""
        self.world, self.all_gather, self.fold = world, all_gather, fold
        self.new_buffer = new_buffer
        self.recv = None
        self.pending = None  # (handle, receive slot) of the step whose fold has not been enqueued yet
        self.slot = 0

    def push(self, partial):
        """start this step's exchange; returns the folded result of the PREVIOUS step (None at the first call)"""
        if self.recv is None:
            self.recv = [self.new_buffer(self.world * partial.numel()) for _ in range(2)]
        handle = self.all_gather(self.recv[self.slot], partial)
        done = self._finish()
        self.pending = (handle, self.slot)
        self.slot ^= 1
        return done

    def flush(self):
        """the folded result of the last pushed step (None when nothing is pending)"""
        return self._finish()

    def _finish(self):
        if self.pending is None:
            return None
        handle, slot = self.pending
        self.pending = None
        if handle is not None:
            handle.wait()
        return self.fold(self.recv[slot], self.world)
