"""Data-parallel layer of the DMVAE step (no reference counterpart: the
reference is single-process).

One process per GPU.  Every loss term is a batch mean of per-row quantities
(base_models.py:74-79, priors.py:145,199), so with equal shards the global
gradient is the mean of the per-rank gradients -- prior tables included.  The
exchange runs on the flat fp32 gradient arena (torch.distributed: backend "nccl"
= RCCL over xGMI on the GPU box, "gloo" on CPU in the tests); the 1/world factor
is folded into the Adam kernel (grad_scale), so no separate scaling pass touches
HBM.  Two forms (make_exchange picks; DMVAE_DP_MODE=sharded|allreduce overrides):

  sharded    the WEIGHT range of the arena (99.7 % of it; the arena keeps every bias
             and the prior tables in a tail of their own, dmvae_plan_grad_buckets):
             reduce-scatter(SUM) of the gradients -> TF-Adam on the OWNED 1/world slice
             (m and v are only ever touched there) -> all-gather of the updated weights,
             on bf16 plans as their bf16 SHADOW: 2 B per parameter on the wire instead of
             4, and no cast pass afterwards.  The tail -- read in fp32 by
             the epilogues and the latent kernel -- is all-reduced whole and updated on
             every rank.  A rank's fp32 weights outside its slice are then stale until
             StepEngine.sync_master() (checkpoint time).  The default for world > 1: the
             optimizer's HBM traffic (30 B per parameter) is divided by the world size.
  allreduce  ONE all-reduce(SUM), every rank applies the whole update (replicated
             Adam): the bit-simplest form, kept as the reference the sharded form is
             tested against (identical bits on the owned slice).

Both come bucketed (segments of the backward pass, each collective started right
behind its segment) or as one collective after the whole backward.  Three buckets
(a collective behind each of decoder / heads / trunk) pay three smaller dW grids +
host hand-overs per step: chosen from 64 MiB (the 4096-wide configuration).  Two
buckets (decoder + heads behind segment 1 as ONE weight-gradient launch, the trunk
behind segment 2) exist and are tested, but priced with a one-rank RCCL
communicator they cost more than they can hide at the MNIST sizes (the table at
make_exchange), so smaller arenas issue one pair after the backward.
DMVAE_DP_OVERLAP=0|1 and DMVAE_DP_BUCKETS=2|3 override (make_exchange).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_rows, rank, world):
    """Rows [lo, hi) of a global batch of n_rows owned by `rank`: equal
    contiguous shards, the remainder spread over the first ranks."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def epoch_plan(order, global_batch, rank, world):
    """The rows of one epoch this rank steps through, and how many steps that is.

    order: the epoch's global row order (identical on every rank).  Returns (local_order, n_full,
    tail, epoch_weight): `n_full` steps of global_batch / world rows each, then -- single process
    only -- one short batch of `tail` rows (includes/utils.py:462-463); epoch_weight multiplies each
    batch loss in the epoch mean (base_models.py:130: 1 / epoch_len).
    world > 1: only the len(order) // global_batch FULL global batches are trained; the ragged tail
    is dropped (its shards would be unequal), so n_full is derived from the GLOBAL row count and is
    the same on every rank -- every rank issues the same number of collectives -- and the epoch mean
    is over those n_full batches.  Raises when not even one full global batch exists."""
    order = np.asarray(order)
    B = int(global_batch)
    if B % world:
        raise ValueError("batch_size %d is not divisible by the world size %d" % (B, world))
    n_full = len(order) // B
    if world == 1:
        tail = len(order) - n_full * B
        return order, n_full, tail, 1.0 / max(1, n_full + (1 if tail else 0))
    if n_full == 0:
        raise ValueError("data parallel: %d rows do not fill one global batch of %d (the ragged tail is dropped when world > 1)"
                         % (len(order), B))
    lo, hi = shard_range(B, rank, world)
    mine = np.concatenate([order[i * B + lo: i * B + hi] for i in range(n_full)])
    return mine, n_full, 0, 1.0 / n_full


class GradExchange:
    """all-reduce(SUM) of the gradient arena; callable usable inside a captured
    HIP graph (RCCL collectives are capturable)."""

    def __init__(self, group=None):
        self.group = group
        # DMVAE_DP_FORCE=1 keeps the exchange on in a world of one rank: the real collectives then run
        # (as identities) on a one-GPU box -- how tests/test_gpu_step.py drives RCCL itself
        force = os.environ.get("DMVAE_DP_FORCE") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def __call__(self, flat_grad):
        if self.enabled:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grad

    # Bucketed form: the backward pass is issued in three segments (dmvae_plan_forward_backward_stage),
    # each completing one contiguous slice of the gradient arena; start() launches that slice's
    # all-reduce on the collective's own stream right behind the segment (it waits for the work
    # already enqueued on the current stream, not for what comes after), finish() makes the current
    # stream wait for all of them before the update.  Only the last bucket (the trunk, 14 % of the
    # arena) is exposed.  DMVAE_DP_OVERLAP=0 falls back to the single all-reduce.
    sharded = False
    _overlap = None           # make_exchange sets it from the arena size; None = the environment switch alone
    n_buckets = 3             # overlapped form: 3 = a collective behind every backward segment; 2 = decoder + heads behind segment 1, trunk behind 2

    @property
    def overlap(self):
        if self._overlap is not None:
            return self.enabled and self._overlap
        return self.enabled and os.environ.get("DMVAE_DP_OVERLAP", "1") != "0"

    def start(self, grad_slice):
        return dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, handle):
        """the current stream waits for one bucket's all-reduce (its Adam can then run while later
        buckets are still on the wire)"""
        if handle is not None:
            handle.wait()

    def finish(self, handles):
        for h in handles:
            self.wait(h)

    def broadcast_(self, tensor, src=0):
        if self.enabled:
            dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def mean_scalars(self, values):
        """average a few python floats over ranks (logging cadence only)"""
        if not self.enabled:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return (t / self.world).tolist()


class ShardedExchange(GradExchange):
    """reduce-scatter -> Adam on the owned shard -> all-gather.  Every range it is given must have a
    length divisible by 64 * world (StepEngine pads the arenas and rounds the bucket bounds, bucket_bounds)."""
    sharded = True

    def __init__(self, group=None):
        super().__init__(group)
        self.align = 64 * self.world

    def padded(self, n):
        return (int(n) + self.align - 1) // self.align * self.align

    def bucket_bounds(self, bounds, n_alloc):
        """bounds = [(lo, hi)] in completion order (StepEngine.grad_buckets: last part of the arena first).  The
        interior boundaries are rounded UP to the alignment: a bucket then only grows into the part of the arena
        that was complete EARLIER, so it is still final when its segment ends.  ONE ENTRY PER SEGMENT, in the
        order given: a bucket that rounding has emptied (fewer than 64 * world elements: tiny models, huge worlds)
        is None -- its elements have moved into the bucket of a LATER segment -- and the caller skips it; dropping
        it would shift every later bucket onto an earlier segment, whose gradients are not written yet."""
        cuts = sorted({lo for lo, _ in bounds} | {hi for _, hi in bounds})
        rounded = [0] + [min(self.padded(c), n_alloc) for c in cuts[1:-1]] + [n_alloc]
        out = []
        for lo, hi in bounds:
            i = cuts.index(lo)
            out.append((rounded[i], rounded[i + 1]) if rounded[i + 1] > rounded[i] else None)
        return out

    def owned(self, lo, hi):
        chunk = (hi - lo) // self.world
        assert chunk * self.world == hi - lo and chunk % 4 == 0, (lo, hi, self.world)
        return lo + self.rank * chunk, lo + (self.rank + 1) * chunk

    # Both collectives run IN PLACE in the one layout NCCL / RCCL define for it: reduce-scatter with
    # recvbuff == sendbuff + rank * recvcount, all-gather with sendbuff == recvbuff + rank * sendcount (the library then
    # skips its local copy).  _in_place asserts exactly that aliasing, so no other overlap of input and output can be
    # passed by accident.  Covered by: gloo at world 2 / 4 / 8 (CPU; gloo copies), two processes on the real kernels, a
    # one-rank RCCL communicator (identity).  NOT yet covered: a multi-rank RCCL run (gpurun gives one GPU) -- the
    # bench line says so (`multi_rank_rccl_verified_before_this_run: false`); DMVAE_DP_MODE=allreduce is the conventional fallback.
    def _in_place(self, flat, lo, hi):
        slo, shi = self.owned(lo, hi)
        whole, mine = flat[lo:hi], flat[slo:shi]
        assert flat.is_contiguous() and mine.numel() * self.world == whole.numel()
        assert mine.data_ptr() == whole.data_ptr() + self.rank * mine.numel() * flat.element_size()
        return whole, mine

    def reduce_scatter(self, flat, lo, hi, async_op=False):
        """sum over ranks of flat[lo:hi]; this rank's slice of the result lands IN PLACE at owned(lo, hi)"""
        whole, mine = self._in_place(flat, lo, hi)
        if not self.enabled:
            return None
        return dist.reduce_scatter_tensor(mine, whole, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_gather(self, flat, lo, hi, async_op=False):
        """every rank's owned slice of flat[lo:hi] to every rank, in place"""
        whole, mine = self._in_place(flat, lo, hi)
        if not self.enabled:
            return None
        return dist.all_gather_into_tensor(whole, mine, group=self.group, async_op=async_op)

    # First-step cross-rank self-check.  After a sharded step every bit a step READS must be the same on
    # every rank: the gathered weights (bf16 shadow on bf16 plans, fp32 else) over [0, sharded_hi) and the all-reduced, replicated
    # tail.  An aliasing fault of the in-place reduce-scatter / all-gather (a slice landing at another rank's offset, a slice not
    # gathered) leaves replicas that disagree and still train -- a plausible number.  So: a checksum of the BITS (the elements viewed
    # as int16 / int32: NaN-proof, and -0.0 != +0.0) -- sum and sum of squares in int64 (modular: exact and independent of the order
    # of the additions), per range -- then all-reduce MIN and MAX of the checksums; unequal on any rank => raise on EVERY rank (the
    # collectives' results are the same everywhere), naming the fallback.
    @staticmethod
    def checksum(t):
        """(sum, sum of squares) mod 2^64 of a tensor's elements viewed as signed integers of its own width"""
        bits = t.contiguous().view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]).to(torch.int64)
        return torch.stack([bits.sum(), (bits * bits).sum()])

    def self_check(self, gathered, sharded_hi, tail):
        """COLLECTIVE (two small all-reduces).  gathered[0:sharded_hi]: what the all-gathers of one sharded step wrote; tail: the
        replicated small-tensor range after its update.  Returns "ok" or raises RuntimeError on every rank."""
        if not self.enabled or self.world == 1:
            return "one rank: nothing to compare"
        cs = torch.cat([self.checksum(gathered[:sharded_hi]), self.checksum(tail)])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            bad = [n for n, a, b in zip(("weights: sum", "weights: sum of squares", "tail: sum", "tail: sum of squares"), lo.tolist(), hi.tolist()) if a != b]
            raise RuntimeError(
                "sharded gradient exchange: after the first step the replicas DISAGREE (%s differ between ranks; this rank: %s, min %s, max %s). "
                "The in-place reduce-scatter / all-gather did not leave identical weights on every rank; rerun with DMVAE_DP_MODE=allreduce "
                "(one all-reduce + replicated Adam)." % (", ".join(bad), cs.tolist(), lo.tolist(), hi.tolist()))
        return "ok"


def make_exchange(param_bytes=0, group=None):
    """the exchange for this process group: None-like (enabled False) in a world of one"""
    mode = os.environ.get("DMVAE_DP_MODE", "sharded")
    ex = ShardedExchange(group) if mode == "sharded" else GradExchange(group)
    ov = os.environ.get("DMVAE_DP_OVERLAP")
    ex._overlap = (param_bytes >= OVERLAP_MIN_BYTES) if ov is None else ov != "0"
    nb = os.environ.get("DMVAE_DP_BUCKETS")
    ex.n_buckets = int(nb) if nb in ("2", "3") else 3
    return ex


# Arena size from which the exchange is cut into buckets that overlap the backward pass: 64 MiB, i.e. the 4096-wide stack (702 MB of
# gradients: the exchange must hide).  The MNIST-sized arenas (19-24 MB) keep ONE reduce-scatter / all-gather pair after the backward.
# MEASURED, round 4, one MI355X, a ONE-RANK RCCL communicator (DMVAE_DP_FORCE=1: every collective is the library call, an identity --
# this prices host issue, stream hand-overs, the extra weight-gradient grids and the stand-alone Adam launches, NOT wire time), ms per
# step (profiles/r04_dp_pricing.txt):
#                      fused single process | one pair | two buckets | three buckets
#     cfg2 (4096 rows)        0.2793        |  0.3241  |   0.3878    |    0.4372
#     cfg4 (8192 rows)        0.6213        |  0.6599  |   0.7274    |    0.8236
# The two-bucket form (decoder + heads behind segment 1 as one weight-gradient launch, the trunk behind segment 2) costs
# +67 us per step over one pair at cfg4; what it can take off the critical path at 8 ranks is the decoder + heads reduce-scatter,
# estimated at ~45 us (17 us of wire at 153 GB/s per link + one collective latency) -- on this evidence it does not
# pay at these sizes, so it is NOT the default.  It stays selectable (DMVAE_DP_OVERLAP=1 DMVAE_DP_BUCKETS=2), is bit-identical to the
# other forms (tests/test_gpu_step.py, the [2] cases) and bench.py prints exchange.exposed_us, so the first real multi-rank run can
# overturn this with a measurement.
OVERLAP_MIN_BYTES = 64 * 2 ** 20
