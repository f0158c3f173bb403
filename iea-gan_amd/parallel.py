"""Data-parallel IEA-GAN over RCCL/xGMI: one process per GPU, events sharded across ranks.

The reference has no multi-GPU code at all, so this is new functionality with the
contract of SURVEY section 8e: every rank holds a full replica of G, D, the Adam moments and the EMA copy
and trains on its own events (BatchNorm statistics, the 40-token RRM and the 40x40 loss Grams are all
intra-event); the ONLY exchange per network and step is one all-reduce (mean) of the flat gradient arena
(17.9 MB for D, 46.8 MB for G in fp32).  D's all-reduce + Adam run on a side HIP stream under the G-phase
generator forward; the main stream waits for them only right before D is evaluated again.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large flat all-reduce per network keeps every
link busy with size/8 per phase instead of hundreds of per-parameter latency-bound collectives.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class GradSync:
    """All-reduce-mean of one flat gradient buffer followed by a callback (the optimiser step),
    optionally on a side stream.  Works on CPU tensors with gloo (tests) and on HIP with RCCL."""

    def __init__(self, group=None, overlap=True, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.force = force and dist.is_initialized()      # exercise the collective path with a 1-rank group (tests)
        self._stream = None
        self._pending = {}
        self.trace = False          # record (side work done, main stream arrives at the wait) event pairs: slack_ms()
        self._slack = []

    def _side_stream(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        return self._stream

    def reduce_then(self, key, flat_grad: torch.Tensor, then=None, blocking=False):
        """Average ``flat_grad`` over the ranks, then call ``then()``.  On HIP with overlap (and not ``blocking``)
        the work is enqueued on a side stream and ``wait(key)`` must be called before the results are consumed."""
        if self.world == 1 and not self.force:
            if then is not None:
                then()
            return
        if flat_grad.is_cuda and self.overlap and not blocking:
            ev = torch.cuda.Event()
            ev.record()
            side = self._side_stream()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
                flat_grad.mul_(1.0 / self.world)
                if then is not None:
                    then()
                done = torch.cuda.Event(enable_timing=self.trace)
                done.record(side)
            self._pending[key] = (done, self.trace)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.mul_(1.0 / self.world)
            if then is not None:
                then()

    def wait(self, key):
        ev, timed = self._pending.pop(key, (None, False))
        if ev is not None:
            if timed and not torch.cuda.is_current_stream_capturing():
                arrive = torch.cuda.Event(enable_timing=True)
                arrive.record()                     # the moment the consumer stream asks for the result
                self._slack.append((key, ev, arrive))
            torch.cuda.current_stream().wait_event(ev)

    def slack_ms(self):
        """Per key, for every traced exchange: milliseconds between the side stream finishing (all-reduce + update) and the consumer
        stream reaching its wait.  Positive = the exchange was hidden completely; negative = the consumer stalled that long."""
        torch.cuda.synchronize()
        out = {}
        for key, done, arrive in self._slack:
            out.setdefault(key, []).append(done.elapsed_time(arrive))
        self._slack = []
        return out

    def wait_all(self):
        for k in list(self._pending):
            self.wait(k)


def broadcast_flat(flat: torch.Tensor, src=0, group=None):
    """Make every rank start from rank ``src``'s state (parameters and buffers: one arena, one call)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def shard_events(n_events: int, rank: int, world: int):
    """Contiguous, balanced split of ``n_events`` independent events over ``world`` ranks."""
    base, rem = divmod(n_events, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def steps_per_rank(n_events: int, world: int) -> int:
    """Iterations per epoch EVERY rank runs: one collective sequence per iteration, so the count must not depend on
    the rank (the ``n_events % world`` tail is dropped).  0 means the run cannot start."""
    return n_events // max(world, 1)


def quiesce():
    """Join the side stream (pending gradient exchanges + updates) and drain the device: call before reading network
    state on the host (checkpoints)."""
    if _CTX is not None:
        _CTX.wait_all()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def shutdown():
    """Leave the process group together: a rank that exits while another still sits in a collective would hang it."""
    if _CTX is not None and torch.cuda.is_available():
        _CTX.wait_all()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not (os.environ.get("IEAGAN_FORCE_DP") and "RANK" in os.environ):
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_CTX = None


def set_context(ctx):
    global _CTX
    _CTX = ctx


def get_context():
    return _CTX
