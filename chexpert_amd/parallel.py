"""Data parallelism for the fused models: one process per GPU, gradients averaged with bucketed
all-reduce (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

The reference is single-device (chexpert.py:453); DP semantics here are DDP's: every rank holds a full
replica and a shard of the minibatch, BatchNorm uses per-rank batch statistics, parameter gradients are
averaged.  Gradients live in ONE flat fp32 buffer in parameter order; backward
produces them from the end of the buffer to the start, so buckets are contiguous ranges handed to the
communicator as soon as the kernels that fill them have been enqueued: the all-reduce of bucket k runs
on a side stream while the compute stream continues with the earlier layers' backward kernels.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grad, bucket_bytes=16 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # rehearsal on a one-GPU box: issue the collectives also on a communicator of one rank (bench.py CHEXPERT_BENCH_FORCE_DP)
        import os
        self.force = dist.is_initialized() and os.environ.get("CHEXPERT_FORCE_COLLECTIVES") == "1"
        self.bucket = max(1, bucket_bytes // flat_grad.element_size())
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self.avg = self.cuda and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.hi = flat_grad.numel()
        self.works = []
        self.ranges = []
        # set by graph.SegmentedTrainStep while it captures the step: a bucket that becomes complete does not start a collective,
        # it CUTS the capture there (the collective is enqueued at that point of every replay)
        self.capture = None
        # called right before a bucket leaves (engine: run the deferred weight-gradient slab sums, so that the bucket is final)
        self.pre_launch = None

    def begin(self):
        self.hi = self.flat.numel()
        self.works, self.ranges = [], []

    def _launch(self, lo, hi):
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        if self.pre_launch is not None:
            self.pre_launch()
        if self.capture is not None:
            self.ranges.append((lo, hi))
            self.capture.cut(("launch", lo, hi))
            return
        view = self.flat[lo:hi]
        self.ranges.append((lo, hi))
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.avg:
                    self.works.append(dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                else:
                    self.works.append(dist.all_reduce(view, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, group=self.group, async_op=True))

    def ready(self, lo):
        """Every gradient at flat offset >= lo is final."""
        if self.hi - lo >= self.bucket or (lo == 0 and self.hi > 0):
            self._launch(lo, self.hi)
            self.hi = lo

    def finish(self):
        """Flush the tail and make the reduced gradients visible to the compute stream."""
        if self.capture is not None:
            # (one cut for the tail bucket and the join: nothing is enqueued between them)
            tail = (self.world > 1 or self.force) and self.hi > 0
            if tail:
                if self.pre_launch is not None:
                    self.pre_launch()
                self.ranges.append((0, self.hi))
            self.capture.cut(*([("launch", 0, self.hi)] if tail else []), ("finish",))
            self.hi = 0
            return
        self.ready(0)
        self.wait()

    def wait(self):
        """The part of finish() after the last launch: join the collectives (a segmented replay calls it between two segments)."""
        for w in self.works:
            w.wait()
        if self.cuda and self.works:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.world > 1 and not self.avg:
            self.flat.mul_(1.0 / self.world)
        self.works = []


def broadcast_module_state(module, src=0, group=None):
    """Initial parameter / buffer broadcast from rank 0 (replicas start identical)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


# ---- data-parallel training loop helpers (chexpert_amd/cli.py) ------------------------------------------------------------------
def dist_info():
    """(rank, world, local_rank) from the torch.distributed.run environment; (0, 1, 0) when launched directly."""
    import os
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_indices(n, rank, world, seed=0, epoch=0, shuffle=True, drop_last=True):
    """This rank's sample indices for one epoch: one seeded permutation shared by all ranks, dealt round-robin (what
    torch.utils.data.DistributedSampler does); drop_last trims to a multiple of `world` so every rank steps equally often."""
    import numpy as np
    order = np.random.RandomState(seed * 1000003 + epoch).permutation(n) if shuffle else np.arange(n)
    if drop_last:
        order = order[:(n // world) * world]
    return order[rank::world].tolist()


def gather_rows(t, group=None):
    """Concatenate per-rank row blocks (N_r, ...) of possibly different N_r in rank order on every rank (sharded evaluation:
    every rank forwards its slice of the validation set, rank 0 computes AUROC on the whole)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device if dist.get_backend(group) == "nccl" else "cpu")
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(v) for v in ns]
    m = max(ns)
    cpu = dist.get_backend(group) != "nccl"
    src = t.cpu() if cpu else t
    pad = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:k] for o, k in zip(outs, ns)]).to(t.device)
