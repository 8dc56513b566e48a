"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed 'nccl') sum
all-reduce of the flat fp32 gradient buffer over xGMI, in buckets that are launched on a side
stream as soon as backward has finished writing them: the decoder gradients (81 MB of 140.6 MB) overlap the
encoder backward, the encoder's 11.8 MB kernels follow layer by layer as their weight gradients finish, so only
the first layers' gradients (12 MB) are exchanged after the last kernel of the step.  The reference has no multi-GPU code
(SURVEY.md 2.1); training shards over batch, all losses are means, BN uses moving statistics,
so averaging the per-rank gradients reproduces the full-batch step.
"""
import torch
import torch.distributed as dist


class GradAllReduce:
    """force=True: run every collective even in a world of one rank (a 1-rank RCCL all-reduce is an identity, but it is the
    real communicator, bucket slices, side-stream joins and flag exchange: how a 1-GPU box exercises the N > 1 code path)."""

    def __init__(self, flat_grad, group=None, force=False):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.buckets = []           # (lo, hi) of every bucket exchanged since the last finish(), in launch order
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self.pending = []

    def bucket_ready(self, lo, hi):
        """flat[lo:hi] holds final local gradients: start summing it across ranks."""
        if not self.active or hi <= lo:
            return
        self.buckets.append((lo, hi))
        view = self.flat[lo:hi]
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for all buckets; returns the world size (the caller scales by 1/world)."""
        if self.cuda and self.active:
            torch.cuda.current_stream().wait_stream(self.stream)
        for w in self.pending:
            w.wait()
        self.pending = []
        self.last_buckets, self.buckets = self.buckets, []
        return self.world

    def all_reduce_max(self, t):
        """In-place MAX over the ranks (the fp16x3 engine's range flag: every rank must take the same branch)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t
