"""Multi-GPU data parallelism of the PDP iteration: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU for tests).  Trajectories are independent: the batch is cut into
contiguous shards, every rank runs the same kernels on its shard with a replicated theta, and ONE collective per iteration follows.  Two forms of it:
  * all-gather of the per-sample gradients and losses `[B/G, p+1]` - every rank ends with the full `[B, p+1]` (BASELINE.json's "all-gather of per-sample PDP
    gradients"); the batch mean the reference takes (PDP/PDP.py:1293-1294, cartpole_PDP.py:77-78) is then a local reduction.  Message size at C3:
    1024 x 10 x 8 B = 80 KB per rank - latency-bound over xGMI, so it is a single collective, never one per parameter;
  * all-reduce of the locally summed row `[p+1]` (allreduce_mean_packed, mode="allreduce") - what a driver that only needs the mean should use: 3.4 KB at
    C5b (p = 420) instead of 27.6 MB received per rank."""
import os

import torch
import torch.distributed as dist


def force_collective():
    """PDP_DIST_FORCE_COLLECTIVE=1: a process group of ONE rank still issues every collective (no world-size-1 short cut anywhere in this module or in bench.py).
    That is how a one-GPU box executes the `nccl` (= RCCL) backend through exactly the calls an 8-GPU run makes: communicator set-up, all_gather_into_tensor /
    all_reduce on device pointers, the side-stream ordering of OverlappedGather (tests/test_gpu_rccl_world1.py)."""
    return os.environ.get("PDP_DIST_FORCE_COLLECTIVE", "0") == "1"


def exchange_active():
    """True when a collective has to be issued: a process group exists and it has more than one rank (or force_collective())"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective())


def host_staged(t):
    """gloo moves host memory: CUDA tensors are staged through it.  That is the TEST mode of the multi-rank path (two ranks sharing one GPU, bench.py with
    PDP_DIST_BACKEND=gloo PDP_DIST_SAME_DEVICE=1: sharding, packing, ordering and overlap logic run, RCCL does not); "nccl" (= RCCL) takes device pointers as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_gather_into(out, inp):
    """dist.all_gather_into_tensor on the current stream, rank r's `inp` landing in the r-th block of the contiguous `out`; through host memory on gloo (see
    host_staged).  Both sides are passed flat: gloo checks that the output splits into pieces of exactly the input's SHAPE ([world, 3] from [3] is refused), RCCL only
    counts elements."""
    assert out.is_contiguous() and out.numel() == inp.numel() * dist.get_world_size()
    inp = inp.contiguous()
    if host_staged(inp):
        h = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(h, inp.cpu().view(-1))
        out.copy_(h.view(out.shape))
    else:
        dist.all_gather_into_tensor(out.view(-1), inp.view(-1))
    return out


def all_reduce_(t, op=None):
    """dist.all_reduce in place; through host memory on gloo"""
    op = dist.ReduceOp.SUM if op is None else op
    if host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def shard_bounds(n_total, world, rank):
    """contiguous block partition; the first (n_total % world) ranks get one extra trajectory"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(x, world=None, rank=None, dim=0):
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(x.shape[dim], world, rank)
    return x.narrow(dim, lo, hi - lo) if hasattr(x, "narrow") else x[lo:hi]


def gather_loss_grad(loss, grad, n_total=None):
    """all-gather per-sample (loss [b], grad [b,p]) of every rank -> (loss [B], grad [B,p]) on every rank.
    Shards may differ by one trajectory (ragged): they are padded to the largest shard for the collective."""
    if not exchange_active():
        return loss, grad
    world, rank = dist.get_world_size(), dist.get_rank()
    b, p = grad.shape
    if n_total is None:
        all_sizes = torch.zeros(world, dtype=torch.int64, device=grad.device)
        all_gather_into(all_sizes, torch.tensor([b], dtype=torch.int64, device=grad.device))
        counts = [int(v) for v in all_sizes.cpu()]
    else:
        counts = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    bmax = max(counts)
    packed = torch.zeros((bmax, p + 1), dtype=grad.dtype, device=grad.device)
    packed[:b, :p] = grad
    packed[:b, p] = loss
    out = torch.empty((world * bmax, p + 1), dtype=grad.dtype, device=grad.device)
    all_gather_into(out, packed)
    out = out.view(world, bmax, p + 1)
    rows = torch.cat([out[r, :counts[r]] for r in range(world)], dim=0)
    return rows[:, p].contiguous(), rows[:, :p].contiguous()


def allreduce_mean_packed(packed, n_total=None):
    """The reference only ever uses the batch MEAN of the per-sample losses and gradients (PDP/PDP.py:1293-1294; Examples/IRL/cartpole/cartpole_PDP.py:77-78), so
    the exchange of a gradient-descent driver need not move the rows at all: every rank sums its [b, p+1] rows (gradient | loss, the layout PDP_OC_PACKED writes)
    on the GPU and ONE all-reduce of p + 1 doubles follows - 3.4 KB at C5b (p = 420) where the all-gather hands every rank 8192 x 421 x 8 B = 27.6 MB.
    Ragged shards need no padding (a sum does not care).  Returns the mean row [p+1] (gradient mean | loss mean) on every rank.
    n_total: the number of trajectories over all ranks (None: it is all-reduced along, as one more entry of the same message)."""
    s = packed.sum(dim=0)
    if not exchange_active():
        return s / float(packed.shape[0] if n_total is None else n_total)
    if n_total is None:
        s = torch.cat([s, torch.tensor([float(packed.shape[0])], dtype=s.dtype, device=s.device)])
        all_reduce_(s)
        return s[:-1] / s[-1]
    all_reduce_(s)
    return s / float(n_total)


def mean_loss_grad(loss, grad, n_total=None, mode="allgather"):
    """the reference's batch mean of (loss, gradient) over ALL trajectories of all ranks.
    mode "allgather": every rank receives every per-sample row and reduces locally (for callers that also want the rows);
    mode "allreduce": local sums, one all-reduce of p + 1 doubles (allreduce_mean_packed) - the exchange a driver that only needs the mean should use.
    The two agree to the rounding of the summation order."""
    if mode == "allreduce":
        m = allreduce_mean_packed(torch.cat([grad, loss[:, None]], dim=1), n_total)
        return m[-1], m[:-1]
    assert mode == "allgather", mode
    L, G = gather_loss_grad(loss, grad, n_total)
    return L.mean(), G.mean(dim=0)


def pdp_iteration(unit, shard_inputs, n_total=None, mode="allgather"):
    """One data-parallel PDP iteration: `unit(**shard_inputs)` must return a dict with per-sample 'loss' [b] and 'grad' [b,p]
    computed on this rank's shard (e.g. OCSys.pdp_grad_batch, or (loss, grad) from ControlPlanning.step_batch / SysID.step_batch);
    a dict with 'packed' [b, p+1] (PDP_OC_PACKED) is reduced without a packing copy in mode "allreduce"."""
    out = unit(**shard_inputs)
    if mode == "allreduce" and isinstance(out, dict) and "packed" in out:
        m = allreduce_mean_packed(out["packed"], n_total)
        return m[-1], m[:-1]
    loss, grad = (out["loss"], out["grad"]) if isinstance(out, dict) else out
    return mean_loss_grad(loss, grad, n_total, mode)


def gather_packed(packed, n_total=None, out=None):
    """all-gather of the [b, p+1] rows the fused kernel writes with PDP_OC_PACKED (gradient | loss): no packing copies on the way in.
    Equal shards (n_total divisible by the world size, or None): ONE collective straight into `out` [B, p+1]; ragged shards go
    through gather_loss_grad's padding.  Returns the [B, p+1] tensor."""
    if not exchange_active():
        return packed
    world = dist.get_world_size()
    b, p1 = packed.shape
    if n_total is not None and n_total % world != 0:
        L, G = gather_loss_grad(packed[:, p1 - 1].contiguous(), packed[:, :p1 - 1].contiguous(), n_total)
        return torch.cat([G, L[:, None]], dim=1)
    if out is None:
        out = torch.empty((world * b, p1), dtype=packed.dtype, device=packed.device)
    return all_gather_into(out, packed.contiguous())


class StepStreams:
    """INDEPENDENT steps of a loop issued on n HIP streams in rotation (n = 2: even steps on one queue, odd steps on the other).

    Why (probes/exchange_overlap.py, profiles/r06_exchange_overlap.txt): the fused OC unit at 1024 trajectories holds every CU for the whole launch (one workgroup per
    CU, all of its LDS), so RCCL's kernel finds room only where a launch ends - and everything a step puts behind its kernel in the SAME queue (the event the side
    stream waits for, the wait for the collective that last read the buffer) is a barrier packet: the next kernel is not dispatched before the previous one has
    drained completely, and the collective then runs in the gap.  Measured through a one-rank RCCL group: 0.0898 ms per step for the kernel alone, 0.1008 ms with the
    all-gather "overlapped" on a side stream (+11 us; 2, 3 or 4 buffers alike), 0.0993 ms with the all-gather simply on the same stream.  With the steps alternating
    between two queues the barrier packets of step k sit in a queue that has nothing else to do until step k + 2: the kernel of step k + 1 moves into each CU the moment
    a workgroup of step k leaves it, the collective takes the first CU that frees (its cost is spread over 256 CUs instead of stalling all of them): 0.0891 ms with
    the all-gather, 0.0882 ms without.  The steps must not share anything they write (outputs, workspace): one prepared call per stream.
    Not for dependent steps (a gradient-descent loop): there the exchange sits on the critical path whatever the stream."""

    def __init__(self, n=2, streams=None):
        """streams: use these (e.g. a pair found by timing a few candidates, as bench.py does: HIP streams share 4 hardware queues in creation order, and a step stream
        that lands in the queue of the side stream - or both step streams in one - loses what the scheme gains); default: n new streams"""
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream() for _ in range(int(n))]
        self.k = 0
        self.fork()

    def fork(self):
        """the step streams wait for what the caller's stream has issued so far (inputs, warm-up)"""
        cur = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(cur)

    def index(self):
        """index of the stream the next step will use"""
        return self.k % len(self.streams)

    def next(self):
        """context manager: the next step's stream as the current stream"""
        s = self.streams[self.k % len(self.streams)]
        self.k += 1
        return torch.cuda.stream(s)

    def join(self):
        """the caller's stream waits for every step issued so far"""
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)


class OverlappedGather:
    """The exchange step of the data-parallel iteration, off the critical path: the all-gather of step k runs on a side stream
    while the kernels of step k+1 run on the compute stream (the per-sample rows of step k are consumed one step later - a
    gradient-descent driver that tolerates one step of staleness, or a benchmark that only needs the exchange done by the end).
    Double-buffered: `submit(packed_k)` copies nothing - the caller alternates between two packed buffers (see `buffers`) - records
    an event on the compute stream, and enqueues the collective on the side stream behind it; `wait(k)` makes the compute stream
    wait for the collective of step k.  On CPU tensors (gloo tests) it degenerates to the blocking collective."""

    def __init__(self, rows, cols, dtype=torch.float64, device="cuda", depth=2, priority=0):
        """depth: number of packed / gathered buffer pairs in rotation (2 = the rows of step k are complete before the kernel of step k + 2 starts);
        priority: stream priority of the side stream (negative = higher; probes/exchange_overlap.py measures what either buys)"""
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.active = exchange_active()
        self.cuda = torch.device(device).type == "cuda"
        self.depth = int(depth)
        self.buffers = [torch.zeros((rows, cols), dtype=dtype, device=device) for _ in range(self.depth)]
        self.gathered = [torch.zeros((self.world * rows, cols), dtype=dtype, device=device) for _ in range(self.depth)]
        self.side = torch.cuda.Stream(priority=priority) if self.cuda else None
        self.done = [None] * self.depth
        self.k = 0

    def next_buffer(self):
        """the packed buffer the next step's kernel should write (its previous collective has been waited for)"""
        i = self.k % self.depth
        if self.cuda and self.done[i] is not None:
            torch.cuda.current_stream().wait_event(self.done[i])
        return self.buffers[i]

    def submit(self):
        """enqueue the all-gather of the buffer handed out by the last next_buffer(); returns the index of the gathered tensor"""
        i = self.k % self.depth
        self.k += 1
        if not self.active:
            self.gathered[i] = self.buffers[i]
            return i
        if not self.cuda:
            all_gather_into(self.gathered[i], self.buffers[i])
            return i
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            all_gather_into(self.gathered[i], self.buffers[i])
            self.done[i] = torch.cuda.Event()
            self.done[i].record(self.side)
        return i

    def result(self, i):
        """the [B, p+1] rows of a submitted step (the compute stream waits for its collective)"""
        if self.cuda and self.done[i] is not None:
            torch.cuda.current_stream().wait_event(self.done[i])
        return self.gathered[i]

    def drain(self):
        if self.cuda:
            for e in self.done:
                if e is not None:
                    torch.cuda.current_stream().wait_event(e)
