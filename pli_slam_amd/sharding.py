"""Frame-batch data parallelism across the GPUs of one node.

Extraction and stereo matching are independent per stereo frame, so a batch is
cut into contiguous shards, one per rank, with no data-path collective; the only
exchange is the gather of the fixed-stride result tables to one rank
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).

With the gloo backend and tables in HBM (several ranks sharing ONE GPU: the single-GPU rehearsal of the multi-GPU path,
bench.py --share-device) the gather and the halo are staged through host memory: gloo moves CPU tensors only.
"""
import torch
import torch.distributed as dist


def shard_range(nframes, rank, world):
    """Contiguous shard [start, start+count) of rank `rank`; the first nframes % world ranks get one extra."""
    base, extra = divmod(nframes, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def gather_tables(table, record_bytes, nframes_local, dst=0, group=None, counts=None):
    """Gather every rank's result table (1-D uint8 tensor, nframes_local records) on rank `dst`.

    Shards may differ by one frame, so tables are padded to the largest shard and cut again on
    the root.  Returns the list of per-rank tables (views trimmed to that rank's frame count) on
    `dst`, None elsewhere.  With world size 1 it returns [table] without communication.
    `counts`: the frame count of every rank when the caller already knows them (equal shards): skips the
    exchange of the counts and its host synchronisation.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [table[: nframes_local * record_bytes]]
    rank = dist.get_rank(group)
    if counts is not None:
        all_counts = [int(c) for c in counts]
        assert len(all_counts) == world and all_counts[rank] == nframes_local
    else:
        mine = torch.tensor([nframes_local], dtype=torch.int64, device=table.device)
        all_counts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(all_counts, mine, group=group)
        all_counts = [int(c.item()) for c in all_counts]
    pad = max(all_counts) * record_bytes
    if table.numel() < pad:
        table = torch.cat([table, table.new_zeros(pad - table.numel())])
    send = table[:pad].contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[: n * record_bytes] for b, n in zip(bufs, all_counts)]


def _host_staged(t, group=None):
    """gloo cannot move device tensors point to point / by gather: stage them through the host."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def exchange_halo(table, record_bytes, nframes_local, group=None, counts=None, force=False):
    """1-frame halo for frame-to-frame matching across shard borders: every rank sends the record of its LAST
    frame to the next rank and receives the record of the frame before its first one.

    `table`: 1-D uint8 tensor with ONE FREE RECORD IN FRONT of the `nframes_local` records of this rank
    (layout [halo | frame 0 | frame 1 | ...]).  The halo slot is filled in place; returns True when it holds a frame
    (every rank but the first; a rank whose predecessor has no frames gets none either), so that the caller runs the
    frame-to-frame matcher over nframes_local + 1 consecutive records starting at the halo, or over nframes_local
    records starting at frame 0.  Point-to-point only (isend / irecv): no collective on the data path.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return False
    # (force: a process group of ONE rank walks the same code — counts, neighbours, requests — and finds no neighbour)
    rank = dist.get_rank(group)
    if counts is not None:                               # the frame count of every rank, known to the caller: no exchange of counts
        all_counts = [int(c) for c in counts]
    else:
        mine = torch.tensor([nframes_local], dtype=torch.int64, device=table.device)
        all_counts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(all_counts, mine, group=group)
        all_counts = [int(c.item()) for c in all_counts]
    reqs = []
    staged = _host_staged(table, group)
    last = table[nframes_local * record_bytes:(nframes_local + 1) * record_bytes]          # record of the last local frame
    if rank + 1 < world and nframes_local > 0:
        reqs.append(dist.isend(last.cpu() if staged else last.contiguous(), dst=rank + 1, group=group))
    got = rank > 0 and all_counts[rank - 1] > 0
    halo_host = None
    if got:
        halo = table[:record_bytes]
        if staged:
            halo_host = torch.empty(record_bytes, dtype=torch.uint8)
            reqs.append(dist.irecv(halo_host, src=rank - 1, group=group))
        else:
            reqs.append(dist.irecv(halo, src=rank - 1, group=group))
    for r in reqs:
        r.wait()
    if halo_host is not None:
        table[:record_bytes].copy_(halo_host)
    return got


class TableGatherer:
    """Double-buffered, asynchronous gather of equal-sized shards to rank `dst`: the gather of step i travels over
    xGMI while the kernels of step i+1 run (a result table is only waited for when its buffer is reused).

        g = TableGatherer(record_bytes * frames_per_rank, device)
        for i in range(steps):
            slot = g.acquire()            # waits for the gather that last used this slot
            ... launch the kernels that fill  g.table(slot) ...
            g.submit(slot)
        g.drain()                         # all gathers done; on dst, g.gathered(slot) holds every rank's table
    """

    def __init__(self, table_bytes, device, depth=2, dst=0, group=None, force=False):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst, self.group, self.depth = dst, group, depth
        # force: issue the collective even in a process group of ONE rank (bench.py --force-dist: the RCCL calls of the multi-GPU
        # path executed on a one-GPU box)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.tables = [torch.zeros(table_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.bufs = [None] * depth
        # gloo + tables in HBM: the shard goes to a (pinned) host copy first and the root receives host buffers
        self.staged = self.active and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        self.host = [torch.empty(table_bytes, dtype=torch.uint8).pin_memory() for _ in range(depth)] if self.staged else None
        bdev = torch.device("cpu") if self.staged else device
        if self.active and self.rank == dst:
            self.bufs = [[torch.empty(table_bytes, dtype=torch.uint8, device=bdev) for _ in range(self.world)]
                         for _ in range(depth)]
        self.works = [None] * depth
        self.next = 0

    def table(self, slot):
        return self.tables[slot]

    def acquire(self):
        slot = self.next
        self.next = (self.next + 1) % self.depth
        if self.works[slot] is not None:
            self.works[slot].wait()
            self.works[slot] = None
        return slot

    def submit(self, slot):
        if self.active:
            src = self.tables[slot]
            if self.staged:
                self.host[slot].copy_(src)               # waits for the kernels on the current stream that fill the table
                src = self.host[slot]
            self.works[slot] = dist.gather(src, self.bufs[slot], dst=self.dst, group=self.group, async_op=True)

    def drain(self):
        for slot in range(self.depth):
            if self.works[slot] is not None:
                self.works[slot].wait()
                self.works[slot] = None

    def gathered(self, slot):
        """On dst: the list of every rank's table of that slot (world size 1: just the local table)."""
        if not self.active:
            return [self.tables[slot]]
        return self.bufs[slot] if self.rank == self.dst else None
