"""Multi-GPU tiling of one frame: interleaved row strips, one process per GPU, ONE RCCL gather.

The reference is single-GPU; pixels are independent (raytracingCs.glsl reads only
gl_GlobalInvocationID, uniforms and read-only buffers), so the frame shards by rows with no
data-path exchange until the final image is assembled.  Equal contiguous tiles are badly
unbalanced (sky rows end at depth 0, floor rows run the full bounce loop), so rank r renders
strips s with s % world == r of `strip_rows` rows each; the kernel maps local
rows to image rows itself (rt_params.stripRows/stripCount/stripIndex) and writes its three
surfaces into one PACKED per-rank buffer

    [ gColor rows*W*16 B | gPosition rows*W*16 B | gNormal rows*W*8 B ]      (rows = whole strips)

which is gathered to rank 0 with a single `torch.distributed.gather` (RCCL over xGMI: each of
the seven links into the root carries one peer's buffer) and put back in image order by a copy
kernel.  The root's inbound links are what bound an 8-GPU 1080p frame, so the buffer that
travels is the 30 B/pixel WIRE format (rt_wire_pack / rt_wire_unpack: rgb f32 | rgb f32 |
rgb f16 -- every surface's alpha is the constant 1.0 and is restored on rank 0) rather than the
40 B/pixel surfaces; the surface-format gather + rt_deinterleave path remains for callers that
want it.  Results are bitwise identical to the single-GPU frame.

Only plumbing lives here (torch owns device memory and the process group); rendering and
re-assembly are C-ABI calls.
"""
from dataclasses import dataclass

from . import layout as L

BPP = (16, 16, 8)   # bytes per pixel of gColor, gPosition, gNormal (raytracingCs.glsl:61-63)


@dataclass(frozen=True)
class StripPlan:
    """The image is a sequence of CYCLES of (root_weight + world - 1) strips of `strip_rows` rows: rank 0 owns the
    first root_weight strips of every cycle (one fat strip), rank r > 0 the strip after that.  root_weight = 1 is
    the plain interleave (strip s -> rank s % world).  A heavier root renders more of the frame itself: its rows
    never cross xGMI, so fewer bytes converge on its inbound links (the bound of a gathered 1080p frame)."""
    width: int
    height: int
    strip_rows: int
    world: int
    root_weight: int = 1

    @property
    def cycle_rows(self):
        return (self.root_weight + self.world - 1) * self.strip_rows

    @property
    def n_cycles(self):
        return (self.height + self.cycle_rows - 1) // self.cycle_rows

    @property
    def n_strips(self):
        return (self.height + self.strip_rows - 1) // self.strip_rows

    def rows_per_cycle(self, rank):
        return (self.root_weight if rank == 0 else 1) * self.strip_rows

    def offset_rows(self, rank):
        return 0 if rank == 0 else (self.root_weight + rank - 1) * self.strip_rows

    def buffer_rows(self, rank):
        """Rows of rank's local surfaces: whole strips (rows beyond the image are written as zeros)."""
        return self.n_cycles * self.rows_per_cycle(rank)

    def local_rows(self, rank):
        """Image rows rank actually owns."""
        rows, rpc, off = 0, self.rows_per_cycle(rank), self.offset_rows(rank)
        for c in range(self.n_cycles):
            lo = c * self.cycle_rows + off
            rows += max(0, min(lo + rpc, self.height) - lo)
        return rows

    @property
    def max_local_rows(self):
        """Rows of a PEER's buffer (equal on every rank r > 0; and on rank 0 when root_weight = 1)."""
        return self.n_cycles * self.strip_rows

    @property
    def rank_bytes(self):
        return self.max_local_rows * self.width * sum(BPP)

    def surface_bytes(self, rank):
        return self.buffer_rows(rank) * self.width * sum(BPP)

    @property
    def rank_pixels(self):
        return self.max_local_rows * self.width

    @property
    def wire_bytes(self):
        """Bytes of one rank's wire buffer (rt_wire_bytes: 30 B/pixel, padded to 16)."""
        return (self.rank_pixels * 30 + 15) // 16 * 16

    def surface_offsets(self, rank=None):
        n = (self.max_local_rows if rank is None else self.buffer_rows(rank)) * self.width
        return 0, n * BPP[0], n * (BPP[0] + BPP[1])

    def global_row(self, rank, local_row):
        rpc = self.rows_per_cycle(rank)
        return (local_row // rpc) * self.cycle_rows + self.offset_rows(rank) + local_row % rpc

    def owner(self, y):
        """(rank, local row) of image row y."""
        c, w = divmod(y, self.cycle_rows)
        root = self.root_weight * self.strip_rows
        if w < root:
            return 0, c * root + w
        j = w - root
        return 1 + j // self.strip_rows, c * self.strip_rows + j % self.strip_rows

    def params(self, base, rank):
        """rt_params for this rank's local surfaces (buffer_rows(rank) x width); rows that fall beyond the
        image are written as zeros by the kernel."""
        return L.copy_params(base, x0=0, y0=0, regionW=self.width, regionH=self.buffer_rows(rank),
                             stripRows=self.rows_per_cycle(rank), stripCount=self.world, stripIndex=rank,
                             stripCycleRows=self.cycle_rows, stripOffsetRows=self.offset_rows(rank))

    def row_index(self):
        """For each image row y: (rank, local row) flattened as rank * max_local_rows + local row
        (equal strips only: every rank buffer has max_local_rows rows)."""
        assert self.root_weight == 1
        idx = []
        for y in range(self.height):
            r, ly = self.owner(y)
            idx.append(r * self.max_local_rows + ly)
        return idx


def default_strip_rows(height, world, n_objects=None):
    """One workgroup tile per strip: the packet kernel renders 8x8-pixel tiles for every scene size.  Short
    strips keep every rank's share of the benchmark scenes within a few percent of equal and the padding of
    the equal-sized gather buffers small (1080p on 8 GPUs: 8-row strips pad 1080 -> 1088 rows)."""
    return 8


def alloc_rank_buffer(plan, device, rank=None):
    """One uint8 tensor holding a rank's three local surfaces (rank=None: a peer-sized buffer)."""
    import torch
    nbytes = plan.rank_bytes if rank is None else plan.surface_bytes(rank)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def surface_views(buf, plan, rank=None):
    """(gColor f32[rows,W,4], gPosition f32[rows,W,4], gNormal f16[rows,W,4]) views of a rank buffer."""
    import torch
    n, w = (plan.max_local_rows if rank is None else plan.buffer_rows(rank)), plan.width
    o0, o1, o2 = plan.surface_offsets(rank)
    end = o2 + n * w * BPP[2]
    col = buf[o0:o1].view(torch.float32).view(n, w, 4)
    pos = buf[o1:o2].view(torch.float32).view(n, w, 4)
    nrm = buf[o2:end].view(torch.float16).view(n, w, 4)
    return col, pos, nrm


def gather_rank_buffers(buf, plan, rank, group=None, dst=0, out=None):
    """One collective per frame: every rank's packed buffer -> rank `dst` ([world, rank_bytes]).
    With the NCCL (= RCCL) backend this is a grouped send/recv gather over xGMI."""
    import torch
    import torch.distributed as dist

    if plan.world == 1:
        return buf.view(1, -1)
    if rank == dst:
        if out is None:
            out = torch.empty((plan.world, plan.rank_bytes), dtype=torch.uint8, device=buf.device)
        dist.gather(buf, gather_list=list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(buf, gather_list=None, dst=dst, group=group)
    return None


def deinterleave_torch(gathered, plan):
    """Reference re-assembly with torch indexing (CPU tests and a cross-check of the HIP copy
    kernel): gathered [world, rank_bytes] uint8 -> (gColor, gPosition, gNormal) full frames."""
    import torch

    idx = torch.tensor(plan.row_index(), dtype=torch.long, device=gathered.device)
    outs = []
    per_rank = [surface_views(gathered[r], plan) for r in range(plan.world)]
    for s in range(3):
        stacked = torch.cat([per_rank[r][s] for r in range(plan.world)], dim=0)
        outs.append(stacked.index_select(0, idx))
    return outs


def deinterleave_hip(tracer, gathered, plan, outs=None, stream=None):
    """Rank-0 re-assembly on the GPU through the C ABI (rt_deinterleave), one launch per surface."""
    import torch

    dev = gathered.device
    if outs is None:
        outs = [torch.empty((plan.height, plan.width, 4), dtype=dt, device=dev)
                for dt in (torch.float32, torch.float32, torch.float16)]
    base = gathered.data_ptr()
    for off, bpp, out in zip(plan.surface_offsets(), BPP, outs):
        tracer.deinterleave(base + off, out.data_ptr(), plan.width, plan.height, bpp, plan.strip_rows,
                            plan.world, plan.rank_bytes, stream=stream)
    return outs


# ---- 30 B/pixel wire format ---------------------------------------------------------------
def alloc_wire_buffer(plan, device):
    import torch
    return torch.empty(plan.wire_bytes, dtype=torch.uint8, device=device)


def pack_wire_hip(tracer, views, wire, plan, stream=None):
    """A PEER's local surfaces -> its wire buffer (rank 0 never packs when its rows stay local)."""
    col, pos, nrm = views
    assert col.shape[0] * col.shape[1] == plan.rank_pixels
    tracer.wire_pack(col.data_ptr(), pos.data_ptr(), nrm.data_ptr(), wire.data_ptr(), plan.rank_pixels, stream=stream)
    return wire


def pack_wire_torch(views, plan, wire=None):
    """Same packing with torch indexing (CPU tests; cross-check of rt_wire_pack)."""
    import torch
    col, pos, nrm = views
    n = plan.rank_pixels
    assert col.shape[0] * col.shape[1] == n
    if wire is None:
        wire = torch.zeros(plan.wire_bytes, dtype=torch.uint8, device=col.device)
    wire[:12 * n].view(torch.float32).view(n, 3).copy_(col.reshape(n, 4)[:, :3])
    wire[12 * n:24 * n].view(torch.float32).view(n, 3).copy_(pos.reshape(n, 4)[:, :3])
    wire[24 * n:30 * n].view(torch.int16).view(n, 3).copy_(nrm.view(torch.int16).reshape(n, 4)[:, :3])
    return wire


def gather_wire(wire, plan, rank, group=None, dst=0, out=None):
    """The one collective per frame, on wire buffers: -> [world, wire_bytes] on rank `dst`.  (The
    root's own contribution is a placeholder when its rows stay local; torch's gather wants one.)"""
    import torch
    import torch.distributed as dist

    if plan.world == 1:
        return wire.view(1, -1)
    if rank == dst:
        if out is None:
            out = torch.empty((plan.world, plan.wire_bytes), dtype=torch.uint8, device=wire.device)
        dist.gather(wire, gather_list=list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(wire, gather_list=None, dst=dst, group=group)
    return None


def unpack_wire_torch(gathered, plan, root_views=None):
    """Reference re-assembly of gathered wire buffers with torch indexing.  root_views = rank 0's own
    local surfaces (its rows are taken from there, as rt_wire_unpack does); None = from wire slot 0."""
    import torch
    n = plan.rank_pixels
    dev = gathered.device
    col = torch.empty((plan.height, plan.width, 4), dtype=torch.float32, device=dev)
    pos = torch.empty_like(col)
    nrm = torch.empty((plan.height, plan.width, 4), dtype=torch.int16, device=dev)
    planes = []
    for r in range(plan.world):
        w = gathered[r]
        planes.append((w[:12 * n].view(torch.float32).view(plan.max_local_rows, plan.width, 3),
                       w[12 * n:24 * n].view(torch.float32).view(plan.max_local_rows, plan.width, 3),
                       w[24 * n:30 * n].view(torch.int16).view(plan.max_local_rows, plan.width, 3)))
    assert root_views is not None or plan.root_weight == 1
    for y in range(plan.height):
        r, ly = plan.owner(y)
        if r == 0 and root_views is not None:
            col[y], pos[y], nrm[y] = root_views[0][ly], root_views[1][ly], root_views[2][ly].view(torch.int16)
            continue
        c3, p3, n3 = planes[r]
        col[y, :, :3], pos[y, :, :3], nrm[y, :, :3] = c3[ly], p3[ly], n3[ly]
        col[y, :, 3] = 1.0
        pos[y, :, 3] = 1.0
        nrm[y, :, 3] = 0x3c00
    return [col, pos, nrm.view(torch.float16)]


def unpack_wire_hip(tracer, gathered, plan, outs=None, root_views=None, stream=None):
    """Rank-0 re-assembly of gathered wire buffers on the GPU (rt_wire_unpack, one launch)."""
    import torch
    dev = gathered.device
    if outs is None:
        outs = [torch.empty((plan.height, plan.width, 4), dtype=dt, device=dev)
                for dt in (torch.float32, torch.float32, torch.float16)]
    root = tuple(v.data_ptr() for v in root_views) if root_views is not None else None
    tracer.wire_unpack(gathered.data_ptr(), plan.wire_bytes, plan.rank_pixels, outs[0].data_ptr(), outs[1].data_ptr(),
                       outs[2].data_ptr(), plan.width, plan.height, plan.strip_rows, plan.world, root=root,
                       root_strips=plan.root_weight, stream=stream)
    return outs
