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
the seven links into the root carries one peer's buffer) and put back in image order by a
16-byte-per-lane copy kernel per surface (rt_deinterleave).  Results are bitwise identical to
the single-GPU frame.

Only plumbing lives here (torch owns device memory and the process group); rendering and
re-assembly are C-ABI calls.
"""
from dataclasses import dataclass

from . import layout as L

BPP = (16, 16, 8)   # bytes per pixel of gColor, gPosition, gNormal (raytracingCs.glsl:61-63)


@dataclass(frozen=True)
class StripPlan:
    width: int
    height: int
    strip_rows: int
    world: int

    @property
    def n_strips(self):
        return (self.height + self.strip_rows - 1) // self.strip_rows

    def local_rows(self, rank):
        rows = 0
        for s in range(rank, self.n_strips, self.world):
            rows += min((s + 1) * self.strip_rows, self.height) - s * self.strip_rows
        return rows

    @property
    def max_local_rows(self):
        """Rows of the (padded) per-rank buffer: whole strips, equal on every rank."""
        per_rank = (self.n_strips + self.world - 1) // self.world
        return per_rank * self.strip_rows

    @property
    def rank_bytes(self):
        return self.max_local_rows * self.width * sum(BPP)

    def surface_offsets(self):
        n = self.max_local_rows * self.width
        return 0, n * BPP[0], n * (BPP[0] + BPP[1])

    def global_row(self, rank, local_row):
        return ((local_row // self.strip_rows) * self.world + rank) * self.strip_rows + local_row % self.strip_rows

    def params(self, base, rank):
        """rt_params for this rank's packed strip buffer (max_local_rows x width); rows that
        fall beyond the image are written as zeros by the kernel."""
        return L.copy_params(base, x0=0, y0=0, regionW=self.width, regionH=self.max_local_rows,
                             stripRows=self.strip_rows, stripCount=self.world, stripIndex=rank)

    def row_index(self):
        """For each image row y: (rank, local row) flattened as rank * max_local_rows + local row."""
        idx = []
        for y in range(self.height):
            s = y // self.strip_rows
            r, ls = s % self.world, s // self.world
            idx.append(r * self.max_local_rows + ls * self.strip_rows + y % self.strip_rows)
        return idx


def default_strip_rows(height, world):
    """16-row strips (one workgroup tile) keep every rank's share of the benchmark scenes within
    a few percent of equal while leaving >= 8 strips per rank at 1080p on 8 GPUs."""
    return 16


def alloc_rank_buffer(plan, device):
    """One uint8 tensor holding this rank's three packed surfaces."""
    import torch
    return torch.empty(plan.rank_bytes, dtype=torch.uint8, device=device)


def surface_views(buf, plan):
    """(gColor f32[rows,W,4], gPosition f32[rows,W,4], gNormal f16[rows,W,4]) views of a rank buffer."""
    import torch
    n, w = plan.max_local_rows, plan.width
    o0, o1, o2 = plan.surface_offsets()
    col = buf[o0:o1].view(torch.float32).view(n, w, 4)
    pos = buf[o1:o2].view(torch.float32).view(n, w, 4)
    nrm = buf[o2:plan.rank_bytes].view(torch.float16).view(n, w, 4)
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
