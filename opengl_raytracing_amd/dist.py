"""Multi-GPU tiling of one frame: interleaved row strips, one process per GPU, one RCCL gather.

The reference is single-GPU; pixels are independent (raytracingCs.glsl reads only
gl_GlobalInvocationID, uniforms and read-only buffers), so the frame shards by rows with no
data-path exchange until the final image is assembled.  Equal contiguous tiles are badly
unbalanced (sky rows end at depth 0, floor rows run the full bounce loop), so rank r renders
strips s with s % world == r of `strip_rows` rows each; the kernel maps local
rows to image rows itself (rt_params.stripRows/stripCount/stripIndex), writing a PACKED strip
buffer that is gathered to rank 0 with a single `torch.distributed.gather` (RCCL over xGMI: all
seven links into the root carry one peer's buffer each) and put back in image order by a
16-byte-per-lane copy kernel (rt_deinterleave).  Results are bitwise identical to the
single-GPU frame.

Only plumbing lives here (torch owns device memory and the process group); rendering is the
C-ABI call.
"""
from dataclasses import dataclass

from . import layout as L


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

    def global_row(self, rank, local_row):
        return ((local_row // self.strip_rows) * self.world + rank) * self.strip_rows + local_row % self.strip_rows

    def params(self, base, rank):
        """rt_params for this rank's packed strip buffer (max_local_rows x width); rows that
        fall beyond the image are written as zeros by the kernel."""
        return L.copy_params(base, x0=0, y0=0, regionW=self.width, regionH=self.max_local_rows,
                             stripRows=self.strip_rows, stripCount=self.world, stripIndex=rank)

    def row_index(self):
        """For each image row y: index into the concatenated [world * max_local_rows] gathered rows."""
        idx = []
        for y in range(self.height):
            s = y // self.strip_rows
            r, ls = s % self.world, s // self.world
            idx.append(r * self.max_local_rows + ls * self.strip_rows + y % self.strip_rows)
        return idx


def default_strip_rows(height, world):
    """16-row strips (one workgroup tile) keep every rank's share within a few percent of
    equal for the benchmark scenes while leaving >= 2 strips per rank at 1080p on 8 GPUs."""
    return 16


def gather_strips(local, plan, rank, group=None, dst=0):
    """`local`: list of this rank's packed surfaces, each a torch tensor [max_local_rows, W, C].
    Returns on rank `dst` a list of tensors [world * max_local_rows, W, C] (rank-major), else None.
    One collective per surface; with the NCCL(=RCCL) backend this is a grouped send/recv gather."""
    import torch
    import torch.distributed as dist

    out = []
    for t in local:
        if plan.world == 1:
            out.append(t)
            continue
        if rank == dst:
            buf = torch.empty((plan.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.gather(t, gather_list=list(buf.unbind(0)), dst=dst, group=group)
            out.append(buf.view((plan.world * t.shape[0],) + tuple(t.shape[1:])))
        else:
            dist.gather(t, gather_list=None, dst=dst, group=group)
    return out if rank == dst else None


def deinterleave_torch(gathered, plan):
    """Reference re-assembly with torch indexing (CPU tests and a cross-check of the HIP copy
    kernel): gathered [world*max_local_rows, W, C] -> [height, W, C]."""
    import torch

    idx = torch.tensor(plan.row_index(), dtype=torch.long, device=gathered.device)
    return gathered.index_select(0, idx)


def deinterleave_hip(tracer, gathered, plan, out=None, stream=None):
    """Rank-0 re-assembly on the GPU through the C ABI (rt_deinterleave)."""
    import torch

    bpp = gathered.shape[2] * gathered.element_size()
    if out is None:
        out = torch.empty((plan.height,) + tuple(gathered.shape[1:]), dtype=gathered.dtype, device=gathered.device)
    tracer.deinterleave(gathered.data_ptr(), out.data_ptr(), plan.width, plan.height, bpp, plan.strip_rows,
                        plan.world, plan.max_local_rows, stream=stream)
    return out
