"""Row-strip decomposition helpers: one process per GPU, strips ordered bottom (rank 0) to top.

`partition_rows` replaces Grid::initialise_2d_topology (LBMGrid.h:347-364) with a 1-D split of the rows, the only
decomposition for which the reference itself is exact. `GlooHalo` is the host-staged transport
(torch.distributed send/recv of the edge rows of each face) used where RCCL is not (CPU tests, ranks
sharing one GPU); the production device path is lbm_comm_init + RCCL inside the library.
"""
import numpy as np

UP = (2, 5, 6)     # populations with c_y = +1: travel north, consumed from the receiver's SOUTH ghost row
DOWN = (4, 7, 8)   # populations with c_y = -1: travel south, consumed from the receiver's NORTH ghost row


def partition_rows(ny, nranks):
    """[(y_start, local_ny)] for ranks 0..nranks-1; the first ny % nranks strips get one extra row."""
    if nranks < 1 or ny < nranks:
        raise ValueError(f"cannot cut {ny} rows into {nranks} strips")
    base, rem = divmod(ny, nranks)
    out, y = [], 0
    for r in range(nranks):
        n = base + (1 if r < rem else 0)
        out.append((y, n))
        y += n
    return out


class GlooHalo:
    """Exchange of the strip edge rows over torch.distributed point-to-point (any backend with CPU tensors)."""

    def __init__(self, rank, world, shape):
        """shape: the per-face message shape, e.g. (2, 9, nx) for Context.halo_export / halo_import."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.shape = rank, world, tuple(shape)

    def exchange(self, export_fn, import_fn):
        """export_fn(south: bool, north: bool) -> (south_out | None, north_out | None), float64 arrays of self.shape;
        import_fn(south=array | None, north=array | None)."""
        torch, dist = self.torch, self.dist
        has_s, has_n = self.rank > 0, self.rank < self.world - 1
        s_out, n_out = export_fn(has_s, has_n)
        ops, s_in, n_in = [], None, None
        if has_n:
            n_in = torch.empty(self.shape, dtype=torch.float64)
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(n_out)), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, n_in, self.rank + 1))
        if has_s:
            s_in = torch.empty(self.shape, dtype=torch.float64)
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(s_out)), self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, s_in, self.rank - 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        import_fn(south=s_in.numpy() if s_in is not None else None, north=n_in.numpy() if n_in is not None else None)
