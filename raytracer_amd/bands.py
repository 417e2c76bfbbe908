"""Row-band sharding of a frame over the ranks of a one-node job and the gather of the bands to rank 0.

Every pixel depends only on (x, y, seed_base) -- the scene is replicated, each pixel re-seeds its own RNG
(reference src/main.cpp:304) -- so ranks render disjoint rows with no data-path exchange; the only collective is the
gather of finished bands to rank 0 (RCCL over xGMI with backend "nccl", gloo in the CPU tests), which plays the
role of the reference's Redis tile list + recompose loop (src/redis.h:27-48, src/main.cpp:375-395).

Bands are dealt cyclically (band b goes to rank b % nranks): contiguous slabs would hand all-sky rows to one GPU and
the rows crossing the glass sphere to another.
"""
import torch
import torch.distributed as dist


def rank_rows(height, band_rows, nranks, rank):
    """Frame rows rendered by `rank`, in the packed order rt_render_rows writes them."""
    return [y for y in range(height) if (y // band_rows) % nranks == rank]


def rows_per_rank(height, band_rows, nranks):
    """Rows of the (padded, equal for every rank) per-rank block of the gather buffer."""
    nbands = (height + band_rows - 1) // band_rows
    return ((nbands + nranks - 1) // nranks) * band_rows


def band_params(renderer, width, height, samples, max_depth, seed_base, band_rows, nranks, rank, flags=0):
    """rt_params selecting this rank's cyclic bands."""
    n = len(rank_rows(height, band_rows, nranks, rank))
    return renderer.params(width, height, samples, max_depth, seed_base, row0=rank * band_rows, rows=n,
                           band_rows=band_rows, band_stride=band_rows * nranks, flags=flags)


def deinterleave(gathered, frame, band_rows, nranks, renderer=None):
    """gathered [nranks, rows_per_rank, W, C] -> frame [H, W, C].  On the GPU this is rt_deinterleave_bands; on CPU
    tensors (gloo tests) a plain index copy."""
    H = frame.shape[0]
    if gathered.is_cuda:
        # (Renderer waits for torch's current stream -- where the gather lands -- before the library's own stream reads)
        W = frame.shape[1]
        bpp = frame.element_size() * (frame.shape[2] if frame.dim() > 2 else 1)
        renderer.deinterleave_bands(gathered, frame, W, H, bpp, band_rows, nranks, gathered.shape[1])
        return frame
    for rank in range(nranks):
        ys = rank_rows(H, band_rows, nranks, rank)
        frame[ys] = gathered[rank, :len(ys)]
    return frame


def gather_frame(local_block, height, band_rows, renderer=None, group=None):
    """Gathers every rank's packed band block ([rows_per_rank, W, C], same shape on all ranks) to rank 0 and
    reassembles the frame there.  Returns the frame on rank 0, None elsewhere."""
    nranks = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if nranks == 1:
        gathered = local_block.unsqueeze(0)
    else:
        gathered = torch.empty((nranks,) + tuple(local_block.shape), dtype=local_block.dtype, device=local_block.device) if rank == 0 else None
        dist.gather(local_block, list(gathered.unbind(0)) if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.empty((height,) + tuple(local_block.shape[1:]), dtype=local_block.dtype, device=local_block.device)
    return deinterleave(gathered, frame, band_rows, nranks, renderer)
