"""Multi-GPU split of one frame: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" for the CPU rehearsal in tests/).

The image partitions by SAMPLE RANGE: rank g renders samples [b_g, e_g) of every pixel into its own
f32 accumulation buffer and the buffers are summed with ONE reduce.  The Philox stream is keyed by the
global sample index (path id = sample*W*H + pixel), so the summed image does not depend on the number
of ranks beyond f32 summation order.  The scene (< 1 MB) is uploaded by every rank itself; there is no
other collective on the path.
"""


def sample_range(spp_total, world_size, rank):
    """Contiguous, balanced split of [0, spp_total) — the first spp_total % world_size ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    q, r = divmod(int(spp_total), int(world_size))
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def render_sharded(render_range, spp_total, accum, dist=None, dst=0):
    """render_range(spp_begin, spp_end, accum) must ADD the samples into `accum` (a torch tensor on this
    rank's device).  Afterwards rank `dst` holds the sum over all ranks.  Returns this rank's range."""
    if dist is not None and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    b, e = sample_range(spp_total, world, rank)
    if e > b:
        render_range(b, e, accum)
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return b, e
