"""Multi-GPU host logic: one process per GPU, photons of one pass sharded by index, one
all-reduce (sum) of the per-pixel flux image per SPPM iteration.

The reference's only parallelism is four independent replicas whose images are added
(raytracer/Raytracer.h:442-458).  Because HitPoint::update() never runs (Raytracer.h:74) and
the resolve is linear in tao (Raytracer.h:288-293), a pass can equally be split by photon
index: every rank runs the cheap eye pass, traces its own photon range, resolves its partial
flux image with the normaliser of the WHOLE pass, and the images add up.
"""
import os


def photon_shard(total_photons, rank, world):
    """[begin, count) of rank's contiguous share of a pass of total_photons photons."""
    base, rem = divmod(int(total_photons), int(world))
    begin = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return begin, count


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def all_reduce_image(pic):
    """Sum the flux image over ranks in place (RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(pic, op=dist.ReduceOp.SUM)
    return pic


def sharded_iteration(ctx, cam, seed, replica, total_rounds, per_round, n_lights, rank, world, pic_tensor=None, stream=None,
                      via_host=False):
    """One SPPM iteration of one replica on this rank's shard.  ctx is any object with the
    SppmContext methods; pic_tensor (device float32 [h,w,3]) receives the partial image and is
    all-reduced in place; without it the host image is returned un-reduced.

    stream: torch's handle of the stream the context queues on (torch.cuda.ExternalStream(ctx.stream_handle()),
    or a torch stream given to the context with ctx.set_stream(stream.cuda_stream)).  The
    collective is then issued under that stream, so it is ordered after the resolve, and whatever the context
    queues next after the collective, without the host waiting: iterations of other contexts keep the GPU busy
    meanwhile.  Without it the host waits for the context before the collective.

    via_host: the process group is gloo (ranks that share one GPU cannot form an RCCL communicator): the image is copied
    to host memory, reduced there and copied back, each step synchronous (gloo only ever sees CPU tensors)."""
    total = total_rounds * per_round * n_lights
    begin, count = photon_shard(total, rank, world)
    ctx.eye_pass(cam, seed, replica)
    if count > 0:
        ctx.photon_pass(total_rounds, per_round, seed, replica, photon_begin=begin, photon_count=count)
    else:
        # photon_count == 0 means "all" at the C-ABI, so an empty shard traces a zero-length range
        ctx.photon_pass(total_rounds, per_round, seed, replica, photon_begin=total, photon_count=1)
    if pic_tensor is not None:
        ctx.resolve_device(pic_tensor.data_ptr())
        if world > 1 and via_host:
            ctx.synchronize()
            host = pic_tensor.cpu()
            all_reduce_image(host)
            pic_tensor.copy_(host)
            return pic_tensor
        if world > 1 and stream is not None:
            import torch
            with torch.cuda.stream(stream):     # the process group orders its own stream after and before this one
                return all_reduce_image(pic_tensor)
        if world > 1:
            # the resolve runs on the library's stream, the collective on torch's: finish the former first
            # (and surface a queue overflow of the queued passes) instead of relying on implicit stream ordering
            ctx.synchronize()
        return all_reduce_image(pic_tensor)
    return ctx.resolve()
