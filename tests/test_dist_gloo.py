"""The path's one exchange step (observer min/max across data-parallel ranks) on world_size-2 gloo,
CPU only: each rank reduces its shard, the packed [max | -min] vector goes through ONE all_reduce(MAX),
and every rank derives the same (scale, offset) as a single process observing the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fakequant_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_minmax(x, ch_axis):
    if ch_axis is None:
        return x.max().reshape(1), (-x.min()).reshape(1), x.abs().max().reshape(1)
    red = tuple(i for i in range(x.dim()) if i != ch_axis)
    return x.amax(dim=red), -x.amin(dim=red), x.abs().amax(dim=red)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "dlmc-quant_amd")]
    from dlmc.quantization.scalar._wrapper import allreduce_minmax
    g = torch.Generator().manual_seed(2333)
    full = torch.randn(8, 6, 5, 5, generator=g)
    full[5, 2, 1, 1] = 9.0       # the extremes live on different ranks
    full[1, 4, 0, 0] = -7.0
    shard = full.chunk(world, dim=0)[rank]
    out = {}
    for ch_axis in (None, 1):
        mx, nmn, ab = _local_minmax(shard, ch_axis)
        gmx, gnmn = allreduce_minmax(mx.clone(), nmn.clone())
        gab, none = allreduce_minmax(ab.clone())
        assert none is None
        out[ch_axis] = (gmx.tolist(), gnmn.tolist(), gab.tolist())     # (plain lists: a tensor in the queue travels as a shared-memory
    q.put((rank, out))                                                  #  file that is gone if this process exits before the parent reads it)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_observer_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(2333)
    full = torch.randn(8, 6, 5, 5, generator=g)
    full[5, 2, 1, 1] = 9.0
    full[1, 4, 0, 0] = -7.0
    for ch_axis in (None, 1):
        wmx, wnmn, wab = _local_minmax(full, ch_axis)
        for rank in range(world):
            gmx, gnmn, gab = (torch.tensor(v) for v in results[rank][ch_axis])
            assert torch.equal(gmx, wmx) and torch.equal(gnmn, wnmn) and torch.equal(gab, wab)
        # and the scale/offset every rank derives == the single-process observer over the whole batch
        gmx, gnmn, gab = (torch.tensor(v) for v in results[0][ch_axis])
        if ch_axis is None:
            s_u, o_u = O.minmax_tensor(full, 8, False)
            s_s, _ = O.minmax_tensor(full, 8, True)
        else:
            s_u, o_u = O.minmax_channel(full, 8, False, ch_axis=1)
            s_s, _ = O.minmax_channel(full, 8, True, ch_axis=1)
        assert torch.equal(((gmx - (-gnmn)) / 255).reshape(-1), s_u.reshape(-1))
        assert torch.equal((-gnmn).reshape(-1), o_u.reshape(-1))
        assert torch.equal((gab / 127).reshape(-1), s_s.reshape(-1))


def test_single_process_is_a_no_op():
    import sys
    from dlmc.quantization.scalar._wrapper import allreduce_minmax
    a, b = torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0])
    x, y = allreduce_minmax(a, b)
    assert x is a and y is b
