"""N > 1 path on CPU: two gloo ranks pack trajectory rows and gather them to the learner rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lifelike_agility_and_play_amd import gather
    n, od, unroll = 8, 207, 4
    g = torch.Generator().manual_seed(100 + rank)
    buf = torch.empty((unroll, n, od + 14))
    for t in range(unroll):
        obs = torch.randn((n, od), generator=g); act = torch.randn((n, 12), generator=g)
        rew = torch.rand((n,), generator=g); done = (torch.rand((n,), generator=g) > 0.8).to(torch.uint8)
        gather.pack_rows(obs, act, rew, done, buf[t])
    out = gather.gather_unroll(buf, dst=0)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_local_packing():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (2, 4, 8, 221)
    for rank in range(world):          # recompute every rank's block locally
        g = torch.Generator().manual_seed(100 + rank)
        for t in range(4):
            obs = torch.randn((8, 207), generator=g); act = torch.randn((8, 12), generator=g)
            rew = torch.rand((8,), generator=g); done = (torch.rand((8,), generator=g) > 0.8).to(torch.uint8)
            row = got[rank, t]
            np.testing.assert_array_equal(row[:, :207], obs.numpy())
            np.testing.assert_array_equal(row[:, 207:219], act.numpy())
            np.testing.assert_array_equal(row[:, 219], rew.numpy())
            np.testing.assert_array_equal(row[:, 220], done.numpy().astype(np.float32))
