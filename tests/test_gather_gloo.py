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


def _ring_worker(rank, world, port, q, emul_lib):
    """Each rank steps its own engine (host emulation library: no GPU here) and hands its trajectory ring to rank 0 with
    the double-buffered asynchronous gather that bench.py uses for N > 1."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import parity_common as pc
    from lifelike_agility_and_play_amd import gather, mocap, urdf_model
    blob, table = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02)
    unroll, n = 3, 6

    def run(seed):
        E = pc.make_engine(blob, table, n, emul_lib, auto_reset=1, seed=seed)
        E.reset()
        T = gather.TrajectoryBuffer(E, unroll, host_memory=True)
        halves = []
        for t in range(2 * unroll + 2):
            E.fill_random_actions(0.3)
            E.step()
            if (t + 1) % unroll == 0:
                k = (t + 1) // unroll - 1
                halves.append(T.half(k).clone())
                if seed == 100 + rank:                     # only the rank's own engine takes part in the collective
                    T.gather_async(k, 0)
        return T, halves
    T, mine = run(100 + rank)
    T.wait()
    if rank == 0:
        got = [o.clone() for o in T.last]                  # the last gathered unroll, one block per rank
        _, other = run(101)                                 # rank 1's engine, recomputed locally
        # the receive side is double-buffered: unroll 0 is still intact after unroll 1 has arrived (a learner has one unroll's time to consume it)
        first = [o.clone() for o in T.received(0)]
        assert all(a is not b for a, b in zip(T.received(0), T.received(1)))
        q.put((got[0].numpy(), mine[-1].numpy(), got[1].numpy(), other[-1].numpy(), T.n_gathered, first[0].numpy(), mine[0].numpy(), first[1].numpy(), other[0].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_async_ring_gather_two_ranks():
    import subprocess
    emul_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
    subprocess.check_call(['make', '-C', emul_dir, '-s', '-j2'])
    emul_lib = os.path.join(emul_dir, '_build', 'libllenv_emul.so')
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, q, emul_lib)) for r in range(world)]
    for p in procs:
        p.start()
    a0, b0, a1, b1, n_g, f0, g0, f1, g1 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert n_g == 2
    np.testing.assert_array_equal(a0, b0)                   # rank 0's own block
    np.testing.assert_array_equal(a1, b1)                   # rank 1's block == rank 1's engine recomputed on rank 0
    np.testing.assert_array_equal(f0, g0)                   # unroll 0 as received, read AFTER unroll 1 arrived: untouched
    np.testing.assert_array_equal(f1, g1)
    assert np.abs(f0 - a0).max() > 0
    assert a0.shape == (6, 3, 224) and np.abs(a0 - a1).max() > 0      # [n_envs][unroll][row]: every env's unroll contiguous
