"""Trajectory hand-off to the learner rank over RCCL/xGMI.

The reference's actor pushes one unroll (unroll_length = 128 steps of the flattened PGData row) to the learner over
ZeroMQ (learning/actors/distill_actor.py:84-176, push at :167).  Here each GPU rank keeps an [unroll][n_envs][row]
ring in HBM and, once per unroll, the rows are gathered to the learner rank with one RCCL collective
(torch.distributed backend "nccl" IS RCCL on ROCm).  Envs never communicate per step.

Row layout (float32, include/llenv.h ll_enable_unrolls): one flattened time step of the learner's data structure,
future[72] | prop | prop_a[36] | action[12] | neglogp | R | V | r | 1 - done   (224 floats for the PMC obs of 207), one env's
unroll contiguous: a rank's block [n_envs][unroll][224] is n_envs of the `unroll_np` arrays the reference pushes.
xGMI is point-to-point: a gather into rank 0 arrives over 7 different links, so it is per-link bound
(~153 GB/s per peer): 4096 envs x 128 steps x 896 B = 470 MB per rank per unroll ~ 3 ms, once per 128 steps.
"""
import numpy as np
import torch
import torch.distributed as dist


class _DevArray(object):
    """Zero-copy view of an engine-owned device buffer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


def host_tensor(ptr, shape):
    """float32 view of HOST memory (the test-only emulation library keeps its "device" buffers on the host)."""
    import ctypes
    n = int(np.prod(shape))
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr))).reshape(shape))


def device_tensor(ptr, shape, dtype=torch.float32, device=None):
    typestr = {torch.float32: '<f4', torch.uint8: '|u1', torch.int32: '<i4', torch.float64: '<f8'}[dtype]
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device if device is not None else torch.device('cuda', torch.cuda.current_device()))


def bind_torch_stream(engine, stream=None):
    """Make torch and the engine queue their work on ONE stream, so that a policy's kernels, the step kernel and the consumers
    of its buffers are ordered without host synchronisation.  torch's default stream has handle 0 and cannot be named through
    ll_set_stream, so a dedicated stream is created (or `stream` used), made torch's current stream, and given to the engine."""
    s = stream if stream is not None else torch.cuda.Stream()
    torch.cuda.set_stream(s)
    engine.set_stream(s.cuda_stream)
    return s


def use_engine_stream(engine):
    """The other direction of bind_torch_stream, for engines without a set_stream entry (EPMC, SEPMC): torch's current stream becomes
    the engine's own stream, so torch ops on the engine's buffers are ordered with its kernels.  The stream belongs to the engine: before
    the engine is closed, give torch another one (torch.cuda.set_stream(torch.cuda.default_stream())) -- torch ops queued on a destroyed
    stream fail with unrelated-looking errors."""
    s = torch.cuda.ExternalStream(int(engine.device_ptrs().stream))
    torch.cuda.set_stream(s)
    return s


def engine_tensors(engine):
    """torch views of the engine's output/action buffers (no copies).  n_envs counts robot rows for the SEPMC engine ([arena][robot])."""
    p = engine.device_ptrs()
    n, od = p.n_envs, p.obs_dim
    out = dict(obs=device_tensor(p.obs, (n, od)), reward=device_tensor(p.reward, (n,)),
               done=device_tensor(p.done, (n,), torch.uint8), actions=device_tensor(p.actions, (n, 12)))
    if p.terminal_obs:
        out['terminal_obs'] = device_tensor(p.terminal_obs, (n, od))
    return out


def pack_rows(obs, actions, reward, done, out):
    """out[n_envs][obs_dim+14] <- obs | action | reward | done   (works for CPU tensors too: used by the gloo test)."""
    od = obs.shape[1]
    out[:, :od].copy_(obs)
    out[:, od:od + 12].copy_(actions)
    out[:, od + 12].copy_(reward)
    out[:, od + 13].copy_(done.to(out.dtype))
    return out


def gather_unroll(local, dst=0, group=None):
    """Gather every rank's [unroll][n_envs][row] block to rank `dst`; returns [world][unroll][n_envs][row] there, else None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local.is_cuda and dist.get_backend(group) == 'gloo':     # test configuration only: gloo gathers host tensors
        local = local.cpu()
    if rank == dst:
        outs = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=outs, dst=dst, group=group)
        return torch.stack(outs, 0)
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


def flatten_unroll(obs_dicts, actions):
    """The reference's wire format for one env's unroll, host-side statement (distill_actor.py:159-162 over a data structure whose
    flatten walks the observation dict in sorted key order): time-major, per time step  future | prop | prop_a | action.
    Pinned by tests/golden/unroll_golden.npz; the device rows written by the step kernel start with exactly these floats."""
    return np.concatenate([np.concatenate([np.asarray(o[k]).reshape(-1) for k in sorted(o)] + [np.asarray(a).reshape(-1)]) for o, a in zip(obs_dicts, actions)])


UNROLL_FIELDS = ('X', 'A', 'neglogp', 'R', 'V', 'r', 'mask')


def split_row(rows, obs_dim):
    """views of an unroll block [..., obs_dim + 17]: the learner's inputs X, A, neglogp, R, V (pmc_net.py:61-96) and r, mask = 1 - done"""
    od = obs_dim
    return dict(X=rows[..., :od], A=rows[..., od:od + 12], neglogp=rows[..., od + 12], R=rows[..., od + 13], V=rows[..., od + 14],
                r=rows[..., od + 15], mask=rows[..., od + 16])


class TrajectoryBuffer(object):
    """The engine's own unroll buffers ([2][n_envs][unroll][row], written inside the step kernel) as a torch tensor, plus the
    hand-off to the learner rank.

    TWO blocks: while the steps of unroll k+1 fill one, the gather of unroll k (the other) is in flight on RCCL's own stream
    (``async_op``), so the collective overlaps with simulation instead of stalling it.  xGMI is point-to-point, so rank 0
    receives over 7 different links at once.

    What the overlap costs is measured, not assumed: every ``wait()`` brackets the point where the engine's stream has to wait for the
    collective with two events (``stall_ms()``: how long the step kernels actually stood still behind a gather) and counts the host time
    spent blocked.  ``mode='blocking'`` (bench.py --gather-mode blocking) waits for every gather before the next step is launched: the
    A/B leg that shows what the double buffer buys.

    The RECEIVE side on the learner rank is double-buffered too: unroll k lands in ``outs[k % 2]``, so a learner has the whole next unroll
    (21 ms at BASELINE config 3) to consume unroll k before anything overwrites it (``received(k)``).

    ``mode='p2p'`` (bench.py --gather-mode p2p) is the same hand-off over a different transport: no collective at all.  Every rank exports
    its unroll allocation and two interprocess events through HIP IPC (include/llenv_xfer.h); the learner rank pulls block k of every rank
    with SDMA copies (hipMemcpyDeviceToDeviceNoCU) on a copy stream of its own, each pull ordered behind the producer's event.  Nothing of
    it occupies a compute unit -- which matters here because the occupancy-1 step kernel shares its SIMDs with nobody: whatever a
    collective keeps resident, the step launches pay in full (profiles/r03_simd_sharing.txt).  Per unroll the ranks meet once on the host
    (a gloo barrier: "my event is recorded"); the producer's stream waits for "block copied" only when it is about to overwrite that block,
    one unroll later.  (ROCm's stream wait on an interprocess event is a HOST-side wait; `_gather_p2p` places the waits accordingly.)"""

    def __init__(self, engine, unroll, host_memory=False, mode='async'):
        ptr, w = engine.enable_unrolls(unroll, 2)
        self.engine = engine
        self.unroll = unroll
        self.mode = mode
        mk = host_tensor if host_memory else device_tensor
        self.buf = mk(ptr, (2, engine.n_envs, unroll, w))
        self.outs = None                  # learner rank: [2][world] receive blocks (unroll k lands in outs[k % 2])
        self.work = None
        self._p2p = None                  # mode 'p2p': IPC handles, events, copy stream (prepare)
        self.last = None
        self.n_gathered = 0
        self._stall_events = []           # (before, after) event pairs around stream-side waits
        self.host_stall_s = 0.0           # host time spent blocked in wait()
        self.extra_gathers = 0            # measurement hook: repeat every gather this many more times
        self.p2p_no_cu = True             # mode 'p2p': SDMA pulls (False: the runtime's default device-to-device path, the A/B leg)
        self._stage = None                # gloo test path: pinned staging buffers, side stream, worker thread
        self._thread = None
        self._thread_errors = {}          # exceptions raised inside the helper threads, one slot per helper ('_thread': gather / pulls, '_watcher'): re-raised on the launching thread at its next join
        self._watcher = None              # mode 'p2p': the thread that turns "the learner has copied block k" (an interprocess event) into the signal word

    def half(self, k):
        return self.buf[k % 2]

    def prepare(self, dst=0, group=None):
        """Allocate everything a gather needs BEFORE a timed region: rank `dst`'s world x block receive buffers (8 x 470 MB at
        BASELINE config 3) and, in the gloo test configuration, the pinned staging blocks.  Idempotent."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        staged = self.buf.is_cuda and dist.get_backend(group) == 'gloo'
        like = self.buf[0]
        if staged and self._stage is None:
            self._stage = dict(host=[torch.empty(like.shape, dtype=like.dtype, pin_memory=True) for _ in range(2)],
                               stream=torch.cuda.Stream(), ev=[torch.cuda.Event(), torch.cuda.Event()])
        if rank == dst and self.outs is None:
            dev = torch.device('cpu') if staged else like.device
            self.outs = [[torch.empty(like.shape, dtype=like.dtype, device=dev) for _ in range(world)] for _ in range(2)]
            for half in self.outs:
                for o in half:
                    o.zero_()                               # touch the pages now, not inside the first timed gather
        if self.mode == 'p2p' and self._p2p is None:
            self._prepare_p2p(dst, group)

    def _prepare_p2p(self, dst, group):
        """Exchange the IPC handles (collective: every rank calls it).  Learner rank: maps every other rank's unroll allocation, opens their
        'block ready' events, owns the copy stream and the two 'block copied' events; the others open those."""
        from . import xfer
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if not self.buf.is_cuda:
            raise RuntimeError("gather mode 'p2p' moves device memory (HIP IPC); the host-memory test configuration has no such path")
        dev = self.buf.device.index if self.buf.device.index is not None else torch.cuda.current_device()
        ctrl = dist.new_group(backend='gloo') if dist.get_backend(group) != 'gloo' else group     # host-side meeting point, no GPU work
        ready = [xfer.IpcEvent(dev) for _ in range(2)]
        mem_h, mem_off = xfer.export_mem(dev, self.buf.data_ptr())
        mine = dict(mem=mem_h, off=mem_off, ready=[e.handle for e in ready], pid=__import__('os').getpid())
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=ctrl)
        st = dict(ctrl=ctrl, ready=ready, dev=dev, dst=dst, block_bytes=int(self.buf[0].numel() * self.buf.element_size()), bases=[])
        # the producer's side without a host-side wait on the launching thread (include/llenv_xfer.h, round 5): the engine's stream waits ON THE DEVICE
        # for a word of signal memory; a watcher thread -- the only one that blocks on the learner's interprocess 'copied' event -- raises it
        st['signal'] = xfer.SignalWord(dev) if (xfer.can_wait_value(dev) and not __import__('os').environ.get('LL_P2P_HOST_WAIT')) else None
        st['aux'] = xfer.CopyStream(dev) if st['signal'] is not None else None
        copied_handles = None
        if rank == dst:
            # one copy stream and one pair of 'block copied' events PER PRODUCER: the pulls from different ranks arrive over different xGMI
            # links and may use different SDMA engines -- on one stream they would queue behind each other
            st['streams'] = [xfer.CopyStream(dev) for _ in range(world)]
            st['copied_all'] = [[xfer.IpcEvent(dev) for _ in range(2)] for _ in range(world)]
            copied_handles = [[e.handle for e in pair] for pair in st['copied_all']]
            st['src'], st['src_ready'] = [], []
            for r, info in enumerate(everyone):
                if r == rank:
                    st['src'].append(int(self.buf.data_ptr())); st['src_ready'].append(ready); st['bases'].append(None)
                else:
                    ptr = xfer.open_mem(dev, info['mem'], info['off'])
                    st['src'].append(ptr); st['bases'].append(ptr - info['off'])
                    st['src_ready'].append([xfer.IpcEvent(dev, h) for h in info['ready']])
        box = [copied_handles]
        dist.broadcast_object_list(box, src=dst, group=ctrl)
        st['copied'] = st['copied_all'][rank] if rank == dst else [xfer.IpcEvent(dev, h) for h in box[0][rank]]     # this rank's own blocks
        self._p2p = st

    def _join(self, which='_thread'):
        """join a helper thread and re-raise, on the launching thread, whatever it died of (a dead pull would otherwise leave the learner with a stale
        block and the producers waiting on an event that is never re-recorded)"""
        t = getattr(self, which)
        if t is not None:
            t.join()
            setattr(self, which, None)
        e = self._thread_errors.pop(which, None)
        if e is not None:
            raise RuntimeError('gather helper thread %s failed: %r' % (which, e)) from e

    def _spawn(self, fn, which='_thread'):
        import threading
        from . import xfer
        dev = self._p2p['dev'] if self._p2p is not None else None

        def run():
            try:
                if dev is not None:
                    xfer.set_device(dev)                     # a new thread starts on device 0
                fn()
            except BaseException as e:                      # noqa: BLE001  (kept for the launching thread: _join)
                self._thread_errors[which] = e
        t = threading.Thread(target=run)
        setattr(self, which, t)
        t.start()

    def _gather_p2p(self, k, dst, group):
        """Hand-off of unroll k without a collective.  ROCm implements hipStreamWaitEvent on an INTERPROCESS event as a host-side wait (the
        call returns once the event has completed), so no such wait is ever issued from the launching thread: the learner's waits for the
        producers' events run in a helper thread that then queues the pulls; a producer's wait for 'block copied' is a DEVICE-side wait of its engine's
        stream on a word of signal memory, which a watcher thread raises once the learner's interprocess event has completed (round 5; before, this
        was a host-side wait on the launching thread: + 5.8 % wall in profiles/r04_p2p_no_cu.txt).  The launching thread goes straight back to
        launching steps; per unroll it meets the other ranks once on the host (a gloo barrier: "my event is recorded")."""
        import time
        st = self._p2p
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        es = int(self.engine.device_ptrs().stream or 0)
        diag = __import__('os').environ.get('LL_P2P_DIAG', '')      # measurement hook (tools/diag_p2p_gaps.py): leave out one element of the hand-off at a time
        if diag == 'nothing':
            return
        if diag != 'norecord':
            st['ready'][k % 2].record(es)                   # behind the kernels that wrote block k (and its TD(lambda) pass)
        t0 = time.perf_counter()
        self._join('_thread')                               # learner rank: the pulls of unroll k - 1 are queued, 'copied' is recorded
        self._join('_watcher')                              # (the watcher of unroll k - 2: long done)
        dist.barrier(group=st['ctrl'])                      # every rank's event is recorded before anybody is told to wait for it
        if rank == dst:
            outs, half, n_pull = self.outs[k % 2], k % 2, 1 + self.extra_gathers

            def pull():
                for r in range(world):
                    cs = st['streams'][r]
                    if diag != 'norecord':
                        st['src_ready'][r][half].make_stream_wait(cs.handle)     # (returns when rank r's unroll k is complete)
                    for _ in range(n_pull if diag not in ('nopull', 'norecord') else 0):      # (extra_gathers: measurement hook -- the residency of more, or slower, peers)
                        cs.pull(outs[r].data_ptr(), st['src'][r] + half * st['block_bytes'], st['block_bytes'], no_cu=self.p2p_no_cu)
                    st['copied_all'][r][half].record(cs.handle)
            self._spawn(pull, '_thread')
            self.last = outs
        if k >= 1:
            # The next unroll (k + 1) overwrites block (k - 1) % 2: not before the learner has copied it.  That copy was queued one unroll ago
            # (its event was recorded before the learner entered this unroll's barrier: the join above).
            copied = st['copied'][(k - 1) % 2]
            if diag == 'nowait':
                pass
            elif st['signal'] is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()                                   # (torch's current stream IS the engine's: gather_async checks it)
                st['signal'].make_stream_wait(es, k)          # device-side: the engine's stream stands still until the word says "unrolls 0 .. k - 1 copied"
                b.record()
                self._stall_events.append((a, b))            # how long it actually stood still: stall_ms()

                def watch():
                    # The engine's stream is ALREADY parked on the word: whatever happens in here, the word is raised (a failed wait would otherwise leave the stream --
                    # and every later sync() or close() -- blocked for good); the exception itself reaches the launching thread at its next join of this helper.
                    try:
                        copied.synchronize()                 # (host-side, in this thread only)
                    finally:
                        st['signal'].write(st['aux'].handle, k)
                self._spawn(watch, '_watcher')
            else:
                copied.make_stream_wait(es)                  # (no hipStreamWaitValue32 on this device: the host-side wait of round 4)
        self.host_stall_s += time.perf_counter() - t0       # barrier + joins (+ the host-side wait of the fallback path)

    def close(self):
        """Join the helper threads, drain the copy streams and give back what prepare() took: IPC mappings, events, streams, the signal word."""
        from . import xfer
        try:
            self._join('_thread'); self._join('_watcher')
        finally:
            st, self._p2p = self._p2p, None
            if st is not None:
                for cs in st.get('streams', []) + ([st['aux']] if st.get('aux') else []):
                    cs.synchronize()
                for base in st.get('bases', []):
                    if base is not None:
                        xfer.close_mem(st['dev'], base)
                evs = {}
                for group_ in [st['ready'], st['copied']] + list(st.get('copied_all', [])) + list(st.get('src_ready', [])):
                    for e in group_:
                        evs[id(e)] = e
                for e in evs.values():
                    e.close()
                for cs in st.get('streams', []) + ([st['aux']] if st.get('aux') else []):
                    cs.close()
                if st.get('signal') is not None:
                    st['signal'].close()

    def received(self, k):
        """Learner rank: the list (one tensor per rank) unroll k was received into; valid until unroll k + 2 is gathered."""
        return self.outs[k % 2]

    def finish(self, k, gamma=0.95, lam=0.95):
        """TD(lambda) returns of unroll k (ll_finish_unroll), bootstrapped from the engine's value buffer: call it after the policy
        has evaluated the observation that follows the unroll and before gathering."""
        self.engine.finish_unroll(k % 2, gamma, lam)

    def wait(self):
        import time
        if self._thread is not None or self._watcher is not None or self._thread_errors:      # helper threads (gloo-staged gather; p2p pulls / watcher)
            t0 = time.perf_counter()
            self._join('_thread'); self._join('_watcher')
            self.host_stall_s += time.perf_counter() - t0
        if self._p2p is not None and 'streams' in self._p2p:
            t0 = time.perf_counter()
            for cs in self._p2p['streams']:
                cs.synchronize()                            # learner rank: every pull queued so far (the helper thread was joined above) has landed
            self.host_stall_s += time.perf_counter() - t0
        if self.work is not None:
            t0 = time.perf_counter()
            if self.buf.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.work.wait()                            # RCCL: the current stream waits for the collective's stream; the host does not
                b.record()
                self._stall_events.append((a, b))
            else:
                self.work.wait()
            self.host_stall_s += time.perf_counter() - t0
            self.work = None

    def stall_ms(self):
        """Total time the engine's stream stood still in wait() behind a gather since the last call (device side, by events)."""
        tot = 0.0
        for a, b in self._stall_events:
            b.synchronize()
            tot += a.elapsed_time(b)
        self._stall_events = []
        return tot

    def gather_async(self, k, dst=0, group=None):
        """Start gathering unroll k (the block the last `unroll` steps wrote); the previous gather must have finished."""
        if self.mode != 'p2p':
            self.wait()                                     # (p2p: ordering is by events; nothing to wait for on the host)
        local = self.half(k)
        if local.is_cuda:
            # the collective (and the staging copy of the gloo test path) is ordered against torch's CURRENT stream only: the step
            # kernels that wrote this block must be on that very stream, or the gather reads rows that are still being written
            es, ts = int(self.engine.device_ptrs().stream or 0), int(torch.cuda.current_stream().cuda_stream)
            if es != ts:
                raise RuntimeError('TrajectoryBuffer.gather_async: the engine launches on stream %#x but torch\'s current stream is %#x; '
                                   'call gather.bind_torch_stream(engine) first' % (es, ts))
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.prepare(dst, group)
        if self.mode == 'p2p':
            self._gather_p2p(k, dst, group)
            self.n_gathered += 1
            return
        if local.is_cuda and dist.get_backend(group) == 'gloo':
            # Test configuration only (a 1-GPU box cannot host two RCCL ranks): gloo gathers host tensors.  The block is copied to pinned
            # memory on a side stream behind the step kernels that wrote it, and a helper thread hands it to gloo once the copy has landed,
            # so that -- as with RCCL -- neither the host nor the engine's stream waits for the gather until the next one is due.
            import threading
            st = self._stage
            host, ev = st['host'][k % 2], st['ev'][k % 2]
            st['stream'].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st['stream']):
                host.copy_(local, non_blocking=True)
                ev.record()
            outs = self.outs[k % 2] if rank == dst else None

            def run():
                try:
                    ev.synchronize()
                    dist.gather(host, gather_list=outs, dst=dst, group=group)
                except BaseException as e:                  # noqa: BLE001  (re-raised on the launching thread: _join)
                    self._thread_errors['_thread'] = e
            self._thread = threading.Thread(target=run)
            self._thread.start()
            if rank == dst:
                self.last = self.outs[k % 2]
        elif rank == dst:
            self.work = dist.gather(local, gather_list=self.outs[k % 2], dst=dst, group=group, async_op=True)
            for _ in range(self.extra_gathers):             # measurement hook (bench.py LL_BENCH_GATHER_REPEAT): RCCL resident for longer
                self.work = dist.gather(local, gather_list=self.outs[k % 2], dst=dst, group=group, async_op=True)
            self.last = self.outs[k % 2]
        else:
            self.work = dist.gather(local, gather_list=None, dst=dst, group=group, async_op=True)
        self.n_gathered += 1
        if self.mode == 'blocking':                         # A/B leg: nothing is launched until the gather has finished
            self.wait()
            if local.is_cuda:
                torch.cuda.current_stream().synchronize()
