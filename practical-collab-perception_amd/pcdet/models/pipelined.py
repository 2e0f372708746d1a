"""Software pipelining of consecutive inference batches on one MI355X (pipeline mode, not in the reference).

A detector forward ends with ONE host read -- how many boxes survive per frame (CenterHead.finalize) -- and a DiscoNet forward starts with
one -- which agents hold points (BEVMaker, reference bev_maker.py:156).  Run batch by batch, each of them drains the GPU: the kernels of the
next batch are launched only after the read returns, and the first ~0.5 ms of every batch are launch-bound (rocprofv3 kernel trace of
bench.py: 10 % of the step idle, profiles/r03_gpu_idle_*.txt).  `PipelinedDetector` keeps the device queue full instead:

  submit(batch i):   [side stream]  copy-in of the points + agent histogram + its 528-byte read  (does not wait for batch i-1's kernels)
                     [main stream]  the whole module chain of batch i, decode, NMS, the gather of the detections, an async copy of the counts
                     then, on the host, the counts of batch i-1 (finished long ago) -> its exact-shape pred_dicts, which it returns
  flush():           the pred_dicts of the last submitted batch

Same kernels, same inputs, same order per batch: every batch's pred_dicts are bit-identical to `model(batch_dict)` run batch by batch
(tests/test_gpu_e2e.py::test_pipelined_detector_*).  A model that corrects the points in place (HunterJr) works on its batch's own buffer
(`points` of submit(), refilled through `copy_from`); not for training.

graph=True (round 6): each replica's whole forward -- static agent discovery, BEV-maker streams as parallel branches, trunk, fusion, head, decode,
NMS, the gather of the detections -- is captured ONCE per (replica, points buffer, batch size, agent-presence pattern) as a hipGraph and replayed per batch: the
Python host then spends ~0.3 ms per batch (copy-in, one graph launch, the counts copy, an event) instead of enqueueing ~210 launches through
ctypes (3.8 ms measured), which is what keeps eight one-rank-per-GPU hosts from competing for cores.  Same kernels, same arguments, same order:
bitwise the eager pipelined detections (tests/test_gpu_bench_mode.py, test_gpu_e2e.py).  What a capture freezes: the row count of the batch,
the batch size and WHICH agents each frame's metadata lists -- a batch that differs in any of them gets its own capture (kept, up to a small
number).  The agents' POSES are data: the makers' pose tables and the fusion's warp affines live in device memory (`PoseTable`,
pcp_select_transform_compact_dev / pcp_warp_nearest_batch_dev) that the runner refreshes from the batch's metadata before every replay, so
one capture serves every pose set.  The mode suits fixed-shape streams (bench.py; a dataloader that pads its clouds to a capacity with
frame index -1 rows).
"""
import os

import torch

from pcp_amd import ops


class PoseTable:
    """Device-resident copies of the pose-derived launch parameters of a captured forward (graph mode).  While the forward is warmed and
    captured, the modules ask `slot(name, host_array)` for the device tensor their kernels read (BEVMaker: pose / presence tables of
    pcp_select_transform_compact_dev; V2XMidFusionDisco: the affines of pcp_warp_nearest_batch_dev).  Before a replay the runner computes the
    same arrays from the new batch's metadata (`model.pose_arrays`) and `upload`s them: two pinned staging buffers per entry, used
    alternately, one async copy each on the replica's stream in front of the replay -- the copies stay OUTSIDE the graph."""

    def __init__(self, device):
        self.device = device
        self.entries = {}                     # name -> dict(host=[pinned, pinned], dev=tensor)

    def slot(self, name, host_array):
        import numpy as np
        a = np.ascontiguousarray(host_array)
        ent = self.entries.get(name)
        if ent is None:
            t = torch.from_numpy(a.copy())
            ent = self.entries[name] = dict(host=[t.clone().pin_memory(), t.clone().pin_memory()], dev=torch.empty(t.shape, dtype=t.dtype, device=self.device))
        assert tuple(ent['dev'].shape) == a.shape and ent['host'][0].numpy().dtype == a.dtype, (name, a.shape, tuple(ent['dev'].shape))
        if not torch.cuda.is_current_stream_capturing():
            ent['host'][0].numpy()[...] = a                  # the warm pass reads real values; under capture the table is left as uploaded
            ent['dev'].copy_(ent['host'][0], non_blocking=True)
        return ent['dev']

    def upload(self, arrays, which):
        """arrays: name -> host array (every entry of the table must be given); `which` alternates 0 / 1 between consecutive uploads"""
        import numpy as np
        assert set(arrays) == set(self.entries), (sorted(arrays), sorted(self.entries))
        for name, a in arrays.items():
            ent = self.entries[name]
            a = np.ascontiguousarray(a)
            assert tuple(ent['dev'].shape) == a.shape, (name, a.shape, tuple(ent['dev'].shape))
            ent['host'][which].numpy()[...] = a
            ent['dev'].copy_(ent['host'][which], non_blocking=True)


class PaddedPoints:
    """Fixed-shape input for the graph mode: a batch's rows at the front of a (capacity, C) buffer, the rows behind them with frame index -1
    (every kernel of the path drops such rows: the pillariser's mask, the agent selection, HunterJr's guards) and agent id -1.  Two buffers,
    used alternately -- the buffer's address is part of a capture's key, and the forward of batch i may still read its buffer while batch
    i + 1 is prepared."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.bufs = [None, None]
        self.n = 0

    def fill(self, points):
        n, c = points.shape
        if n > self.capacity:
            raise ValueError('batch of %d rows exceeds the padded capacity %d' % (n, self.capacity))
        k = self.n & 1
        self.n += 1
        buf = self.bufs[k]
        if buf is None or buf.shape[1] != c or buf.device != points.device:
            buf = self.bufs[k] = torch.zeros((self.capacity, c), dtype=torch.float32, device=points.device)
        buf[:n].copy_(points)
        if n < self.capacity:
            buf[n:, 0] = -1.0
            buf[n:, c - 1] = -1.0
        return buf


class PipelinedDetector:
    @staticmethod
    def supports(model):
        """CenterPoint-style detectors (module chain + CenterHead with a deferred finalize) without a point corrector"""
        head = getattr(model, 'dense_head', None)
        return (hasattr(model, '_run_modules') and not PipelinedDetector._corrector_with_makers(model) and head is not None
                and hasattr(head, 'gather_pending') and hasattr(head, 'device_postprocess')
                and not PipelinedDetector._head_writes_exchange_data(head))

    @staticmethod
    def _head_writes_exchange_data(head):
        # GENERATING_EXCHANGE_DATA / RETURN_MODAR_POINTS are served AFTER the finalize of CenterHead.forward (the *_modar.pth files and
        # batch_dict['mo_pts'] of the reference's exchange-database workflow, center_head.py:207-223); a deferred finalize would skip
        # them silently, so such a head runs batch by batch
        cfg = getattr(head, 'model_cfg', None)
        get = (lambda k: cfg.get(k, False)) if hasattr(cfg, 'get') else (lambda k: getattr(cfg, k, False))
        return bool(cfg is not None and (get('GENERATING_EXCHANGE_DATA') or get('RETURN_MODAR_POINTS')))

    @staticmethod
    def _corrector_with_makers(model):
        # a corrector (HunterJr) rewrites the batch's points in place: fine on the batch's own buffer, not beside BEV makers that read them
        return getattr(model, 'corrector', None) is not None and any(type(m).__name__ == 'BEVMaker' for m in model.module_list)

    MAX_GRAPHS = 8

    def __init__(self, model, replicas=1, graph=False):
        """replicas = 2: consecutive batches alternate between the model and a deep copy of it (same weights, its own persistent buffers and
        packed weight forms), each on its own HIP stream -- batch i+1's whole forward may then run beside batch i's instead of behind it
        (launches that leave CUs idle, e.g. the one-round layers at four frames, the head, decode and NMS, are filled by the other batch).
        Batches of one replica stay in order on its stream; the two replicas share nothing mutable (the library's scratch buffers are keyed
        by stream)."""
        assert not model.training
        if self._corrector_with_makers(model):
            raise NotImplementedError('PipelinedDetector: a point corrector beside BEV makers runs batch by batch')
        if self._head_writes_exchange_data(model.dense_head):
            raise NotImplementedError('PipelinedDetector: a head that writes exchange data (MoDAR) runs batch by batch')
        self.model = model
        self.models = [model]
        if replicas > 1:
            import copy
            streams = getattr(model, '_maker_streams', None)
            model._maker_streams = None                      # HIP streams are not copyable: every replica makes its own
            try:
                self.models += [copy.deepcopy(model) for _ in range(replicas - 1)]
            except Exception as e:                           # a model that cannot be copied still pipelines, on one replica
                import warnings
                warnings.warn('PipelinedDetector: the model could not be deep-copied (%s): one replica' % e)
                self.models = [model]
            finally:
                model._maker_streams = streams
        self.mains = None
        self.head = model.dense_head
        self.side = None
        self._pending = None          # (ob, os_, ol, counts_host, event, batch_size, outputs are static)
        self._pinned = {}             # two pinned count buffers per batch size, used alternately
        self._events = None           # two blocking events, used alternately (at most two batches are pending: the one being read and the newest)
        self._n = 0
        self.graph = bool(graph)
        self._graphs = {}             # (replica, points ptr, shape, batch size, agent-presence pattern) -> captured forward
        self._evicted = []            # captures that made room for newer ones, kept alive until a device synchronisation
        # PCP_PIPELINE_EARLY_MAKERS=0: the maker streams of batch i+1 wait for the main stream (i.e. for batch i's tail), as `model()` does
        self.early_makers = os.environ.get('PCP_PIPELINE_EARLY_MAKERS', '1') != '0'
        self._has_makers = any(type(m).__name__ == 'BEVMaker' and m.maker_type in ('rsu', 'car') for m in model.module_list)

    def _discover(self, points, batch_dict):
        """the agent histogram of a DiscoNet batch on the side stream: its host read waits for that stream only"""
        ids, rows = ops.column_id_counts(points, -1)
        batch_dict['_pcp_agent_ids'] = (points, ids, rows)

    @torch.no_grad()
    def submit(self, points, batch_size, metadata, copy_from=None, extra=None):
        """points: (N, C) CUDA tensor the forward reads (it must stay untouched until the NEXT submit returns); copy_from: optional source
        tensor copied into `points` first (on the side stream), e.g. the upload of the batch; extra: the other entries of a dataloader's
        batch_dict (frame ids, ...), passed through to the modules.  Returns the pred_dicts of the PREVIOUS batch (None for the first)."""
        if self.side is None:
            self.side = torch.cuda.Stream()
            if len(self.models) > 1 or self.graph:           # a capture needs a stream of its own (never the legacy default stream)
                # PCP_PIPELINE_PRIO (diagnostic): 'trunk' = the replicas' streams above the BEV-maker streams, 'first' = replica 0 above replica 1
                prio = os.environ.get('PCP_PIPELINE_PRIO', '')
                pr = [(-1 if (prio == 'trunk' or (prio == 'first' and i == 0)) else 0) for i in range(len(self.models))]
                self.mains = [torch.cuda.Stream(priority=p_) for p_ in pr]
        r = self._n % len(self.models)
        model = self.models[r]
        head = model.dense_head
        main = self.mains[r] if self.mains is not None else torch.cuda.current_stream()
        if self.mains is not None:
            main.wait_stream(torch.cuda.current_stream())    # whatever the caller queued before (first batches: nothing)
        bd = dict(extra) if extra is not None else {}
        bd.update({'points': points, 'batch_size': batch_size, 'metadata': metadata})
        if self.graph:
            return self._submit_graph(r, model, main, bd, copy_from)
        if copy_from is not None or self._has_makers:
            if copy_from is None:
                # the caller filled `points` itself: whatever did that on the caller's stream (a non_blocking upload, a preprocessing
                # kernel) is ordered in front of the side stream's reads.  With copy_from the contract is a host-visible source
                # (pinned / pageable memory or a tensor that is complete): the side stream must NOT wait for the caller's stream, which
                # with one replica is the stream batch i-1 still runs on
                self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                if copy_from is not None:
                    points.copy_(copy_from, non_blocking=True)
                if self._has_makers:
                    self._discover(points, bd)
                ready = self.side.record_event()
            main.wait_event(ready)
            if copy_from is not None and self.early_makers:
                bd['_pcp_points_ready'] = ready              # the BEV-maker streams start from here, not from the main stream's position
        if self._has_makers and os.environ.get('PCP_PIPELINE_STATIC_AGENTS') == '1':
            bd['_pcp_static_agents'] = True                  # diagnostic: the graph mode's device-side agent discovery in the eager runner
        with torch.cuda.stream(main):
            head.defer_finalize = True
            try:
                bd = model._run_modules(bd)
            finally:
                head.defer_finalize = False
            ob, os_, ol, cnt = head.gather_pending(bd['_pcp_pending_head'], batch_size)
            key = (tuple(cnt.shape), cnt.dtype)
            if key not in self._pinned:
                self._pinned[key] = [torch.empty(cnt.shape, dtype=cnt.dtype, pin_memory=True) for _ in range(2)]
            counts_host = self._pinned[key][self._n & 1]
            self._n += 1
            counts_host.copy_(cnt, non_blocking=True)
            # a BLOCKING event (hipEventBlockingSync): the host thread that waits for batch i-1's counts sleeps instead of spinning on the
            # event -- with one Python host per GPU and eight GPUs per box, a spinning waiter is a core taken from another rank's enqueueing
            ev = self._events[self._n & 1] if self._events else None
            if ev is None:
                self._events = [torch.cuda.Event(blocking=True), torch.cuda.Event(blocking=True)]
                ev = self._events[self._n & 1]
            ev.record(main)
        prev, self._pending = self._pending, (ob, os_, ol, counts_host, ev, batch_size, False)
        return self._finish(prev)

    # ---- graph mode ---------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _structure_key(metadata):
        """which agents each frame's metadata lists: the part of the metadata a capture freezes (it decides the maker passes, the warped pairs
        and the zero-filled maps).  The POSES themselves are data: they live in the graph's pose table and are refreshed before every replay."""
        return tuple(tuple(sorted(int(a) for a in (meta.get('se3_from_ego', None) or {}))) if isinstance(meta, dict) else () for meta in metadata)

    def _capture(self, model, main, bd):
        """one eager forward in the capture's form (static agent discovery: no host read), then the capture itself, both on the replica's stream"""
        head = model.dense_head
        batch_size = bd['batch_size']

        table = PoseTable(bd['points'].device) if hasattr(model, 'pose_arrays') else None

        def run():
            d = dict(bd)
            if self._has_makers or any(type(m).__name__ == 'BEVMaker' for m in model.module_list):
                d['_pcp_static_agents'] = True
                if table is not None:
                    d['_pcp_pose_table'] = table
            head.defer_finalize = True
            # the BEV-maker passes are captured IN SEQUENCE on the replica's stream, not as parallel branches: replaying a graph with
            # three side branches per replica measured 10.84 ms per headline step against 10.43 ms for the sequential capture (hipGraph
            # schedules the branches' kernels with gaps; the eager runner's overlapped makers are worth 1 % -- profiles/r06_pipeline_graph_ab.txt)
            overlap, model.overlap_makers = getattr(model, 'overlap_makers', False), False
            try:
                d = model._run_modules(d)
            finally:
                head.defer_finalize = False
                model.overlap_makers = overlap
            return head.gather_pending(d['_pcp_pending_head'], batch_size)
        with torch.cuda.stream(main):
            run()                                            # lazy allocations / packed forms of the static form happen here, not under capture
        main.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            ob, os_, ol, cnt = run()
        if table is not None and not table.entries:
            table = None                                     # a model without pose-derived parameters (single-agent configs)
        return dict(graph=g, ob=ob, os=os_, ol=ol, cnt=cnt, table=table, uploads=0)

    def _submit_graph(self, r, model, main, bd, copy_from):
        points, batch_size = bd['points'], bd['batch_size']
        key = (r, points.data_ptr(), tuple(points.shape), int(batch_size), self._structure_key(bd['metadata']))
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.MAX_GRAPHS:
                # the oldest capture makes room.  Its last replay may still be running (it can even be the pending batch's): the entry is parked,
                # and parked entries are only dropped behind a device synchronisation -- destroying an executing graph is not defined
                self._evicted.append(self._graphs.pop(next(iter(self._graphs))))
                if len(self._evicted) > 4:
                    torch.cuda.synchronize()
                    del self._evicted[:-1]
            if copy_from is not None:
                with torch.cuda.stream(main):
                    points.copy_(copy_from, non_blocking=True)     # the capture's eager pass reads real points
            ent = self._graphs[key] = self._capture(model, main, bd)
        with torch.cuda.stream(main):
            if copy_from is not None:
                points.copy_(copy_from, non_blocking=True)
            if ent['table'] is not None:
                # this batch's poses -> the device tables the captured kernels read (host arithmetic only; a staging buffer is reused two
                # uploads later, when the batch that read it has been waited for)
                ent['table'].upload(model.pose_arrays(bd['metadata'], batch_size), ent['uploads'] & 1)
                ent['uploads'] += 1
            ent['graph'].replay()
            cnt = ent['cnt']
            ckey = (tuple(cnt.shape), cnt.dtype)
            if ckey not in self._pinned:
                self._pinned[ckey] = [torch.empty(cnt.shape, dtype=cnt.dtype, pin_memory=True) for _ in range(2)]
            counts_host = self._pinned[ckey][self._n & 1]
            self._n += 1
            counts_host.copy_(cnt, non_blocking=True)
            if not self._events:
                self._events = [torch.cuda.Event(blocking=True), torch.cuda.Event(blocking=True)]
            ev = self._events[self._n & 1]
            ev.record(main)
        # the graph's output tensors are static (the replica's next replay overwrites them): _finish hands out copies
        prev, self._pending = self._pending, (ent['ob'], ent['os'], ent['ol'], counts_host, ev, batch_size, True)
        return self._finish(prev)

    def prepare(self, points, batch_size, metadata):
        """one forward per replica (results discarded): each replica builds its packed weight forms and persistent buffers on first use --
        setup work, like building the model, that must not land in a measured or latency-critical batch"""
        for _ in self.models:
            self.submit(points, batch_size, metadata)
        self.flush()
        torch.cuda.synchronize()
        self._n = 0

    def flush(self):
        prev, self._pending = self._pending, None
        return self._finish(prev)

    @staticmethod
    def _wait(ev):
        """hipEventSynchronize spins (with the blocking flag it yields in a loop: still a busy core, measured with time.thread_time); the
        batch waited for is the PREVIOUS one while the newest is already queued, so nothing is lost by sleeping in 0.1 ms steps instead"""
        import time
        while not ev.query():
            time.sleep(1e-4)

    @classmethod
    def _finish(cls, p):
        if p is None:
            return None
        ob, os_, ol, counts_host, ev, batch_size, static = p
        cls._wait(ev)
        cur = torch.cuda.current_stream()
        if static:
            # graph mode: copies, queued on the caller's stream now (the data are complete: the event was waited for) -- the replica's next
            # replay waits for the caller's stream at its submit(), i.e. for these copies
            ob, os_, ol = ob.clone(), os_.clone(), ol.clone()
        else:
            for t in (ob, os_, ol):
                t.record_stream(cur)                             # allocated on the replica's stream, consumed on the caller's
        counts = counts_host.numpy().copy()
        return [dict(pred_boxes=ob[b, :int(counts[b])], pred_scores=os_[b, :int(counts[b])], pred_labels=ol[b, :int(counts[b])])
                for b in range(batch_size)]
