"""Multi-stream software pipeline over independent fragment pairs.

The front end of a pair (grid subsampling + radius searches) is made of many small, latency-bound
kernels and needs host round trips (each subsampled level's row count sizes the next level's tensors);
the model forward is a few hundred kernels enqueued by one call.  Three things overlap here:

  * a front-end worker thread builds pyramids on its own HIP stream (its host round trips wait with the
    GIL released, so they do not stall the threads that enqueue forwards).  A pyramid needs four host round
    trips (three subsampled row counts, one for the table widths), during which its stream would sit idle -- a
    third of the time: the worker therefore INTERLEAVES the pyramids of `interleave` consecutive pairs on that one
    stream (pyramid.pyramid_steps is a generator that yields at each round trip), so the kernels of pair k+1 run
    while the host waits for pair k.  More front-end STREAMS (front_streams > 1) were measured and lose: every
    stream beyond front end + three model streams costs more than it overlaps;
  * forwards of consecutive pairs alternate between two model streams, so the coarse levels of one pair
    (a few hundred points, few workgroups) overlap with the fine levels of the next;
  * one event per pair hands the finished batch dict from the front-end stream to a model stream.

Pairs are independent (one pair per batch, per-pair InstanceNorm statistics and GNN), so
nothing else crosses streams.  Requests are served strictly in order.

Two ways to drive it:
  * request() / next_prepared() / run(): the caller's thread enqueues every forward (one C call that
    launches ~400 kernels, ~2.5 ms of host time for an S30k pair);
  * submit() / result(): one forward-worker thread per model stream makes that call (ctypes releases the
    GIL), so the host-side launch cost of consecutive pairs overlaps as well.  Pair k is handled by
    worker k mod W and results come back in submission order.

The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with the front-end
stream, three model streams and the default stream that is one too few and two streams serialise -- set
GPU_MAX_HW_QUEUES=8 in the environment before the first HIP call (bench.py does)."""
import collections
import queue
import threading
import time

import torch

from .pyramid import check_tie_status, pyramid_steps


def _tensors(batch):
    for v in batch.values():
        if isinstance(v, torch.Tensor):
            yield v
        elif isinstance(v, (list, tuple)):
            for t in v:
                if isinstance(t, torch.Tensor):
                    yield t


class _Mailbox:
    """Items keyed by sequence number; get(k) blocks until item k has been put (front-end workers finish out of
    order, a forward worker consumes its pairs in order)."""

    def __init__(self):
        self._items, self._cv = {}, threading.Condition()

    def put(self, key, item):
        with self._cv:
            self._items[key] = item
            self._cv.notify_all()

    def get(self, key):
        with self._cv:
            while key not in self._items and None not in self._items:
                self._cv.wait()
            return self._items.pop(key) if key in self._items else None


class PairPipeline:
    def __init__(self, net, config, neighborhood_limits, device=None, model_streams=2, threaded=True, front_streams=1,
                 interleave=2):
        self.net, self.config, self.limits = net, config, neighborhood_limits
        self.device = torch.device(device if device is not None else "cuda")
        self.fronts = [torch.cuda.Stream(device=self.device) for _ in range(max(1, front_streams))]
        self.front = self.fronts[0]
        self._interleave = max(1, int(interleave))
        self.models = [torch.cuda.Stream(device=self.device) for _ in range(max(1, model_streams))]
        self._turn = 0
        self._requests = [queue.Queue() for _ in self.fronts]
        self._ready = queue.Queue()
        self._outstanding = 0
        self._worker = None
        self._front_workers = []
        self._threaded = threaded
        # submit()/result() mode: per-forward-worker mailbox of prepared pairs and queue of finished forwards
        self._fwd_in = [_Mailbox() for _ in self.models]
        self._fwd_out = [queue.Queue() for _ in self.models]
        self._fwd_workers = []
        self._submitted = 0
        self._returned = 0
        # tie-order status words of pairs in flight: (pinned host copy, event) -- read once the event has passed
        self._tie_pending = []
        self._tie_lock = threading.Lock()
        # build the runner's weight descriptor once, here, before any worker thread can race for it
        if getattr(net, "use_runner", False) and next(net.parameters()).is_cuda:
            with torch.cuda.device(self.device):
                net.runner().descriptor()
        if threaded:
            for f in range(len(self.fronts)):
                t = threading.Thread(target=self._serve, args=(f,), name=f"pcrcg-front-end-{f}", daemon=True)
                t.start()
                self._front_workers.append(t)
            self._worker = self._front_workers[0]
            for w in range(len(self.models)):
                t = threading.Thread(target=self._serve_forward, args=(w,), name=f"pcrcg-forward-{w}", daemon=True)
                t.start()
                self._fwd_workers.append(t)

    # ---- front end -----------------------------------------------------------------------------
    def _steps(self, points, lengths):
        return pyramid_steps(points, lengths, self.config, self.limits, defer_tie_check=True, defer_restore=True)

    def _finish(self, batch, f):
        """Called under the front-end stream once a pyramid generator has returned its batch."""
        done = torch.cuda.Event()
        done.record(self.fronts[f])
        self._check_tie_status(wait=False)
        return batch, done

    def prepare(self, points, lengths, f=0):
        """Build the pyramid of one pair on front-end stream f (blocking variant)."""
        with torch.cuda.stream(self.fronts[f]):
            steps = self._steps(points, lengths)
            try:
                while True:
                    next(steps).synchronize()
            except StopIteration as fin:
                return self._finish(fin.value, f)

    def _serve(self, f):
        """Front-end worker: up to `interleave` pyramids in flight on stream f, advanced round-robin -- the oldest one
        is resumed as soon as the value it waits for has arrived, the others' kernels keep the stream busy."""
        torch.cuda.set_device(self.device)
        active = collections.deque()          # [generator, event it waits for, request]
        closing = False
        stats = self.front_stats = {"wait_s": 0.0, "advance_s": 0.0, "idle_s": 0.0, "pairs": 0, "resumed_ready": 0, "resumes": 0}

        def deliver(item, result):
            if len(item) == 4:                # submit(): hand the pair to its forward worker (in-order mailbox)
                self._fwd_in[item[2]].put(item[3], result)
            else:
                self._ready.put(result)

        def advance(entry):
            """Resume a generator; True if it is still running."""
            try:
                with torch.cuda.stream(self.fronts[f]):
                    try:
                        entry[1] = next(entry[0])
                        return True
                    except StopIteration as fin:
                        deliver(entry[2], self._finish(fin.value, f))
            except BaseException as e:        # surfaced by result() / next_prepared()
                deliver(entry[2], e)
            return False

        while True:
            # request()/next_prepared() deliver through one queue in completion order: keep those strictly serial
            while not closing and len(active) < self._interleave and all(len(e[2]) == 4 for e in active):
                t0 = time.perf_counter()
                try:
                    item = self._requests[f].get(block=not active)
                except queue.Empty:
                    break
                finally:
                    stats["idle_s"] += time.perf_counter() - t0
                if item is None:
                    closing = True
                    break
                entry = [self._steps(item[0], item[1]), None, item]
                stats["pairs"] += 1
                t0 = time.perf_counter()
                running = advance(entry)
                stats["advance_s"] += time.perf_counter() - t0
                if running:
                    active.append(entry)
                if len(item) != 4:
                    break
            if not active:
                if closing:
                    return
                continue
            entry = active.popleft()
            t0 = time.perf_counter()
            stats["resumes"] += 1
            stats["resumed_ready"] += 1 if entry[1].query() else 0
            entry[1].synchronize()
            t1 = time.perf_counter()
            running = advance(entry)
            stats["wait_s"] += t1 - t0
            stats["advance_s"] += time.perf_counter() - t1
            if running:
                if len(entry[2]) == 4:
                    active.append(entry)
                else:
                    active.appendleft(entry)      # serial mode: finish this pair before admitting the next

    def _check_tie_status(self, wait):
        """Raise if restoring the reference's tie order failed for an earlier pair (pyramid.check_tie_status)."""
        with self._tie_lock:
            pending, self._tie_pending = self._tie_pending, []
        bad, keep = 0, []
        for host, ev in pending:
            if wait:
                ev.synchronize()
            if ev.query():
                bad = bad or int(host[0])
            else:
                keep.append((host, ev))
        with self._tie_lock:
            self._tie_pending = keep + self._tie_pending
        check_tie_status(bad)

    def _serve_forward(self, w):
        torch.cuda.set_device(self.device)
        seq = w
        while True:
            item = self._fwd_in[w].get(seq)
            seq += len(self.models)
            if item is None:
                return
            try:
                if isinstance(item, BaseException):
                    raise item
                self._fwd_out[w].put(self._forward(item, self.models[w]))
            except BaseException as e:
                self._fwd_out[w].put(e)

    def _order_inputs(self, points, lengths, f):
        """The front-end stream reads the inputs: make it wait for whatever the caller's stream still has in flight on
        them (an asynchronous upload, a preprocessing kernel) and keep their memory from being reused under it."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.fronts[f].wait_event(ev)             # enqueued before the worker can enqueue this pair's first kernel
        points.record_stream(self.fronts[f])
        lengths.record_stream(self.fronts[f])

    def request(self, points, lengths):
        """Ask the front-end worker for the pyramid of one more pair (served in order)."""
        self._order_inputs(points, lengths, 0)
        self._outstanding += 1
        if self._worker is None:
            self._ready.put(self.prepare(points, lengths))
        else:
            self._requests[0].put((points, lengths))      # served in order: one front-end worker

    def next_prepared(self):
        item = self._ready.get()
        self._outstanding -= 1
        if isinstance(item, BaseException):
            raise item
        return item

    # ---- model ---------------------------------------------------------------------------------
    def _forward(self, prepared, stream):
        batch, done = prepared
        restore, touched = batch.pop("restore", (None, ()))
        for t in list(_tensors(batch)) + list(touched):    # allocated on the front-end stream, consumed on `stream`
            if t.is_cuda:
                t.record_stream(stream)
        stream.wait_event(done)
        with torch.cuda.stream(stream), torch.no_grad():
            if restore is not None:
                # the reference's order inside tie groups (csrc/tieorder.hip), here rather than on the front-end
                # stream: that stream is the pipeline's bottleneck, the model streams wait for pyramids half of the time
                status = restore()
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(status, non_blocking=True)    # asynchronous: the word is 0 on sane clouds
                ev = torch.cuda.Event()
                ev.record(stream)
                with self._tie_lock:
                    self._tie_pending.append((host, ev))
            out = self.net(batch)
        return out

    def run(self, prepared):
        """Enqueue the forward of a prepared pair on the next model stream; returns the result dict."""
        stream = self.models[self._turn % len(self.models)]
        self._turn += 1
        return self._forward(prepared, stream)

    # ---- submit / result: front end and forward both off the caller's thread -------------------
    def submit(self, points, lengths):
        """Queue one pair for pyramid build + forward; results are returned by result() in this order."""
        seq = self._submitted
        w = seq % len(self.models)
        self._submitted += 1
        self._order_inputs(points, lengths, seq % len(self.fronts) if self._threaded else 0)
        if self._threaded:
            self._requests[seq % len(self.fronts)].put((points, lengths, w, seq))
        else:
            self._fwd_out[w].put(self._forward(self.prepare(points, lengths), self.models[w]))

    def result(self):
        """Result dict of the oldest submitted pair (its kernels are enqueued on a model stream, not
        necessarily finished: synchronize() or use the tensors on a stream that waits for it)."""
        if self._returned >= self._submitted:
            raise RuntimeError("PairPipeline.result(): nothing submitted")
        w = self._returned % len(self.models)
        self._returned += 1
        out = self._fwd_out[w].get()
        if isinstance(out, BaseException):
            raise out
        return out

    def drain(self):
        """Wait for every outstanding request / submission, dropping the results, then for all streams."""
        while self._outstanding > 0:
            self.next_prepared()
        while self._returned < self._submitted:
            self.result()
        self.synchronize()

    def synchronize(self):
        for s in self.fronts:
            s.synchronize()
        for s in self.models:
            s.synchronize()
        self._check_tie_status(wait=True)

    def close(self):
        for f, t in enumerate(self._front_workers):
            self._requests[f].put(None)
            t.join(timeout=10)
        self._front_workers, self._worker = [], None
        for w, t in enumerate(self._fwd_workers):
            self._fwd_in[w].put(None, None)
            t.join(timeout=10)
        self._fwd_workers = []
