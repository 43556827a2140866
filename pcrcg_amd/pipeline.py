"""Multi-stream software pipeline over independent fragment pairs.

The front end of a pair (grid subsampling + radius searches) is made of many small, latency-bound
kernels and needs host round trips (each subsampled level's row count sizes the next level's tensors);
the model forward is a few hundred kernels enqueued by one call.  Three things overlap here:

  * a front-end worker thread builds pyramids on its own HIP stream (its host round trips wait with the
    GIL released, so they do not stall the thread that enqueues forwards);
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
import queue
import threading

import torch

from .pyramid import build_pyramid


def _tensors(batch):
    for v in batch.values():
        if isinstance(v, torch.Tensor):
            yield v
        elif isinstance(v, (list, tuple)):
            for t in v:
                if isinstance(t, torch.Tensor):
                    yield t


class PairPipeline:
    def __init__(self, net, config, neighborhood_limits, device=None, model_streams=2, threaded=True):
        self.net, self.config, self.limits = net, config, neighborhood_limits
        self.device = torch.device(device if device is not None else "cuda")
        self.front = torch.cuda.Stream(device=self.device)
        self.models = [torch.cuda.Stream(device=self.device) for _ in range(max(1, model_streams))]
        self._turn = 0
        self._requests = queue.Queue()
        self._ready = queue.Queue()
        self._outstanding = 0
        self._worker = None
        self._threaded = threaded
        # submit()/result() mode: per-worker queues of prepared pairs and of finished forwards
        self._fwd_in = [queue.Queue() for _ in self.models]
        self._fwd_out = [queue.Queue() for _ in self.models]
        self._fwd_workers = []
        self._submitted = 0
        self._returned = 0
        if threaded:
            self._worker = threading.Thread(target=self._serve, name="pcrcg-front-end", daemon=True)
            self._worker.start()
            for w in range(len(self.models)):
                t = threading.Thread(target=self._serve_forward, args=(w,), name=f"pcrcg-forward-{w}", daemon=True)
                t.start()
                self._fwd_workers.append(t)

    # ---- front end -----------------------------------------------------------------------------
    def prepare(self, points, lengths):
        """Build the pyramid of one pair on the front-end stream (blocking variant)."""
        with torch.cuda.stream(self.front):
            batch = build_pyramid(points, lengths, self.config, self.limits)
            done = torch.cuda.Event()
            done.record(self.front)
        return batch, done

    def _serve(self):
        torch.cuda.set_device(self.device)
        while True:
            item = self._requests.get()
            if item is None:
                return
            dest = self._ready
            if len(item) == 3:              # submit(): hand the pair to its forward worker
                dest, item = self._fwd_in[item[2]], item[:2]
            try:
                dest.put(self.prepare(*item))
            except BaseException as e:      # surfaced by next_prepared() / result()
                dest.put(e)

    def _serve_forward(self, w):
        torch.cuda.set_device(self.device)
        while True:
            item = self._fwd_in[w].get()
            if item is None:
                return
            try:
                if isinstance(item, BaseException):
                    raise item
                self._fwd_out[w].put(self._forward(item, self.models[w]))
            except BaseException as e:
                self._fwd_out[w].put(e)

    def request(self, points, lengths):
        """Ask the front-end worker for the pyramid of one more pair (served in order)."""
        self._outstanding += 1
        if self._worker is None:
            self._ready.put(self.prepare(points, lengths))
        else:
            self._requests.put((points, lengths))

    def next_prepared(self):
        item = self._ready.get()
        self._outstanding -= 1
        if isinstance(item, BaseException):
            raise item
        return item

    # ---- model ---------------------------------------------------------------------------------
    def _forward(self, prepared, stream):
        batch, done = prepared
        for t in _tensors(batch):          # allocated on the front-end stream, consumed on `stream`
            if t.is_cuda:
                t.record_stream(stream)
        stream.wait_event(done)
        with torch.cuda.stream(stream), torch.no_grad():
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
        w = self._submitted % len(self.models)
        self._submitted += 1
        if self._threaded:
            self._requests.put((points, lengths, w))
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
        self.front.synchronize()
        for s in self.models:
            s.synchronize()

    def close(self):
        if self._worker is not None:
            self._requests.put(None)
            self._worker.join(timeout=10)
            self._worker = None
        for w, t in enumerate(self._fwd_workers):
            self._fwd_in[w].put(None)
            t.join(timeout=10)
        self._fwd_workers = []
