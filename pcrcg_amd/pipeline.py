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
nothing else crosses streams.  Requests are served strictly in order."""
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
        if threaded:
            self._worker = threading.Thread(target=self._serve, name="pcrcg-front-end", daemon=True)
            self._worker.start()

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
            try:
                self._ready.put(self.prepare(*item))
            except BaseException as e:      # surfaced by next_prepared()
                self._ready.put(e)

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
    def run(self, prepared):
        """Enqueue the forward of a prepared pair on the next model stream; returns the result dict."""
        batch, done = prepared
        stream = self.models[self._turn % len(self.models)]
        self._turn += 1
        for t in _tensors(batch):          # allocated on the front-end stream, consumed on `stream`
            if t.is_cuda:
                t.record_stream(stream)
        stream.wait_event(done)
        with torch.cuda.stream(stream), torch.no_grad():
            out = self.net(batch)
        return out

    def drain(self):
        """Wait for every outstanding request, dropping the batches, then for all streams."""
        while self._outstanding > 0:
            self.next_prepared()
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
