"""Two-stream software pipeline over independent fragment pairs.

The front end of a pair (grid subsampling + radius searches) is made of many small, latency-bound
kernels and needs three host round trips (each subsampled level's row count sizes the next level's
tensors); the model forward is a few long kernels enqueued by one call.  Running the front end of
pair i+1 on its own HIP stream while the forward of pair i runs on another keeps the GPU busy during
those round trips.  Pairs are independent, so nothing but the hand-off of a finished
batch dict crosses the streams (one event)."""
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
    def __init__(self, net, config, neighborhood_limits, device=None):
        self.net, self.config, self.limits = net, config, neighborhood_limits
        self.device = torch.device(device if device is not None else "cuda")
        self.front = torch.cuda.Stream(device=self.device)     # pyramid builder
        self.model = torch.cuda.Stream(device=self.device)     # KPFCNN + GCN forward

    def prepare(self, points, lengths):
        """Enqueue (and, for the row counts, wait for) the pyramid of one pair on the front-end stream."""
        with torch.cuda.stream(self.front):
            batch = build_pyramid(points, lengths, self.config, self.limits)
            done = torch.cuda.Event()
            done.record(self.front)
        for t in _tensors(batch):          # produced on `front`, consumed on `model`
            if t.is_cuda:
                t.record_stream(self.model)
        return batch, done

    def run(self, prepared):
        """Enqueue the forward of a prepared pair on the model stream; returns the result dict."""
        batch, done = prepared
        self.model.wait_event(done)
        with torch.cuda.stream(self.model), torch.no_grad():
            out = self.net(batch)
        return out

    def synchronize(self):
        self.front.synchronize()
        self.model.synchronize()
