"""Pair engine: independent fragment pairs through two stages of host threads, each stage a call into
libpcrcg_hip.so that releases the GIL.

  front stage   F threads share ONE front-end HIP stream (+ its side streams).  A thread builds the pyramid of its pairs with
                pcrcg_pyramid_build (the whole front end in one call; levels are sized from a row bound, the call waits
                ONCE, at the end of the chain, for the row and column counts).  One stream on purpose: the front-end
                kernels are latency-bound and partly persistent (the KD-forest's task queue); several pyramids side
                by side, or pyramids on the model streams, slow everything down (measured: every pair on its own
                stream, 4 streams: 217 pairs/s; this topology: see DESIGN.md).  F = 1 by default.  F > 1 on the one stream =
                OVERLAPPED builds: the library keeps the chains whole (a per-stream enqueue lock, released before a call
                waits for its round trip), so one thread's host turn overlaps the next chain and the stream never
                stands idle between chains -- which buys nothing (585 against 591 pairs/s): the rate is the chip's, not
                the front thread's (profiles/r06_ab_overlapped_builds.txt).
  model stage   M threads, one HIP stream each, enqueue the forwards (pcrcg_kpfcnn_forward) of pairs k, k+M, ...;
                the coarse levels of one pair overlap the fine levels of the next.  M = 3: a fourth stream with
                forwards is slower again on this GPU.

One event per pair hands the finished tables from the front-end stream to a model stream; the arena holding them
goes back to its front thread's ring once the forward has passed.  Pairs are independent, so nothing
else crosses streams.  This replaced round 1's generator-interleaving pipeline, whose front-end
worker spent 1.7 ms of interpreter time per pair.

    eng = PairStreams(net, config, limits, device)
    eng.submit(points, lengths); ...; out = eng.result()      # results come back in submission order

Streams and hardware (round 6): gfx950's command processor has FOUR compute dispatchers; a stream's hardware queue belongs
to one, and a dispatcher hands out the workgroups of one kernel at a time, so two busy streams on one dispatcher take turns
kernel by kernel (profiles/r06_queue_pipes.txt).  The engine therefore CHOOSES its streams: it classifies a dozen
candidates by measurement (pcrcg_stream_pipe_classes, ~50 ms at construction) and takes the front-end streams from one
class and the three model streams from the other three (_pick_streams; `pipe_classes` records the choice, a
RuntimeWarning says when it could not be made).  That holds with the runtime's default of four hardware queues and with
GPU_MAX_HW_QUEUES=8 alike; rounds 1-5 took streams in creation order and needed the variable set before the first HIP
call.  It also lifted the "one engine per process" rule of those rounds (a fresh engine 453 pairs/s, the third one created
in the same process 355): four engines created one after the other in one process run at 586-592 pairs/s each
(scripts/engines_in_one_process.py).  Two engines ACTIVE at once still share the four dispatchers."""
import os
import queue
import threading
import time

import torch

from .pyramid import NativePyramid, check_tie_status


class _Mailbox:
    """Items keyed by sequence number; get(k) blocks until item k has been put."""

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


class PairStreams:
    ARENAS = 4        # per front thread: pairs whose tables may be alive at once (built, forward not yet passed)

    def __init__(self, net, config, neighborhood_limits, device=None, model_streams=3, front_threads=1, tie_order=None,
                 pairs_per_build=4, up_nearest=False, front_streams=1, front_priority=0, pairs_per_forward=4,
                 adaptive_jobs=False, forest_stream=None, pipes=None):
        """pairs_per_forward = 2 .. 4: pairs that were built together also go through the network together, up to that
        many per pcrcg_kpfcnn_forward_group call on one model stream, in which every product with a weight matrix runs
        once for all of them (the pairs never mix; outputs equal separate forwards up to summation order).  1: one call
        per pair.  pairs_per_build (1 .. 4): pairs one front-end kernel chain carries.  The defaults are four and four
        since the end of round 5 (two and two before): +2.6 % on 30 000-point pairs, +5 % on the voxelised-scan-like ones
        (profiles/r05_ab_group_size.txt); 2 x 120 000-point pairs run 1 % better with three and three.  Keep at least
        three builds' worth of pairs submitted ahead (bench.py: 24).
        up_nearest: the engine's internal pyramids carry ONE-column upsample tables (the nearest coarse point: the only
        column KPFCNN.forward reads, ref:models/blocks.py:77-87) instead of the batch contract's [N, limit] tables:
        ~1 % less front-end work, same outputs.  Off by default: the tables are then exactly what build_pyramid()
        hands to any other consumer.
        adaptive_jobs (default OFF since round 5): when on, whether two pairs of one build share a forward call is
        decided from the state of the queues and of the model streams at that moment, i.e. from host and GPU TIMING --
        grouping changes the summation order of the weight products, so the low-order bits of a pair's outputs then differ
        from run to run (never beyond the 1e-5 the grouped and the single forward differ by).  It bought 1.5 % on 20-step
        regions in round 4 and buys nothing any more (profiles/r05_ab_adaptive_jobs_20_step.txt: 510-515 pairs/s either way,
        as does building an empty engine's first pair alone), so the default is the fixed grouping: which pairs share a
        call depends on the configuration only, and together with PCRCG_DEBUG=deterministic=1 (atomics-free sums) the
        outputs are a function of the inputs alone."""
        self.net, self.config, self.limits = net, config, neighborhood_limits
        self.device = torch.device(device if device is not None else "cuda")
        if not getattr(net, "use_runner", False):
            raise RuntimeError("pcrcg_amd.PairStreams needs the C++ runner path (use_batch_norm=True)")
        self.runner = net.runner()
        with torch.cuda.device(self.device):
            self.runner.descriptor()           # built once, here, before any worker thread can race for it
        # front_streams / front_priority (< 0: the front-end chain ahead of the forwards) exist for measurements: more
        # than one front-end stream, and a prioritised one, both measured slower
        # side streams of the front end (round 6: the chain is a DAG -- the subsamplings and the KD-forests of the tie-order
        # restore step need nothing from the searches): forest_stream (PCRCG_FOREST_STREAM) = 0 (default) everything in line on
        # the front-end stream, 1 subsamplings + forests on one side stream, 2 on one each.  Alone on the GPU a chain spread
        # over streams of OTHER dispatchers takes half the time (profiles/r06_chain_latency_alone.txt); inside the engine the
        # other dispatchers belong to the forwards, side streams of the front end's own dispatcher add no concurrency, and the
        # chain in line measures 1.5-2 % better at 20-step regions, equal at 480 (profiles/r06_ab_fill_streams.txt)
        if forest_stream is None:
            forest_stream = int(os.environ.get("PCRCG_FOREST_STREAM", "0"))
        self._pick_streams(max(1, int(front_streams)), max(1, int(model_streams)), int(forest_stream), int(front_priority),
                           os.environ.get("PCRCG_ENGINE_PIPES", "auto") if pipes is None else pipes)
        self.front = self.fronts[0]
        nf = max(1, int(front_threads))
        # overlapped builds: the front threads share one stream; pcrcg_pyramid_build keeps the chains whole (the stream's
        # enqueue lock) and one thread's round trip + host work overlaps the next chain
        self._overlap = nf > 1 and len(self.fronts) == 1 and not self.sides
        # every front thread owns a ring of builders (arena + pinned scratch each)
        self.up_nearest = bool(up_nearest)
        self._pyr = [[NativePyramid(config, neighborhood_limits, tie_order, up_nearest=self.up_nearest)
                      for _ in range(self.ARENAS)]
                     for _ in range(nf)]
        # per arena: a one-slot queue holding the event after which it may be overwritten (None: never used); the
        # front thread TAKES it before building into the arena, the model thread puts the forward's event back
        if self.sides:
            for f, ring in enumerate(self._pyr):
                for pyr in ring:
                    pyr.set_side_streams(*self.sides[f % len(self.sides)])
        self._free = [[queue.Queue() for _ in range(self.ARENAS)] for _ in range(nf)]
        for ring in self._free:
            for q in ring:
                q.put(None)
        self._in = queue.Queue()               # one queue: a front thread takes up to `pairs_per_build` consecutive pairs
        self._take = threading.Lock()
        self._per_build = min(8, max(1, int(pairs_per_build)))     # (pcrcg_pyramid_build: at most 16 clouds)
        self._users = [[1] * self.ARENAS for _ in range(nf)]   # forwards that read the arena's current contents
        self._per_forward = min(4, max(1, int(pairs_per_forward)))
        self._adaptive = bool(adaptive_jobs)
        self._last_done = [None] * len(self.models)       # per model stream: the event behind its newest forward
        self._mid = [_Mailbox() for _ in self.models]     # jobs (one or two pairs) by job index: thread m serves m, m + M, ...
        self._jobs = 0                                     # job indices are handed out under self._take, with the pairs
        self._queued = [0] * len(self.models)              # jobs handed out to a model thread whose forward is not enqueued yet
        self._qlock = threading.Lock()                     # (its own lock: a front thread WAITS for input holding self._take)
        self._results = _Mailbox()                         # (outputs, done event) or an exception, by submission index
        self._submitted = self._returned = 0
        self._pending = []                     # (status tensor, slot, event) of pairs whose tie status is unread
        self._lock = threading.Lock()
        self.stats = {"pairs": 0, "builds": 0, "front_idle_s": 0.0, "arena_wait_s": 0.0, "build_s": 0.0, "model_idle_s": 0.0,
                      "launch_s": 0.0}      # host seconds per stage, summed over the threads of the stage
        self._stats_lock = threading.Lock()    # the stage threads all add to `stats`
        self._threads = []
        for f in range(nf):
            t = threading.Thread(target=self._serve_front, args=(f,), name=f"pcrcg-front-{f}", daemon=True)
            t.start()
            self._threads.append(t)
        for m in range(len(self.models)):
            t = threading.Thread(target=self._serve_model, args=(m,), name=f"pcrcg-model-{m}", daemon=True)
            t.start()
            self._threads.append(t)

    def _pick_streams(self, n_front, n_model, want_side, front_priority, pipes):
        """The engine's streams by hardware DISPATCHER (round 6).  gfx950's command processor has four compute dispatchers;
        a stream's hardware queue belongs to one, and a dispatcher hands out the workgroups of one kernel at a time -- two
        busy streams on one dispatcher take turns kernel by kernel (profiles/r06_queue_pipes.txt; this is the wall rounds
        3-5 ran into with a fourth model stream: it shared the front-end stream's dispatcher, and the front-end chain stood
        behind every GEMM's dispatch).  pipes="auto": a dozen candidate streams are classified by measurement
        (ops.stream_pipe_classes, ~50 ms once) and the engine takes the front-end stream and its KD-forest stream from
        ONE class -- both run small latency-bound kernels whose dispatch is over at once -- and the model streams from the
        other classes, in turn.  pipes="off" (or PCRCG_ENGINE_PIPES=off): streams in creation order, as rounds 1-5 did.
        self.pipe_classes records what was chosen."""
        dev = self.device
        if pipes == "off":
            self.fronts = [torch.cuda.Stream(device=dev, priority=front_priority) for _ in range(n_front)]
            self.sides = [(torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev) if want_side > 1 else None)
                          for _ in self.fronts] if want_side else []
            self.models = [torch.cuda.Stream(device=dev) for _ in range(n_model)]
            self.pipe_classes = None
            return
        from . import ops
        cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
        fcands = cands if front_priority == 0 else [torch.cuda.Stream(device=dev, priority=front_priority) for _ in range(4)]
        try:
            try:
                cls = ops.stream_pipe_classes(cands + (fcands if fcands is not cands else []))
            except RuntimeError:        # other work on the GPU while the probe ran (an engine just closed, another thread's kernels):
                torch.cuda.synchronize(dev)         # its timing means nothing then -- once more on a drained device
                time.sleep(0.05)
                cls = ops.stream_pipe_classes(cands + (fcands if fcands is not cands else []))
        except RuntimeError as e:       # the probe is a measurement: if it cannot be made, run as rounds 1-5 did and say so
            import warnings
            warnings.warn("pcrcg_amd.PairStreams: could not classify streams by dispatcher (%s); taking them in creation order" % e,
                          RuntimeWarning)
            return self._pick_streams(n_front, n_model, want_side, front_priority, "off")
        ccls, fcls = cls[:len(cands)], (cls[len(cands):] if fcands is not cands else cls[:len(cands)])
        by = {}
        for s_, c in zip(cands, ccls):
            by.setdefault(c, []).append(s_)
        # the front end's class: the one that offers most streams (it needs 1 + want_side per front-end stream; the probe's
        # classes are not evenly filled: torch hands out pool streams, and the runtime maps them to hardware queues as it likes)
        if fcands is cands:
            front_class = max(sorted(by), key=lambda c: len(by[c]))
        else:
            front_class = fcls[0]
        self.fronts, used = [], set()
        for s_, c in zip(fcands, fcls):                       # front-end streams: all from that class
            if c == front_class and len(self.fronts) < n_front:
                self.fronts.append(s_)
                used.add(id(s_))
        while len(self.fronts) < n_front:
            self.fronts.append(self.fronts[-1])
        free_front = [s_ for s_ in by.get(front_class, []) if id(s_) not in used]
        self.sides = []
        if want_side:
            # side streams come from the front end's class or not at all: a stream of unknown class may sit on a model
            # stream's dispatcher (seen: the fourth engine of a process got two candidates of its class, took a third stream
            # blindly for the KD-forests, and ran 13 % slower).  Too few: the forests share the subsamplings' stream, or
            # everything stays in line.
            spare = list(free_front)
            for i in range(len(self.fronts)):
                sub = spare.pop(0) if spare else None
                forest = spare.pop(0) if (spare and want_side > 1) else None
                self.sides.append((sub, forest) if sub is not None else None)
            if any(p_ is None for p_ in self.sides):
                self.sides = []
        others = [c for c in sorted(by) if c != front_class] or [front_class]
        if pipes == "front4":                                 # measurement aid: model streams on ALL classes, the front end's too
            others = sorted(by)
        take = {c: 0 for c in by}
        self.models = []
        for m in range(n_model):
            c = others[m % len(others)]
            taken = [x for pair in self.sides for x in pair if x is not None]
            pool = [s_ for s_ in by[c] if id(s_) not in used and all(s_ is not x for x in taken)]
            self.models.append(pool[take[c] % len(pool)] if pool else torch.cuda.Stream(device=dev))
            take[c] += 1
        mcls = [others[m % len(others)] for m in range(n_model)]
        # the engine's premise, verified: the front end and every model stream on a dispatcher of its own (with up to three
        # model streams).  It holds with the runtime's default of four hardware queues as well as with GPU_MAX_HW_QUEUES=8
        # (tests/test_pairstream_gpu.py runs both in fresh processes) -- the streams are CHOSEN by class, not taken in creation
        # order, which is what made rounds 1-5 depend on that variable (448 against 536 pairs/s).  If the probe finds fewer
        # classes than the engine needs streams, two of them will take turns: say so instead of running slowly in silence.
        distinct = len(set(mcls)) == min(n_model, 3) and front_class not in mcls
        if not distinct:
            import warnings
            warnings.warn("pcrcg_amd.PairStreams: the probe found dispatcher classes %s among its candidate streams; the front-end "
                          "stream (class %d) and the model streams (classes %s) share a hardware dispatcher and will take turns "
                          "kernel by kernel (pcrcg_stream_pipe_classes, profiles/r06_queue_pipes.txt)"
                          % (sorted(by), front_class, mcls), RuntimeWarning)
        self.pipe_classes = {"candidates": ccls, "front": front_class, "model": mcls, "distinct": bool(distinct),
                             "side_streams": sum(1 for x in (self.sides[0] if self.sides else ()) if x is not None),
                             "side_class": front_class if self.sides else None,
                             "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "(unset: the runtime's default, 4)")}

    def set_up_nearest(self, on):
        """Switch the engine's internal upsample tables between the batch contract's [N, limit] form (off) and the
        one-column form (see __init__); the engine must be drained."""
        if self._returned < self._submitted:
            raise RuntimeError("PairStreams.set_up_nearest(): pairs in flight -- drain() first")
        self.synchronize()
        self.up_nearest = bool(on)
        for ring in self._pyr:
            for pyr in ring:
                pyr.cfg.up_nearest = int(self.up_nearest)

    @staticmethod
    def job_sizes(n_pairs, per_forward, one_each):
        """Pairs per forward job for a build of `n_pairs` consecutive pairs: up to `per_forward` each, or one each."""
        per = 1 if one_each else max(1, per_forward)
        return [min(per, n_pairs - i) for i in range(0, n_pairs, per)]

    def _idle_models(self):
        """Model streams with nothing queued or running: no job waiting in the stream's mailbox and its newest forward passed."""
        with self._qlock:
            waiting = list(self._queued)
        n = 0
        for m in range(len(self.models)):
            ev = self._last_done[m]
            if waiting[m] == 0 and (ev is None or ev.query()):
                n += 1
        return n

    def _stat(self, **add):
        with self._stats_lock:
            for k, v in add.items():
                self.stats[k] += v

    def reset_stats(self):
        with self._stats_lock:
            for k in self.stats:
                self.stats[k] = 0

    def stats_snapshot(self):
        with self._stats_lock:
            return dict(self.stats)

    # ---- workers -------------------------------------------------------------------------------
    def _serve_front(self, f):
        torch.cuda.set_device(self.device)
        turn = 0
        front = self.fronts[f % len(self.fronts)]
        while True:
            t0 = time.perf_counter()
            with self._take:                   # consecutive pairs go to one thread
                items = [self._in.get()]
                while items[-1] is not None and len(items) < self._per_build:
                    try:
                        items.append(self._in.get_nowait())
                    except queue.Empty:
                        break
                if items[-1] is None:
                    self._in.put(None)         # pass the shutdown token on to the other front threads
                    items.pop()
                # the pairs' forward jobs, in submission order: consecutive pairs of this build, up to per_forward each; with
                # adaptive_jobs (off by default) one pair each while model streams stand idle or nothing else is queued
                sizes = self.job_sizes(len(items), self._per_forward, self._adaptive and len(items) > 1 and
                                       (self._in.empty() or self._idle_models() >= 2))
                job0 = self._jobs
                self._jobs += len(sizes)
                with self._qlock:
                    for j in range(len(sizes)):
                        self._queued[(job0 + j) % len(self.models)] += 1
            if not items:
                return
            self._stat(front_idle_s=time.perf_counter() - t0)
            a = turn % self.ARENAS
            turn += 1
            k = len(items)
            # hand-back accounting of the arena: `owed` tokens (one per forward that reads its current contents) are
            # in its queue or still to come; `taken` of them have been consumed by this build so far
            owed, taken, claimed = self._users[f][a], 0, False
            posted = 0                                            # jobs of this build already handed to a model thread
            try:
                with torch.cuda.stream(front), torch.no_grad():
                    for _, points, lengths, ready, _images in items:
                        front.wait_event(ready)                   # inputs may still be in flight on the caller's stream
                        points.record_stream(front)
                        lengths.record_stream(front)
                    t0 = time.perf_counter()
                    while taken < owed:                           # blocks until those forwards have been enqueued ...
                        freed = self._free[f][a].get()
                        taken += 1
                        if freed is not None:
                            front.wait_event(freed)               # ... and the stream waits until they have passed
                    self._users[f][a] = k
                    claimed = True
                    pyr = self._pyr[f][a]
                    t1 = time.perf_counter()
                    built = None
                    if k == 1:
                        # one pair: its tie-order restore step goes to the pair's model stream
                        b, arena, lens_h, slot, deferred = pyr.build(items[0][1], items[0][2], defer_restore=True)
                        batches = [b]
                    elif self._overlap and len(sizes) == 1:
                        # overlapped builds (two front threads, one stream): when build() returns, its round trip has seen
                        # the chain through -- the tables are complete, nothing of this build is left in the front-end
                        # stream, and what IS in that stream by now is the other thread's next chain: no event from it.
                        # The restore step goes to the model stream of the build's one forward job.
                        batches, arena, lens_h, slot, deferred = pyr.build([it[1] for it in items], [it[2] for it in items],
                                                                           group=2, defer_restore=True)
                        built = False
                    else:
                        # several pairs stacked into ONE kernel chain (the chain is latency-bound: k pairs cost little
                        # more than one); the restore step covers all of them and runs here
                        # (the pairs are handed over as PARTS: the builder copies them into its arena itself, no torch.cat)
                        batches, arena, lens_h, slot = pyr.build([it[1] for it in items], [it[2] for it in items], group=2)
                        deferred = None
                    self._stat(arena_wait_s=t1 - t0, build_s=time.perf_counter() - t1, pairs=k, builds=1)
                    if built is None:
                        built = torch.cuda.Event()
                        built.record(front)
                start = 0
                for j, n in enumerate(sizes):
                    seqs = [it[0] for it in items[start:start + n]]
                    imgs = [it[4] for it in items[start:start + n]]
                    self._mid[(job0 + j) % len(self.models)].put(job0 + j, (seqs, (batches, start, n), arena, built, pyr, slot,
                                                                             deferred, f, a, (lens_h, imgs)))
                    start += n
                    posted = j + 1
            except BaseException as e:                            # surfaced by result()
                if claimed:
                    # the arena was taken over for k forwards that will not happen: one token for each of them
                    for _ in items:
                        self._free[f][a].put(None)
                else:
                    # the previous contents' readers still owe `owed - taken` tokens; the next build waits for exactly
                    # those (or, if none is left, for one free token)
                    left = owed - taken
                    if left == 0:
                        self._free[f][a].put(None)
                        left = 1
                    self._users[f][a] = left
                start = 0
                for j, n in enumerate(sizes):
                    if j >= posted:                               # (a job index is posted exactly once)
                        self._mid[(job0 + j) % len(self.models)].put(job0 + j, ([it[0] for it in items[start:start + n]], e))
                    start += n

    def _serve_model(self, m):
        torch.cuda.set_device(self.device)
        from . import _lib
        _lib.lib().pcrcg_thread_shares_gpu(1)      # this thread's forwards run beside the other streams': no split-K
        stream, job = self.models[m], m
        while True:
            t0 = time.perf_counter()
            item = self._mid[m].get(job)
            job += len(self.models)
            if item is None:
                return
            self._stat(model_idle_s=time.perf_counter() - t0)
            seqs = item[0]
            if isinstance(item[1], BaseException):
                with self._qlock:
                    self._queued[m] -= 1
                for q in seqs:
                    self._results.put(q, item[1])
                continue
            _, b, arena, built, pyr, slot, deferred, f, a, (lens_h, imgs) = item
            try:
                with torch.cuda.stream(stream), torch.no_grad():
                    if built:
                        stream.wait_event(built)
                    t0 = time.perf_counter()
                    # the reference's order inside tie groups (KD-forest + reorder), here rather than on the front-end
                    # stream: that stream's serial kernel chain is the pipeline's bottleneck, the model streams have slack
                    if deferred is not None:
                        pyr.restore(deferred, slot)
                    batches, start, n = b
                    feats = self._image_inputs(batches, start, n, lens_h, imgs)
                    if n >= 2:
                        outs = self.runner.launch_group(batches, n, self.device, start)     # these pairs in ONE call
                    else:
                        outs = [self.runner.launch(batches[start], self.device)]
                    self._stat(launch_s=time.perf_counter() - t0)
                    done = torch.cuda.Event()
                    done.record(stream)
                self._last_done[m] = done
                for q, out in zip(seqs, outs):
                    self._free[f][a].put(done)                 # one token per pair that read the arena
                    out["_tie_status"] = (pyr.status, slot)
                    with self._lock:
                        self._pending.append((pyr.status, slot, done))
                    out["_keep"] = (b, arena, feats)
                    self._results.put(q, (out, done))
            except BaseException as e:
                for q in seqs:
                    self._free[f][a].put(None)
                    self._results.put(q, e)
            finally:
                with self._qlock:
                    self._queued[m] -= 1

    # ---- caller --------------------------------------------------------------------------------
    def _image_inputs(self, batches, start, n, lens_h, imgs):
        """PCR-CG's shipped configuration (image_feature: ref:configs/test/indoor.yaml:21-34, ref:models/architectures.py:
        195-514): the [N, 129] input of every pair of a forward job, injected on the job's model stream from the pair's
        2-D feature maps and projections (pcrcg_inject_image_features) in rows of KPFCNN.IMAGE_WIDTH floats, and handed to
        the runner in place of the pyramid's all-ones [N, 1] features.  -> the tensors (kept alive with the outputs)."""
        if not getattr(self.net, "image_feature", False):
            if any(im is not None for im in imgs):
                raise RuntimeError("PairStreams.submit(images=...): the network was built without image_feature")
            return None
        from . import ops
        keep = []
        for g in range(n):
            im = imgs[g]
            if im is None:
                raise RuntimeError("PairStreams.submit(): this network takes image features -- pass images=[...] with every pair")
            bt = batches[start + g]
            first = 2 * (start + g) if len(lens_h[0]) > 2 else 0       # the pair's first cloud in a grouped build
            x = ops.inject_image_features(int(bt.n_points[0]), int(lens_h[0][first]), im, channels=128, width=self.net.IMAGE_WIDTH)
            bt.features, bt.feat_dim = x.data_ptr(), int(x.shape[1])
            keep.append(x)
        return keep

    def submit(self, points, lengths, images=None):
        """Queue one pair (points [N,3] f32, lengths [2] i32 on the device) for pyramid build + forward.
        images: for a network with image_feature (PCR-CG's shipped configuration) the pair's projections as
        ops.inject_image_features takes them -- a list, in the reference's write order (KPFCNN.image_list builds it from a
        reference batch dict), of dicts with fmap [128,H,W] f32, inds2d [n,2] i64, inds3d [n] i64, target (bool) and
        optionally valid [W,H] f32, all on the device.  The 2-D backbone that produces the maps is the caller's."""
        seq = self._submitted
        self._submitted += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self._in.put((seq, points, lengths, ready, images))

    def result(self, wait=True):
        """Outputs of the oldest submitted pair, plus out["done_event"] (a torch.cuda.Event recorded behind their last
        kernel).  The kernels run on a model stream.  wait=True (the default, the safe behaviour): the caller's current
        stream waits for them, so any op or copy the caller enqueues next sees finished data.
        wait=False: nothing is made to wait -- the tensors may still be being written; the caller orders its own
        consumer behind out["done_event"] (or calls check() / synchronize()).  A throughput loop that keeps submitting
        from the same stream wants this form (bench.py): a wait queued on the caller's stream also delays the `ready`
        events that later submit() calls record on it, i.e. it ties the start of pair k+depth's pyramid to the end of
        pair k's forward (measured on the null stream: 240 instead of 340 pairs/s)."""
        if self._returned >= self._submitted:
            raise RuntimeError("PairStreams.result(): nothing submitted")
        item = self._results.get(self._returned)
        self._returned += 1
        if isinstance(item, BaseException):
            raise item
        out, done = item
        cur = torch.cuda.current_stream(self.device)
        if wait:
            cur.wait_event(done)
        for t in out.values():
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
        out["done_event"] = done
        self._check_status(wait=False)
        return out

    @staticmethod
    def check(out):
        """Wait for THIS pair's kernels and raise if its tie-order restore step reported a capacity status (the status
        word is written on the device after result() has returned the dict; result() / synchronize() raise it too, but
        for whichever pair has finished by then -- use this to tie the check to one pair)."""
        out["done_event"].synchronize()
        status, slot = out["_tie_status"]
        check_tie_status(int(status[slot]))
        return out

    def _check_status(self, wait):
        """Tie-order restore status of finished pairs (pcrcg_pyramid_build writes it asynchronously)."""
        with self._lock:
            pending, self._pending = self._pending, []
        bad, keep = 0, []
        for status, slot, ev in pending:
            if wait:
                ev.synchronize()
            if ev.query():
                bad = bad or int(status[slot])
            else:
                keep.append((status, slot, ev))
        with self._lock:
            self._pending = keep + self._pending
        check_tie_status(bad)

    def drain(self):
        while self._returned < self._submitted:
            self.result(wait=False)
        self.synchronize()

    def synchronize(self):
        for s in self.fronts + [x for pair in self.sides for x in pair if x is not None] + self.models:
            s.synchronize()
        self._check_status(wait=True)

    def close(self):
        self._in.put(None)
        for mb in self._mid + [self._results]:
            mb.put(None, None)
        for t in self._threads:
            t.join(timeout=10)
        self._threads = []
