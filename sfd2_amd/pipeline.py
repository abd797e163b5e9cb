"""Pipelined host drivers: what keeps one MI355X fed from image files and feature stores.

The reference hides the JPEG decoder behind 4 DataLoader worker processes (extract_localization.py:230-233) and then
runs decode -> network -> numpy -> HDF5 strictly in turn per image (:240-272); per pair it reads both groups, casts and
uploads them and launches one matcher call (hloc/match_features.py:90-119).  At 1.5 ms of device time per 1600x1200
image and 4 us per pair those loops, not the kernels, set the rate.  This module is the host side that removes them:

  extract:  decode pool (threads; PIL releases the GIL inside the decoder) -> pinned uint8 buffers -> asynchronous
            sfd2_extract (the library's copy stream uploads image i + 1 under the network of image i; outputs land in
            pinned slots; sfd2_extract_record_async brings the count and the per-image range verdict) -> writer
            threads (float64 containers, key-point rescale, store append).
  match:    pairs grouped by their first image; every descriptor set is converted ONCE (sfd2_desc_pack) and kept
            resident in HBM as fp16 [n][128] in an LRU sized from free device memory; one sfd2_match_batch per query
            group; a reader pool prefetches the sets the next groups need; a writer thread casts and appends.

Both produce exactly what the serial loops of extract_localization.main / match_features.main produce (tests compare
the stores dataset by dataset).  torch is used for pinned / device allocations and for an event on the library's stream.
"""
import collections
import ctypes
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib


# ----------------------------------------------------------------------------------------------- buffers
class PinnedPool:
    """A fixed number of growable pinned host buffers (uint8).  acquire() blocks until one is free: that is the
    back-pressure that bounds how far the decoders run ahead of the device."""

    def __init__(self, count, nbytes=0):
        import torch
        self._torch = torch
        self._free = queue.Queue()
        for _ in range(count):
            self._free.put(_PinnedBuf(torch, nbytes))

    def acquire(self, block=True):
        try:
            return self._free.get(block=block)
        except queue.Empty:
            return None

    def release(self, buf):
        self._free.put(buf)


class _PinnedBuf:
    def __init__(self, torch, nbytes):
        self._torch = torch
        self.t = None
        self.np = None
        if nbytes:
            self.reserve(nbytes)

    def reserve(self, nbytes):
        """uint8 numpy view of at least nbytes of pinned memory (reallocated only when too small)."""
        if self.t is None or self.t.numel() < nbytes:
            self.t = None
            self.t = self._torch.empty(int(nbytes), dtype=self._torch.uint8, pin_memory=True)
            self.np = self.t.numpy()
        return self.np[:nbytes]


# ----------------------------------------------------------------------------------------------- extract
class _Slot:
    """Pinned output arrays of one image in flight."""

    def __init__(self, torch, cap):
        self.cap = cap
        pin = dict(pin_memory=True)
        self.kp_t = torch.empty((cap, 2), dtype=torch.float32, **pin)
        self.sc_t = torch.empty((cap,), dtype=torch.float32, **pin)
        # descriptors as the reference STORES them: float64 [128][cap] (SFD2_FLAG_DESC_STORE64: the device does the cast and the transposition that
        # extract_localization.py:253,269-272 do on the host -- the writer threads' largest cost, tools/host_soak.py)
        self.de_t = torch.empty((128, cap), dtype=torch.float64, **pin)
        self.rec_t = torch.zeros((4,), dtype=torch.int32, **pin)
        self.kp, self.sc, self.de64, self.rec = self.kp_t.numpy(), self.sc_t.numpy(), self.de_t.numpy(), self.rec_t.numpy()
        self.event = torch.cuda.Event()
        self.meta = None
        self.inbuf = None
        self.sync_result = None


class AsyncExtractor:
    """Several single-scale, mask-free extractions in flight on one context (extract_resnet_return's arithmetic, the
    C-ABI's asynchronous form).  submit() queues one decoded uint8 image; results come back in submission order from
    finish(), as the float32 arrays the synchronous call fills."""

    def __init__(self, model, top_k, conf_th, depth=3, slots=8, lanes=1):
        """lanes > 1: that many contexts (model.lanes(n): own HIP stream, workspace and staging slots; the replicas are made once per
        model) take the images in turn; results do not depend on the lane.  Measured in THIS loop on pre-decoded images (f16c, 16 threads,
        512 images; CHANGELOG round 5): 678 images/s with one lane, 728-743 with two (4-6 images in flight).  (Rounds 4-5 had "two lanes are
        slower, 393 against 547": that run created the replica -- weight upload, packing, probes: ~0.3 s -- inside its timed region.)"""
        import torch
        if top_k <= 0:
            raise ValueError("the pipelined extractor needs a key-point capacity (max_keypoints > 0)")
        self.torch = torch
        self.model, self.ctx = model, model.context
        self.top_k, self.conf_th, self.depth = int(top_k), float(conf_th), int(depth)
        self.flags = 0 if getattr(model, "require_stability", True) else _lib.FLAG_NO_STABILITY
        self.device = torch.device("cuda", self.ctx.device)
        nl = max(1, int(lanes))
        self.models = model.lanes(nl) if hasattr(model, "lanes") else [model] + [model.replica() for _ in range(nl - 1)]
        with torch.cuda.device(self.device):
            self._free = queue.Queue()
            for _ in range(max(slots, depth + 1)):
                self._free.put(_Slot(torch, self.top_k))
            self.streams = [torch.cuda.ExternalStream(m.context.stream, device=self.device) for m in self.models]
        self._resized = [None] * len(self.models)   # device float32 [3][h][w] of the image being resized (stream ordered: one per lane is enough)
        self._turn = 0
        self.inflight = collections.deque()
        self.repeats = 0                # images re-run synchronously (range fallback)

    def submit(self, image_u8, H, W, resize, meta, inbuf=None):
        """image_u8: uint8 [H,W,3] RGB or [H,W,4] RGBX view (pinned for a truly asynchronous upload); resize (w, h) or None."""
        if not image_u8.flags.c_contiguous:
            image_u8 = np.ascontiguousarray(image_u8)      # (a transposed view multiplied / cast keeps its operand's layout)
        slot = self._free.get()         # blocks while the writers are behind: back-pressure
        lane = self._turn % len(self.models)
        self._turn += 1
        ctx, torch = self.models[lane].context, self.torch
        lib = ctx.lib
        slot.lane = lane
        slot.meta, slot.inbuf, slot.sync_result = meta, inbuf, None
        slot.image = image_u8           # kept for a synchronous repeat
        src, on_dev, h, w = image_u8.ctypes.data, 0, H, W
        flags = self.flags | _lib.FLAG_ASYNC
        px4 = _lib.FLAG_IMG_U8_X if image_u8.shape[-1] == 4 else 0
        if resize is not None and tuple(resize) != (W, H):
            w, h = int(resize[0]), int(resize[1])
            if self._resized[lane] is None or self._resized[lane].numel() < 3 * h * w:
                ctx.sync()              # nothing may still read the old buffer when it is replaced
                self._resized[lane] = torch.empty(3 * h * w, dtype=torch.float32, device=self.device)
            _lib.check(lib.sfd2_preprocess(ctx.h, src, 0, H, W, _lib.FLAG_ASYNC | px4, h, w, self._resized[lane].data_ptr()))
            src, on_dev = self._resized[lane].data_ptr(), 1
        else:
            flags |= _lib.FLAG_IMG_U8_HWC | px4
        slot.size = (w, h)
        slot.resize = None if on_dev == 0 else (w, h)
        n = ctypes.c_int(0)
        _lib.check(lib.sfd2_extract(ctx.h, src, on_dev, h, w, self.conf_th, self.top_k, flags | _lib.FLAG_DESC_STORE64, slot.kp.ctypes.data,
                                    slot.sc.ctypes.data, slot.de64.ctypes.data, 0, slot.cap, ctypes.byref(n)))
        _lib.check(lib.sfd2_extract_record_async(ctx.h, slot.rec.ctypes.data, 0))
        slot.event.record(self.streams[lane])
        self.inflight.append(slot)

    def finish(self):
        """Oldest image in flight -> its slot (n, kp, sc, de valid; give it back with release())."""
        slot = self.inflight.popleft()
        slot.event.synchronize()
        n, _, saturated, flags = (int(v) for v in slot.rec.view(np.uint32))
        if flags & 1:
            raise RuntimeError("libsfd2hip: candidate buffer overflow in a pipelined extract")
        if saturated:
            # SFD2_PREC_F16C left its range on this image: the synchronous call repeats it in SFD2_PREC_F16X3 by itself
            from .extractor import extract_resnet_return
            img, model = slot.image, self.models[slot.lane]
            if img.shape[-1] == 4:
                img = np.ascontiguousarray(img[:, :, :3])
            if slot.resize is not None:
                from .extract_localization import preprocess
                img = preprocess(model, img, slot.resize)
            slot.sync_result = extract_resnet_return(model, img=img, topK=self.top_k, conf_th=self.conf_th)
            self.repeats += 1
        slot.n = n
        slot.image = None
        return slot

    def release(self, slot):
        self._free.put(slot)


def slot_arrays(slot):
    """(keypoints [n,2], scores [n], descriptors [n,128]) as float64 -- the containers extract_resnet_return returns
    (nets/extractor.py:322-337)."""
    if slot.sync_result is not None:
        r = slot.sync_result
        return r["keypoints"], r["scores"], r["descriptors"]
    n = slot.n
    # descriptors: the transposed VIEW of the slot's float64 [128][cap] block -- the caller's .transpose() (extract_localization.py:253) gives the stored
    # [128, n] array back without a copy when the image filled its capacity (n == cap: every BASELINE workload); the store copies it out before the slot is reused
    return slot.kp[:n].astype(np.float64), slot.sc[:n].astype(np.float64), slot.de64[:, :n].T


class OrderedPrefetch:
    """load(idx, resource) for idx in `indices` on `workers` threads, results handed out in order.  `window` bounds the
    loads running or waiting.  `claim` (optional, non-blocking, returns None when nothing is free) is called in ORDER on
    the consumer's thread before a load is submitted -- it deals the pinned input buffers, so a late item can never
    starve behind buffers held by its successors, and the consumer (who also gives buffers back) never blocks on one."""

    def __init__(self, load, indices, workers, window, claim=None):
        self._load, self._it, self._claim = load, iter(indices), claim
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="sfd2-decode")
        self._pending = collections.deque()
        self._window = max(1, window)
        self._done = False
        self._held = None           # next index, waiting for a resource
        self._have_held = False

    def top_up(self):
        while not self._done and len(self._pending) < self._window:
            if not self._have_held:
                try:
                    self._held = next(self._it)
                    self._have_held = True
                except StopIteration:
                    self._done = True
                    return
            res = None
            if self._claim is not None:
                res = self._claim()
                if res is None:
                    return          # nothing free now; retried on the next call
            idx, self._have_held = self._held, False
            self._pending.append(self._pool.submit(self._load, idx, res))

    @property
    def ready(self):
        return bool(self._pending)

    @property
    def exhausted(self):
        return self._done and not self._pending

    def pop(self):
        """Next item in order (blocks until its load has finished)."""
        return self._pending.popleft().result()

    def close(self):
        for f in self._pending:
            f.cancel()
        self._pool.shutdown(wait=True)


class WriterPool:
    """`workers` threads calling fn(job) for queued jobs; errors surface on the producer's next put() / on close().  After an error the
    remaining jobs are not written but still handed to `on_drop(job)` (optional), which must give back whatever the job carries (an
    extractor slot, a ring of pinned buffers): a producer waiting for one of those then wakes up and meets the error in its next put()
    instead of waiting for ever."""

    def __init__(self, fn, workers=1, maxsize=0, name="sfd2-writer", on_drop=None):
        self._fn, self._q, self._err, self._on_drop = fn, queue.Queue(maxsize), None, on_drop
        self._threads = [threading.Thread(target=self._run, name=f"{name}-{i}", daemon=True) for i in range(max(1, workers))]
        for t in self._threads:
            t.start()

    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            if self._err is None:
                try:
                    self._fn(job)
                except BaseException as e:      # noqa: BLE001 - handed to the producer
                    if self._err is None:
                        self._err = e
            elif self._on_drop is not None:
                try:
                    self._on_drop(job)
                except BaseException:           # noqa: BLE001 - the first error is the one reported
                    pass

    def put(self, job):
        if self._err is not None:
            raise self._err
        self._q.put(job)

    def close(self):
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        if self._err is not None:
            raise self._err


# ----------------------------------------------------------------------------------------------- match
class _PinnedStage:
    """One pinned host buffer of the descriptor reads + the event behind its last asynchronous use."""

    def __init__(self, torch):
        self._torch, self.t, self.np = torch, None, None
        self.event, self.busy = torch.cuda.Event(), False

    def reserve(self, nbytes):
        if self.busy:
            self.event.synchronize()
            self.busy = False
        if self.t is None or self.t.numel() < nbytes:
            self.t = self._torch.empty(int(max(nbytes, 1)), dtype=self._torch.uint8, pin_memory=True)
            self.np = self.t.numpy()
        return self.np[:nbytes]


class ResidentSets:
    """Descriptor sets of a feature store as fp16 [n][128] in HBM, least recently used evicted.  `budget` bytes
    (default: 60 % of the device memory free at construction).  get(name) -> (device pointer, n); a set is converted
    once by sfd2_desc_pack from the store's float64 [128, n] dataset -- the (float)(double) then fp16 rounding
    hloc/match_features.py:105's .float() followed by the matcher's own conversion performs."""

    def __init__(self, ctx, feats, budget=None, readers=4, lock_reads=None):
        import torch
        self.torch, self.ctx, self.feats = torch, ctx, feats
        self.device = torch.device("cuda", ctx.device)
        if budget is None:
            free, _ = torch.cuda.mem_get_info(self.device)
            budget = int(free * 0.6)
        self.budget, self.used = int(budget), 0
        self._sets = collections.OrderedDict()      # name -> (tensor, n, last use seq)
        self._pending = {}                          # name -> future of the host read
        self._pool = ThreadPoolExecutor(max_workers=max(1, readers), thread_name_prefix="sfd2-read")
        # h5py files are not safe to read from several threads; the stand-in stores are
        safe = getattr(feats, "threadsafe_reads", False) if lock_reads is None else not lock_reads
        self._rlock = None if safe else threading.Lock()
        self.loads = self.hits = self.evictions = 0
        self.completed_seq = -1                     # last query whose device work is known to be finished
        # pinned staging for the reads: a set read into pageable memory cost its query ~0.4 ms of synchronous upload (4 MB of float64) on the
        # driver's thread; read into a pinned buffer by the reader thread, the conversion kernel's upload is asynchronous.  A buffer is reused once
        # the event recorded behind its conversion has passed.
        self._stage_free = queue.Queue()
        for _ in range(max(64, 4 * max(1, readers))):       # (buffers are allocated on first use: a driver that keeps few reads pending touches few)
            self._stage_free.put(_PinnedStage(torch))
        self._stream = torch.cuda.ExternalStream(ctx.stream, device=self.device)

    def _read(self, name):
        if self._rlock is not None:
            with self._rlock:
                src = self.feats[name]['descriptors'].__array__()
        else:
            src = self.feats[name]['descriptors'].__array__()
        if src.dtype not in (np.float64, np.float32, np.float16) or src.ndim != 2:
            return np.ascontiguousarray(src), None
        # never WAIT for a stage: the driver's thread is the one that gives stages back, and it may be waiting for this very read (a caller is free to
        # prefetch in one order and consume in another).  With none free the set goes through pageable memory (a synchronous upload: slower, never stuck).
        try:
            st = self._stage_free.get_nowait()
        except queue.Empty:
            return np.ascontiguousarray(src), None
        dst = st.reserve(src.nbytes).view(src.dtype).reshape(src.shape)       # (waits for the buffer's previous conversion)
        np.copyto(dst, src)
        return dst, st

    def prefetch(self, names):
        for name in names:
            if name not in self._sets and name not in self._pending:
                self._pending[name] = self._pool.submit(self._read, name)

    def get(self, name, seq):
        """seq: number of the query group being assembled (sets it uses are not evicted while it is)."""
        ent = self._sets.get(name)
        if ent is not None:                 # (the hit path runs ~50 times per query group on the driver's thread: no tensor calls, no tuple rebuild)
            self._sets.move_to_end(name)
            ent[2] = seq
            self.hits += 1
            return ent[3], ent[1]
        fut = self._pending.pop(name, None)
        d, stage = fut.result() if fut is not None else self._read(name)
        if d.ndim != 2:
            raise ValueError(f"descriptors of {name!r}: expected [dim, n]")
        dim, n = d.shape
        dt = {np.dtype(np.float64): _lib.DT_F64, np.dtype(np.float32): _lib.DT_F32, np.dtype(np.float16): _lib.DT_F16}.get(d.dtype)
        if dt is None:
            d, dt = d.astype(np.float32), _lib.DT_F32
            if stage is not None:
                self._stage_free.put(stage)
                stage = None
        nbytes = max(n, 1) * 128 * 2
        self._make_room(nbytes, seq)
        t = self.torch.empty(max(n, 1) * 128, dtype=self.torch.float16, device=self.device)
        src = _lib.DescSet(d.ctypes.data, n, dt, _lib.LAYOUT_DN, 0, None, 0, 0)
        if stage is not None:       # pinned source: the upload rides on the stream; the buffer goes back behind an event
            _lib.check(self.ctx.lib.sfd2_desc_pack(self.ctx.h, ctypes.byref(src), dim, t.data_ptr(), _lib.FLAG_ASYNC))
            stage.event.record(self._stream)
            stage.busy = True
            self._stage_free.put(stage)
        else:                       # synchronous: `d` is a temporary
            _lib.check(self.ctx.lib.sfd2_desc_pack(self.ctx.h, ctypes.byref(src), dim, t.data_ptr(), 0))
        self._sets[name] = [t, n, seq, t.data_ptr()]
        self.used += nbytes
        self.loads += 1
        return self._sets[name][3], n

    def _make_room(self, nbytes, cur_seq):
        synced = False
        while self.used + nbytes > self.budget:
            victim = next((k for k, v in self._sets.items() if v[2] != cur_seq), None)   # oldest first; never the group in assembly
            if victim is None:
                return                      # one group needs more than the budget: over it for this group
            t, n, seq, _ = self._sets[victim]
            if seq > self.completed_seq and not synced:
                self.ctx.sync()             # the victim may still be read by a queued batch
                synced = True
            del self._sets[victim]
            self.used -= max(n, 1) * 128 * 2
            self.evictions += 1

    def close(self):
        self._pool.shutdown(wait=True)
        self._sets.clear()
