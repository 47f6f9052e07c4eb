"""Scene loader for the data path in front of the hot path.

The reference feeds the model through a torch DataLoader with 8 worker processes, pin_memory and the PyG collate of
``HierarchicalData`` (datasets/scannetcolorgraph_dataloader.py:190-208, utils/data_utils.py:11-42), one scene (3-D
training) or a batch of crops per step.  This module is the MI355X-side equivalent, designed around 288 GB of HBM:

* a background thread reads / collates / pins the NEXT batches while the GPU trains (``prefetch`` deep);
* host->device copies run on their own HIP stream; the compute stream only waits for the copy event;
* the graph part of a scene (int64 index tensors AND the CSR plan built from them) never changes between epochs or
  between masks, so it is kept RESIDENT in HBM in an LRU cache bounded in bytes - a revisited scene uploads only its
  per-vertex features (x, color, mask) and skips the plan build entirely (~1.4 ms of a 10 ms step at 200k vertices);
* scenes are sharded over data-parallel ranks like a DistributedSampler (seeded permutation per epoch, padded so that
  every rank runs the same number of steps - the gradient all-reduce needs that).

Items are ``(graph_path, mask_path)`` pairs in the reference's on-disk schema (scene_io.load_scene), already built CPU
``HierarchicalBatch`` objects, or zero-argument callables returning one.
"""
import collections
import concurrent.futures
import ctypes
import queue
import threading

import torch

from . import plan as _plan
from .data import HierarchicalBatch, collate, sample_keys
from .scene_io import load_scene

_FEATURE_KEYS = ('x', 'color', 'mask', 'batch', 'name')


def shard_indices(n, epoch, seed=0, shuffle=True, rank=0, world_size=1, sizes=None):
    """The item order of one epoch for one rank: seeded permutation of range(n), padded by wrap-around to a multiple of
    world_size, every world_size-th element starting at rank (torch.utils.data.DistributedSampler semantics).

    sizes (optional, one number per item - e.g. the level-0 vertex count): SIZE-BALANCED sharding for the synchronous
    gradient all-reduce.  ScanNet scenes spread over 12 k...28 k (crops) / 50 k...400 k (scenes) vertices and a step takes
    as long as its slowest rank, so the items are sorted by size, cut into consecutive buckets of world_size items (the
    ranks of one step get neighbours in size), the ORDER of the buckets is shuffled per epoch and the items of a bucket
    are dealt to the ranks by a per-bucket seeded rotation (no rank always gets the largest of its bucket).  Same
    coverage / padding guarantees as the plain form."""
    if sizes is not None and world_size > 1 and n > 0:
        assert len(sizes) == n
        by_size = sorted(range(n), key=lambda i: (-float(sizes[i]), i))
        total = (n + world_size - 1) // world_size * world_size
        by_size = by_size + by_size[:total - n]
        buckets = [by_size[b:b + world_size] for b in range(0, total, world_size)]
        g = torch.Generator().manual_seed(int(seed) + int(epoch))
        order = torch.randperm(len(buckets), generator=g).tolist() if shuffle else list(range(len(buckets)))
        rot = torch.randint(0, world_size, (len(buckets),), generator=g).tolist() if shuffle else [0] * len(buckets)
        return [buckets[b][(rank + rot[b]) % world_size] for b in order]
    if shuffle:
        g = torch.Generator().manual_seed(int(seed) + int(epoch))
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    if world_size > 1 and n > 0:
        total = (n + world_size - 1) // world_size * world_size
        order = order + order[:total - n]
    return order[rank::world_size]


def _tensor_bytes(obj):
    seen, total = set(), 0
    stack = [obj]
    while stack:
        o = stack.pop()
        if torch.is_tensor(o):
            st = o.untyped_storage()
            if st.data_ptr() not in seen:
                seen.add(st.data_ptr())
                total += st.nbytes()
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif isinstance(o, (list, tuple)):
            stack.extend(o)
        elif hasattr(o, '__dict__') or hasattr(o, '__slots__'):
            for k in list(getattr(o, '__dict__', {}).keys()) + list(getattr(o, '__slots__', ())):
                if k in ('_sample',):
                    continue
                v = getattr(o, k, None)
                if v is not None and not callable(v):
                    stack.append(v)
    return total


class ResidentGraphCache:
    """LRU of the immutable graph part of scenes, resident in HBM: {key: (index tensors on the device, GraphPlan)}."""

    def __init__(self, capacity_bytes):
        self.capacity = int(capacity_bytes)
        self.used = 0
        self.hits = self.misses = 0
        self._d = collections.OrderedDict()

    def get(self, key):
        e = self._d.get(key)
        if e is None:
            self.misses += 1
            return None
        self._d.move_to_end(key)
        self.hits += 1
        return e

    def put(self, key, graph_tensors, plan, nbytes):
        if nbytes > self.capacity:
            return
        while self.used + nbytes > self.capacity and self._d:
            _, (_, _, b) = self._d.popitem(last=False)
            self.used -= b
        self._d[key] = (graph_tensors, plan, nbytes)
        self.used += nbytes

    def __len__(self):
        return len(self._d)


class _StagingSlot:
    """One pinned staging area of the upload ring: a persistent pinned byte buffer per key (grown on demand) and the
    event after which the H2D copies issued from it have completed - the worker waits for it before overwriting the slot.
    Staging through persistent buffers keeps the worker thread's work at one GIL-releasing memcpy per tensor."""

    def __init__(self):
        self.bufs = {}
        self.done = None

    def stage(self, key, t):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(int(nbytes * 1.25), 64), dtype=torch.uint8, pin_memory=True)
        view = buf[:nbytes].view(t.dtype).view(t.shape)
        # a plain single-threaded memcpy with the GIL released (ctypes): torch's CPU copy_ fans out over the intra-op
        # thread pool, and waking ~256 sleeping OpenMP threads for a few megabytes cost 7 ms per tensor AND slowed the
        # training thread (step enqueue 5 -> 11-16 ms) while they spun
        if nbytes:
            ctypes.memmove(view.data_ptr(), t.data_ptr(), nbytes)
        return view


class SceneLoader:
    """Iterate GPU-resident training batches: ``for sample in loader.epoch(e): loss = step(sample)``."""

    def __init__(self, items, device, batch_size=1, shuffle=True, seed=0, rank=0, world_size=1, prefetch=2,
                 cache_bytes=32 << 30, end_level=3, cropped=False, model=None, host_cache_bytes=64 << 30, worker_threads=4, workers=2,
                 sizes=None, locality_order=True):
        self.items = list(items)
        # locality_order (round 6, default on): the plan of a scene whose graph part stays resident in HBM (the cache below)
        # is built with the vertices renumbered by locality (plan.GraphPlan(reorder=True): Morton order at level 0, first-child order
        # above).  The renumbering is paid ONCE per scene - the permutation lives in the cached plan - and every later step gathers
        # neighbour rows from L2 instead of across the fabric (edge backward: no 16-byte mask gathers in foreign 64-byte sectors).
        # Invisible at the model boundary: the network enters the plan's order at its input and returns colours in the caller's
        # vertex order (surfacetextureinpaintingnet.forward, PermuteRowsFn).  Without a cache (plans rebuilt every step) the
        # renumbering costs more than it gains and stays off.
        self.locality_order = bool(locality_order)
        self.device = torch.device(device)
        self.batch_size, self.shuffle, self.seed = int(batch_size), bool(shuffle), int(seed)
        self.rank, self.world_size, self.prefetch = int(rank), int(world_size), max(1, int(prefetch))
        self.end_level, self.cropped = end_level, cropped
        # per-item sizes (level-0 vertex counts) -> size-balanced rank sharding (shard_indices); in-memory items carry
        # their size, file items need the caller's list (e.g. from the dataset index)
        if sizes is None and self.world_size > 1 and all(isinstance(it, HierarchicalBatch) for it in self.items):
            sizes = [int(it.x.shape[0]) for it in self.items]
        self.sizes = None if sizes is None else [float(v) for v in sizes]
        self.model = model                                   # optional: lets the loader build the plan (model.prefetch_plan)
        self.worker_threads = max(1, int(worker_threads))    # intra-op threads of each WORKER thread's torch ops (collate ...)
        self.workers = max(1, int(workers))                  # worker threads preparing batches side by side (order is kept)
        self._host_lock = threading.Lock()
        self.cache = ResidentGraphCache(cache_bytes) if (cache_bytes and self.batch_size == 1) else None
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        # parsed scenes (file items) kept in host RAM, LRU in bytes: a revisit costs a dict lookup instead of ~30 ms of
        # torch.load + layout conversion in the worker thread (ScanNet: ~1200 scenes x ~48 MB = 58 GB for one rank)
        self._host_cache = collections.OrderedDict()
        self._host_cache_cap, self._host_cache_used = int(host_cache_bytes or 0), 0
        self._ring = [_StagingSlot() for _ in range(self.prefetch + self.workers + 2)] if self.device.type == 'cuda' else []

    # ---- CPU side (background thread) ---------------------------------------------------------------------------
    def _load(self, i):
        it = self.items[i]
        if isinstance(it, HierarchicalBatch):
            return it
        if callable(it):
            return it()
        with self._host_lock:
            hit = self._host_cache.get(i)
            if hit is not None:
                self._host_cache.move_to_end(i)
                return hit[0]
        scene = load_scene(it[0], it[1], end_level=self.end_level, cropped=self.cropped)
        if self._host_cache_cap:
            nb = _tensor_bytes({k: scene[k] for k in sample_keys(scene)})
            if nb <= self._host_cache_cap:
                with self._host_lock:
                    while self._host_cache_used + nb > self._host_cache_cap and self._host_cache:
                        _, (_, b) = self._host_cache.popitem(last=False)
                        self._host_cache_used -= b
                    if i not in self._host_cache:
                        self._host_cache[i] = (scene, nb)
                        self._host_cache_used += nb
        return scene

    def steps_per_epoch(self):
        n = len(shard_indices(len(self.items), 0, self.seed, False, self.rank, self.world_size, self.sizes))
        return (n + self.batch_size - 1) // self.batch_size

    def _batch_ids(self, epoch):
        idx = shard_indices(len(self.items), epoch, self.seed, self.shuffle, self.rank, self.world_size, self.sizes)
        return [idx[b:b + self.batch_size] for b in range(0, len(idx), self.batch_size)]

    def _prepare(self, turn, ids):
        """CPU side of batch number `turn` (a worker thread): load, collate, stage into the turn's pinned slot."""
        samples = [self._load(i) for i in ids]
        batch = samples[0] if len(samples) == 1 else collate(samples)
        if self.device.type != 'cuda':
            return ids, batch
        key = ids[0] if self.cache is not None else None
        cached = key is not None and key in self.cache._d       # peek (no LRU update from a worker thread)
        keys = [k for k in sample_keys(batch) if (k in _FEATURE_KEYS or not cached)]
        slot = self._ring[turn % len(self._ring)]
        if slot.done is not None:
            slot.done.synchronize()                             # the uploads issued from this slot have left it
            slot.done = None
        pinned = HierarchicalBatch(**{k: (slot.stage(k, batch[k]) if torch.is_tensor(batch[k]) else batch[k]) for k in keys})
        object.__setattr__(pinned, '_slot', slot)            # not a data key: stays out of the sample and the cache
        return ids, pinned

    def _cpu_batches(self, epoch):
        for turn, ids in enumerate(self._batch_ids(epoch)):
            yield self._prepare(turn, ids)

    # ---- device side ------------------------------------------------------------------------------------------------
    def _to_device(self, ids, cpu_batch):
        if self.device.type != 'cuda':
            return cpu_batch
        main = torch.cuda.current_stream(self.device)
        key = ids[0] if self.cache is not None else None
        entry = self.cache.get(key) if key is not None else None
        with torch.cuda.stream(self._copy_stream):
            dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v)
                   for k, v in ((k, cpu_batch[k]) for k in sample_keys(cpu_batch))}
        for v in dev.values():
            if torch.is_tensor(v):
                v.record_stream(main)
        slot = cpu_batch.__dict__.get('_slot')
        if slot is not None:
            slot.done = self._copy_stream.record_event()        # the worker may overwrite the slot after this
        uploaded = self._copy_stream.record_event()
        main.wait_stream(self._copy_stream)
        if entry is not None:                                 # resident graph part + its plan: nothing to upload or build
            graph, plan, _ = entry
            out = HierarchicalBatch(**graph)
            for k in _FEATURE_KEYS:
                if k in dev:
                    out[k] = dev[k]
            plan._sample = out
            out._plan_cache = plan
            return out
        if key is not None and any(k not in dev for k in ('edge_index', 'num_vertices')):
            # the worker skipped the graph tensors expecting a cache hit, but the entry was evicted meanwhile: reload
            return self._to_device(ids, next(self._reload(ids)))
        out = HierarchicalBatch(**dev)
        if 'num_vertices' in cpu_batch:
            out._nv_host = cpu_batch['num_vertices'].clone()    # level sizes for the plan without a device sync (a copy:
                                                                # the staging slot is overwritten by a later batch)
        # the batch is handed out while the previous step is still executing (the host runs ahead of the GPU): with a model
        # the whole CSR plan is built NOW on the plan side streams, after the upload and beside that step, and the compute
        # stream joins it when this batch's forward starts
        plan = None
        if self.model is not None and hasattr(self.model, 'build_plan'):
            plan = out._plan_cache = self.model.build_plan(out, after=uploaded, reorder=self._reorder_flag())
        if self.cache is not None:
            if plan is None:
                plan = self.model.prefetch_plan(out, reorder=self._reorder_flag()) if self.model is not None else _plan.plan_for(out)
            graph = {k: v for k, v in dev.items() if k not in _FEATURE_KEYS}
            self._pending = (key, graph, plan)
        return out

    def _reorder_flag(self):
        """True where the plan being built will be kept (graph cache on): renumber by locality once; None = the module default."""
        return True if (self.locality_order and self.cache is not None) else None

    def _reload(self, ids):
        s = self._load(ids[0])
        yield HierarchicalBatch(**{k: (s[k].pin_memory() if torch.is_tensor(s[k]) else s[k]) for k in sample_keys(s)})

    def _commit_pending(self):
        """Insert the previous step's graph into the cache once its plan is complete (all lazily built pieces exist)."""
        p = getattr(self, '_pending', None)
        if p is not None:
            key, graph, plan = p
            self.cache.put(key, graph, plan, _tensor_bytes(graph) + _tensor_bytes(plan))
            self._pending = None

    def epoch(self, epoch=0):
        """Batches of one epoch in order.  `workers` threads prepare the next `prefetch` batches side by side (their torch
        CPU ops limited to `worker_threads` intra-op threads each, thread-locally: waking the process-wide pool of
        hundreds of threads costs milliseconds per op and slows the training thread); slot `turn % ring` of the pinned
        staging ring belongs to batch `turn`, and a batch is only submitted after batch `turn - prefetch - 1` was handed
        to the GPU, so a slot is never overwritten before its uploads were issued (and `slot.done` covers their
        completion)."""
        tasks = self._batch_ids(epoch)

        def init():
            torch.set_num_threads(self.worker_threads)

        ex = concurrent.futures.ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix='stin-loader', initializer=init)
        window = collections.deque()
        try:
            nxt = 0
            while nxt < len(tasks) or window:
                while nxt < len(tasks) and len(window) <= self.prefetch:
                    window.append(ex.submit(self._prepare, nxt, tasks[nxt]))
                    nxt += 1
                item = window.popleft().result()                 # re-raises a worker's exception here
                if self.cache is not None:
                    self._commit_pending()
                yield self._to_device(*item)
            if self.cache is not None:
                self._commit_pending()
        finally:
            for f in window:
                f.cancel()
            ex.shutdown(wait=True)
