"""Graph plan: the per-sample CSR structures every kernel of the hot path runs on.

Built ONCE per sample on the GPU (stin_csr_from_coo_i64) and cached on the sample
object, shared by all blocks of a level:
  * per edge set: destination CSR (row i -> sources j, original edge order) for the
    forward gather / dA pass, source CSR (row j -> targets i) for the dB pass, and
    inv_deg = 1 / max(1, in-degree);
  * per pooling level: int32 trace, children CSR (fine vertices of each coarse
    vertex, ascending), inv_count;
  * per level: instance-norm row groups (gid / slice ids) for batched samples.
Indices are int64 at the Python boundary (as in the reference) and int32 inside.
"""
import collections
import ctypes
import struct

import torch

from . import _lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t):
    """hipStream_t of torch's current stream on t's device (the fast C accessor when this torch has it)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    if t is None:
        return 0
    p = t.data_ptr()
    if p == 0 and t.untyped_storage().nbytes() > 0:       # zero-element view (an empty edge set): still a valid address
        p = t.untyped_storage().data_ptr() + t.storage_offset() * t.element_size()
    return p


_CONSTANTS = collections.OrderedDict()        # (device, dtype, shape, bytes) -> device tensor, LRU
_CONSTANTS_MAX = 1024


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _upload(host_tensor, device):
    """Small host tensor (row-pointer arrays, 1 / count vectors: tens of values) -> read-only device constant without
    stalling the host: pinned staging buffer + non-blocking copy (the caching host allocator keeps the staging block
    alive until the copy has run).  Constants are kept in a small LRU keyed by their bytes: a batch composition seen
    before costs a dict lookup, and a step being captured into a HIP graph (train_step.TrainStep(graph=True)) - where
    host-to-device copies cannot be recorded - finds the arrays its eager warm-up pass uploaded."""
    if torch.device(device).type != 'cuda':
        return host_tensor.to(device)
    host_tensor = host_tensor.contiguous()
    key = (str(device), host_tensor.dtype, tuple(host_tensor.shape), host_tensor.numpy().tobytes())
    hit = _CONSTANTS.get(key)
    if hit is not None:
        _CONSTANTS.move_to_end(key)
        return hit
    if _capturing():
        raise RuntimeError('a host array must be uploaded while a HIP graph is being captured: run the step once eagerly '
                           'on a sample of the same composition first (TrainStep(graph=True) does)')
    pinned = torch.empty(host_tensor.shape, dtype=host_tensor.dtype, pin_memory=True)
    pinned.copy_(host_tensor)
    dev = pinned.to(device, non_blocking=True)
    _CONSTANTS[key] = dev
    while len(_CONSTANTS) > _CONSTANTS_MAX:
        _CONSTANTS.popitem(last=False)
    return dev


FLAG_POOL = 4096
_FLAG_POOLS = {}


def _flag_word(device):
    """-> a [1] int32 out-of-range flag for one plan.

    Eager plans take a word of the device's zero-initialised pool (round-robin: no fill kernel per plan).  A plan built
    while a HIP graph is being captured gets a tensor of its own inside the capture instead: the allocation belongs to the
    graph's private pool and its zero fill is a memset node, so EVERY replay starts from a clean flag (a pool word would keep
    the 1 of a bad sample for all later replays, and its address - baked into the graph - would come round again for an eager
    plan FLAG_POOL plans later)."""
    if _capturing():
        return torch.zeros(1, dtype=torch.int32, device=device)
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ent = _FLAG_POOLS.get(key)
    if ent is None:
        pool = torch.zeros(FLAG_POOL, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()        # side streams may use a word before the creating stream is joined
        ent = _FLAG_POOLS[key] = [pool, 0]
    ent[1] = (ent[1] + 1) % FLAG_POOL
    half = FLAG_POOL // 2
    if ent[1] % half == 0:
        # entering a half of the pool: its words were handed out FLAG_POOL/2 .. FLAG_POOL plans ago, so whatever plan dirtied
        # one without ever validating (validate=False) has long been dropped.  Builds on the side streams may still be running
        # for RECENT plans (the other half), so the fill is fenced on both sides by a device-wide wait - once per FLAG_POOL/2
        # plans - instead of racing them on the current stream.
        torch.cuda.synchronize(device)
        ent[0][ent[1]:ent[1] + half].zero_()
        torch.cuda.current_stream(device).synchronize()
    return ent[0][ent[1]:ent[1] + 1]


_SIDE_STREAMS = {}


def _side_streams(device, n=4):
    """A small per-device pool of HIP streams for building the independent CSR structures of a sample side by side
    (each build is a chain of ~13 short, latency-bound kernels)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


class CSR:
    __slots__ = ('rowptr', 'col', 'perm', 'inv_deg', 'n_rows', 'n_entries')

    def __init__(self, rowptr, col, perm, inv_deg, n_rows, n_entries):
        self.rowptr, self.col, self.perm, self.inv_deg = rowptr, col, perm, inv_deg
        self.n_rows, self.n_entries = n_rows, n_entries


_JOB = struct.Struct('<2Q3q3QQ3QQQ2i')     # stin_plan_job_t (include/stin_hip.h): 120 bytes
JOBS_MAX = 16                              # STIN_PLAN_MAX_JOBS


class PlanJobs:
    """CSR builds collected for ONE batched launch sequence (stin_plan_build_many: one memset + four kernels for all of them)."""

    def __init__(self):
        self.blobs, self.total_e, self.total_cnt, self.keep = [], 0, 0, []

    def add(self, a, b, E, N, b_limit, pair, rowptr0, col0, perm0, inv_deg0, rowptr1=None, col1=None, xslot=None, w_src=None,
            narrow_out=None):
        self.blobs.append(_JOB.pack(_ptr(a), _ptr(b), E, N, b_limit, _ptr(rowptr0), _ptr(col0), _ptr(perm0), _ptr(inv_deg0),
                                    _ptr(rowptr1), _ptr(col1), _ptr(xslot), _ptr(w_src), _ptr(narrow_out), int(pair), 0))
        self.total_e += E
        self.total_cnt += (2 if pair else 1) * N + 1
        self.keep.append((a, b))                 # contiguous copies of the index rows must outlive the launches

    def run(self, bad, device):
        """Enqueue the builds on torch's current stream of `device`."""
        if not self.blobs:
            return
        lib = _lib.load()
        for i in range(0, len(self.blobs), JOBS_MAX):
            chunk = self.blobs[i:i + JOBS_MAX]
            ws_bytes = lib.stin_plan_build_workspace_bytes(self.total_e, self.total_cnt)     # (upper bound for a chunk)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            buf = ctypes.create_string_buffer(b''.join(chunk), len(chunk) * _JOB.size)
            _lib.check(lib.stin_plan_build_many(buf, len(chunk), _ptr(bad), _ptr(ws), ws_bytes, _stream(ws)), 'stin_plan_build_many')
        self.blobs, self.keep = [], []


def build_csr(key, val, n_rows, val_limit, bad, want_perm=False, jobs=None):
    """key/val: int64 device tensors [E] (val may be None) -> CSR grouped by key.  jobs: a PlanJobs to add the build to
    (the caller runs it); None = build now."""
    assert key.dtype == torch.int64 and key.is_cuda and key.is_contiguous()
    E = key.numel()
    dev = key.device
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if want_perm else None
    inv_deg = torch.empty(max(n_rows, 1), dtype=torch.float32, device=dev)
    if val is not None:
        assert val.dtype == torch.int64 and val.is_contiguous() and val.numel() == E
    own = jobs is None
    if own:
        jobs = PlanJobs()
    jobs.add(key, val, E, n_rows, val_limit, 0, rowptr, col, perm, inv_deg)
    if own:
        jobs.run(bad, dev)
    return CSR(rowptr, col[:E], None if perm is None else perm[:E], inv_deg[:n_rows], n_rows, E)


class EdgeSet:
    """One directed edge set of one level (edge_index[0] = source j, [1] = target i)."""

    def __init__(self, edge_index, n, bad, jobs=None):
        assert edge_index.dim() == 2 and edge_index.shape[0] == 2, 'edge_index must be [2, E]'
        self.n = n
        src = edge_index[0].contiguous()
        dst = edge_index[1].contiguous()
        assert src.dtype == torch.int64 and src.is_cuda
        E = dst.numel()
        dev = src.device
        e1 = max(E, 1)
        idx = torch.empty(2 * (n + 1) + 3 * e1, dtype=torch.int32, device=dev)   # one allocation for both CSRs
        rp_d, rp_s = idx[:n + 1], idx[n + 1:2 * (n + 1)]
        col_d = idx[2 * (n + 1):2 * (n + 1) + e1]
        col_s = idx[2 * (n + 1) + e1:2 * (n + 1) + 2 * e1]
        xslot = idx[2 * (n + 1) + 2 * e1:]
        inv_deg = torch.empty(max(n, 1) + e1, dtype=torch.float32, device=dev)
        w_src = inv_deg[max(n, 1):]
        own = jobs is None
        if own:
            jobs = PlanJobs()
        # side 0 groups by dst (value = src), side 1 groups by src (value = dst)
        jobs.add(dst, src, E, n, n, 1, rp_d, col_d, None, inv_deg, rp_s, col_s, xslot, w_src)
        if own:
            jobs.run(bad, dev)
        self.by_dst = CSR(rp_d, col_d[:E], None, inv_deg[:n], n, E)     # rows = targets, col = sources
        self.by_src = CSR(rp_s, col_s[:E], None, None, n, E)            # rows = sources, col = targets
        self.inv_deg = self.by_dst.inv_deg                               # 1 / max(1, in-degree)
        self.w_src = w_src[:E]                                           # 1/max(1, indeg(target)) per src-CSR slot
        self.xslot = xslot[:E]                                           # src-CSR slot -> dst-CSR slot of the same edge
        self.n_edges = E
        self._storage = (idx, inv_deg)

    def tensors(self):
        """The allocations behind this edge set (for record_stream when built on a side stream)."""
        return self._storage


class PoolMap:
    """trace: fine vertex -> coarse vertex (hierarchy_trace_index_l)."""

    def __init__(self, trace, n_fine, n_coarse, bad, jobs=None, renumbered=None):
        """renumbered = (coarse_new, fine_new, trace_new): the map after a vertex renumbering (GraphPlan._ensure_order) -
        coarse_new[v] / fine_new[v] = the NEW ids of trace[v] / v for v in ORIGINAL order (so every children list keeps the
        original order of its members: the arg-first rule of max pooling picks the same child as before the renumbering),
        trace_new int32 [n_fine] = new fine id -> new coarse id."""
        assert trace.numel() == n_fine
        trace = trace.contiguous()
        assert trace.dtype == torch.int64 and trace.is_cuda
        dev = trace.device
        self.n_fine, self.n_coarse = n_fine, n_coarse
        rowptr = torch.empty(n_coarse + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(n_fine, 1), dtype=torch.int32, device=dev)
        inv_deg = torch.empty(max(n_coarse, 1), dtype=torch.float32, device=dev)
        own = jobs is None
        if own:
            jobs = PlanJobs()
        if renumbered is not None:
            coarse_new, fine_new, self.trace = renumbered
            jobs.add(coarse_new, fine_new, n_fine, n_coarse, n_fine, 0, rowptr, col, None, inv_deg)
            if own:
                jobs.run(bad, dev)
            self.children = CSR(rowptr, col[:n_fine], None, inv_deg[:n_coarse], n_coarse, n_fine)
            self.inv_count = self.children.inv_deg
            return
        self.trace = torch.empty(max(n_fine, 1), dtype=torch.int32, device=dev)[:n_fine]
        # children CSR (col = fine ids, ascending) and the int32 trace in the same counting pass
        jobs.add(trace, None, n_fine, n_coarse, 0, 0, rowptr, col, None, inv_deg, narrow_out=self.trace)
        if own:
            jobs.run(bad, dev)
        self.children = CSR(rowptr, col[:n_fine], None, inv_deg[:n_coarse], n_coarse, n_fine)
        self.inv_count = self.children.inv_deg

    def tensors(self):
        c = self.children
        return tuple(t for t in (c.rowptr, c.col, c.perm, c.inv_deg, self.trace) if t is not None)


_INV_COUNT = collections.OrderedDict()


def _inv_count_const(device, n_rows):
    """The 1-element device tensor 1 / n_rows of a single-graph norm, from a small per-device LRU: a plan is built for every
    sample, its levels usually repeat the sizes of earlier ones, and torch.full was one framework fill kernel per level and step
    (4 of the 7 fills of a headline step).  Read-only for every consumer."""
    if device.type == 'cuda' and _capturing():             # (a capture's allocations belong to the graph's private pool)
        return torch.full((1,), 1.0 / max(int(n_rows), 1), dtype=torch.float32, device=device)
    key = (device.type, device.index, int(n_rows))
    t = _INV_COUNT.get(key)
    if t is None:
        t = torch.full((1,), 1.0 / max(int(n_rows), 1), dtype=torch.float32, device=device)
        _INV_COUNT[key] = t
        while len(_INV_COUNT) > 256:
            _INV_COUNT.popitem(last=False)
    else:
        _INV_COUNT.move_to_end(key)
    return t


class NormGroups:
    """Row groups for the per-graph instance norm at one level.

    Single graph (or the whole-batch norm of the io blocks): one range,
    ids None.  Batched: ``gid`` = graph id per row (centring/scaling, true counts),
    ``ptr``/``sid`` = the row ranges the SUMS run over - the reference's
    ``linspace(0, N, B+1)`` slices when ``linspace_quirk`` (fastinstancenorm.py:53-82,
    SURVEY Q2) else the true per-graph ranges."""

    def __init__(self, n_rows, device, batch=None, counts=None, linspace_quirk=True):
        self.n_rows = n_rows
        self.device = device
        if batch is None:
            self.B = 1
            self.gid = self.sid = self.ptr_sum = self.ptr_true = None
            self.inv_cnt = _inv_count_const(device, n_rows)
            self.quirk = False
            return
        B = int(counts.numel())
        self.B = B
        true_ptr = torch.zeros(B + 1, dtype=torch.int64)
        true_ptr[1:] = torch.cumsum(counts.to(torch.int64), 0)
        if linspace_quirk:
            sum_ptr = torch.linspace(0, n_rows, B + 1, dtype=torch.int).to(torch.int64)
        else:
            sum_ptr = true_ptr
        self.quirk = bool((sum_ptr != true_ptr).any())
        # one pinned, non-blocking upload for both pointer arrays (a pageable .to(device) would stall the host until the
        # GPU has drained everything queued before it - once per level and step)
        ptrs = _upload(torch.stack([true_ptr, sum_ptr]).to(torch.int32), device)
        self.ptr_true, self.ptr_sum = ptrs[0], ptrs[1]
        if torch.device(device).type == 'cuda' and batch.dtype == torch.int64 and B <= 8192:
            # graph id and slice id (= number of slice boundaries <= row) per row in ONE launch (stin_norm_group_ids_i64; the
            # framework route below is five launches - cast, arange, cast, searchsorted, cast - per level and step)
            batch = batch.contiguous()
            self.gid = torch.empty(n_rows, dtype=torch.int32, device=device)
            self.sid = torch.empty(n_rows, dtype=torch.int32, device=device) if self.quirk else self.gid
            _lib.check(_lib.load().stin_norm_group_ids_i64(_ptr(batch), _ptr(self.ptr_sum), B, n_rows, _ptr(self.gid),
                                                          _ptr(self.sid) if self.quirk else None, _stream(batch)),
                       'stin_norm_group_ids_i64')
        else:
            self.gid = batch.to(torch.int32).contiguous()
            if self.quirk:      # slice id per row = number of slice boundaries <= row (on the device: no host-side expansion)
                rows = torch.arange(n_rows, device=device, dtype=torch.int64)
                self.sid = torch.searchsorted(self.ptr_sum[1:].to(torch.int64), rows, right=True).to(torch.int32)
            else:
                self.sid = self.gid
        self.inv_cnt = _upload(1.0 / counts.to(torch.float32).clamp(min=1), device)


class GraphPlan:
    """All CSR structures of one (possibly batched) hierarchical sample."""

    def __init__(self, sample, linspace_quirk=True, validate=True, validation=None, positions=None, reorder=None):
        """reorder: None = the module default (REORDER, off); True / False = renumber the vertices by locality inside THIS plan
        (see _ensure_order) - what a caller that builds a plan ONCE and keeps it resident asks for (loader.SceneLoader's graph
        cache: the renumbering is paid once per scene, every later step gathers L2-resident neighbour rows)."""
        x = sample.x
        assert x.is_cuda, 'the HIP path needs the sample on the GPU (sample.to("cuda"))'
        self.device = x.device
        self.linspace_quirk = linspace_quirk
        nv = getattr(sample, '_nv_host', None)            # host copy kept by HierarchicalBatch.to(): no device sync
        if nv is None:
            nv = sample.num_vertices                      # foreign sample types: one tiny D2H copy (a host sync) per plan
        nv = nv.detach().to('cpu', torch.int64)           # [B, L]
        if nv.dim() == 1:
            nv = nv.view(1, -1)
        self.num_vertices = nv
        self.level_sizes = [int(v) for v in nv.sum(dim=0)]
        self.num_graphs = int(nv.shape[0])
        self._sample = sample
        # out-of-range flag, zeroed on a side stream so that side-stream builds (prefetch) never have to wait for the
        # compute still queued on the main stream; the main stream joins it here (one event wait)
        # out-of-range flag: one word of a per-device pool that is zeroed ONCE (round 4: a fresh torch.zeros per plan was a fill
        # kernel + an event pair per step).  Kernels only ever set a flag on an index error, and the error path clears the
        # word again before it raises (_clear_flag), so a word handed out again FLAG_POOL plans later is zero.
        self._bad = _flag_word(self.device)
        self._pending = []
        self._edges = {}
        self._pools = {}
        self._norms = {}
        self._batch = {}
        self._validate = validate
        self._validation = validation or VALIDATION          # 'sync' | 'deferred', see validate()
        self._validated = False
        self._flag_host = None
        self._gen = 0                      # bumped by every build that can set the out-of-range flag; _flag_gen = the one the pending copy saw
        self._flag_gen = -1
        # vertex renumbering by locality ("optional vertex reordering, inverted at the boundary"): see _ensure_order
        self._pos_cols = positions
        self._ranks = None                 # per level: int64 [n_l + 1], old id -> new id (last entry = n_l: the out-of-range sentinel)
        self.order0 = self.rank0 = None    # int32: new -> old / old -> new at level 0, for the gathers at the model's boundary
        self._reorder = bool(positions is not None and (REORDER if reorder is None else reorder) and self.num_graphs == 1
                             and self.level_sizes[0] >= (REORDER_MIN if reorder is None else 2)
                             and x.dim() == 2 and x.shape[1] >= positions[1] and not _capturing())

    # ---- vertex renumbering -------------------------------------------------------------
    def _ensure_order(self):
        """Renumber the vertices of every level so that memory order follows space (round 3).  The gathers of the edge
        kernels read neighbour rows; on a mesh whose vertex numbering has no locality (ScanNet order, the benchmark's random
        permutation) every gathered row comes over the fabric, with a coherent numbering most are L2 hits (level-2 forward
        45 -> 36 us, level 0 118 -> 103 at the headline size).  Level 0: stable sort by the 30-bit Morton code of the vertex
        positions (columns `positions` of x - [6, 9) in the reference's feature layout,
        datasets/scannetcolorgraph_dataloader.py:113-121); level l + 1: by the smallest new id among a vertex's children.
        Only index VALUES change: every edge list and trace keeps its order, so each CSR row keeps the neighbour order of the
        reference's sequential scatter and each children list the order the arg-first max rule needs - per-row results are
        bit-identical, only the row order of the column reductions (instance-norm statistics, fp64 partial sums) differs.
        Single graphs only (the linspace-slice statistics of a batch depend on the row order).  Runs on torch's current stream
        (the plan side stream when prefetched)."""
        if not self._reorder or self._ranks is not None:
            return
        s, dev = self._sample, self.device
        a, b = self._pos_cols
        if REORDER_IMPL != 'torch' and s.x.dtype == torch.float32 and s.x.stride(1) == 1:
            # HIP path (csrc/stin_order.hip): bounding box, Morton keys, one radix sort per level, rank scatter - ~12 launches from
            # one foreign call instead of ~60 framework kernels; same permutation as the torch formulation below (tested)
            lib = _lib.load()
            sizes = self.level_sizes
            ranks = [torch.empty(n + 1, dtype=torch.int32, device=dev) for n in sizes]
            order0 = torch.empty(sizes[0], dtype=torch.int32, device=dev)
            ws_bytes = lib.stin_vertex_order_workspace_bytes(max(sizes))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            st = struct.Struct('<q3Q')
            blob, keep = [], []
            for lvl, n in enumerate(sizes):
                tr = None
                if lvl > 0:
                    tr = s['hierarchy_trace_index_%d' % lvl].contiguous()
                    keep.append(tr)
                blob.append(st.pack(n, _ptr(tr), _ptr(ranks[lvl]), _ptr(order0) if lvl == 0 else 0))
            buf = ctypes.create_string_buffer(b''.join(blob), len(blob) * st.size)
            _lib.check(lib.stin_vertex_order_f32(s.x.data_ptr() + 4 * a, s.x.stride(0), buf, len(sizes), _ptr(ws), ws_bytes, _stream(ws)),
                       'stin_vertex_order_f32')
            self._order_keep = (ws, keep)
            self._ranks = ranks
            self.order0 = order0
            self.rank0 = ranks[0][:-1]
            return
        pos = s.x[:, a:b].to(torch.float32)
        lo, hi = pos.amin(0, keepdim=True), pos.amax(0, keepdim=True)
        q = ((pos - lo) / (hi - lo).clamp_min(1e-20) * 1024.0).to(torch.int64).clamp_(0, 1023)

        def spread(v):                                    # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        order = torch.argsort(code, stable=True)
        n = order.numel()
        rank = torch.empty(n + 1, dtype=torch.int64, device=dev)
        rank[order] = torch.arange(n, device=dev)
        rank[n] = n
        ranks = [rank]
        for lvl in range(1, len(self.level_sizes)):
            nc = self.level_sizes[lvl]
            tr = s['hierarchy_trace_index_%d' % lvl]
            ok = (tr >= 0) & (tr < nc)
            first = torch.full((nc,), n, dtype=torch.int64, device=dev)
            first.scatter_reduce_(0, torch.where(ok, tr, 0), torch.where(ok, ranks[-1][:-1], n), 'amin', include_self=True)
            c_order = torch.argsort(first, stable=True)
            c_rank = torch.empty(nc + 1, dtype=torch.int64, device=dev)
            c_rank[c_order] = torch.arange(nc, device=dev)
            c_rank[nc] = nc
            ranks.append(c_rank)
            n = nc
        self._ranks = ranks
        self.order0 = order.to(torch.int32)
        self.rank0 = ranks[0][:-1].to(torch.int32)

    def _relabel_many(self, items):
        """items: [('e', edge tensor [2, E] int64, level) | ('p', trace [n_fine] int64, level)] -> relabelled tensors in ONE launch
        per 16 arrays (stin_relabel_many_i64): 'e' -> new [2, E] int64; 'p' -> (coarse_new, fine_new, trace_new) as PoolMap wants."""
        lib = _lib.load()
        st = struct.Struct('<QqQqQQQQ')
        blobs, outs, keep = [], [], []
        for kind, t, level in items:
            t = t.contiguous()
            keep.append(t)
            rank = self._ranks[level]
            if kind == 'e':
                out = torch.empty_like(t)
                blobs.append(st.pack(_ptr(t), t.numel(), _ptr(rank), self.level_sizes[level], _ptr(out), 0, 0, 0))
                outs.append(out)
            else:
                nf = t.numel()
                coarse_new = torch.empty(max(nf, 1), dtype=torch.int64, device=self.device)[:nf]
                fine_new = torch.empty(max(nf, 1), dtype=torch.int64, device=self.device)[:nf]
                trace_new = torch.empty(max(nf, 1), dtype=torch.int32, device=self.device)[:nf]
                blobs.append(st.pack(_ptr(t), nf, _ptr(rank), self.level_sizes[level], _ptr(coarse_new), _ptr(self._ranks[level - 1]),
                                     _ptr(fine_new), _ptr(trace_new)))
                outs.append((coarse_new, fine_new, trace_new))
        for i in range(0, len(blobs), 16):
            chunk = blobs[i:i + 16]
            buf = ctypes.create_string_buffer(b''.join(chunk), len(chunk) * st.size)
            _lib.check(lib.stin_relabel_many_i64(buf, len(chunk), _stream(self._bad)), 'stin_relabel_many_i64')
        self._relabel_keep = keep
        return outs

    def _hip_order(self):
        return self._reorder and self._ranks is not None and self._ranks[0].dtype == torch.int32

    def _relabelled(self, idx, level):
        """Index tensor of vertex ids of `level` in the new numbering; out-of-range ids map to n_level, which the CSR build
        flags and leaves out exactly as it does an out-of-range original id."""
        r = self._ranks[level]
        n = r.numel() - 1
        return r[idx.clamp(-1, n)]                        # (-1 wraps to the last entry = the sentinel n)

    def _edge_tensor(self, key, level):
        ei = self._sample.edge_index if key == 'edge_index' else self._sample[key]
        if self._reorder:
            self._ensure_order()
            if self._hip_order():
                return self._relabel_many([('e', ei, level)])[0]
            ei = self._relabelled(ei, level)
        return ei

    def _pool_map(self, level, jobs=None):
        trace = self._sample['hierarchy_trace_index_%d' % level]
        nf, nc = self.level_sizes[level - 1], self.level_sizes[level]
        if not self._reorder:
            return PoolMap(trace, nf, nc, self._bad, jobs)
        self._ensure_order()
        if self._hip_order():
            return PoolMap(trace, nf, nc, self._bad, jobs, renumbered=self._relabel_many([('p', trace, level)])[0])
        coarse_new = self._relabelled(trace.contiguous(), level)          # [nf], original fine order
        fine_new = self._ranks[level - 1][:-1]
        trace_new = torch.empty(max(nf, 1), dtype=torch.int32, device=self.device)[:nf]
        trace_new[fine_new] = torch.where(coarse_new < nc, coarse_new, 0).to(torch.int32)
        return PoolMap(trace, nf, nc, self._bad, jobs, renumbered=(coarse_new, fine_new.contiguous(), trace_new))

    def order_tensors(self):
        """The renumbering's tensors (record_stream when built on a side stream)."""
        if self._ranks is None:
            return ()
        return tuple(self._ranks) + (self.order0, self.rank0)

    def _todo(self, edge_items, pool_levels):
        return ([('e', k, l) for (k, l) in edge_items if k not in self._edges] +
                [('p', l, l) for l in pool_levels if l not in self._pools])

    def _build_batch(self, todo):
        """All listed structures as ONE batch of launches (PlanJobs) on torch's current stream -> the new objects."""
        jobs, made = PlanJobs(), []
        pre = None
        if self._reorder:
            self._ensure_order()
            if self._hip_order():                          # every index array of the batch relabelled in one launch
                items = []
                for kind, key, level in todo:
                    if kind == 'e':
                        items.append(('e', self._sample.edge_index if key == 'edge_index' else self._sample[key], level))
                    else:
                        items.append(('p', self._sample['hierarchy_trace_index_%d' % level], level))
                pre = self._relabel_many(items)
        for i, (kind, key, level) in enumerate(todo):
            if kind == 'e':
                ei = pre[i] if pre is not None else self._edge_tensor(key, level)
                obj = self._edges[key] = EdgeSet(ei, self.level_sizes[level], self._bad, jobs)
            elif pre is not None:
                trace = self._sample['hierarchy_trace_index_%d' % level]
                obj = self._pools[level] = PoolMap(trace, self.level_sizes[level - 1], self.level_sizes[level], self._bad, jobs,
                                                   renumbered=pre[i])
            else:
                obj = self._pools[level] = self._pool_map(level, jobs)
            made.append(obj)
        jobs.run(self._bad, self.device)
        self._validated = False
        self._gen += 1
        return made

    def ensure(self, edge_items=(), pool_levels=()):
        """Build whatever of the listed structures is still missing, as one batch on the current stream (forward() calls
        this once instead of building each piece at its first use: five launches for a whole sample)."""
        if self._pending:
            self.join()
        todo = self._todo(edge_items, pool_levels)
        if todo:
            self._build_batch(todo)
        return self

    def prefetch(self, edge_items=(), pool_levels=(), inputs_ready=False, join=True, after=None):
        """Build the listed edge sets [(key, level), ...] and pool maps [level, ...] NOW as one batch on a side stream
        instead of at first use on the compute stream.  inputs_ready=True asserts that the sample's index tensors are
        already complete in memory (a loader handed over resident tensors): the build then does not wait for work still
        queued on the compute stream and overlaps with it (the previous step's tail); otherwise it starts after
        everything queued so far.  after = an event the index tensors are complete at (a loader's upload stream): the
        build waits for it instead.  join=False leaves the compute stream alone: it waits for the build when the plan is
        first USED (a plan built for the NEXT step while the current one is still being enqueued - TrainStep.prefetch).
        (Rounds 1-2 spread ~72 small launches over four side streams; the batched build is five launches on one.)"""
        todo = self._todo(edge_items, pool_levels)
        if not todo:
            return self
        if _capturing():                                  # captured step: built at first use on the captured stream
            return self
        main = torch.cuda.current_stream(self.device)
        s = _side_streams(self.device)[0]                 # (the stream the out-of-range flag was zeroed on)
        if after is not None:
            s.wait_event(after)
        elif not inputs_ready:
            s.wait_stream(main)
        with torch.cuda.stream(s):
            made = self._build_batch(todo)
            if self._validate and self._validation == 'deferred':
                # (round 4) the flag travels to the host from the BUILD stream: the compute stream sees neither the 4-byte copy nor
                # an event record (10 us of its time per step); validate() only copies again if something was built lazily since
                self._copy_flag()
        for obj in made:
            for t in obj.tensors():
                t.record_stream(main)            # allocated on s, consumed on the compute stream: defer reuse accordingly
        for t in self.order_tensors():
            t.record_stream(main)
        if join:
            main.wait_stream(s)
        else:
            self._pending.append(s.record_event())
        return self

    def join(self):
        """Make the current stream wait for builds started with prefetch(join=False)."""
        if self._pending:
            main = torch.cuda.current_stream(self.device)
            for ev in self._pending:
                main.wait_event(ev)
            self._pending = []

    # ---- lazily built pieces ------------------------------------------------------
    def edges(self, key, level):
        if self._pending:
            self.join()
        if key not in self._edges:
            self._edges[key] = EdgeSet(self._edge_tensor(key, level), self.level_sizes[level], self._bad)
            self._validated = False
            self._gen += 1
        return self._edges[key]

    def edges_from_tensor(self, edge_index, n):
        key = ('tensor', edge_index.data_ptr(), tuple(edge_index.shape), n)
        if key not in self._edges:
            self._edges[key] = EdgeSet(edge_index, n, self._bad)
            self._validated = False
            self._gen += 1
        return self._edges[key]

    def pool(self, level):
        """level >= 1: map from level-1 (fine) to level (coarse)."""
        if self._pending:
            self.join()
        if level not in self._pools:
            self._pools[level] = self._pool_map(level)
            self._validated = False
            self._gen += 1
        return self._pools[level]

    def batch_vector(self, level):
        """int64 graph id per vertex of `level` (None for a single graph), propagated exactly
        like the reference: scatter_max(batch, trace) going down
        (models/surfacetextureinpaintingnet.py:421-422)."""
        if self.num_graphs == 1:
            return None
        if level not in self._batch:
            if level == 0:
                self._batch[0] = self._sample.batch.contiguous()
            else:
                from . import functional as F_
                self._batch[level] = F_.batch_pool(self.batch_vector(level - 1), self.pool(level))
        return self._batch[level]

    def norm_groups(self, level, whole_batch=False):
        key = (level, whole_batch)
        if key not in self._norms:
            n = self.level_sizes[level]
            if self.num_graphs == 1 or whole_batch:
                self._norms[key] = NormGroups(n, self.device)
            else:
                self._norms[key] = NormGroups(n, self.device, self.batch_vector(level), self.num_vertices[:, level],
                                              self.linspace_quirk)
        return self._norms[key]

    def validate(self):
        """Raise like the reference's CPU IndexError when any index was out of range (the kernels leave such pairs out,
        so nothing faulted).  validation = 'sync' (default): one host sync per freshly built plan, the error surfaces
        in the forward call that used the bad sample.  'deferred': the flag is copied to pinned host memory without
        stalling the host; `check_deferred()` - called at the start of the next forward and by TrainStep after the
        optimizer step - raises then, i.e. at the latest one step later, and the host keeps running ahead of the GPU."""
        if not self._validate or self._validated:
            return
        if self._validation == 'deferred':
            if self._flag_host is None or self._flag_gen != self._gen:
                self._copy_flag()
            return
        if int(self._bad.item()) != 0:
            self._bad.zero_()                                  # the pool word goes back clean
            raise IndexError('edge / trace index out of range for the level sizes in sample.num_vertices')
        self._validated = True

    def _copy_flag(self):
        """Deferred validation: the flag word as it stands on the CURRENT stream -> pinned host memory, behind an event.  The flag is
        sticky until it is checked, so a later copy (after a lazy build) supersedes an earlier one."""
        self._flag_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self._flag_host.copy_(self._bad, non_blocking=True)
        self._flag_event = torch.cuda.current_stream(self.device).record_event()
        self._flag_gen = self._gen
        if not any(p is self for p in _PENDING_CHECKS):
            _PENDING_CHECKS.append(self)

    def _check_now(self, wait):
        if self._flag_host is None or self._validated:
            return True
        if not wait and not self._flag_event.query():
            return False
        self._flag_event.synchronize()
        bad = int(self._flag_host[0])
        self._flag_host = None
        if bad != 0:
            self._bad.zero_()                                  # the pool word goes back clean
            raise IndexError('edge / trace index out of range for the level sizes in sample.num_vertices '
                             '(reported by the deferred plan validation: of an earlier forward call, or of a plan that was built ahead)')
        self._validated = self._flag_gen == self._gen          # (a lazy build after the copy is checked by the next copy)
        return True


# 'sync' | 'deferred' (see GraphPlan.validate); STIN_PLAN_VALIDATION overrides the default
import os as _os
VALIDATION = _os.environ.get('STIN_PLAN_VALIDATION', 'sync')
# vertex renumbering by locality inside the plan build (GraphPlan._ensure_order): OFF by default, STIN_REORDER=1 enables it for
# level-0 sizes from STIN_REORDER_MIN up.  Measured (round 3, 200 704 vertices): the edge kernels gain 0.26 ms per step (level-0
# forward 117 -> 96 us, level 1 73 -> 61, level 2 46 -> 41) but the renumbering itself - a stable sort per level, the rank
# scatters and the relabelling of every index array, on the plan stream beside the step - costs more when it is redone for
# EVERY step (framework ops: 7.53 -> 7.67 ms; the HIP implementation csrc/stin_order.hip: 7.69 -> 7.79 ms, 1 M vertices bf16
# 25.9 -> 26.8).  It pays where a plan is built once and reused (a scene cache); the data
# pipeline's cheaper route is to renumber a scene ONCE when it is read (scene_io.load_scene(locality_order=True),
# synthetic.renumber_by_locality) - the bench line's `vertex_locality` figure.
REORDER = _os.environ.get('STIN_REORDER', '0') == '1'
REORDER_MIN = 65536
REORDER_IMPL = 'hip'          # 'hip' (csrc/stin_order.hip) | 'torch' (framework ops, the cross-check)
_PENDING_CHECKS = []


def check_deferred(wait=False):
    """Resolve the deferred index checks whose flag copy has completed (all of them with wait=True)."""
    keep = []
    try:
        while _PENDING_CHECKS:
            p = _PENDING_CHECKS.pop(0)
            if not p._check_now(wait):
                keep.append(p)
    finally:
        _PENDING_CHECKS[:0] = keep


def plan_for(sample, **kw):
    """The cached GraphPlan of `sample` (built on first use)."""
    plan = getattr(sample, '_plan_cache', None)
    if plan is None or plan.device != sample.x.device:
        plan = GraphPlan(sample, **kw)
        try:
            sample._plan_cache = plan
        except Exception:  # foreign sample types without attribute assignment
            pass
    return plan
