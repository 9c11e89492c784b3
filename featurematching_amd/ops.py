"""Functional wrappers over the C ABI: torch tensors in, torch tensors out.

torch is used for device memory and streams only; every computation below happens in
the hand-written HIP kernels of libfmatch_hip.so.  All functions enqueue on the current
torch stream of the inputs' device.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_DTYPES = {torch.float32: _lib.FM_F32, torch.float16: _lib.FM_F16, torch.bfloat16: _lib.FM_BF16}


def _desc(t: torch.Tensor, name: str) -> torch.Tensor:
    """Coarse descriptors go to the kernels in the type they come in (float32, float16 or bfloat16: no up-cast
    pass); anything else is converted to float32."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype not in _DTYPES:
        t = t.float()
    return t.contiguous()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


@dataclass
class CoarseBuffers:
    """Capacity-sized device outputs of the coarse stage plus the device-side count."""
    b_ids: torch.Tensor
    i_ids: torch.Tensor
    j_ids: torch.Tensor
    mkpts0_c: torch.Tensor
    mkpts1_c: torch.Tensor
    mconf: torch.Tensor
    count: torch.Tensor       # int32[2] on device: {M, status bits}
    cap: int
    workspace: torch.Tensor   # kept alive until the stream has consumed it
    conf_matrix: Optional[torch.Tensor] = None   # dense [N,L,S] when requested

    def read_count(self) -> int:
        """The single host sync of the path: returns M (raises on a device-side status)."""
        lib = _lib.load()
        m, info = C.c_int32(0), C.c_int32(0)
        st = lib.fm_read_count_info(_ptr(self.count), self.cap, C.byref(m), C.byref(info), _stream(self.count.device))
        self.info = int(info.value)          # raw FM_DEV_* bits, the informational ones included (FM_DEV_ALL_DENSE)
        if st != _lib.FM_OK:
            err = _lib.FMatchError(st, "fm_coarse_match")
            err.required = int(m.value)
            raise err
        return int(m.value)

    def cell_maps(self):
        """((map0, pitch0, ties0), (map1, pitch1, ties1)): device addresses of the cell -> match-index+1
        maps and tie lists of image 0 and image 1 inside the workspace (valid while this object is alive);
        feed them to gather_windows(cells=...) for the cell-ordered crops."""
        lib = _lib.load()
        if not getattr(self, '_has_cell_maps', True):
            raise RuntimeError("this coarse call ran with cell_maps=False (FM_MODE_NO_CELL_MAPS)")
        n, l, s, c, slots = self._shape
        p0, p1, t0, t1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        q0, q1 = C.c_int(), C.c_int()
        base = self.workspace.data_ptr() + ((-self.workspace.data_ptr()) % 256)
        _lib.check(lib.fm_coarse_cell_maps(C.c_void_p(base), n, l, s, c, slots, C.byref(p0), C.byref(q0), C.byref(t0),
                                           C.byref(p1), C.byref(q1), C.byref(t1)), "fm_coarse_cell_maps")
        return (p0.value, q0.value, t0.value), (p1.value, q1.value, t1.value)

    def softmax_stats(self):
        """(nm_r, sum_r, pitch_r, nm_c, sum_c, pitch_c): device addresses of the stabilisers and denominators of every
        row / column inside the workspace (fm_coarse_softmax_stats; valid while this object is alive).  Needs
        stats=True or conf_matrix=True."""
        if not getattr(self, '_has_stats', False):
            raise RuntimeError("this coarse call ran without stats=True / conf_matrix=True: no softmax statistics")
        lib = _lib.load()
        n, l, s, c, slots = self._shape
        nr, sr, nc, sc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        qr, qc = C.c_int(), C.c_int()
        base = self.workspace.data_ptr() + ((-self.workspace.data_ptr()) % 256)
        _lib.check(lib.fm_coarse_softmax_stats(C.c_void_p(base), n, l, s, c, slots, C.byref(nr), C.byref(sr), C.byref(qr),
                                               C.byref(nc), C.byref(sc), C.byref(qc)), "fm_coarse_softmax_stats")
        return C.c_void_p(nr.value), C.c_void_p(sr.value), qr.value, C.c_void_p(nc.value), C.c_void_p(sc.value), qc.value

    def sliced(self, m: int) -> dict:
        return dict(b_ids=self.b_ids[:m], i_ids=self.i_ids[:m], j_ids=self.j_ids[:m],
                    mkpts0_c=self.mkpts0_c[:m], mkpts1_c=self.mkpts1_c[:m], mconf=self.mconf[:m])


_WS_BYTES = {}       # (N, L, S, C, slots, mode, conf) -> fm_coarse_workspace_bytes_mode


def coarse_match_async(feat_c0: torch.Tensor, feat_c1: torch.Tensor, hw0_c, hw1_c, scale_px: float,
                       thr: float = 0.2, border_rm: int = 2, temperature: float = 0.1,
                       scale0: Optional[torch.Tensor] = None, scale1: Optional[torch.Tensor] = None,
                       cap: Optional[int] = None, cand_slots: Optional[int] = None,
                       conf_matrix: bool = False, exact_screening: bool = False, dense: bool = False,
                       cell_maps: bool = True, exact_step: bool = False, stats: bool = False,
                       flat: bool = False, side_map: Optional[torch.Tensor] = None,
                       side_scratch: Optional[torch.Tensor] = None, alone: bool = False) -> CoarseBuffers:
    """Enqueue the coarse stage (coarse_matching_new.py:43-143, eval) and return the
    capacity-sized device buffers without synchronising.  feat_c0 / feat_c1 may be float32, float16 or bfloat16
    (fm_coarse_match_dtype: half-precision values are exact in float32, so the result equals the float32 call on
    the up-cast tensors).  The common path is four launches (prep, int8 max pass, screening kernel, assignment);
    `dense` (FM_MODE_DENSE) adds the float16 planes and the dense sum kernel for samples with flat similarity (without
    it such samples report FM_E_DENSE through read_count), `exact_screening` (FM_MODE_EXACT_SCREENING) the two kernels
    that re-screen the candidates with exact softmax denominators (without it rows / columns that overflow their
    candidate slots report FM_E_CANDIDATES).  cell_maps=False (FM_MODE_NO_CELL_MAPS) skips the cell -> match maps the
    cell-ordered window crops read (CoarseBuffers.cell_maps() is then meaningless).  `exact_step`
    (FM_MODE_EXACT_STEP) derives the int8 screening step from the images' true maxima (one more small kernel) instead
    of from a sample of rows: the answer to FM_E_STEP (an outlier descriptor outside the sample).  `stats`
    (FM_MODE_STATS) leaves the log-softmax offsets of every row and column in the workspace
    (CoarseBuffers.softmax_stats(): what dual_softmax_at and its backward read).  `flat` (FM_MODE_FLAT, implies
    dense) is the hint that every sample has flat similarity: the screening sweep is skipped, the planes come out of the
    prep kernel and all samples go to the dense sum kernel (two launches fewer); the result does not depend on it.
    `side_map` (fm_coarse_match_maps): image 1's NCHW float32 fine map [N, 64, Hf, Wf] - its channels-last copy, which
    fine_match_maps would make as its first launch, rides in the assignment kernel's launch instead and lands in
    `side_scratch` (allocated when not given; CoarseBuffers.side_scratch) - pass that to fine_match_maps(prepared=...).
    `alone` (FM_MODE_ALONE): this call has the GPU to itself - launches that cannot fill the chip take the grid that is
    fastest for the kernel alone instead of the small footprint that leaves room for other streams' kernels."""
    lib = _lib.load()
    f0 = _desc(feat_c0, "feat_c0")
    f1 = _desc(feat_c1, "feat_c1")
    if f1.dtype != f0.dtype:
        f1 = f1.to(f0.dtype)
    n, l, c = f0.shape
    s = f1.shape[1]
    if f1.shape[0] != n or f1.shape[2] != c:
        raise ValueError(f"feat_c0 {tuple(f0.shape)} and feat_c1 {tuple(f1.shape)} disagree")
    dev = f0.device
    if cap is None:
        cap = n * min(l, s)
    if cand_slots is None:
        cand_slots = lib.fm_default_cand_slots(float(thr))
    mode = (_lib.FM_MODE_EXACT_SCREENING if exact_screening else 0) | (_lib.FM_MODE_DENSE if dense else 0) | \
           (0 if cell_maps else _lib.FM_MODE_NO_CELL_MAPS) | (_lib.FM_MODE_EXACT_STEP if exact_step else 0) | \
           (_lib.FM_MODE_STATS if stats else 0) | (_lib.FM_MODE_FLAT if flat else 0) | (_lib.FM_MODE_ALONE if alone else 0)
    wkey = (n, l, s, c, cand_slots, mode, bool(conf_matrix))
    ws_bytes = _WS_BYTES.get(wkey)
    if ws_bytes is None:                       # (a pure function of the shapes: asked once per shape)
        nb = C.c_size_t(0)
        _lib.check(lib.fm_coarse_workspace_bytes_mode(n, l, s, c, cand_slots, mode, int(bool(conf_matrix)), C.byref(nb)),
                   "fm_coarse_workspace_bytes_mode")
        if len(_WS_BYTES) > 256:
            _WS_BYTES.clear()
        ws_bytes = _WS_BYTES[wkey] = int(nb.value)
    nbytes = C.c_size_t(ws_bytes)
    # (one block for the workspace and the outputs, carved into typed views, was tried: the slicing costs a module
    # caller more than the nine allocations it replaces - 5.1 k against 6.4 k pairs/s)
    ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    ws_ptr = C.c_void_p(ws.data_ptr() + off)
    i64 = dict(dtype=torch.int64, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    out = CoarseBuffers(torch.empty(cap, **i64), torch.empty(cap, **i64), torch.empty(cap, **i64),
                        torch.empty(cap, 2, **f32), torch.empty(cap, 2, **f32), torch.empty(cap, **f32),
                        torch.empty(2, dtype=torch.int32, device=dev), cap, ws)
    if conf_matrix:      # data['conf_matrix'] (coarse_matching_new.py:70): one more sweep + N*L*S*4 bytes
        out.conf_matrix = torch.empty(n, l, s, dtype=torch.float32, device=dev)
    sc0 = None if scale0 is None else _f32c(scale0.to(dev), "scale0")
    sc1 = None if scale1 is None else _f32c(scale1.to(dev), "scale1")
    args = (_ptr(f0), _ptr(f1), _DTYPES[f0.dtype], n, l, s, c, int(hw0_c[0]), int(hw0_c[1]), int(hw1_c[0]),
            int(hw1_c[1]), float(temperature), float(thr), int(border_rm), float(scale_px),
            _ptr(sc0), _ptr(sc1), ws_ptr, nbytes.value, cand_slots, mode,
            _ptr(out.b_ids), _ptr(out.i_ids), _ptr(out.j_ids), _ptr(out.mkpts0_c),
            _ptr(out.mkpts1_c), _ptr(out.mconf), cap, _ptr(out.count), _ptr(out.conf_matrix))
    if side_map is not None:
        if not (side_map.is_cuda and side_map.dtype == torch.float32 and side_map.dim() == 4 and side_map.shape[1] == 64
                and side_map.is_contiguous()):
            raise ValueError("side_map: a contiguous NCHW float32 GPU map with 64 channels")
        need = side_map.numel() * 4
        if side_scratch is None or side_scratch.numel() * side_scratch.element_size() < need:
            side_scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        st = lib.fm_coarse_match_maps(*args, _ptr(side_map), int(side_map.shape[0]), 64, int(side_map.shape[2]),
                                      int(side_map.shape[3]), _ptr(side_scratch), _stream(dev))
        out.side_scratch = side_scratch
        out._side_map = side_map
    else:
        st = lib.fm_coarse_match_dtype(*args, _stream(dev))
    _lib.check(st, "fm_coarse_match_dtype")
    out._keep = (f0, f1, sc0, sc1)   # inputs must outlive the enqueued kernels
    out._shape = (n, l, s, c, cand_slots)
    out._has_cell_maps = bool(cell_maps)
    out._has_stats = bool(stats or conf_matrix)
    out._temperature = float(temperature)
    return out


class HintMemory:
    """The `hint_io` word of fm_coarse_match_auto, kept per problem kind.

    fm_coarse_match_auto (C) answers what the data asks for by itself - flat similarity, candidate overflow, a clipped
    int8 step - and leaves the mode that served the call in a hint word; a call that starts from that word does not
    repeat the failing attempts (flat data runs FM_MODE_FLAT from its second call on).  This class only stores the word:
      * keyed by (shapes, thr, temperature); at most `capacity` keys, least recently used dropped first;
      * it DECAYS: every `reprobe`-th call of a remembered key passes no hint; when the common path then serves the
        call the key is forgotten, otherwise the new word replaces the old one;
      * guarded by a lock (module callers may run on several threads / streams);
      * visible: `snapshot()` decodes what is remembered, `clear()` forgets it.
    A hint is never wrong, only possibly slower than the common path.  Callers that pass modes themselves bypass it."""

    def __init__(self, capacity: int = 64, reprobe: int = 64):
        import threading
        from collections import OrderedDict
        self._lock = threading.Lock()
        self._d = OrderedDict()          # key -> [hint word, calls]
        self.capacity, self.reprobe = capacity, reprobe

    def start(self, key):
        """(hint word to begin the call with, probing?)"""
        with self._lock:
            e = self._d.get(key)
            if e is None:
                return 0, False
            self._d.move_to_end(key)
            e[1] += 1
            if e[1] % self.reprobe == 0:
                return 0, True
            return e[0], False

    def finish(self, key, hint: int):
        """store what fm_coarse_match_auto left in hint_io (0 = the common path served the call: forget the key)"""
        hint &= 0xffff                   # (mode bits + slots; the attempts byte is per call)
        with self._lock:
            if hint == 0:
                self._d.pop(key, None)
                return
            e = self._d.get(key)
            if e is None:
                self._d[key] = [hint, 0]
            else:
                e[0] = hint
            self._d.move_to_end(key)
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)

    def clear(self):
        with self._lock:
            self._d.clear()

    @staticmethod
    def decode(hint: int) -> dict:
        return {'dense': bool(hint & (_lib.FM_MODE_DENSE | _lib.FM_MODE_FLAT)), 'exact': bool(hint & _lib.FM_MODE_EXACT_SCREENING),
                'step': bool(hint & _lib.FM_MODE_EXACT_STEP), 'flat': bool(hint & _lib.FM_MODE_FLAT),
                'slots': (hint >> 8) & 0xff, 'wide': ((hint >> 8) & 0xff) >= 16, 'attempts': (hint >> 24) & 0xff}

    def snapshot(self):
        with self._lock:
            return {k: dict(self.decode(v[0]), calls=v[1], hint=v[0]) for k, v in self._d.items()}


def hint_key(shape0, shape1, thr=0.2, temperature=0.1, dtype=torch.float32, conf_matrix=False, stats=False):
    """The problem kind MODE_MEMORY keys a hint word by: shapes, thr, temperature, descriptor dtype, and whether the call
    asks for the conf_matrix / the softmax statistics (such a call starts at 16 slots and runs the denominator
    reduction whatever the data: its word must not follow plain inference calls of the same shape, nor the reverse)."""
    return (tuple(shape0), tuple(shape1), float(thr), float(temperature), dtype, bool(conf_matrix), bool(stats))


MODE_MEMORY = HintMemory()
# how often a correct-but-slow route served a call in this process (bench.py prints them in `extra.slow_paths`, so a
# silent detour shows in the record): the torch formula of the conf_matrix backward (three [N,L,S] temporaries)
SLOW_PATHS = {"conf_matrix_grad_torch_formula": 0}
_AUTO_WS_BYTES = {}


def coarse_match(feat_c0, feat_c1, hw0_c, hw1_c, scale_px, thr=0.2, border_rm=2, temperature=0.1,
                 scale0=None, scale1=None, conf_matrix: bool = False, exact_screening: Optional[bool] = None,
                 dense: Optional[bool] = None, exact_step: Optional[bool] = None, stats: bool = False,
                 flat: Optional[bool] = None, cell_maps: bool = True, alone: bool = True) -> dict:
    """Synchronous form: sliced outputs.  A thin caller of fm_coarse_match_auto - the ONE C entry point that serves any
    data (flat similarity, candidate overflow, a clipped int8 step and the assignment's bounded wait are answered inside
    it, behind the host sync the reference's torch.where has at coarse_matching_new.py:109).  What is left here: the
    allocations, FM_E_CAPACITY (exact ties can exceed N*min(L,S): larger output buffers, once more) and the optional
    hint word per problem kind (MODE_MEMORY: flat data starts its second call where the first ended).
    exact_screening / dense / exact_step / flat: None = left to the library (and the hint); True = the mode to START
    with (a caller that knows its data).  conf_matrix / stats as in coarse_match_async.  `alone` (FM_MODE_ALONE, default
    on HERE): the synchronous form waits for its result on the host - the reference's single-pair caller,
    demo/demo.py:95-116 - so its launches take the grids that are fastest for a kernel alone on the device; a process
    that runs several such calls side by side (threads, streams) passes alone=False."""
    lib = _lib.load()
    f0 = _desc(feat_c0, "feat_c0")
    f1 = _desc(feat_c1, "feat_c1")
    if f1.dtype != f0.dtype:
        f1 = f1.to(f0.dtype)
    n, l, c = f0.shape
    s = f1.shape[1]
    if f1.shape[0] != n or f1.shape[2] != c:
        raise ValueError(f"feat_c0 {tuple(f0.shape)} and feat_c1 {tuple(f1.shape)} disagree")
    dev = f0.device
    key = hint_key(feat_c0.shape, feat_c1.shape, thr, temperature, f0.dtype, conf_matrix, stats)
    managed = exact_screening is None and dense is None and exact_step is None and flat is None
    mode = (_lib.FM_MODE_EXACT_SCREENING if exact_screening else 0) | (_lib.FM_MODE_DENSE if dense else 0) | \
           (_lib.FM_MODE_EXACT_STEP if exact_step else 0) | (_lib.FM_MODE_FLAT if flat else 0) | \
           (0 if cell_maps else _lib.FM_MODE_NO_CELL_MAPS) | (_lib.FM_MODE_STATS if stats else 0) | \
           (_lib.FM_MODE_ALONE if alone else 0)
    hint0, probing = MODE_MEMORY.start(key) if managed else (0, False)
    sc0 = None if scale0 is None else _f32c(scale0.to(dev), "scale0")
    sc1 = None if scale1 is None else _f32c(scale1.to(dev), "scale1")
    i64 = dict(dtype=torch.int64, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    cap = n * min(l, s)
    max_slots = 16                      # (64 only when 16 slots and the exact re-screening still overflow: thr < 1/16)
    conf = torch.empty(n, l, s, **f32) if conf_matrix else None
    for _ in range(4):
        wkey = (n, l, s, c, max_slots)
        ws_bytes = _AUTO_WS_BYTES.get(wkey)
        if ws_bytes is None:
            nb = C.c_size_t(0)
            _lib.check(lib.fm_coarse_workspace_bytes_auto(n, l, s, c, max_slots, C.byref(nb)), "fm_coarse_workspace_bytes_auto")
            if len(_AUTO_WS_BYTES) > 256:
                _AUTO_WS_BYTES.clear()
            ws_bytes = _AUTO_WS_BYTES[wkey] = int(nb.value)
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        out = CoarseBuffers(torch.empty(cap, **i64), torch.empty(cap, **i64), torch.empty(cap, **i64),
                            torch.empty(cap, 2, **f32), torch.empty(cap, 2, **f32), torch.empty(cap, **f32),
                            torch.empty(2, dtype=torch.int32, device=dev), cap, ws)
        out.conf_matrix = conf
        hint, m, info = C.c_int32(hint0), C.c_int32(0), C.c_int32(0)
        st = lib.fm_coarse_match_auto(_ptr(f0), _ptr(f1), _DTYPES[f0.dtype], n, l, s, c, int(hw0_c[0]), int(hw0_c[1]),
                                      int(hw1_c[0]), int(hw1_c[1]), float(temperature), float(thr), int(border_rm),
                                      float(scale_px), _ptr(sc0), _ptr(sc1), C.c_void_p(ws.data_ptr() + off), ws_bytes,
                                      max_slots, mode, _ptr(out.b_ids), _ptr(out.i_ids), _ptr(out.j_ids),
                                      _ptr(out.mkpts0_c), _ptr(out.mkpts1_c), _ptr(out.mconf), cap, _ptr(out.count),
                                      _ptr(conf), C.byref(hint), C.byref(m), C.byref(info), _stream(dev))
        if st == _lib.FM_E_CAPACITY:
            cap, hint0 = int(m.value), int(hint.value) & 0xffff
            continue
        if st == _lib.FM_E_CANDIDATES and max_slots < 64:
            max_slots, hint0 = 64, int(hint.value) & 0xffff
            continue
        _lib.check(st, "fm_coarse_match_auto")
        if managed:
            MODE_MEMORY.finish(key, int(hint.value))
        h = int(hint.value)
        slots_used = (h >> 16) & 0xff          # the slot count the serving attempt ran with (always written)
        out.info, out.hint, out.attempts = int(info.value), h & 0xffff, (h >> 24) & 0xff
        out._keep = (f0, f1, sc0, sc1)
        out._shape = (n, l, s, c, slots_used)
        out._has_cell_maps = bool(cell_maps)
        out._has_stats = bool(stats or conf_matrix)
        out._temperature = float(temperature)
        res = out.sliced(int(m.value))
        if conf_matrix:
            res['conf_matrix'] = conf
        res['_coarse_buffers'] = out          # keeps the workspace (and its cell maps) alive
        return res
    raise RuntimeError("coarse_match: overflow persisted after retries")


def _dsm_backward(f0, f1, temperature, buffers, b_ids, i_ids, j_ids, gc):
    """(dL/df0, dL/df1) of sum_e g_e conf_e from gc = g * conf at the entries (fm_dual_softmax_backward): two launches of
    one tiled kernel that recomputes the similarities tile by tile - no [N, L, S] array."""
    lib = _lib.load()
    x0, x1 = _f32c(f0, "feat_c0"), _f32c(f1, "feat_c1")
    n, l, c = x0.shape
    s = x1.shape[1]
    stats = buffers.softmax_stats()
    need = int(lib.fm_dual_softmax_backward_workspace_bytes(n, l, s, c))
    ws = torch.empty(need + 256, dtype=torch.uint8, device=x0.device)
    off = (-ws.data_ptr()) % 256
    d0, d1 = torch.empty_like(x0), torch.empty_like(x1)
    k = int(b_ids.shape[0])
    b64 = lambda t: t.to(torch.int64).contiguous()
    bb, ii, jj, g = b64(b_ids), b64(i_ids), b64(j_ids), _f32c(gc, "gc")
    _lib.check(lib.fm_dual_softmax_backward(_ptr(x0), _ptr(x1), n, l, s, c, float(temperature), *stats,
                                            _ptr(bb), _ptr(ii), _ptr(jj), _ptr(g), k,
                                            C.c_void_p(ws.data_ptr() + off), need, _ptr(d0), _ptr(d1), _stream(x0.device)),
               "fm_dual_softmax_backward")
    d0._keep = (ws, bb, ii, jj, g, buffers)
    return d0.to(f0.dtype), d1.to(f1.dtype)


class _DualSoftmaxAt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_c0, feat_c1, b_ids, i_ids, j_ids, temperature, buffers):
        lib = _lib.load()
        x0, x1 = _f32c(feat_c0, "feat_c0"), _f32c(feat_c1, "feat_c1")
        n, l, c = x0.shape
        s = x1.shape[1]
        stats = buffers.softmax_stats()
        k = int(b_ids.shape[0])
        b64 = lambda t: t.to(torch.int64).contiguous()
        bb, ii, jj = b64(b_ids), b64(i_ids), b64(j_ids)
        conf = torch.empty(k, dtype=torch.float32, device=x0.device)
        _lib.check(lib.fm_dual_softmax_conf_at(_ptr(x0), _ptr(x1), n, l, s, c, float(temperature), *stats,
                                               _ptr(bb), _ptr(ii), _ptr(jj), k, _ptr(conf),
                                               _stream(x0.device)), "fm_dual_softmax_conf_at")
        ctx.save_for_backward(feat_c0, feat_c1, bb, ii, jj, conf)
        ctx.temperature, ctx.buffers = float(temperature), buffers
        return conf

    @staticmethod
    def backward(ctx, grad):
        f0, f1, bb, ii, jj, conf = ctx.saved_tensors
        d0, d1 = _dsm_backward(f0, f1, ctx.temperature, ctx.buffers, bb, ii, jj, grad.float() * conf)
        return d0, d1, None, None, None, None, None


def dual_softmax_at(feat_c0: torch.Tensor, feat_c1: torch.Tensor, b_ids, i_ids, j_ids, buffers: CoarseBuffers) -> torch.Tensor:
    """conf_matrix[b_ids, i_ids, j_ids] (coarse_matching_new.py:64-68) as a differentiable function of the descriptors,
    without the matrix: `buffers` = the CoarseBuffers of a coarse call on the same descriptors that ran with stats=True
    (out['_coarse_buffers'] of ops.coarse_match(..., stats=True)).  What the reference's coarse loss with sparse
    supervision reads (losses/loss.py:57-61: conf[pos_mask]); forward and backward allocate O(N (L + S) C)."""
    return _DualSoftmaxAt.apply(feat_c0, feat_c1, b_ids, i_ids, j_ids, buffers._temperature, buffers)


def _dsm_backward_dense(f0, f1, temperature, buffers, grad):
    """(dL/df0, dL/df1) for a DENSE dL/dconf (fm_dual_softmax_backward_dense): three tiled sweeps that recompute conf from
    exact float32 dot products and the forward call's softmax statistics - no [N, L, S] temporary."""
    lib = _lib.load()
    x0, x1 = _f32c(f0, "feat_c0"), _f32c(f1, "feat_c1")
    n, l, c = x0.shape
    s = x1.shape[1]
    g = _f32c(grad, "grad")
    stats = buffers.softmax_stats()
    need = int(lib.fm_dual_softmax_backward_workspace_bytes(n, l, s, c))
    ws = torch.empty(need + 256, dtype=torch.uint8, device=x0.device)
    off = (-ws.data_ptr()) % 256
    d0, d1 = torch.empty_like(x0), torch.empty_like(x1)
    _lib.check(lib.fm_dual_softmax_backward_dense(_ptr(x0), _ptr(x1), n, l, s, c, float(temperature), *stats, _ptr(g),
                                                  C.c_void_p(ws.data_ptr() + off), need, _ptr(d0), _ptr(d1), _stream(x0.device)),
               "fm_dual_softmax_backward_dense")
    d0._keep = (ws, g, buffers)
    return d0.to(f0.dtype), d1.to(f1.dtype)


def _count_nonzero_bounded(t: torch.Tensor, chunk_elems: int = 1 << 20) -> int:
    """number of non-zero entries of a large tensor with temporaries of at most `chunk_elems` bytes (torch.count_nonzero
    materialises a boolean mask of the whole tensor) and one host sync"""
    flat = t.reshape(-1)
    tot = torch.zeros((), dtype=torch.int64, device=t.device)
    for o in range(0, flat.numel(), chunk_elems):
        tot += torch.count_nonzero(flat[o:o + chunk_elems])
    return int(tot)


class _ConfMatrixGrad(torch.autograd.Function):
    """Attaches the gradient of the dual softmax to the conf_matrix the HIP forward produced, so that the
    reference's coarse loss (losses/loss.py:27-67 reads data['conf_matrix']) trains the descriptors.

    conf = A * B with A = softmax(sim, dim 1), B = softmax(sim, dim 2), sim = f0 . f1^T / (C T)
    (coarse_matching_new.py:64-68).  With G = dL/dconf and c = conf:
        dL/dsim = 2 G c - A u - B v,   u_j = sum_i (G c)_ij,   v_i = sum_j (G c)_ij
        dL/df0  = dL/dsim . f1 / (C T),   dL/df1 = dL/dsim^T . f0 / (C T)
    Both forms of G are served by HIP kernels that recompute A and B tile by tile from the softmax statistics of the
    forward call - no [N, L, S] temporary:
      * the reference's default loss reads conf at the supervised entries only (sparse_spvs: loss.py:57-61), so G is zero
        almost everywhere: its non-zero entries go to fm_dual_softmax_backward;
      * a G that is dense (the focal / cross-entropy terms over ALL negatives, loss.py:44-50, 62-65: more than 16 entries
        per row on average) goes to fm_dual_softmax_backward_dense, which streams G through the same tiled sweep.
    Without the forward call's buffers (no statistics) the plain torch formula is all that is left."""

    @staticmethod
    def forward(ctx, feat_c0, feat_c1, conf, temperature, buffers):
        ctx.save_for_backward(feat_c0, feat_c1, conf)
        ctx.temperature, ctx.buffers = float(temperature), buffers
        return conf.view_as(conf)

    @staticmethod
    def backward(ctx, grad):
        f0, f1, conf = ctx.saved_tensors
        n, l, s = conf.shape
        if ctx.buffers is not None:
            if _count_nonzero_bounded(grad) <= 16 * n * max(l, s):
                nz = torch.nonzero(grad, as_tuple=True)
                gc = grad[nz] * conf[nz]
                d0, d1 = _dsm_backward(f0, f1, ctx.temperature, ctx.buffers, nz[0], nz[1], nz[2], gc)
            else:
                d0, d1 = _dsm_backward_dense(f0, f1, ctx.temperature, ctx.buffers, grad)
            return d0, d1, None, None, None
        import warnings
        SLOW_PATHS["conf_matrix_grad_torch_formula"] += 1
        warnings.warn("attach_conf_matrix_grad without the forward call's buffers: the torch formula (three [N,L,S] temporaries)")
        k = 1.0 / (f0.shape[-1] * ctx.temperature)
        sim = torch.bmm(f0.float(), f1.float().transpose(1, 2)) * k
        gc = grad * conf
        a = torch.softmax(sim, dim=1)
        dsim = 2.0 * gc - a * gc.sum(dim=1, keepdim=True)
        del a
        dsim -= torch.softmax(sim, dim=2) * gc.sum(dim=2, keepdim=True)
        del sim, gc
        g0 = torch.bmm(dsim, f1.float()) * k
        g1 = torch.bmm(dsim.transpose(1, 2), f0.float()) * k
        return g0.to(f0.dtype), g1.to(f1.dtype), None, None, None


def attach_conf_matrix_grad(feat_c0: torch.Tensor, feat_c1: torch.Tensor, conf_matrix: torch.Tensor,
                            temperature: float, buffers: Optional[CoarseBuffers] = None) -> torch.Tensor:
    """conf_matrix (from coarse_match(..., conf_matrix=True)) as a differentiable function of the descriptors;
    `buffers` = out['_coarse_buffers'] of that call (its softmax statistics serve the HIP backward)."""
    return _ConfMatrixGrad.apply(feat_c0, feat_c1, conf_matrix, temperature, buffers)


def gather_windows(feat_f: torch.Tensor, b_ids: torch.Tensor, ids: torch.Tensor, w: int, stride: int,
                   w_c: int, pad: int = 2, count: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, cells=None, h_c: Optional[int] = None) -> torch.Tensor:
    """Window crop (fine_preprocess.py:43-50) of the selected coarse cells only.
    feat_f is the logical [N,Cf,Hf,Wf] tensor, stored NCHW-contiguous or channels_last.
    cells = (map address, pitch, tie-list address) of this image (CoarseBuffers.cell_maps())
    selects the cell-ordered kernel when the shape allows it (NCHW, Cf 64, W 5/7): the windows are
    visited in raster order of THIS image's cells, which keeps each XCD's reads inside a band of the map."""
    lib = _lib.load()
    if not feat_f.is_cuda:
        raise RuntimeError("feat_f must live on the GPU: the HIP path has no CPU fallback")
    n, cf, hf, wf = feat_f.shape
    if feat_f.dtype not in _DTYPES:             # float16 / bfloat16 maps go to the kernels as they are (no up-cast pass)
        feat_f = feat_f.float()
    if feat_f.is_contiguous():
        layout = 0
    elif feat_f.is_contiguous(memory_format=torch.channels_last):
        layout = 1
    else:
        feat_f, layout = feat_f.contiguous(), 0
    m_max = int(b_ids.shape[0])
    if out is None:
        out = torch.empty(m_max, w * w, cf, dtype=torch.float32, device=feat_f.device)
    if m_max == 0:
        return out
    if feat_f.dtype != torch.float32:
        st = lib.fm_gather_windows_dtype(_ptr(feat_f), _DTYPES[feat_f.dtype], n, cf, hf, wf, layout, w, stride, pad, w_c,
                                         _ptr(b_ids), _ptr(ids), _ptr(count), m_max, _ptr(out), _stream(feat_f.device))
        _lib.check(st, "fm_gather_windows_dtype")
        return out
    if cells is not None and layout == 0 and cf == 64 and w in (5, 7) and h_c:
        st = lib.fm_gather_windows_cells(_ptr(feat_f), n, cf, hf, wf, w, stride, pad, int(h_c), int(w_c),
                                         C.c_void_p(cells[0]), int(cells[1]), C.c_void_p(cells[2]), _ptr(b_ids), _ptr(ids),
                                         _ptr(count), m_max, _ptr(out), _stream(feat_f.device))
        if st != -3:                       # FM_E_UNSUPPORTED: fall through to the per-window kernel
            _lib.check(st, "fm_gather_windows_cells")
            return out
    st = lib.fm_gather_windows(_ptr(feat_f), n, cf, hf, wf, layout, w, stride, pad, w_c, _ptr(b_ids), _ptr(ids),
                               _ptr(count), m_max, _ptr(out), _stream(feat_f.device))
    _lib.check(st, "fm_gather_windows")
    return out


def pack_merge_weights(merge_w: torch.Tensor) -> torch.Tensor:
    """merge_feat.weight [64, 128] -> the 16 KiB MFMA fragment buffer fm_gather_merge_windows reads."""
    lib = _lib.load()
    w = _f32c(merge_w, "merge_w")
    if tuple(w.shape) != (64, 128):
        raise ValueError(f"merge_feat.weight must be [64, 128], got {tuple(w.shape)}")
    packed = torch.empty(16384, dtype=torch.uint8, device=w.device)
    _lib.check(lib.fm_merge_pack_weights(_ptr(w), 64, _ptr(packed), _stream(w.device)), "fm_merge_pack_weights")
    return packed


def gather_merge_windows(feat_f: torch.Tensor, packed_w: torch.Tensor, ctx_bias: torch.Tensor, b_ids: torch.Tensor,
                         ids: torch.Tensor, w: int, stride: int, h_c: int, w_c: int, pad: int = 2,
                         count: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                         cells=None) -> torch.Tensor:
    """Window crop fused with FinePreprocess's context merge (fine_preprocess.py:43-60): returns
    merge_feat(cat[window, down_proj(feat_c)]) for the selected cells, [M, WW, 64].  ctx_bias [N, h_c*w_c, 64]
    is the position-independent half (W_c . down_proj(feat_c) + bias) per coarse cell."""
    lib = _lib.load()
    if not feat_f.is_cuda:
        raise RuntimeError("feat_f must live on the GPU: the HIP path has no CPU fallback")
    n, cf, hf, wf = feat_f.shape
    feat_f = _f32c(feat_f, "feat_f")
    ctx_bias = _f32c(ctx_bias, "ctx_bias")
    if tuple(ctx_bias.shape) != (n, h_c * w_c, 64):
        raise ValueError(f"ctx_bias must be [{n}, {h_c * w_c}, 64], got {tuple(ctx_bias.shape)}")
    m_max = int(b_ids.shape[0])
    if out is None:
        out = torch.empty(m_max, w * w, cf, dtype=torch.float32, device=feat_f.device)
    if m_max == 0:
        return out
    cm, cp, ct = (C.c_void_p(cells[0]), int(cells[1]), C.c_void_p(cells[2])) if cells is not None else (None, 0, None)
    st = lib.fm_gather_merge_windows(_ptr(feat_f), n, cf, hf, wf, w, stride, pad, int(h_c), int(w_c), cm, cp, ct,
                                     _ptr(packed_w), _ptr(ctx_bias), _ptr(b_ids), _ptr(ids), _ptr(count), m_max,
                                     _ptr(out), _stream(feat_f.device))
    _lib.check(st, "fm_gather_merge_windows")
    return out


def gather_windows_pair(feat_f0: torch.Tensor, feat_f1: torch.Tensor, b_ids, i_ids, j_ids, w: int, stride: int,
                        hw0_c, hw1_c, cells, pad: int = 2, count: Optional[torch.Tensor] = None, out0=None, out1=None,
                        packed_w: Optional[torch.Tensor] = None, ctx0=None, ctx1=None):
    """Both images' window crops in one launch (cell order; cells = CoarseBuffers.cell_maps()); with
    packed_w / ctx0 / ctx1 the crop is fused with the context merge.  NCHW maps, Cf = 64, W in {5,7}."""
    lib = _lib.load()
    f0, f1 = _f32c(feat_f0, "feat_f0"), _f32c(feat_f1, "feat_f1")
    n, cf, hf0, wf0 = f0.shape
    hf1, wf1 = f1.shape[2:]
    m_max = int(b_ids.shape[0])
    if out0 is None:
        out0 = torch.empty(m_max, w * w, cf, dtype=torch.float32, device=f0.device)
    if out1 is None:
        out1 = torch.empty(m_max, w * w, cf, dtype=torch.float32, device=f0.device)
    if m_max == 0:
        return out0, out1
    if packed_w is not None:
        ctx0, ctx1 = _f32c(ctx0, "ctx0"), _f32c(ctx1, "ctx1")
    (m0, p0, t0), (m1, p1, t1) = cells
    st = lib.fm_gather_windows_pair(_ptr(f0), _ptr(f1), n, cf, hf0, wf0, hf1, wf1, w, stride, pad, int(hw0_c[0]),
                                    int(hw0_c[1]), int(hw1_c[0]), int(hw1_c[1]), C.c_void_p(m0), int(p0), C.c_void_p(t0),
                                    C.c_void_p(m1), int(p1), C.c_void_p(t1), _ptr(packed_w), _ptr(ctx0), _ptr(ctx1),
                                    _ptr(b_ids), _ptr(i_ids), _ptr(j_ids), _ptr(count), m_max, _ptr(out0), _ptr(out1),
                                    _stream(f0.device))
    _lib.check(st, "fm_gather_windows_pair")
    return out0, out1


def fine_match(win0: torch.Tensor, win1: torch.Tensor, mix0: torch.Tensor, mix1: torch.Tensor,
               mkpts0_c: torch.Tensor, mkpts1_c: torch.Tensor, scale_f: float,
               count: Optional[torch.Tensor] = None):
    """Fine stage (fine_matching_new.py:50-79).  mix0/mix1 = float32 [WW+1] (weight, bias).
    Returns (mkpts0_f, mkpts1_f), each [M,3] = (x, y, std)."""
    lib = _lib.load()
    win0 = _f32c(win0, "win0")
    win1 = _f32c(win1, "win1")
    m_max, ww, cf = win0.shape
    dev = win0.device
    out0 = torch.empty(m_max, 3, dtype=torch.float32, device=dev)
    out1 = torch.empty(m_max, 3, dtype=torch.float32, device=dev)
    if m_max == 0:
        return out0, out1
    k0 = _f32c(mkpts0_c, "mkpts0_c")
    k1 = _f32c(mkpts1_c, "mkpts1_c")
    st = lib.fm_fine_match(_ptr(win0), _ptr(win1), m_max, _ptr(count), ww, cf, _ptr(_f32c(mix0, "mix0")),
                           _ptr(_f32c(mix1, "mix1")), _ptr(k0), _ptr(k1), float(scale_f), _ptr(out0), _ptr(out1),
                           _stream(dev))
    _lib.check(st, "fm_fine_match")
    return out0, out1


def _map_layout(t: torch.Tensor):
    """(tensor, layout) of a logical [N,Cf,Hf,Wf] fine map: 0 = NCHW-contiguous, 1 = channels-last storage.  float32,
    float16 and bfloat16 maps are taken as they are (fm_fine_match_maps_dtype: no up-cast pass)."""
    if t.dtype not in _DTYPES:
        t = t.float()
    if t.is_contiguous():
        return t, 0
    if t.is_contiguous(memory_format=torch.channels_last):
        return t, 1
    return t.contiguous(), 0


def fine_match_maps(feat_f0: torch.Tensor, feat_f1: torch.Tensor, b_ids, i_ids, j_ids, w: int, stride: int, w0c: int,
                    w1c: int, mix0: torch.Tensor, mix1: torch.Tensor, mkpts0_c: torch.Tensor, mkpts1_c: torch.Tensor,
                    scale_f: float, pad: int = 2, count: Optional[torch.Tensor] = None,
                    scratch: Optional[torch.Tensor] = None, prepared: Optional[torch.Tensor] = None):
    """Window crop + fine stage from the maps in one call (fm_fine_match_maps_dtype; fine_preprocess.py:43-50 with plain
    windows + fine_matching_new.py:50-79): no window tensors.  Channels-last maps are read in place; of NCHW float32
    maps image 1 is first copied to channels-last storage in `scratch` (allocated when not given), of NCHW float16 /
    bfloat16 maps (an autocast backbone, network/net.py:56-57) both images are, in their own element type.
    `prepared`: the scratch buffer a coarse_match_async(side_map=feat_f1) call filled (FM_LAYOUT_NCHW_PREPARED: NCHW
    float32 maps, image 1's copy exists already - no transpose launch here).
    Returns (mkpts0_f, mkpts1_f), float32."""
    lib = _lib.load()
    if not feat_f0.is_cuda:
        raise RuntimeError("feat_f0 must live on the GPU: the HIP path has no CPU fallback")
    f0, lay0 = _map_layout(feat_f0)
    f1, lay1 = _map_layout(feat_f1)
    if f1.dtype != f0.dtype:
        f1 = f1.to(f0.dtype)
    if lay0 != lay1:                                 # one layout per call
        f1, lay1 = (f1.contiguous(), 0) if lay0 == 0 else (f1.contiguous(memory_format=torch.channels_last), 1)
    n, cf, hf0, wf0 = f0.shape
    hf1, wf1 = f1.shape[2:]
    dev = f0.device
    dt = _DTYPES[f0.dtype]
    m_max = int(b_ids.shape[0])
    out0 = torch.empty(m_max, 3, dtype=torch.float32, device=dev)
    out1 = torch.empty(m_max, 3, dtype=torch.float32, device=dev)
    if m_max == 0:
        return out0, out1
    need = int(lib.fm_fine_maps_scratch_bytes_dtype(n, cf, hf0, wf0, hf1, wf1, lay0, dt))
    if prepared is not None:
        if lay0 != 0 or dt != _lib.FM_F32 or prepared.numel() * prepared.element_size() < need:
            raise ValueError("prepared: NCHW float32 maps and the scratch a coarse_match_async(side_map=...) call filled")
        scratch, lay0 = prepared, _lib.FM_LAYOUT_NCHW_PREPARED
    if need and (scratch is None or scratch.numel() * scratch.element_size() < need):
        scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    st = lib.fm_fine_match_maps_dtype(_ptr(f0), _ptr(f1), dt, lay0, n, cf, hf0, wf0, hf1, wf1, w, stride, pad, int(w0c),
                                      int(w1c), _ptr(b_ids), _ptr(i_ids), _ptr(j_ids), _ptr(count), m_max,
                                      _ptr(_f32c(mix0, "mix0")), _ptr(_f32c(mix1, "mix1")), _ptr(_f32c(mkpts0_c, "mkpts0_c")),
                                      _ptr(_f32c(mkpts1_c, "mkpts1_c")), float(scale_f), _ptr(scratch) if need else None,
                                      _ptr(out0), _ptr(out1), _stream(dev))
    _lib.check(st, "fm_fine_match_maps_dtype")
    out0._keep = (f0, f1, scratch)
    return out0, out1


_TF_NAMES = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight", "mlp.2.weight",
             "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def pack_coarse_transformer(state_dict: dict, n_layers: int, device) -> torch.Tensor:
    """state dict of a coarse LocalFeatureTransformer (d_model 256, 8 heads, `n_layers` encoder layers) -> the
    A-operand fragments fm_coarse_transformer reads (fm_coarse_tf_pack_weights)."""
    lib = _lib.load()
    keep, arrs = [], []
    want = [(256, 256)] * 4 + [(512, 512), (256, 512)] + [(256,)] * 4
    for l in range(n_layers):
        ts = [state_dict[f"layers.{l}.{name}"].detach().to(device=device, dtype=torch.float32).contiguous()
              for name in _TF_NAMES]
        if [tuple(t.shape) for t in ts] != want:
            raise ValueError(f"fm_coarse_transformer serves d_model 256 only, got {[tuple(t.shape) for t in ts]}")
        keep.extend(ts)
        arrs.append((C.c_void_p * 10)(*[t.data_ptr() for t in ts]))
    table = (C.POINTER(C.c_void_p) * n_layers)(*[C.cast(a, C.POINTER(C.c_void_p)) for a in arrs])
    nbytes = int(lib.fm_coarse_tf_packed_bytes(n_layers))
    if nbytes == 0:
        raise ValueError(f"fm_coarse_transformer: unsupported layer count {n_layers}")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _lib.check(lib.fm_coarse_tf_pack_weights(C.cast(table, C.c_void_p), n_layers, _ptr(packed), _stream(packed.device)),
               "fm_coarse_tf_pack_weights")
    torch.cuda.current_stream(packed.device).synchronize()       # the sources in `keep` may be freed after this
    return packed


def coarse_transformer(feat0: torch.Tensor, feat1: torch.Tensor, packed: torch.Tensor, layer_names, nhead: int = 8,
                       workspace: Optional[torch.Tensor] = None, mask0: Optional[torch.Tensor] = None,
                       mask1: Optional[torch.Tensor] = None):
    """The coarse context layers (network/net.py:74) on feat0 [N,L,256], feat1 [N,S,256]; layer_names as in the
    reference's config (['self', 'cross', ...]).  mask0 [N,L] / mask1 [N,S]: the reference's optional padding masks
    (transformer.py:78-96; True = a real token), served by the same kernels (fm_coarse_transformer_masked)."""
    lib = _lib.load()
    feat0, feat1 = _f32c(feat0, "feat0"), _f32c(feat1, "feat1")
    n, l, c = feat0.shape
    s = feat1.shape[1]
    if feat1.shape[0] != n or feat1.shape[2] != c:
        raise ValueError(f"feat0 {tuple(feat0.shape)} and feat1 {tuple(feat1.shape)} do not belong together")
    kinds = (C.c_int * len(layer_names))(*[{'self': 0, 'cross': 1}[k] for k in layer_names])
    nbytes = C.c_size_t()
    _lib.check(lib.fm_coarse_tf_workspace_bytes(n, l, s, C.byref(nbytes)), "fm_coarse_tf_workspace_bytes")
    if workspace is None or workspace.numel() < nbytes.value:
        workspace = torch.empty(nbytes.value, dtype=torch.uint8, device=feat0.device)
    out0, out1 = torch.empty_like(feat0), torch.empty_like(feat1)

    def as_bytes(m, length, name):
        if m is None:
            return None
        if tuple(m.shape) != (n, length):
            raise ValueError(f"{name} must be [{n}, {length}], got {tuple(m.shape)}")
        return (m != 0).to(device=feat0.device, dtype=torch.uint8).contiguous()
    m0, m1 = as_bytes(mask0, l, "mask0"), as_bytes(mask1, s, "mask1")
    _lib.check(lib.fm_coarse_transformer_masked(_ptr(feat0), _ptr(feat1), _ptr(m0), _ptr(m1), n, l, s, c, nhead, kinds,
                                                len(layer_names), _ptr(packed), _ptr(workspace), workspace.numel(),
                                                _ptr(out0), _ptr(out1), _stream(feat0.device)), "fm_coarse_transformer_masked")
    out0._keep = (m0, m1)
    return out0, out1


def pack_fine_transformer(state_dict: dict, device) -> torch.Tensor:
    """state dict of a fine LocalFeatureTransformer (layers.0 = 'self', layers.1 = 'cross'; d_model 64, 8 heads) ->
    the packed operand fragments fm_fine_transformer reads (fm_fine_tf_pack_weights)."""
    lib = _lib.load()
    keep, arrs = [], []
    for l in range(2):
        ptrs = []
        for name in _TF_NAMES:
            t = state_dict[f"layers.{l}.{name}"].detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            ptrs.append(t.data_ptr())
        arrs.append((C.c_void_p * 10)(*ptrs))
    shapes = [tuple(t.shape) for t in keep[:10]]
    if shapes != [(64, 64)] * 4 + [(128, 128), (64, 128)] + [(64,)] * 4:
        raise ValueError(f"fm_fine_transformer serves d_model 64 only, got {shapes}")
    packed = torch.empty(int(lib.fm_fine_tf_packed_bytes()), dtype=torch.uint8, device=device)
    _lib.check(lib.fm_fine_tf_pack_weights(arrs[0], arrs[1], _ptr(packed), _stream(packed.device)), "fm_fine_tf_pack_weights")
    torch.cuda.current_stream(packed.device).synchronize()       # the sources in `keep` may be freed after this
    return packed


def fine_transformer(win0: torch.Tensor, win1: torch.Tensor, packed: torch.Tensor, count: Optional[torch.Tensor] = None,
                     status: Optional[torch.Tensor] = None, start_scale: int = 8):
    """The fine context layers (network/net.py:79-80) on the windows [M, WW, 64] of both images, WW in {25, 49}.
    `status` = a zeroed int32 device tensor: the kernel ORs FM_DEV_RANGE into element 0 when a value did not fit its
    float16 operand halves at any of its activation scales (see fmatch.h) - the outputs must then be discarded; with two
    elements, element 1 receives by how much the matches went below `start_scale` (log2 of the first attempt's activation
    scale: 8, 4, 0 or -4; fm_fine_transformer_start)."""
    lib = _lib.load()
    win0, win1 = _f32c(win0, "win0"), _f32c(win1, "win1")
    m, ww, cf = win0.shape
    out0, out1 = torch.empty_like(win0), torch.empty_like(win1)
    if m:
        lowered = None
        if status is not None and status.numel() > 1:
            lowered = C.c_void_p(status.data_ptr() + 4)
        _lib.check(lib.fm_fine_transformer_start(_ptr(win0), _ptr(win1), m, _ptr(count), ww, cf, _ptr(packed), _ptr(out0),
                                                 _ptr(out1), _ptr(status), int(start_scale), lowered, _stream(win0.device)),
                   "fm_fine_transformer")
    return out0, out1
