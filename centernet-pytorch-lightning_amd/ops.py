"""Host-side operators: thin autograd glue over the C ABI (no arithmetic happens in Python/ATen).

Activations are contiguous NHWC tensors in the compute dtype (torch.bfloat16 or torch.float32) whose channel
count is padded to a multiple of 16; parameters, images, head maps and gradients stay NCHW fp32 like the
reference.  Every op enqueues hand-written HIP kernels on the current stream through `_hip.call`.
"""
import torch
from torch.autograd import Function

from . import _hip
from ._hip import Hooks, call, dtype_code

import os as _os

_NO_DGRAD_S2_TILE = bool(_os.environ.get("CN_DISABLE_DGRAD3X3_S2"))

_DCN_UNFUSED = bool(_os.environ.get("CN_DCN_UNFUSED"))
BN_MOMENTUM = 0.1  # msra_resnet.py:11, pose_dla_dcn.py:13
BN_EPS = 1e-5


def rup(x, m):
    return (x + m - 1) // m * m


class WeightsEpoch:
    """Counter of every parameter / BN-buffer update that PyTorch's `_version` cannot see: FlatAdam writes the parameters and
    cn_bn_train_fwd the running statistics through raw device pointers.  The no-grad caches (packed / BN-folded weights in
    nn.Conv2d.infer, nn.DCN.infer, nn.BatchNorm2d.folded) key on it next to `_version`."""
    value = 0

    @classmethod
    def bump(cls):
        cls.value += 1


def zeros(shape, dtype, device):
    """torch.zeros on the device without an ATen launch: torch.empty + cn_zero on the current stream (CPU tensors: torch.zeros)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if not t.is_cuda:
        return t.zero_()
    if t.numel():
        call("cn_zero", t, t.numel() * t.element_size())
    return t


def zeros_like(t, dtype=None):
    return zeros(t.shape, dtype or t.dtype, t.device)


def _empty_like_shape(x, shape):
    return torch.empty(shape, dtype=x.dtype, device=x.device)


def _bn_ws(npix, C, device):
    n = _hip.query("cn_bn_workspace_bytes", int(npix), int(C))
    return _hip.workspace(n, device, "bn"), n


# ------------------------------------------------------------------------------------------------ packing
def _pack_dims(shape, mode):
    A, B, KH, KW = shape
    taps = KH * KW
    if mode == 0:
        rows, inner = B, A
    elif mode == 1:
        rows, inner = A, B
    else:
        rows, inner = taps * B, A
    rows_pad, inner_pad = rup(rows, 32), rup(inner, 16)
    return rows_pad, inner_pad, (inner_pad if mode == 2 else taps * inner_pad)


_FAR_BUFFERS = {}


class FarFlags:
    """The per-call `far_flag` words of the DCN backward (set by the offset / mask gradient kernel when a sample left the tile window,
    read by the data gradient kernel).  Every launch-stream kernel boundary of a replayed step costs ~3.7 us: inside a TrainStep the 16
    one-word zero fills become ONE fill of a persistent 64-word array at the start of the step, each DCN layer takes the next word."""
    SLOTS = 64
    enabled = not _os.environ.get("CN_DISABLE_FAR_FLAG_ARRAY")
    _buf, _cur, _next, active = {}, None, 0, False

    @classmethod
    def begin(cls, device, owner=None):
        """owner: the TrainStep (a captured graph bakes the array's address in: every step object has its own)"""
        if not cls.enabled:
            return
        key = (str(device), owner)
        buf = cls._buf.get(key)
        if buf is None:
            buf = cls._buf[key] = torch.empty(cls.SLOTS, dtype=torch.int32, device=device)
        call("cn_zero", buf, cls.SLOTS * 4)
        cls._cur, cls._next, cls.active = buf, 0, True

    @classmethod
    def end(cls):
        cls.active, cls._cur = False, None

    @classmethod
    def take(cls, device):
        buf = cls._cur
        if not cls.active or buf is None or buf.device != device or cls._next >= cls.SLOTS:
            return zeros((1,), torch.int32, device)
        cls._next += 1
        return buf[cls._next - 1:cls._next]


def _far_buffer(shape, device):
    """persistent all-zero fp32 scratch for DCN far samples (cn_dcn_bwd_dx restores the zeros it consumes)"""
    key = (tuple(shape), str(device))
    buf = _FAR_BUFFERS.get(key)
    if buf is None:
        buf = _FAR_BUFFERS[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return buf


class BnStats:
    """Training-mode conv + BN: the producing kernel accumulates the batch statistics of its output in its epilogue (the `bn_part`
    fields of cn_hooks).  `launch` hands a sink to ONE C-ABI call through its hooks argument and reads back whether the kernel
    took it (`bn_taken`); the op wrappers pick that up (`pop`) and tag the output tensor for `batch_norm_act`.

    Sinks (fp32 [slots][2][C], persistent, all-zero when handed out) are reduced by the kernel that CONSUMES them
    (cn_bn_train_fwd_sink / cn_bn_train_bwd_sink: no finalize launch), and a kernel cannot clear what its own workgroups are still
    reading — so the sinks form a chain on the launch stream: a consumed sink is `retired`, and the NEXT sink-consuming launch
    zeroes it on the way.  `acquire` hands out a sink of the right width that is known to be clean (never a retired or a
    handed-out one); at any time exactly one sink per namespace is dirty between steps (the last one consumed), which a replayed hipGraph
    reproduces, so host state and device state stay in step.  `ns` (set by a TrainStep) keeps the chains of different steps /
    graphs apart."""
    enabled = not _os.environ.get("CN_DISABLE_BN_EPILOGUE_STATS")
    fused = not _os.environ.get("CN_DISABLE_BN_SINK_FINALIZE")      # A/B: finalize launches instead of consumer-side reduction
    slots = int(_os.environ.get("CN_BN_SLOTS", 0))
    last = None
    ns = None
    _rings, _state = {}, {}

    @classmethod
    def _st(cls, device):
        """dirty set + retired chain of (namespace, device): a launch on one device is never asked to clear another device's sink
        (round-3 ADVICE).  A namespace runs on ONE stream per device (a TrainStep's launch stream) — the chain is a stream order."""
        key = (cls.ns, str(device))
        st = cls._state.get(key)
        if st is None:
            st = cls._state[key] = {"dirty": set(), "retired": []}
        return st

    @classmethod
    def acquire(cls, kind, C, device):
        """-> a clean sink for `kind` ('f' forward statistics / 'b' backward sums); marks it dirty"""
        if not cls.slots:
            # consumer-side reduction: every workgroup of the apply pass reads slots * 2 * C floats from L2, so few rows (DLA-34 step:
            # 43.0 / 42.7 / 42.1 / 42.1 / 42.2 ms with 128 / 64 / 32 / 16 / 8 rows; finalize launches + 128 rows: 42.9)
            cls.slots = 32 if cls.fused else int(_hip.query("cn_bn_stats_slots"))
        ring = cls._rings.setdefault((cls.ns, kind, int(C), str(device)), [])
        st = cls._st(device)
        for buf in ring:
            if buf.data_ptr() not in st["dirty"]:
                break
        else:
            buf = torch.zeros((cls.slots, 2, int(C)), dtype=torch.float32, device=device)
            ring.append(buf)
        st["dirty"].add(buf.data_ptr())
        return buf

    @classmethod
    def release(cls, buf):
        """the sink was handed out but nothing wrote to it (or its consumer cleared it itself)"""
        cls._st(buf.device)["dirty"].discard(buf.data_ptr())

    @classmethod
    def retire(cls, buf):
        """buf was just consumed by a *_sink launch -> (sink that launch should clear | None).  Call BEFORE the launch."""
        st = cls._st(buf.device)
        clear = st["retired"].pop(0) if st["retired"] else None
        while st["retired"]:                       # (does not happen in a regular step: at most one sink waits to be cleared)
            extra = st["retired"].pop(0)
            extra.zero_()
            st["dirty"].discard(extra.data_ptr())
        st["retired"].append(buf)
        if clear is not None:
            st["dirty"].discard(clear.data_ptr())
        return clear

    @classmethod
    def launch(cls, want, y, name, *args, hooks=None):
        """`hooks`: the call's other per-call extras (input pre-affine, backward-statistics sink), if any"""
        cls.last = None
        if not (want and cls.enabled and y.dtype == torch.bfloat16):
            return call(name, *args, hooks=hooks)
        part = cls.acquire("f", y.shape[-1], y.device)
        hooks = (hooks or Hooks()).set(bn_part=part, bn_slots=part.shape[0], bn_C=part.shape[2])
        try:
            call(name, *args, hooks=hooks)
        except BaseException:
            cls.release(part)
            raise
        if hooks.bn_taken:                   # (0: the kernel this shape dispatched to has no hook, or the width is not a multiple of 8)
            cls.last = part
        else:
            cls.release(part)

    @classmethod
    def pop(cls, y):
        """tag y with the sink its producer filled (None: no hook in that kernel)"""
        part, cls.last = cls.last, None
        if part is not None:
            y._bn_part = part
        return y

    @classmethod
    def reset(cls):
        """after an aborted step: a producer may have filled a sink nobody consumed"""
        for ring in cls._rings.values():
            for buf in ring:
                buf.zero_()
        cls._state.clear()
        cls.last = None
        BnBwdSinks.entries = []                # (their buffers are in the rings above)


class BnBwdSinks:
    """Backward-statistics sinks filled by the kernel that PRODUCED a gradient tensor (the `bnb_part` fields of cn_hooks), kept next to that very
    tensor (same Python object, same version: the side-channel rule of SparseRows) until the BN backward that receives it picks the
    sink up.  What nobody picked up when the backward pass ends is zeroed and handed back."""
    enabled = not _os.environ.get("CN_DISABLE_BN_BWD_EPILOGUE_STATS")
    entries = []
    _lock = __import__("threading").Lock()

    @classmethod
    def note(cls, t, sink):
        with cls._lock:
            first = not cls.entries
            cls.entries.append((t, t._version, sink))
        if first:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.clear)
            except RuntimeError:              # not inside a backward pass
                pass

    @classmethod
    def take(cls, g):
        with cls._lock:
            for i, (t, v, sink) in enumerate(cls.entries):
                if t is g:
                    cls.entries.pop(i)
                    if g._version == v:
                        return sink
                    sink.zero_()              # the gradient was modified in place after the sums were taken: they are stale
                    BnStats.release(sink)
                    return None
        return None

    @classmethod
    def clear(cls):
        with cls._lock:
            left, cls.entries = cls.entries, []
        for _, _, sink in left:
            sink.zero_()
            BnStats.release(sink)


class GradReady:
    """Deposit notifications for the data-parallel exchange.  Inside a TrainStep most parameter gradients never pass through
    autograd: weight-gradient kernels (side stream) and the BN backward (launch stream) write straight into the flat gradient
    buffer.  Every such site calls `GradReady.note(*params)` right after ENQUEUEING its kernels, on the stream that runs them;
    `engine.GradSync` installs `sink` for the duration of a backward pass and launches a bucket's all-reduce when its last
    parameter has been noted.  A site whose deposit is DEFERRED (side-stream closures run one submit later) first `claim`s its
    parameters from inside the autograd node: autograd's post-accumulate hook fires for a parameter even when the node returned
    None for it (torch 2.10), i.e. before the deferred kernels exist, and must not count as the deposit."""
    sink = None
    claim_sink = None

    @classmethod
    def note(cls, *params):
        if cls.sink is not None:
            for p in params:
                if p is not None:
                    cls.sink(p)

    @classmethod
    def claim(cls, *params):
        if cls.claim_sink is not None:
            for p in params:
                if p is not None:
                    cls.claim_sink(p)


class GradCell:
    """Gradient accumulator of ONE activation tensor that has several consumers (`share`).

    Autograd sums the gradients of a multiply-used tensor with one element-wise add per extra consumer (25 bf16 passes per DLA-34
    step, 3 tensor passes each).  Here the consumers' backward functions talk to each other instead: whoever finishes first `give`s
    its gradient to the cell (free), every later one `take`s what is there and adds it in the EPILOGUE of its own data-gradient
    kernel (the residual input of the conv kernels, cn_bn_train_bwd_acc, cn_maxpool_bwd_acc) before giving the sum back, and
    all of them return None to autograd; `ShareFn.backward`, which autograd runs once every consumer is done, hands the cell's
    content to the producer.  A consumer without an epilogue slot (DCNv2's sampling gradient, a pass-through residual) that finds
    the cell occupied pays one cn_add; consumers that know nothing about cells return their gradient as usual and are summed in
    `ShareFn.backward` — the result is the same sum in every case (bf16 rounding of partial sums as with autograd's own adds)."""
    __slots__ = ("partial", "sparse")
    enabled = not _os.environ.get("CN_DISABLE_GRAD_CELLS")
    adds = 0            # cn_add launches spent on cells (tests / profiling)

    def __init__(self):
        self.partial = None
        self.sparse = []      # closures fn(total) that scatter-add a consumer's few non-zero gradient rows into the finished sum (HeadFn)

    def take(self, like=None):
        """-> the gradient accumulated so far (None if nothing), handing its ownership to the caller; with `like` given only when
        it has exactly that tensor's shape and dtype (an epilogue slot reads it element for element)"""
        g = self.partial
        if g is None or (like is not None and (g.shape != like.shape or g.dtype != like.dtype)):
            return None
        self.partial = None
        return g

    def give(self, g):
        """add g to the cell; -> None (what the consumer returns to autograd for this input)"""
        if g is None:
            return None
        if self.partial is None:
            self.partial = g
        else:
            self.partial = _add_tensors(self.partial, g)
        return None


def _add_tensors(a, b):
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float32):
        return a + b
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    call("cn_add", a, b, out, a.numel(), dtype_code(a.dtype))
    GradCell.adds += 1
    return out


class ShareFn(Function):
    """Identity whose output carries a GradCell (see there); the producer's gradient is the cell's content plus whatever
    cell-unaware consumers returned through autograd."""

    @staticmethod
    def forward(ctx, x, cell):
        ctx.cell = cell
        ctx.meta = (x.shape, x.dtype, x.device)
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        total = ctx.cell.take()
        if g is not None:
            total = g if total is None else _add_tensors(total, g)
        if ctx.cell.sparse:
            fns, ctx.cell.sparse = ctx.cell.sparse, []
            if total is None:
                total = zeros(ctx.meta[0], ctx.meta[1], ctx.meta[2])
            elif total is g:
                total = g.clone()                 # autograd's tensor is not ours to modify
            for fn in fns:
                fn(total)
        return total, None


def cell_of(t):
    return getattr(t, "_grad_cell", None) if t is not None else None


def share(x):
    """Declare that activation x feeds several layers (DLA trees: pose_dla_dcn.py:245-262; IDAUp: :482-488; the task heads:
    heads.py:38-43).  No-op without autograd, or when x is already shared."""
    if not (GradCell.enabled and torch.is_grad_enabled() and x.requires_grad) or cell_of(x) is not None:
        return x
    cell = GradCell()
    y = ShareFn.apply(x, cell)
    y._grad_cell = cell
    for attr in ("_cn_nhwc", "_bn_part"):
        if hasattr(x, attr):
            setattr(y, attr, getattr(x, attr))
    return y


class PackArena:
    """Every weight packing a training step needs, produced by ONE launch at the start of the step.

    A step packs each convolution weight once per use (forward operand, data-gradient operand, the DCN variants): ~160
    launches of a few microseconds each.  The weights only change in the optimizer, so a `TrainStep` owns one arena:
    it records the requests of its first step (`recording`), `build()`s persistent destinations plus the device table,
    and from then on calls `repack()` once per step; `pack_weight` then returns the resident copy.  `PackArena.current`
    is the arena of the TrainStep whose step is running (None otherwise -> direct packs); entries are keyed by
    (storage pointer, shape, mode, dtype)."""
    current = None

    def __init__(self):
        self.recording = False
        self.requests, self.slots, self.table, self.n_blocks, self.dtype, self.block_record = {}, {}, None, 0, None, None

    @staticmethod
    def key(w, mode, dtype):
        return (w.data_ptr(), tuple(w.shape), mode, dtype)

    def build(self):
        self.recording = False
        reqs = list(self.requests.items())
        self.requests = {}
        if not reqs:
            return
        dts = {k[3] for k, _ in reqs}
        if len(dts) != 1:                     # mixed compute dtypes: keep the per-layer packs
            return
        self.dtype = dts.pop()
        rows, blk = [], 0
        for k, w in reqs:
            mode = k[2]
            A, B, KH, KW = w.shape
            taps = KH * KW
            tile = int(_hip.query("cn_pack_weight_tile", taps, mode))         # AA | BB << 8 (see pack_weight_batch_kernel)
            if not tile:
                continue                          # (too many taps for a tile: that weight keeps its own cn_pack_weight launch)
            AA, BB = tile & 255, tile >> 8
            rows_pad, inner_pad, cols = _pack_dims(w.shape, mode)
            wp = torch.empty((rows_pad, cols), dtype=self.dtype, device=w.device)
            self.slots[k] = wp
            a_range, b_range = (rows_pad, inner_pad) if mode == 1 else ((inner_pad, rows_pad) if mode == 0 else (inner_pad, B))
            tiles_b = -(-b_range // BB)
            rows.append([w.data_ptr(), wp.data_ptr(), A, B, taps, mode, rows_pad, inner_pad, blk, tile | (tiles_b << 16)])
            blk += -(-a_range // AA) * tiles_b + (1 if (mode == 2 and rows_pad > taps * B) else 0)
        if not rows:
            return
        first = torch.tensor([r[8] for r in rows] + [blk], dtype=torch.int64)
        self.block_record = torch.repeat_interleave(torch.arange(len(rows), dtype=torch.int32), first[1:] - first[:-1]).to(reqs[0][1].device)
        self.table = torch.tensor(rows, dtype=torch.int64).to(reqs[0][1].device)
        self.n_blocks = blk

    def repack(self):
        if self.table is not None:
            call("cn_pack_weight_batch", self.table, self.table.shape[0], self.n_blocks, self.block_record, dtype_code(self.dtype))


def pack_weight(w, mode, dtype, row_scale=None):
    """w: fp32 [A,B,KH,KW] parameter -> packed GEMM operand (see cn_pack_weight)."""
    arena = PackArena.current
    if arena is not None and row_scale is None:
        k = PackArena.key(w, mode, dtype)
        hit = arena.slots.get(k)
        if hit is not None:
            return hit
        if arena.recording:
            arena.requests[k] = w.detach()
    A, B, KH, KW = w.shape
    rows_pad, inner_pad, cols = _pack_dims(w.shape, mode)
    wp = torch.empty((rows_pad, cols), dtype=dtype, device=w.device)
    call("cn_pack_weight", w.detach().contiguous(), wp, A, B, KH, KW, mode, rows_pad, inner_pad, row_scale, dtype_code(dtype))
    return wp


def unpack_wgrad(dwp, A, B, KH, KW, into=None):
    """packed fp32 gradient -> parameter layout; `into` (a .grad view) is accumulated in place."""
    dw = into if into is not None else torch.empty((A, B, KH, KW), dtype=torch.float32, device=dwp.device)
    call("cn_unpack_wgrad", dwp, dw, A, B, KH, KW, rup(B, 16), int(into is not None))
    return dw


class SideGrads:
    """Weight gradients on a second HIP stream.

    A layer's weight gradient only feeds the optimizer, while its data gradient is on the critical path of backward.
    With this switch on, every op launches its weight-gradient kernels on a side stream (forked after dY is ready) and
    deposits the result directly into `param.grad` (the flat gradient buffer) instead of returning it to autograd, so
    the two chains overlap and fill each other's latency bubbles; `join()` merges the streams before the optimizer.
    Under hipGraph capture this becomes a fork/join of graph branches.  Not used together with grad-ready hooks.
    """
    stream = None
    more = []             # further weight-gradient streams (CN_SIDE_STREAMS > 1): submissions go round robin over [stream] + more
    active = False        # only true while a TrainStep (which joins afterwards) is running its backward

    @classmethod
    def enable(cls, on=True, fp32=False):
        """fp32 (parity mode): the generic fp32 weight-gradient kernels are several times longer — they keep the wider grid"""
        if on and cls.stream is None:
            cls.stream = torch.cuda.Stream()
            cls.more = [torch.cuda.Stream() for _ in range(int(_os.environ.get("CN_SIDE_STREAMS", 1)) - 1)]
        # background-shaped weight-gradient grids while they share the GPU with the data-gradient chain.  Round 3 (slab-form 3x3
        # kernels at two waves per SIMD, matrix-core DCN weight gradient): the side stream has slack, so the fewer CUs it occupies the
        # faster the critical chain runs — DLA-34 bs 64: 1 413 / 1 432 / 1 437 / 1 460 / 1 480 / 1 478 / 1 368 / 1 165 images/s with
        # 768 / 512 / 384 / 192 / 160 / 128 / 96 / 64 workgroups (below ~128 the side stream itself becomes the critical path)
        cls.thin = cls.grid = int(_os.environ.get("CN_WGRAD_BLOCKS", (384 if fp32 else 160) if on else 1536))
        return on

    thin = 1536
    grid = 1536           # cn_hooks.wgrad_blocks of the weight-gradient calls issued now (`wgrad_hooks`): `thin`, or the tail's wide grid
    fwd_order = 0         # convs seen in this step's forward pass (TrainStep resets it): the first few are the LAST of backward
    TAIL_LAYERS = int(_os.environ.get("CN_TAIL_LAYERS", 2))      # re-swept on the final tree: 3 -> 2: -0.1 ms (five interleaved pairs)

    @classmethod
    def next_order(cls):
        cls.fwd_order += 1
        return cls.fwd_order - 1

    @classmethod
    def wide_if_tail(cls, fn, order):
        """Weight gradients of the network's first layers are produced when the data-gradient chain is (all but) finished:
        nothing is left to share the GPU with, so they get a full-width grid instead of the background-shaped one (the
        last millisecond of a step was three thin kernels running one after the other on an otherwise idle GPU)."""
        if order >= cls.TAIL_LAYERS or cls.thin >= 1536 or _os.environ.get("CN_THIN_TAIL"):
            return fn

        def wide():
            cls.grid = int(_os.environ.get("CN_TAIL_BLOCKS", 1536))
            try:
                fn()
            finally:
                cls.grid = cls.thin
        return wide

    @classmethod
    def wgrad_hooks(cls, pre=None, Ci=0):
        """cn_hooks of one weight-gradient call: the split-K grid of the moment (+ the input pre-affine of a deferred BN)"""
        h = Hooks()
        h.wgrad_blocks = cls.grid
        if pre is not None:
            h.set(pre_ss=pre[0], pre_C=Ci, pre_relu=int(pre[1]))
        return h

    @classmethod
    def usable(cls, *params):
        return cls.active and cls.stream is not None and all(p is None or p.grad is not None for p in params)

    pending = []          # (event on the main stream, closure, tensors) not yet launched

    @classmethod
    def submit(cls, fn, *tensors, claims=()):
        """Run `fn` (weight-gradient launches) on the side stream once everything enqueued on the main stream so far is
        done.  The launch itself is DEFERRED to the next submit()/join(): by then the main stream has enqueued its own
        continuation (the data gradient), so under hipGraph capture that continuation is the FIRST child of the
        producer node and keeps the producer's queue — when the side branch was captured first, replay put main-chain
        kernels behind weight-gradient kernels on the same hardware queue (5 ms of main-stream stalls per step)."""
        GradReady.claim(*claims)          # `fn` deposits (and reports) these parameters' gradients later
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        cls._flush()
        cls.pending.append((ev, fn, tensors))

    @classmethod
    def all_streams(cls):
        return ([cls.stream] if cls.stream is not None else []) + cls.more

    rr = 0

    @classmethod
    def _flush(cls):
        todo, cls.pending = cls.pending, []
        streams = cls.all_streams()
        for ev, fn, tensors in todo:
            st = streams[cls.rr % len(streams)]       # layers' weight gradients are independent of each other
            cls.rr += 1
            st.wait_event(ev)
            for t in tensors:
                if t is not None:
                    t.record_stream(st)
            with torch.cuda.stream(st):
                fn()

    stamps = None         # int64[4] device tensor (tools/tail_stamps.py): [step start, launch-stream end, side-stream end, joined]

    @classmethod
    def join(cls):
        if cls.stream is not None:
            cls._flush()
        if cls.active and cls.stream is not None:
            if cls.stamps is not None:
                call("cn_stamp", cls.stamps[1:])
                with torch.cuda.stream(cls.stream):
                    call("cn_stamp", cls.stamps[2:])
            for st in cls.all_streams():
                torch.cuda.current_stream().wait_stream(st)
            if cls.stamps is not None and not _os.environ.get("CN_STAMP_END_ONLY"):
                call("cn_stamp", cls.stamps[3:])
        cls.active = False
        cls.rr = 0


def conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


_SMALLK_WIDTHS = (64, 128, 256, 512, 1024, 2048)


def _igemm(x, wp, bias, residual, Co, KH, KW, stride, pad, transposed, relu, OH, OW, out_dtype=None, bn_stats=False, pre=None, hooks=None):
    """pre = (ss fp32 [2][Ci], relu): x is the raw output of the previous conv; the kernel applies that layer's BN (+ ReLU) on load
    (a shape without that hook raises).  hooks: further per-call extras (the caller reads their `*_taken` fields afterwards)."""
    N, H, W, Ci = x.shape
    cp = rup(Co, 16)
    out_dtype = out_dtype or x.dtype
    y = torch.empty((N, OH, OW, cp), dtype=out_dtype, device=x.device)    # the kernels write the channel padding as zeros
    if pre is not None:
        hooks = (hooks or Hooks()).set(pre_ss=pre[0], pre_C=Ci, pre_relu=int(pre[1]))
    BnStats.launch(bn_stats, y, "cn_conv2d_fwd", x, wp, bias, residual, y, N, H, W, Ci, Ci, OH, OW, Co, cp,
                   residual.shape[-1] if residual is not None else 0, KH, KW, stride, pad, int(transposed), int(relu),
                   dtype_code(x.dtype), dtype_code(out_dtype), hooks=hooks)
    return y


def _wgrad(x, dy, Co, KH, KW, stride, pad, want_bias, db_into=None, pre=None):
    """-> (dwp fp32 [rup32(Co)][KH*KW*Ci], db fp32[Co] | None); x [N,H,W,Ci], dy [N,OH,OW,>=Co].
    `db_into`: fp32[Co] buffer the bias gradient is ACCUMULATED into (e.g. bias.grad)."""
    N, H, W, Ci = x.shape
    _, OH, OW, ld = dy.shape
    dwp = zeros((rup(Co, 32), KH * KW * Ci), torch.float32, x.device)
    db = db_into if db_into is not None else (zeros((Co,), torch.float32, x.device) if want_bias else None)
    call("cn_conv2d_wgrad", x, dy, dwp, db, N, H, W, Ci, Ci, OH, OW, Co, ld, KH, KW, stride, pad, dtype_code(x.dtype),
         hooks=SideGrads.wgrad_hooks(pre, Ci))
    return dwp, db


def _wgrad_param(x, dy, Co, Ci, KH, KW, stride, pad, into=None, want_bias=False, db_into=None, pre=None):
    """Weight (+ bias) gradient in PARAMETER layout: -> (dw fp32 [Co,Ci,KH,KW] — `into` when given, accumulated in place —, db).
    Shapes whose kernel has the slab form (cn_conv2d_wgrad_direct: bf16 3x3 / stride 1) need neither a pre-zeroed packed gradient
    nor an unpack launch; everything else goes through cn_conv2d_wgrad + cn_unpack_wgrad."""
    N, H, W, Cx = x.shape
    _, OH, OW, ld = dy.shape
    dt = dtype_code(x.dtype)
    if Cx == Ci and x.dtype == torch.bfloat16 and pre is None:
        hooks = SideGrads.wgrad_hooks()        # (the slab count, hence the scratch size, follows the grid)
        n = int(_hip.query("cn_conv2d_wgrad_direct_bytes_h", N, H, W, Ci, Cx, OH, OW, Co, ld, KH, KW, stride, pad, dt, hooks.wgrad_blocks))
        if n:
            ws = _hip.workspace(n, x.device, "wgrad_slabs")
            dw = into if into is not None else torch.empty((Co, Ci, KH, KW), dtype=torch.float32, device=x.device)
            db = db_into if db_into is not None else (zeros((Co,), torch.float32, x.device) if want_bias else None)
            call("cn_conv2d_wgrad_direct", x, dy, dw, db, int(into is not None), ws, n, N, H, W, Ci, Cx, OH, OW, Co, ld, KH, KW,
                 stride, pad, dt, hooks=hooks)
            return dw, db
    dwp, db = _wgrad(x, dy, Co, KH, KW, stride, pad, want_bias, db_into=db_into, pre=pre)
    if into is not None:
        return unpack_wgrad(dwp, Co, Ci, KH, KW, into=into), db
    dw = unpack_wgrad(dwp, Co, Cx, KH, KW)[:, :Ci].contiguous() if Cx != Ci else unpack_wgrad(dwp, Co, Ci, KH, KW)
    return dw, db


# ------------------------------------------------------------------------------------------------ conv
class Conv2dFn(Function):
    """nn.Conv2d (+bias, +ReLU) on NHWC.  weight fp32 [Co,Ci,KH,KW]; x channels = rup(Ci,16) (zero padded)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, mask_dx=False, defer_relu_bwd=False, passthrough=False, bn_stats=False, pre=None):
        """pre = (ss, relu): x is the RAW output of the previous conv whose training-mode BN (+ ReLU) was deferred (BnDeferFn): the
        forward and the weight-gradient kernels apply it on load; the data gradient is w.r.t. the normalised activation.
        mask_dx: x is the output of a fused-ReLU producer that was built with defer_relu_bwd=True; this layer's data
        gradient is then masked by (x > 0) in its own epilogue (relu mode 2) and the producer skips its cn_relu_bwd pass.
        passthrough: also return x itself (for a residual connection around this conv): the gradient of that second use
        then arrives in THIS backward and is added in the data-gradient kernel's epilogue instead of by a separate
        element-wise pass of the autograd engine."""
        Co, Ci, KH, KW = weight.shape
        N, H, W, Cx = x.shape
        assert Cx == rup(Ci, 16), f"conv input has {Cx} channels, weight expects {Ci}"
        wp = pack_weight(weight, 1, x.dtype)
        OH, OW = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
        y = _igemm(x, wp, bias, None, Co, KH, KW, stride, pad, False, relu, OH, OW, bn_stats=bn_stats, pre=pre)
        ctx.save_for_backward(x, weight, y if (relu and not defer_relu_bwd) else None)
        ctx.pre = pre
        ctx.cfg = (stride, pad, relu and not defer_relu_bwd, bias is not None)
        ctx.mask_dx = mask_dx
        ctx.bias_ref = bias
        ctx.order = SideGrads.next_order()
        ctx.passthrough = passthrough
        ctx.cell = cell_of(x)
        if passthrough:
            ctx.set_materialize_grads(False)      # a skip path that reports to x's GradCell sends None here, not a zero tensor
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, y = ctx.saved_tensors
        if dy is None:                            # only the skip path carried a gradient
            if ctx.cell is not None:
                dskip = ctx.cell.give(dskip)
            return dskip, None, None, None, None, None, None, None, None, None, None
        stride, pad, relu, has_bias = ctx.cfg
        dy = dy.contiguous()
        if relu:
            g = torch.empty_like(dy)
            call("cn_relu_bwd", dy, y, g, dy.numel(), dtype_code(dy.dtype))
            dy = g
        dx, dw, db = _conv2d_bwd(x, weight, ctx.bias_ref, dy, stride, pad, has_bias, ctx.mask_dx, ctx.cell, ctx.order,
                                 ctx.needs_input_grad[:3], dskip, pre=ctx.pre)
        return dx, dw, db, None, None, None, None, None, None, None, None


def _conv2d_bwd(x, weight, bias_ref, dy, stride, pad, has_bias, mask_dx, cell, order, needs, dskip=None, pre=None):
    """Backward of one NHWC convolution for the output gradient dy (ReLU already undone): -> (dx, dw, db).  Weight / bias gradients
    go to the side stream and straight into `.grad` inside a TrainStep (then dw = db = None); dx goes to x's GradCell when it has one
    (then dx = None).  needs = (dx?, dw?, db?).  Shared by Conv2dFn and HeadFn."""
    Co, Ci, KH, KW = weight.shape
    N, H, W, Cx = x.shape
    dx = dw = db = None
    if needs[1] and Cx == Ci and SideGrads.usable(weight, bias_ref):
        def side_work(x=x, dy=dy, bias=bias_ref):
            _wgrad_param(x, dy, Co, Ci, KH, KW, stride, pad, into=weight.grad, db_into=bias.grad if has_bias else None, pre=pre)
            GradReady.note(weight, bias)
        SideGrads.submit(SideGrads.wide_if_tail(side_work, order), x, dy, claims=(weight, bias_ref))
    elif needs[1]:
        dw, db = _wgrad_param(x, dy, Co, Ci, KH, KW, stride, pad, want_bias=has_bias and needs[2], pre=pre)
    if needs[0]:
        if pre is not None and mask_dx:
            raise RuntimeError("mask_dx on a conv whose input BN is deferred: the masks would be applied twice")
        wpd = pack_weight(weight, 0, x.dtype)          # rows = Ci, k = tap*rup(Co,16) + co
        if Cx != Ci:
            raise RuntimeError("data gradient through a channel-padded conv input is not supported")
        small = Co <= 4 and Ci in _SMALLK_WIDTHS                     # 1- / 2-channel heads -> streaming VALU kernel
        mid = 4 < Co <= 96 and Ci == 256 and dy.shape[-1] <= 96      # class / keypoint heads -> streaming MFMA kernel
        if (KH == 1 and KW == 1 and stride == 1 and pad == 0 and (small or mid) and x.dtype == torch.bfloat16
                and not _os.environ.get("CN_DISABLE_CONV_SMALLK")):
            # a short contraction with a 256-wide output is epilogue / overhead bound on the GEMM kernel
            dx = torch.empty((N, H, W, Ci), dtype=x.dtype, device=x.device)
            call("cn_conv1x1_smallk", dy, wpd, x if mask_dx else None, dx, N * H * W, Co, dy.shape[-1], Ci, Ci,
                 x.shape[-1] if mask_dx else 0, 2 if mask_dx else 0, dtype_code(x.dtype))
        elif mask_dx:
            dx = _igemm(dy, wpd, None, x, Ci, KH, KW, stride, pad, True, 2, H, W)
        else:
            skip = dskip.contiguous() if (dskip is not None and dskip.dtype == x.dtype and dskip.shape == x.shape) else None
            if skip is not None:
                dskip = None                                  # folded into the epilogue
            elif cell is not None and not (stride == 2 and KH == 3 and ((Co, Ci) == (32, 16) or ((Co, Ci) == (64, 32) and _NO_DGRAD_S2_TILE))):
                # x is shared: what its other consumers sent rides in the epilogue's residual slot (the stride-2 shapes on the
                # direct-from-global data-gradient kernel keep it: that kernel has no residual input)
                skip = cell.take(like=x)
            sink = hooks = None
            if (pre is not None and len(pre) > 2 and skip is None and BnBwdSinks.enabled and BnStats.fused and x.dtype == torch.bfloat16
                    and Cx == pre[2].shape[1]):
                # x is a raw conv output behind a deferred BN: ask the data-gradient kernel for that BN's backward statistics
                sink = BnStats.acquire("b", Cx, x.device)
                hooks = Hooks().set(bnb_part=sink, bnb_slots=sink.shape[0], bnb_C=Cx, bnb_x=x, bnb_stats=pre[2], bnb_relu=int(pre[1]))
            try:
                dx = _igemm(dy, wpd, None, skip, Ci, KH, KW, stride, pad, True, False, H, W, hooks=hooks)
            except BaseException:
                if sink is not None:
                    BnStats.release(sink)
                raise
            if sink is not None:
                if hooks.bnb_taken:
                    BnBwdSinks.note(dx, sink)
                else:
                    BnStats.release(sink)
        if dskip is not None:
            dx = _add_tensors(dx, dskip)
    elif dskip is not None:
        dx = dskip
    if cell is not None:
        dx = cell.give(dx)
    return dx, dw, db


def _cat_args(xs):
    """(x0..x5, c0..c5, nsrc) of cn_conv1x1_cat_fwd"""
    if len(xs) > 6:
        raise RuntimeError("conv1x1_cat: at most 6 sources")
    ptrs = list(xs) + [None] * (6 - len(xs))
    chans = [int(t.shape[-1]) for t in xs] + [0] * (6 - len(xs))
    return (*ptrs, *chans, len(xs))


def conv1x1_cat_raw(xs, wp, bias, residual, Co, relu, bn_stats=False):
    """y = act(conv1x1(cat(xs, channel)) + bias + residual) without the concatenated tensor (one launch)"""
    N, H, W, _ = xs[0].shape
    cp = rup(Co, 16)
    y = torch.empty((N, H, W, cp), dtype=xs[0].dtype, device=xs[0].device)
    BnStats.launch(bn_stats, y, "cn_conv1x1_cat_fwd", *_cat_args(xs), wp, bias, residual, y, N, H, W, Co, cp,
                   residual.shape[-1] if residual is not None else 0, int(relu), dtype_code(xs[0].dtype))
    return y


class Conv1x1CatFn(Function):
    """1x1 conv over torch.cat(xs, 1) (DLA Root, pose_dla_dcn.py:180-188) with the concatenation never materialised: the forward
    GEMM walks the sources in its K loop; each source's data gradient is its own GEMM over the matching ROWS of the packed
    data-gradient operand and its weight gradient lands in the matching COLUMNS of weight.grad (no split copies either).
    weight fp32 [Co, sum C_s, 1, 1]; every C_s a multiple of 16."""

    @staticmethod
    def forward(ctx, weight, bn_stats, *xs):
        Co = weight.shape[0]
        cells = [cell_of(t) for t in xs]
        xs = tuple(t.contiguous() for t in xs)
        assert sum(t.shape[-1] for t in xs) == weight.shape[1], "Root conv: channel counts of the children do not add up"
        y = conv1x1_cat_raw(xs, pack_weight(weight, 1, xs[0].dtype), None, None, Co, False, bn_stats=bn_stats)
        ctx.save_for_backward(weight, *xs)
        ctx.order = SideGrads.next_order()
        ctx.cells = cells
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, *xs = ctx.saved_tensors
        Co, Ct = weight.shape[:2]
        N, H, W, _ = xs[0].shape
        dy = dy.contiguous()
        dt = xs[0].dtype
        chans = [t.shape[-1] for t in xs]
        offs = [sum(chans[:i]) for i in range(len(xs))]
        dw = None
        if ctx.needs_input_grad[0]:
            side = SideGrads.usable(weight)
            if not side:
                dw = zeros((Co, Ct), torch.float32, dy.device)

            def wgrads(into):
                for x, c, k0 in zip(xs, chans, offs):
                    dwp, _ = _wgrad(x, dy, Co, 1, 1, 1, 0, False)               # [rup32(Co)][c]
                    call("cn_unpack_wgrad_cols", dwp, into.view(Co, Ct)[:, k0:], Co, c, c, Ct, int(into is weight.grad))
            if side:
                def side_work():
                    wgrads(weight.grad)
                    GradReady.note(weight)
                SideGrads.submit(side_work, dy, *xs, claims=(weight,))
            else:
                wgrads(dw)
                dw = dw.view(Co, Ct, 1, 1)
        dxs = [None] * len(xs)
        if any(ctx.needs_input_grad[2:]):             # inputs: weight, bn_stats flag, *xs
            wpd = pack_weight(weight, 0, dt)          # rows = input channel (of the concatenation), k = rup(Co, 16)
            for i, (c, k0) in enumerate(zip(chans, offs)):
                if ctx.needs_input_grad[2 + i]:
                    cell = ctx.cells[i]
                    acc = cell.take(like=xs[i]) if cell is not None else None      # shared child: sum in the epilogue
                    dxs[i] = _igemm(dy, wpd[k0:k0 + c], None, acc, c, 1, 1, 1, 0, True, False, H, W)
                    if cell is not None:
                        dxs[i] = cell.give(dxs[i])
        return (dw, None, *dxs)


def conv1x1_cat(xs, weight, bn_stats=False):
    return BnStats.pop(Conv1x1CatFn.apply(weight, bn_stats, *xs)) if bn_stats else Conv1x1CatFn.apply(weight, False, *xs)


class ConvTranspose2dFn(Function):
    """nn.ConvTranspose2d(Ci,Co,4,stride=2,padding=1,bias=False) — msra_resnet.py:178-187.  weight [Ci,Co,KH,KW]."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        Ci, Co, KH, KW = weight.shape
        N, H, W, _ = x.shape
        wp = pack_weight(weight, 0, x.dtype)               # rows = Co, k = tap*Ci + ci
        OH, OW = (H - 1) * stride - 2 * pad + KH, (W - 1) * stride - 2 * pad + KW
        y = _igemm(x, wp, None, None, Co, KH, KW, stride, pad, True, False, OH, OW)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        Ci, Co, KH, KW = weight.shape
        N, H, W, _ = x.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[1] and SideGrads.usable(weight):
            def side_work(x=x, dy=dy):
                dwp, _ = _wgrad(dy, x, Ci, KH, KW, stride, pad, False)
                unpack_wgrad(dwp, Ci, Co, KH, KW, into=weight.grad)
                GradReady.note(weight)
            SideGrads.submit(side_work, x, dy, claims=(weight,))
        elif ctx.needs_input_grad[1]:
            # dW[ci][co][t] = sum x[n,ih,iw,ci] * dy[n, ih*s-p+kh, iw*s-p+kw, co]: the wgrad kernel with roles swapped
            dwp, _ = _wgrad(dy, x, Ci, KH, KW, stride, pad, False)
            dw = unpack_wgrad(dwp, Ci, Co, KH, KW)
        if ctx.needs_input_grad[0]:
            wpd = pack_weight(weight, 1, x.dtype)          # rows = Ci, k = tap*Co + co
            dx = _igemm(dy, wpd, None, None, Ci, KH, KW, stride, pad, False, False, H, W)
        return dx, dw, None, None


def stem_conv_infer(img, weight, scale, bias, stride, pad, relu, dtype):
    """no-grad stem with an eval-mode BN (+ReLU) folded into the epilogue"""
    Co, Ci, KH, KW = weight.shape
    N, _, H, W = img.shape
    OH, OW = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
    y = torch.empty((N, OH, OW, Co), dtype=dtype, device=img.device)
    call("cn_stem_conv_fwd", img.contiguous(), weight.detach().contiguous(), scale, bias, y, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, int(relu),
         dtype_code(dtype))
    return y


class StemConvFn(Function):
    """7x7 conv on the NCHW fp32 image (3 channels) -> NHWC activations; no data gradient (it is the input)."""

    @staticmethod
    def forward(ctx, img, weight, stride, pad, dtype, bn_stats=False):
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = img.shape
        OH, OW = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
        img = img.contiguous()
        y = torch.empty((N, OH, OW, Co), dtype=dtype, device=img.device)
        BnStats.launch(bn_stats, y, "cn_stem_conv_fwd", img, weight.detach().contiguous(), None, None, y, N, Ci, H, W, Co, KH, KW, stride,
                       pad, OH, OW, 0, dtype_code(dtype))
        ctx.save_for_backward(img, weight)
        ctx.cfg = (stride, pad)
        ctx.order = SideGrads.next_order()
        return y

    @staticmethod
    def backward(ctx, dy):
        img, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = img.shape
        dy = dy.contiguous()
        if SideGrads.usable(weight):
            def side_work(img=img, dy=dy):  # the kernel accumulates with atomics: deposit straight into weight.grad
                call("cn_stem_conv_wgrad", img, dy, weight.grad, N, Ci, H, W, Co, KH, KW, stride, pad, dy.shape[1], dy.shape[2],
                     dtype_code(dy.dtype), hooks=SideGrads.wgrad_hooks())
                GradReady.note(weight)
            SideGrads.submit(SideGrads.wide_if_tail(side_work, ctx.order), img, dy, claims=(weight,))
            return None, None, None, None, None, None
        dw = zeros_like(weight, torch.float32)
        call("cn_stem_conv_wgrad", img, dy, dw, N, Ci, H, W, Co, KH, KW, stride, pad, dy.shape[1], dy.shape[2], dtype_code(dy.dtype),
             hooks=SideGrads.wgrad_hooks())
        return None, dw, None, None, None, None


# ------------------------------------------------------------------------------------------------ batch norm
class BatchNormActFn(Function):
    """Training-mode BatchNorm2d + optional residual add + optional ReLU (one fused apply pass)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, relu, part=None):
        """part: the statistics sink the kernel that produced x filled in its epilogue (BnStats) — x is then read once, by the
        apply pass, instead of twice"""
        C = x.shape[-1]
        npix = x.numel() // C
        y = torch.empty_like(x)
        stats = torch.empty((4, C), dtype=torch.float32, device=x.device)     # mean, invstd, scale, shift
        mean, invstd, ss = stats[0], stats[1], stats[2:]
        ws, n = _bn_ws(npix, C, x.device)
        if part is not None and BnStats.fused:
            clear = BnStats.retire(part)          # one launch: the apply kernel reduces the sink, and zeroes the one consumed before it
            call("cn_bn_train_fwd_sink", x, residual, y, gamma.detach(), beta.detach(), running_mean, running_var, mean, invstd, ss,
                 part, part.shape[0], clear, clear.numel() if clear is not None else 0, npix, C, BN_MOMENTUM, BN_EPS, int(relu),
                 dtype_code(x.dtype))
        elif part is not None:
            call("cn_bn_train_fwd_stats", x, residual, y, gamma.detach(), beta.detach(), running_mean, running_var, mean, invstd, ss,
                 part, part.shape[0], npix, C, BN_MOMENTUM, BN_EPS, int(relu), dtype_code(x.dtype), ws, n)
            BnStats.release(part)                 # handed back all-zero
        else:
            call("cn_bn_train_fwd", x, residual, y, gamma.detach(), beta.detach(), running_mean, running_var, mean, invstd, ss,
                 npix, C, BN_MOMENTUM, BN_EPS, int(relu), dtype_code(x.dtype), ws, n)
        # ReLU backward mask: without a residual input it is recomputed from x and the saved affine (y is not re-read)
        need_y = relu and residual is not None
        ctx.save_for_backward(x, y if need_y else None, gamma, stats)
        ctx.beta_ref = beta
        ctx.cfg = (relu, residual is not None)
        ctx.res_cell = cell_of(residual)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, stats = ctx.saved_tensors
        mean, invstd, ss = stats[0], stats[1], stats[2:]
        relu, has_res = ctx.cfg
        C = x.shape[-1]
        npix = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        ws, n = _bn_ws(npix, C, x.device)
        beta = ctx.beta_ref
        cell = ctx.res_cell if has_res else None
        racc = cell.take(like=x) if cell is not None else None       # shared residual input: its other consumers' sum joins in the store
        sink = clear = None
        filled = BnBwdSinks.take(dy)               # the kernel that produced dy already summed the statistics in its epilogue
        if filled is not None and (has_res or y is not None):
            filled.zero_()
            BnStats.release(filled)
            filled = None
        if filled is not None:
            sink = filled
            clear = BnStats.retire(sink)
        elif BnStats.fused and x.dtype == torch.bfloat16 and C % 8 == 0:
            # two launches instead of three: the statistics pass adds into a sink that the apply pass reduces itself
            sink = BnStats.acquire("b", C, x.device)
            clear = BnStats.retire(sink)

        def run(dgamma, dbeta, accumulate):
            if filled is not None:
                call("cn_bn_train_bwd_apply", dy, x, y, gamma.detach(), mean, invstd, ss, dx, dres, racc, dgamma, dbeta, accumulate,
                     sink, sink.shape[0], clear, clear.numel() if clear is not None else 0, npix, C, int(relu), dtype_code(x.dtype))
            elif sink is not None:
                call("cn_bn_train_bwd_sink", dy, x, y, gamma.detach(), mean, invstd, ss, dx, dres, racc, dgamma, dbeta, accumulate,
                     sink, sink.shape[0], clear, clear.numel() if clear is not None else 0, npix, C, int(relu), dtype_code(x.dtype))
            else:
                call("cn_bn_train_bwd_acc", dy, x, y, gamma.detach(), mean, invstd, ss, dx, dres, racc, dgamma, dbeta, accumulate,
                     npix, C, int(relu), dtype_code(x.dtype), ws, n)

        if SideGrads.usable(gamma, beta):      # inside a TrainStep: deposit straight into the flat gradient buffer
            run(gamma.grad, beta.grad, 1)
            GradReady.note(gamma, beta)
            return dx, None, None, None, None, (cell.give(dres) if cell is not None else dres), None, None
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        run(dgamma, dbeta, 0)
        return dx, dgamma, dbeta, None, None, (cell.give(dres) if cell is not None else dres), None, None


class BnDeferFn(Function):
    """Training-mode BatchNorm2d (+ ReLU) whose apply pass is left to the CONSUMER: forward only finalizes the batch statistics the
    producing kernel accumulated (`part`) and hands the raw tensor on; the consumer (a 16-input-channel 3x3 conv: its forward and its
    weight gradient, `pre=`) computes bf16(relu(fma(x, scale, shift))) on load — bit-identical to what the apply pass would have
    stored — so the normalised activation is never written or read.  Backward is BatchNormActFn's (ReLU mask recomputed from x and
    the saved affine) on the gradient w.r.t. that virtual activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, part, relu):
        C = x.shape[-1]
        npix = x.numel() // C
        stats = torch.empty((4, C), dtype=torch.float32, device=x.device)     # mean, invstd, scale, shift
        call("cn_bn_finalize_sink", part, part.shape[0], gamma.detach(), beta.detach(), running_mean, running_var, stats[0], stats[1],
             stats[2:], npix, C, BN_MOMENTUM, BN_EPS)
        BnStats.release(part)                     # handed back all-zero
        ctx.save_for_backward(x, None, gamma, stats)
        ctx.beta_ref = beta
        ctx.cfg = (relu, False)
        ctx.res_cell = None
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)          # (no zero tensor for the statistics output's gradient: an ATen fill per step otherwise)
        return x.view_as(x), stats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:
            return None, None, None, None, None, None, None
        r = BatchNormActFn.backward(ctx, dy)
        return r[0], r[1], r[2], None, None, None, None


class StemBnDeferFn(Function):
    """7x7 stem conv -> training-mode BatchNorm2d (+ ReLU) as ONE autograd node whose BN passes live in the neighbouring kernels
    (pose_dla_dcn.py:283-287 base_layer): forward = the stem kernel (batch statistics from its epilogue) + a finalize launch, the raw
    conv output is handed on for the next conv to normalise on load (BnDeferFn's protocol); backward = statistics pass + a
    coefficient launch, then the stem's weight-gradient kernel forms the BN input gradient on load (cn_stem_conv_wgrad_bn) — neither
    the normalised activation nor the BN input gradient ever exists in memory."""

    @staticmethod
    def forward(ctx, img, weight, gamma, beta, running_mean, running_var, stride, pad, dtype, relu):
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = img.shape
        OH, OW = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
        img = img.contiguous()
        y = torch.empty((N, OH, OW, Co), dtype=dtype, device=img.device)
        BnStats.launch(True, y, "cn_stem_conv_fwd", img, weight.detach().contiguous(), None, None, y, N, Ci, H, W, Co, KH, KW, stride,
                       pad, OH, OW, 0, dtype_code(dtype))
        part, BnStats.last = BnStats.last, None
        if part is None:
            raise RuntimeError("StemBnDeferFn: the stem kernel for this shape has no statistics hook (use stem conv + BatchNormActFn)")
        npix = N * OH * OW
        stats = torch.empty((4, Co), dtype=torch.float32, device=img.device)  # mean, invstd, scale, shift
        call("cn_bn_finalize_sink", part, part.shape[0], gamma.detach(), beta.detach(), running_mean, running_var, stats[0], stats[1],
             stats[2:], npix, Co, BN_MOMENTUM, BN_EPS)
        BnStats.release(part)
        ctx.save_for_backward(img, weight, gamma, y, stats)
        ctx.beta_ref = beta
        ctx.cfg = (stride, pad, relu)
        ctx.order = SideGrads.next_order()
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:
            return None, None, None, None, None, None, None, None, None, None
        img, weight, gamma, y, stats = ctx.saved_tensors
        beta = ctx.beta_ref
        stride, pad, relu = ctx.cfg
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = img.shape
        OH, OW = y.shape[1], y.shape[2]
        npix = N * OH * OW
        dt = dtype_code(y.dtype)
        dy = dy.contiguous()
        mean, invstd, ss = stats[0], stats[1], stats[2:]
        sink = BnBwdSinks.take(dy)                 # filled by the epilogue of the kernel that produced dy, where it has the hook
        filled = sink is not None
        if not filled:
            sink = BnStats.acquire("b", Co, y.device)
        clear = BnStats.retire(sink)
        if not filled:
            call("cn_bn_bwd_stats", dy, y, None, mean, invstd, ss, sink, sink.shape[0], npix, Co, int(relu), dt)
        coef = torch.empty((5, Co), dtype=torch.float32, device=y.device)
        direct = SideGrads.usable(gamma, beta)
        dgamma = gamma.grad if direct else torch.empty(Co, dtype=torch.float32, device=y.device)
        dbeta = beta.grad if direct else torch.empty(Co, dtype=torch.float32, device=y.device)
        call("cn_bn_bwd_coef_sink", sink, sink.shape[0], gamma.detach(), mean, invstd, ss, dgamma, dbeta, int(direct), coef, clear,
             clear.numel() if clear is not None else 0, npix, Co)
        if direct:
            GradReady.note(gamma, beta)

        def wgrad(dw):
            call("cn_stem_conv_wgrad_bn", img, dy, y, coef, dw, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, int(relu), dt,
                 hooks=SideGrads.wgrad_hooks())

        if SideGrads.usable(weight):
            def side_work():
                wgrad(weight.grad)
                GradReady.note(weight)
            SideGrads.submit(SideGrads.wide_if_tail(side_work, ctx.order), img, dy, y, coef, claims=(weight,))
            dw = None
        else:
            dw = zeros_like(weight, torch.float32)
            wgrad(dw)
        return None, dw, (None if direct else dgamma), (None if direct else dbeta), None, None, None, None, None, None


def stem_bn_defer(img, weight, bn, stride, pad, dtype, relu=True):
    """-> the stem's RAW output tagged `_cn_pre` (see batch_norm_defer)"""
    y, stats = StemBnDeferFn.apply(img, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, stride, pad, dtype, relu)
    y._cn_pre = (stats[2:], bool(relu), stats)
    return y


def batch_norm_defer(x, bn, relu=True):
    """-> x itself (raw) tagged with `_cn_pre = (scale | shift, relu)` for the conv that consumes it; x must carry the statistics sink
    of its producer (`_bn_part`)"""
    y, stats = BnDeferFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, x._bn_part, relu)
    y._cn_pre = (stats[2:], bool(relu), stats)
    return y


class ScaleShiftActFn(Function):
    """Eval-mode BN as a per-channel affine (+residual, +ReLU).  Gradients flow to x / residual only: the statistics
    and the affine are frozen in eval mode."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        C = x.shape[-1]
        y = torch.empty_like(x)
        call("cn_scale_shift_act", x, residual, y, scale, shift, x.numel() // C, C, int(relu), dtype_code(x.dtype))
        ctx.save_for_backward(y if relu else None, scale)
        ctx.cfg = (relu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        relu, has_res = ctx.cfg
        dy = dy.contiguous()
        C = dy.shape[-1]
        if relu:
            g = torch.empty_like(dy)
            call("cn_relu_bwd", dy, y, g, dy.numel(), dtype_code(dy.dtype))
            dy = g
        dx = torch.empty_like(dy)
        zero = zeros_like(scale)
        call("cn_scale_shift_act", dy, None, dx, scale, zero, dy.numel() // C, C, 0, dtype_code(dy.dtype))
        return dx, None, None, (dy if has_res else None), None


def scale_shift_act(x, scale, shift, residual, relu):
    """Affine + residual + ReLU (eval BN that was not folded into a conv)."""
    return ScaleShiftActFn.apply(x, scale, shift, residual, relu)


# ------------------------------------------------------------------------------------------------ pool / up-sample / glue
class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        N, H, W, C = x.shape
        OH, OW = conv_out(H, k, stride, pad), conv_out(W, k, stride, pad)
        y = _empty_like_shape(x, (N, OH, OW, C))
        need = ctx.needs_input_grad[0]
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device) if need else None
        call("cn_maxpool_fwd", x, y, idx, N, H, W, C, k, stride, pad, OH, OW, dtype_code(x.dtype))
        ctx.save_for_backward(idx)
        ctx.cfg = (k, stride, pad, OH, OW, (N, H, W, C), x.dtype)
        ctx.cell = cell_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        k, stride, pad, OH, OW, (N, H, W, C), dt = ctx.cfg
        dx = torch.empty((N, H, W, C), dtype=dt, device=dy.device)
        acc = ctx.cell.take(like=dx) if ctx.cell is not None else None
        call("cn_maxpool_bwd_acc", idx, dy.contiguous(), acc, dx, N, H, W, C, k, stride, pad, OH, OW, dtype_code(dt))
        return (ctx.cell.give(dx) if ctx.cell is not None else dx), None, None, None


class DwDeconvFn(Function):
    """Depthwise ConvTranspose2d(o,o,2f,stride=f,padding=f//2,groups=o,bias=False); weight fp32 [C,1,k,k]."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, residual=None):
        N, H, W, C = x.shape
        k = weight.shape[-1]
        OH, OW = (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
        y = _empty_like_shape(x, (N, OH, OW, C))
        call("cn_dwdeconv_fwd", x, weight.detach().contiguous(), None if residual is None else residual.contiguous(), y, N, H, W, C, k,
             stride, pad, OH, OW, dtype_code(x.dtype))
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, OH, OW)
        ctx.has_res = residual is not None
        ctx.cells = (cell_of(x), cell_of(residual))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, OH, OW = ctx.cfg
        N, H, W, C = x.shape
        k = weight.shape[-1]
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("cn_dwdeconv_bwd_input", dy, weight.detach().contiguous(), dx, N, H, W, C, k, stride, pad, OH, OW, dtype_code(x.dtype))
        if ctx.needs_input_grad[1] and SideGrads.usable(weight):
            def side_work(x=x, dy=dy):
                _dwdeconv_wgrad(x, dy, weight.grad, N, H, W, C, k, stride, pad, OH, OW)
                GradReady.note(weight)
            SideGrads.submit(side_work, x, dy, claims=(weight,))
        elif ctx.needs_input_grad[1]:
            dw = zeros_like(weight, torch.float32)
            _dwdeconv_wgrad(x, dy, dw, N, H, W, C, k, stride, pad, OH, OW)
        dres = dy if ctx.has_res else None
        if ctx.cells[0] is not None and dx is not None:
            dx = ctx.cells[0].give(dx)
        if ctx.cells[1] is not None:
            dres = ctx.cells[1].give(dres)
        return dx, dw, None, None, dres


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        call("cn_add", a, b, out, a.numel(), dtype_code(a.dtype))
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class UpsampleAddFn(Function):
    """y = a + nn.Upsample(scale_factor=2)(low)  — the hourglass merge (large_hourglass.py:196-204); a may be None."""

    @staticmethod
    def forward(ctx, a, low):
        N, H, W, C = low.shape
        low = low.contiguous()
        y = _empty_like_shape(low, (N, 2 * H, 2 * W, C))
        call("cn_upsample2x_add", None if a is None else a.contiguous(), low, y, N, H, W, C, dtype_code(low.dtype))
        ctx.has_a = a is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        N, OH, OW, C = dy.shape
        dy = dy.contiguous()
        dlow = _empty_like_shape(dy, (N, OH // 2, OW // 2, C))
        call("cn_sumpool2x2", dy, dlow, N, OH // 2, OW // 2, C, dtype_code(dy.dtype))
        return (dy if ctx.has_a else None), dlow


class ConcatFn(Function):
    """torch.cat(xs, channel) on NHWC (pose_dla_dcn.py:182)."""

    @staticmethod
    def forward(ctx, *xs):
        N, H, W, _ = xs[0].shape
        chans = [t.shape[-1] for t in xs]
        out = _empty_like_shape(xs[0], (N, H, W, sum(chans)))
        off, npix = 0, N * H * W
        for t, c in zip(xs, chans):
            call("cn_copy_channels", t, c, 0, out, sum(chans), off, npix, c, dtype_code(t.dtype))
            off += c
        ctx.chans = chans
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        N, H, W, tot = dy.shape
        outs, off = [], 0
        for i, c in enumerate(ctx.chans):
            if ctx.needs_input_grad[i]:
                g = _empty_like_shape(dy, (N, H, W, c))
                call("cn_copy_channels", dy, tot, off, g, c, 0, N * H * W, c, dtype_code(dy.dtype))
                outs.append(g)
            else:
                outs.append(None)
            off += c
        return tuple(outs)


class ToNCHWFn(Function):
    """NHWC activations -> public NCHW fp32 tensor with C real channels (drops channel padding)."""

    @staticmethod
    def forward(ctx, x, C):
        N, H, W, ld = x.shape
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        call("cn_nhwc_to_nchw", x, out, N, C, H, W, ld, dtype_code(x.dtype))
        ctx.cfg = (ld, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dy):
        ld, dt = ctx.cfg
        N, C, H, W = dy.shape
        dx = torch.empty((N, H, W, ld), dtype=dt, device=dy.device)
        call("cn_nchw_to_nhwc", dy.contiguous(), dx, N, C, H, W, ld, dtype_code(dt))
        return dx, None


class FromNCHWFn(Function):
    """Public NCHW fp32 tensor [B,C,H,W] -> NHWC activations (channels zero-padded to a multiple of 16), differentiable: the
    inverse of ToNCHWFn.  This is the seam a third-party (plain torch, NCHW) backbone enters the HIP heads through."""

    @staticmethod
    def forward(ctx, x, dtype):
        N, C, H, W = x.shape
        cpad = rup(C, 16)
        out = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
        call("cn_nchw_to_nhwc", x.contiguous().float(), out, N, C, H, W, cpad, dtype_code(dtype))
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dy):
        N, H, W, ld = dy.shape
        dx = torch.empty((N, ctx.C, H, W), dtype=torch.float32, device=dy.device)
        call("cn_nhwc_to_nchw", dy.contiguous(), dx, N, ctx.C, H, W, ld, dtype_code(dy.dtype))
        return dx, None


def mark_nhwc(t, channels=None):
    """Tag an engine-internal NHWC activation tensor (shape [B,H,W,Cpad]) so that a consumer at the package boundary can tell it
    from a public NCHW tensor [B,C,H,W]; `channels` = real channel count (without the padding)."""
    t._cn_nhwc = int(channels if channels is not None else t.shape[-1])
    return t


def is_nhwc(t):
    return getattr(t, "_cn_nhwc", None) is not None


def emit_maps(maps, out_channels, nchw_out):
    """What a backbone returns: the reference contract `list[Tensor[B,out_channels,H/4,W/4]]` fp32 NCHW when `nchw_out`
    (models/__init__.py:14-19), else the engine's NHWC handles, tagged for `heads.CenterHead`."""
    if nchw_out:
        return [ToNCHWFn.apply(m, out_channels) for m in maps]
    # a map that is a shared alias inside the backbone (GradCell) leaves as a plain view: the tensor a caller holds must receive its
    # gradient through autograd (retain_grad / torch.autograd.grad on it), not have it routed past it through a cell
    return [mark_nhwc(m.view_as(m) if cell_of(m) is not None else m, out_channels) for m in maps]


def to_nhwc(x_nchw, dtype, cpad=None):
    """Public NCHW fp32 tensor -> NHWC activations (no autograd: used for inputs)."""
    N, C, H, W = x_nchw.shape
    cpad = cpad or rup(C, 16)
    out = torch.empty((N, H, W, cpad), dtype=dtype, device=x_nchw.device)
    call("cn_nchw_to_nhwc", x_nchw.contiguous(), out, N, C, H, W, cpad, dtype_code(dtype))
    return out


# ------------------------------------------------------------------------------------------------ DCNv2
class DCNv2Fn(Function):
    """DCN(chi, cho, 3x3, stride 1, pad 1, dilation 1, deformable_groups 1) — pose_dla_dcn.py:441-449.
    weight [Co,Ci,3,3], bias [Co], om_weight [27,Ci,3,3], om_bias [27]."""

    @staticmethod
    def forward(ctx, x, weight, bias, om_weight, om_bias, bn_stats=False):
        Co, Ci, _, _ = weight.shape
        N, H, W, _ = x.shape
        dt = dtype_code(x.dtype)
        # offsets / mask logits stay fp32 in both compute modes: sampling coordinates must not be quantised to bf16
        om = _igemm(x, pack_weight(om_weight, 1, x.dtype), om_bias.detach(), None, 27, 3, 3, 1, 1, False, False, H, W,
                    out_dtype=torch.float32)
        wp = pack_weight(weight, 1, x.dtype)                  # [Co_pad][9*Ci]
        if _DCN_UNFUSED:
            col = torch.empty((N, H, W, 9 * Ci), dtype=x.dtype, device=x.device)
            call("cn_dcn_im2col", x, om, col, N, H, W, Ci, Ci, om.shape[-1], dt)
            y = _igemm(col, wp, bias.detach(), None, Co, 1, 1, 1, 0, False, False, H, W)   # a 1x1 conv over the columns
        else:
            # fused: bilinear sampling writes the MFMA operand tile in LDS, no column tensor
            cp = rup(Co, 16)
            y = torch.empty((N, H, W, cp), dtype=x.dtype, device=x.device) if cp == Co else zeros((N, H, W, cp), x.dtype, x.device)
            BnStats.launch(bn_stats, y, "cn_dcn_fwd", x, om, wp, bias.detach(), y, N, H, W, Ci, Ci, Co, cp, om.shape[-1], 0, dt)
            col = None
        ctx.save_for_backward(x, om, col, weight, om_weight)
        ctx.params = (bias, om_weight, om_bias)
        ctx.cell = cell_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, col, weight, om_weight = ctx.saved_tensors
        Co, Ci, _, _ = weight.shape
        N, H, W, _ = x.shape
        dt = dtype_code(x.dtype)
        dy = dy.contiguous()
        # main weight / bias: 1x1 wgrad over the sampled columns
        side = SideGrads.usable(weight, ctx.params[0], ctx.params[1], ctx.params[2])
        dw = db = dw_om = db_om = None
        def main_wgrad(db_into, want_bias):
            """dW / db of the deformable conv: fused re-sampling kernel in bf16, im2col + GEMM in fp32 parity mode"""
            if x.dtype == torch.bfloat16 and not _DCN_UNFUSED:
                dwp_ = zeros((rup(Co, 32), 9 * Ci), torch.float32, x.device)
                call("cn_dcn_wgrad", x, om, dy, dwp_, N, H, W, Ci, Ci, Co, dy.shape[-1], om.shape[-1], dt, hooks=SideGrads.wgrad_hooks())
                db_ = db_into if db_into is not None else (zeros((Co,), torch.float32, x.device) if want_bias else None)
                if db_ is not None:
                    call("cn_colsum", dy, db_, N * H * W, Co, dy.shape[-1], dt)
                return dwp_, db_
            c = col
            if c is None:
                c = torch.empty((N, H, W, 9 * Ci), dtype=x.dtype, device=x.device)
                call("cn_dcn_im2col", x, om, c, N, H, W, Ci, Ci, om.shape[-1], dt)
            return _wgrad(c, dy, Co, 1, 1, 1, 0, want_bias, db_into=db_into)

        if side:
            def side_work():
                dwp, _ = main_wgrad(ctx.params[0].grad, False)
                unpack_wgrad(dwp, Co, Ci, 3, 3, into=weight.grad)
                GradReady.note(weight, ctx.params[0])
            SideGrads.submit(side_work, x, om, dy, col, claims=(weight, ctx.params[0]))
        else:
            dwp, db = main_wgrad(None, True)
            dw = unpack_wgrad(dwp, Co, Ci, 3, 3)
        dx_s = torch.empty_like(x)
        if _DCN_UNFUSED:
            dx_far = zeros((N, H, W, Ci), torch.float32, x.device)
            dom32 = zeros_like(om)
            # reference pipeline kept for A/B profiling: materialise dcol, then source + gather kernels
            wpd = pack_weight(weight, 2, x.dtype)                 # [9*Ci][Co]
            dcol = _igemm(dy, wpd, None, None, 9 * Ci, 1, 1, 1, 0, False, False, H, W)
            dx_tile = torch.empty((N, H, W, Ci), dtype=torch.float32, device=x.device)
            call("cn_dcn_col2im", dcol, x, om, dx_tile, dx_far, dom32, N, H, W, Ci, Ci, om.shape[-1], dt)
            del dcol
            call("cn_add_f32_to", dx_tile, dx_far, dx_s, dx_tile.numel(), dt)
            del dx_tile
        else:
            # fused: the 9x-wide column gradient never reaches HBM
            #   dom  <- epilogue of the GEMM dY x W^T (against the bilinear corner differences of x)
            #   dx   <- adjoint bilinear gather of dY (LDS hit lists) contracted with W, + far samples
            # dx_far (samples displaced > 3 px: rare) follows the lazy protocol of the header: one persistent all-zero
            # buffer per shape + a per-call flag, instead of clearing and re-reading 4*P*Ci bytes per layer per step
            dx_far = _far_buffer((N, H, W, Ci), x.device)
            far_flag = FarFlags.take(x.device)
            slabs = _hip.query("cn_dcn_bwd_dom_slabs", int(Ci), int(dy.shape[-1]), dt)
            direct = (x.dtype == torch.bfloat16 and Ci == 64 and slabs == 1 and dy.shape[-1] in (64, 128)
                      and not _os.environ.get("CN_DISABLE_DOM_TILE"))
            if direct:
                # one channel block: the tile kernel writes the final bf16 offset / mask gradient itself
                slabs, dom32 = 0, torch.empty(om.shape, dtype=x.dtype, device=x.device)
            elif x.dtype == torch.bfloat16 and slabs == Ci // 64:
                # tile kernel: one fp32 copy of dom per 64-channel block of x, plain stores (no atomics, nothing to clear)
                dom32 = torch.empty((slabs,) + tuple(om.shape), dtype=torch.float32, device=x.device)
            else:
                slabs, dom32 = 1, zeros_like(om)
            call("cn_dcn_bwd_dom", dy, pack_weight(weight, 2, x.dtype), x, om, dom32, slabs, dx_far, far_flag, N, H, W, Ci, Co,
                 dy.shape[-1], Ci, om.shape[-1], dt)
            call("cn_dcn_bwd_dx", dy, pack_weight(weight, 0, x.dtype), om, dx_far, far_flag, dx_s, N, H, W, Ci,
                 dy.shape[-1], om.shape[-1], dt)
            if direct:
                dom = dom32
            elif slabs > 1 or x.dtype != torch.float32:
                dom = torch.empty(om.shape, dtype=x.dtype, device=x.device)
                call("cn_sum_slabs", dom32, dom, slabs, om.numel(), dt)
            else:
                dom = dom32
        del dx_far
        if _DCN_UNFUSED:
            if x.dtype == torch.float32:
                dom = dom32
            else:
                dom = torch.empty(om.shape, dtype=x.dtype, device=x.device)
                call("cn_cast", dom32, 0, dom, dt, dom32.numel())
        # offset/mask conv backward (its data gradient is added to the sampling gradient through `residual`)
        if side:
            def side_work_om(x=x, dom=dom, p1=ctx.params[1], p2=ctx.params[2]):
                _wgrad_param(x, dom, 27, Ci, 3, 3, 1, 1, into=p1.grad, db_into=p2.grad)
                GradReady.note(p1, p2)
            SideGrads.submit(side_work_om, x, dom, claims=(ctx.params[1], ctx.params[2]))
        else:
            dw_om, db_om = _wgrad_param(x, dom, 27, Ci, 3, 3, 1, 1, want_bias=True)
        wpo = pack_weight(om_weight, 0, x.dtype)              # rows = Ci, k = tap*32 + c
        dx = torch.empty_like(x)
        call("cn_conv2d_fwd", dom, wpo, None, dx_s, dx, N, H, W, om.shape[-1], om.shape[-1], H, W, Ci, Ci, Ci,
             3, 3, 1, 1, 1, 0, dt, dt)
        if ctx.cell is not None:
            dx = ctx.cell.give(dx)
        return dx, dw, db, dw_om, db_om, None


# ------------------------------------------------------------------------------------------------ losses
class SigmoidClampFn(Function):
    """utils/decode.py:43-45: sigmoid IN PLACE on x, returns clamp(x, 1e-4, 1-1e-4)."""

    @staticmethod
    def forward(ctx, x, lo):
        assert x.dtype == torch.float32 and x.is_contiguous()
        y = torch.empty_like(x)
        call("cn_sigmoid_clamp_fwd", x, y, x.numel(), float(lo))
        ctx.mark_dirty(x)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x)       # x now holds sigmoid(z): the clamp passes gradient where lo <= x <= 1-lo
        ctx.lo = float(lo)
        return x, y

    @staticmethod
    def backward(ctx, dx_unused, dy):
        (s,) = ctx.saved_tensors
        if dx_unused is not None:
            raise RuntimeError("gradient through the in-place sigmoid alias is not supported; use the clamped output")
        if dy is None:
            return None, None
        dz = torch.empty_like(s)
        call("cn_sigmoid_clamp_bwd", dy.contiguous(), s, dz, s.numel(), ctx.lo)
        return dz, None


class FocalLossFn(Function):
    """utils/losses.py:14-39 on sigmoid-clamped predictions."""

    @staticmethod
    def forward(ctx, pred, gt):
        pred, gt = pred.contiguous(), gt.contiguous().float()
        B, C = pred.shape[:2]
        HW = pred[0, 0].numel()
        out = torch.empty(4, dtype=torch.float32, device=pred.device)
        n = _hip.query("cn_focal_workspace_bytes", pred.numel())
        ws = _hip.workspace(n, pred.device, "focal")
        call("cn_focal_fwd", pred, gt, out, B, C, HW, gt.shape[0], gt.shape[1], ws, n)
        ctx.save_for_backward(pred, gt, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, gt, out = ctx.saved_tensors
        B, C = pred.shape[:2]
        dpred = torch.empty_like(pred)
        call("cn_focal_bwd", pred, gt, out, g.contiguous().float().reshape(1), dpred, B, C, pred[0, 0].numel(), gt.shape[0], gt.shape[1])
        return dpred, None


class SigmoidFocalFn(Function):
    """sigmoid_clamped (utils/decode.py:43-45) + FocalLoss (utils/losses.py:14-39) as ONE autograd node: the forward runs the
    same two kernels (x becomes sigmoid(x) in place, y the clamped copy, the loss is computed on y), the backward is a single pass
    instead of focal-backward -> 335 MB of d loss / d p -> sigmoid/clamp-backward."""

    @staticmethod
    def forward(ctx, x, gt, lo):
        assert x.dtype == torch.float32 and x.is_contiguous()
        y = torch.empty_like(x)
        gt = gt.contiguous().float()
        B, C = x.shape[:2]
        HW = x[0, 0].numel()
        out = torch.empty(4, dtype=torch.float32, device=x.device)
        n = _hip.query("cn_focal_workspace_bytes", x.numel())
        ws = _hip.workspace(n, x.device, "focal")
        if (gt.shape == x.shape and x.numel() % 4 == 0 and not ((x.data_ptr() | y.data_ptr() | gt.data_ptr()) & 15)
                and not _os.environ.get("CN_DISABLE_FUSED_FOCAL_FWD")):
            call("cn_sigmoid_clamp_focal_fwd", x, y, gt, out, x.numel(), float(lo), ws, n)      # one pass instead of two
        else:
            call("cn_sigmoid_clamp_fwd", x, y, x.numel(), float(lo))
            call("cn_focal_fwd", y, gt, out, B, C, HW, gt.shape[0], gt.shape[1], ws, n)
        ctx.mark_dirty(x)
        ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, gt, out)
        ctx.lo = float(lo)
        # the second (NHWC, compute dtype) copy of d loss / d logits is only worth writing when the map comes out of a bf16 HeadFn,
        # the one consumer that takes it (fp32 parity mode and foreign heads discard it: round-3 ADVICE)
        ctx.dual = getattr(x, "_cn_head_dtype", None) == torch.bfloat16
        return x, y, out[0]

    @staticmethod
    def backward(ctx, dx_unused, dy_unused, g):
        s, gt, out = ctx.saved_tensors
        if dx_unused is not None:
            raise RuntimeError("gradient through the in-place sigmoid alias is not supported")
        if g is None:
            return None, None, None
        B, C = s.shape[:2]
        dz = torch.empty_like(s)
        gout = g.contiguous().float().reshape(1)
        if DualLayout.enabled and ctx.dual and s.dim() == 4 and gt.shape == s.shape:
            # the usual consumer is a head's backward, which wants NHWC in the compute dtype: leave that copy next to the fp32 map
            # (the consumer recognises `dz` itself — DualLayout — and skips its layout-change pass)
            alt = torch.empty((B, s.shape[2], s.shape[3], rup(C, 16)), dtype=torch.bfloat16, device=s.device)
            if _hip.try_call("cn_sigmoid_focal_bwd_dual", s, gt, out, gout, dz, alt, B, C, s[0, 0].numel(), alt.shape[-1], ctx.lo):
                DualLayout.note(dz, alt)
                return dz, None, None
        call("cn_sigmoid_focal_bwd", s, gt, out, gout, dz, B, C, s[0, 0].numel(), gt.shape[0], gt.shape[1], ctx.lo)
        return dz, None, None


def conv1x1_to_nchw(h, wp, bias, C):
    """A head's last conv straight into the public layout: fp32 NCHW [N,C,H,W] = conv1x1(h) + bias.  One launch where the
    streaming kernel takes the shape (cn_conv1x1_nchw_fwd), else the NHWC conv followed by the layout change."""
    N, H, W, Ch = h.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=h.device)
    if h.dtype == torch.bfloat16 and _hip.try_call("cn_conv1x1_nchw_fwd", h, wp, bias, out, N, H, W, Ch, Ch, C, dtype_code(h.dtype)):
        return out
    y = _igemm(h, wp, bias, None, C, 1, 1, 1, 0, False, False, H, W)
    call("cn_nhwc_to_nchw", y, out, N, C, H, W, y.shape[-1], dtype_code(y.dtype))
    return out


def head2_infer(x, hidden, out):
    """No-grad 2-channel head in ONE launch (cn_head2_fwd): -> fp32 NCHW [N, 2, H, W], or None when the kernel declines the shape.
    The hidden conv's packed weights come from its own no-grad cache (nn.Conv2d.infer_key)."""
    N, H, W, Cx = x.shape
    Ch = hidden.weight.shape[0]
    if Cx != 64 or hidden.weight.shape[1] != 64 or Ch % 64 or out.weight.shape[0] != 2:
        return None
    if _os.environ.get("CN_DISABLE_HEAD2") or torch.are_deterministic_algorithms_enabled():
        return None                 # the one-launch form sums per-wave partials with fp32 atomics: not bit-reproducible run to run
    c = hidden.infer_key(x)
    y = zeros((N, 2, H, W), torch.float32, x.device)
    w2 = out.weight.detach().reshape(2, Ch)
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    if _hip.try_call("cn_head2_fwd", x, c["wp"], c["b"], w2, out.bias.detach(), y, N, H, W, 64, Cx, Ch, dtype_code(x.dtype)):
        return y
    return None


class SparseRows:
    """Side channel next to autograd for gradients that are dense tensors by contract but zero outside a few known pixels: the
    backward of a gather-type loss `note`s (its dense gradient map, the gathered indices); a consumer that can work on rows
    (`HeadFn.backward`) `take`s the indices when the tensor autograd hands it IS that very map, unmodified — same Python object
    (the registry's reference keeps it alive, so the engine neither frees nor accumulates into it in place) and same version
    counter.  Anything else (the map was summed with another gradient, a different tensor) finds nothing and runs dense."""
    entries = []
    enabled = not _os.environ.get("CN_DISABLE_SPARSE_HEAD_BWD")
    _lock = __import__("threading").Lock()      # autograd may run backward nodes of several graphs on several threads

    @classmethod
    def note(cls, t, ind):
        if not cls.enabled or t.dim() != 4 or not t.is_cuda:
            return
        with cls._lock:
            if len(cls.entries) >= 16:
                cls.entries.pop(0)
            first = not cls.entries
            cls.entries.append((t, t._version, ind))
        if first:
            # an entry nobody takes (the consumer is not a HeadFn: a plain-torch head, CN_DISABLE_HEAD_FN) must not pin its
            # gradient map (335 MB at C3) beyond the backward pass that made it: drop what is left when this pass ends
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.clear)
            except RuntimeError:              # not inside a backward pass (tests calling backward() of a Function by hand)
                pass

    @classmethod
    def take(cls, g):
        with cls._lock:
            for i, (t, v, ind) in enumerate(cls.entries):
                if t is g:
                    cls.entries.pop(i)
                    return ind if g._version == v else None
        return None

    @classmethod
    def clear(cls):
        with cls._lock:
            del cls.entries[:]


class DualLayout(SparseRows):
    """The same side channel for a gradient map whose producer also left it in a second layout: note(fp32 NCHW map, NHWC copy in
    the compute dtype); `take` hands the copy to the consumer that receives that very map."""
    entries = []
    enabled = not _os.environ.get("CN_DISABLE_DUAL_LAYOUT_GRAD")
    _lock = __import__("threading").Lock()


class HeadFn(Function):
    """One task head (heads.py:4-25): conv3x3 + bias -> ReLU -> conv1x1 + bias -> public NCHW fp32 map, as ONE autograd node.

    Forward: two launches — the 3x3 conv with the ReLU in its epilogue, and the 1x1 conv writing the public NCHW fp32 map itself
    (`conv1x1_to_nchw`; where that kernel declines the shape: NHWC conv + layout change, as the separate nodes did).  Backward:
    when the incoming gradient is the map of a gather-type loss (`SparseRows`: zero except at ind[b, :], <= 128 of the 16 384
    pixels of a 512x512 image) everything is computed on the R = B*M rows that can be non-zero — csrc/head_sparse.hip: compact
    operands by cn_head_sparse_gather, both weight gradients and the nine-tap data gradient as small 1x1 GEMMs over R "pixels",
    the data gradient scattered into the shared input's gradient sum — instead of pushing a 537 MB hidden-layer gradient through
    two dense convolutions per head.  Any other gradient takes the dense path (`_conv2d_bwd` twice, exactly what the separate
    nodes ran).  The sums are the reference's (heads.py under autograd), taken in a different order."""

    sparse_runs = 0       # backward passes that took the row path (tests / profiling)
    fused2 = not _os.environ.get("CN_DISABLE_HEAD2")      # A/B: 2-channel heads through the one-launch forward (cn_head2_fwd)

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        Ch, Ci, KH, KW = w1.shape
        C = w2.shape[0]
        N, H, W, Cx = x.shape
        assert (KH, KW) == (3, 3) and tuple(w2.shape[1:]) == (Ch, 1, 1) and Cx == rup(Ci, 16)
        wp1 = pack_weight(w1, 1, x.dtype)
        h = out = None
        if (HeadFn.fused2 and not torch.are_deterministic_algorithms_enabled() and C == 2 and x.dtype == torch.bfloat16 and Ci == 64 and Cx == 64 and Ch % 64 == 0 and b1 is not None
                and b2 is not None):
            # a 2-channel head (width_height / regression) in ONE launch: ReLU + 1x1 in the 3x3 kernel's epilogue, the hidden
            # activation (537 MB at C3) is never stored; its backward recomputes the few hidden rows it needs from the input patches
            w2c = w2.detach().reshape(2, Ch)
            out = zeros((N, 2, H, W), torch.float32, x.device)
            if not _hip.try_call("cn_head2_fwd", x, wp1, b1.detach(), w2c if w2c.is_contiguous() else w2c.contiguous(), b2.detach(), out,
                                 N, H, W, 64, Cx, Ch, dtype_code(x.dtype)):
                out = None
        if out is None:
            h = _igemm(x, wp1, b1, None, Ch, 3, 3, 1, 1, False, True, H, W)
            out = conv1x1_to_nchw(h, pack_weight(w2, 1, x.dtype), b2, C)
        ctx.has_h = h is not None
        ctx.save_for_backward(x, h if h is not None else x.new_empty(0), w1, w2)
        ctx.refs = (b1, b2)
        ctx.ld2 = rup(C, 16)
        ctx.orders = (SideGrads.next_order(), SideGrads.next_order())
        ctx.cell = cell_of(x)
        return out

    @staticmethod
    def backward(ctx, g):
        x, h, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.refs
        need = ctx.needs_input_grad
        N, H, W, Cx = x.shape
        Ch, Ci = w1.shape[:2]
        C = w2.shape[0]
        ind = SparseRows.take(g)
        if (ind is not None and g.dtype == torch.float32 and g.is_contiguous() and ind.shape[0] == N and C <= 64 and Cx == Ci
                and (not ctx.has_h or h.shape[-1] == Ch) and 4 * ind.shape[1] <= H * W):
            return HeadFn._backward_rows(ctx, g, ind.contiguous())
        if not ctx.has_h:      # dense gradient behind a one-launch forward (not a gather-type loss): the hidden activation is recomputed
            h = _igemm(x, pack_weight(w1, 1, x.dtype), b1.detach(), None, Ch, 3, 3, 1, 1, False, True, H, W)
        dyn = DualLayout.take(g)          # the loss's backward may have left the map in this layout already (SigmoidFocalFn)
        if dyn is None or dyn.dtype != x.dtype or tuple(dyn.shape) != (N, H, W, ctx.ld2):
            dyn = torch.empty((N, H, W, ctx.ld2), dtype=x.dtype, device=x.device)
            call("cn_nchw_to_nhwc", g.contiguous().float(), dyn, N, C, H, W, ctx.ld2, dtype_code(x.dtype))
        dh, dw2, db2 = _conv2d_bwd(h, w2, b2, dyn, 1, 0, True, True, None, ctx.orders[1], (need[0] or need[1] or need[2], need[3], need[4]))
        dx = dw1 = db1 = None
        if dh is not None:
            dx, dw1, db1 = _conv2d_bwd(x, w1, b1, dh, 1, 1, True, False, ctx.cell, ctx.orders[0], need[:3])
        return dx, dw1, db1, dw2, db2

    @staticmethod
    def _backward_rows(ctx, g, ind):
        x, h, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.refs
        need = ctx.needs_input_grad
        N, H, W, Cx = x.shape
        Ch, Ci = w1.shape[:2]
        C, M = w2.shape[0], ind.shape[1]
        R, K, Cq, dt = N * M, 9 * Ci, rup(C, 16), x.dtype
        HeadFn.sparse_runs += 1
        dhc = torch.empty((1, 1, R, Ch), dtype=dt, device=x.device)
        xg = torch.empty((1, 1, R, K), dtype=dt, device=x.device)
        gq = torch.empty((1, 1, R, Cq), dtype=dt, device=x.device)
        w2c = w2.detach().contiguous()
        if ctx.has_h:
            hg = torch.empty((1, 1, R, Ch), dtype=dt, device=x.device)
            call("cn_head_sparse_gather", h, x, ind, g, w2c, hg, dhc, xg, gq, N, M, C, H, W, Ch, h.shape[-1], Ci, Cx, Cq, dtype_code(dt))
        else:
            # no hidden activation was stored: patches first, the R hidden rows relu(patch W1^T + b1) as a 1x1 convolution over R
            # "pixels" (w1 read as [Ch, Ci*9], the patches' own column order), then the masked hidden gradient
            call("cn_head_sparse_gather_rows", None, x, ind, g, w2c, None, xg, gq, N, M, C, H, W, Ch, Ch, Ci, Cx, Cq, 1, dtype_code(dt))
            hg = _igemm(xg, pack_weight(w1.detach().view(Ch, K, 1, 1), 1, dt), b1.detach(), None, Ch, 1, 1, 1, 0, False, True, 1, R)
            call("cn_head_sparse_gather_rows", hg, x, ind, g, w2c, dhc, None, None, N, M, C, H, W, Ch, hg.shape[-1], Ci, Cx, Cq, 2, dtype_code(dt))
        dw1 = db1 = dw2 = db2 = dx = None
        if need[1] or need[3]:
            if SideGrads.usable(w1, b1, w2, b2):
                def side_work():
                    _wgrad_param(hg, gq, C, Ch, 1, 1, 1, 0, into=w2.grad, db_into=b2.grad)
                    _wgrad_param(xg, dhc, Ch, K, 1, 1, 1, 0, into=w1.grad.view(Ch, K, 1, 1), db_into=b1.grad)   # K = ci*9 + tap: w1's own layout
                    GradReady.note(w1, b1, w2, b2)
                SideGrads.submit(side_work, hg, gq, xg, dhc, claims=(w1, b1, w2, b2))
            else:
                dw2, db2 = _wgrad_param(hg, gq, C, Ch, 1, 1, 1, 0, want_bias=True)
                dw1, db1 = _wgrad_param(xg, dhc, Ch, K, 1, 1, 1, 0, want_bias=True)
                dw1 = dw1.view(Ch, Ci, 3, 3)
        if need[0]:
            # the nine-tap data gradient of the rows = the data gradient of the 1x1 convolution whose weight is w1 read as [Ch, Ci*9]
            wpd = pack_weight(w1.detach().view(Ch, K, 1, 1), 0, dt)
            dxc = _igemm(dhc, wpd, None, None, K, 1, 1, 1, 0, True, False, 1, R, out_dtype=torch.float32)

            def scatter(total, dxc=dxc):
                call("cn_scatter3x3_add", dxc, ind, total, N, M, H, W, Ci, total.shape[-1], dtype_code(total.dtype))
            if ctx.cell is not None:
                ctx.cell.sparse.append(scatter)       # applied to the finished sum of x's consumers (ShareFn.backward)
            else:
                dx = zeros_like(x)
                scatter(dx)
        return dx, dw1, db1, dw2, db2


class GatherL1Fn(Function):
    """utils/losses.py:53-63 / 81-91: masked L1 between rows gathered at `ind` and the targets."""

    @staticmethod
    def forward(ctx, feat, mask, ind, target):
        feat = feat.contiguous()
        B, C = feat.shape[:2]
        HW = feat[0, 0].numel()
        N = ind.shape[1]
        mask8 = mask.contiguous()
        mask8 = mask8.view(torch.uint8) if mask8.dtype == torch.bool else mask8.to(torch.uint8)    # bool is one byte: no copy kernel
        has_c = int(mask.dim() == 3)
        ind = ind.contiguous().long()
        target = target.contiguous().float()
        out = torch.empty(3, dtype=torch.float32, device=feat.device)
        call("cn_gather_l1_fwd", feat, ind, mask8, target, out, B, C, HW, N, has_c)
        ctx.save_for_backward(feat, ind, mask8, target, out)
        ctx.has_c = has_c
        return out[0]

    @staticmethod
    def backward(ctx, g):
        feat, ind, mask8, target, out = ctx.saved_tensors
        B, C = feat.shape[:2]
        dfeat = zeros_like(feat)
        call("cn_gather_l1_bwd", feat, ind, mask8, target, out, g.contiguous().float().reshape(1), dfeat, B, C,
             feat[0, 0].numel(), ind.shape[1], ctx.has_c)
        SparseRows.note(dfeat, ind)       # zero outside ind[b, :]: a HeadFn behind `feat` works on those rows only
        return dfeat, None, None, None


def _dwdeconv_wgrad(x, dy, dw, N, H, W, C, k, stride, pad, OH, OW):
    """depthwise up-conv weight gradient: the row-walking slab kernel where it takes the shape (the x2 layers), else the general one"""
    n = _hip.query("cn_dwdeconv_wgrad_ws_bytes", N, OH, C)
    if n and x.dtype == torch.bfloat16:
        ws = _hip.workspace(n, x.device, "dwwgrad")
        if _hip.try_call("cn_dwdeconv_bwd_weight_rows", x, dy, dw, ws, n, N, H, W, C, k, stride, pad, OH, OW, dtype_code(x.dtype),
                         hooks=SideGrads.wgrad_hooks()):
            return
    call("cn_dwdeconv_bwd_weight", x, dy, dw, N, H, W, C, k, stride, pad, OH, OW, dtype_code(x.dtype))


class WeightedSumFn(Function):
    """total = sum_i w_i * term_i over scalar loss terms: ONE launch forward and ONE backward (cn_weighted_sum) instead of the ~16
    ATen scalar kernels of `hm_weight * hm_loss + wh_weight * wh_loss + ...` (centernet_detection.py:108-116), which sit at the
    forward / backward seam of a step where nothing overlaps them."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights = tuple(float(w) for w in weights)
        n = len(terms)
        out = torch.empty((), dtype=torch.float32, device=terms[0].device)
        pad = [None] * (8 - n)
        call("cn_weighted_sum", *terms, *pad, *ctx.weights, *([0.0] * (8 - n)), n, out)
        return out

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        out = torch.empty((8,), dtype=torch.float32, device=g.device)
        call("cn_weighted_sum_bwd", g.contiguous().float(), *ctx.weights, *([0.0] * (8 - n)), n, out)
        return (None,) + tuple(out[i] for i in range(n))


def weighted_sum(terms, weights):
    """sum_i weights[i] * terms[i] for scalar tensors (python numbers among the terms are folded on the host)"""
    ts, ws, const = [], [], 0.0
    for t, w in zip(terms, weights):
        if isinstance(t, torch.Tensor):
            ts.append(t); ws.append(w)
        else:
            const += float(t) * float(w)
    dev_ok = (0 < len(ts) <= 8 and all(t.is_cuda and t.dim() == 0 and t.dtype == torch.float32 for t in ts))
    if not dev_ok or const != 0.0:
        tot = const
        for t, w in zip(ts, ws):
            tot = tot + w * t
        return tot
    return WeightedSumFn.apply(tuple(ws), *ts)


# ------------------------------------------------------------------------------------------------ functional aliases
def conv2d(x, weight, bias=None, stride=1, pad=0, relu=False, mask_dx=False, defer_relu_bwd=False, passthrough=False, bn_stats=False, pre=None):
    """bn_stats: the output feeds a training-mode BatchNorm — ask the kernel for the batch statistics (BnStats);
    pre: x is a raw conv output whose BN (+ ReLU) the kernels apply on load (batch_norm_defer)"""
    out = Conv2dFn.apply(x, weight, bias, stride, pad, relu, mask_dx, defer_relu_bwd, passthrough, bn_stats, pre)
    if bn_stats:
        BnStats.pop(out[0] if passthrough else out)
    if passthrough and cell_of(x) is not None:
        out[1]._grad_cell = cell_of(x)       # the skip path of a shared tensor reports to the same cell (its consumer runs first)
    return out


def conv_transpose2d(x, weight, stride=2, pad=1):
    return ConvTranspose2dFn.apply(x, weight, stride, pad)


def batch_norm_act(x, bn, residual=None, relu=True):
    return BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, relu, getattr(x, "_bn_part", None))


def max_pool(x, k, stride, pad=0):
    return MaxPoolFn.apply(x, k, stride, pad)


def add(a, b):
    return AddFn.apply(a, b)


def upsample2x_add(a, low):
    return UpsampleAddFn.apply(a, low)


def concat(xs):
    return xs[0] if len(xs) == 1 else ConcatFn.apply(*xs)
