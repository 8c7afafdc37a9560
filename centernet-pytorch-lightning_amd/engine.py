"""Training runtime around the hot path: fused flat-buffer Adam, RCCL data-parallel gradient exchange, a small trainer.

Data parallelism follows Lightning-DDP semantics: one process per GPU, each rank normalises its loss by
its LOCAL num_pos / mask sums, gradients are SUM-all-reduced over RCCL (torch.distributed backend "nccl") in flat
~25 MB buckets launched from grad-ready hooks so they overlap the rest of backward, and the 1/world average is folded
into the Adam kernel's gradient scale (no extra pass).  BN statistics stay per-rank (no SyncBN in the reference).
"""
import os

import torch
import torch.distributed as dist

from . import _hip
from . import ops
from .ops import SideGrads, PackArena, WeightsEpoch, GradReady


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(defaults) semantics on ONE contiguous fp32 parameter buffer -> one cn_adam_step launch.

    Parameters are re-pointed at views of `flat_p` and their `.grad` at views of `flat_g`, so autograd accumulates
    straight into the flat gradient buffer that the bucketed all-reduce and the Adam kernel consume.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params if p.requires_grad]
        # the remaining keys are torch.optim.Adam's defaults (what the reference runs with): carried in the param group so that
        # a checkpoint of this optimizer loads into torch.optim.Adam unchanged
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        self.params = params
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.offsets, self.numel = offs, n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)
        self.t = 0
        self.grad_scale = 1.0
        # {lr, 1-b1^t, 1-b2^t, t} live in HBM and are advanced ON THE DEVICE (cn_adam_advance) inside the captured graph:
        # a host that runs many replays ahead must not race per-step uploads
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._lr_dev = None

    def zero_grad(self, set_to_none=False):
        if self.flat_g.is_cuda and SideGrads.stamps is not None and not os.environ.get("CN_STAMP_END_ONLY"):
            ops.call("cn_stamp", SideGrads.stamps)
        if self.flat_g.is_cuda:
            ops.call("cn_zero", self.flat_g, self.flat_g.numel() * 4)      # no ATen fill inside the (captured) step
        else:
            self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):   # autograd may have replaced .grad (e.g. after set_to_none elsewhere)
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)

    @torch.no_grad()
    def prepare_step(self):
        """Host half of a step (never captured in a graph): mirror t, upload the learning rate only when it changed."""
        g = self.param_groups[0]
        self.t += 1
        WeightsEpoch.bump()             # the parameters are about to change through raw pointers: eval-mode caches expire
        if self._lr_dev != float(g["lr"]):
            self._lr_dev = float(g["lr"])
            self.hyper[0:1].fill_(self._lr_dev)

    @torch.no_grad()
    def set_step(self, t):
        """restore the step counter (checkpoint resume): host mirror + device copy"""
        self.t = int(t)
        self.hyper[3:4].copy_(torch.tensor([self.t], dtype=torch.int32).view(torch.float32))

    def state_dict(self):
        """torch.optim.Adam's checkpoint format (`state[i] = {step, exp_avg, exp_avg_sq}` per parameter, `param_groups`), so a
        checkpoint written under this optimizer resumes under torch.optim.Adam (the reference's, centernet.py:94-95) and back.
        The moments are copies of the views into the flat buffers."""
        state = {}
        if self.t > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.t)),
                            "exp_avg": self.flat_m[o:o + n].view_as(p).clone(),
                            "exp_avg_sq": self.flat_v[o:o + n].view_as(p).clone()}
        groups = [{**{k: v for k, v in g.items() if k != "params"}, "params": list(range(len(self.params)))}
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Inverse of `state_dict` (also accepts a torch.optim.Adam checkpoint of the same parameter list): moments into the
        flat buffers, step counter to the host mirror and the device copy, lr / betas / eps from the param group."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one param group with "
                             f"{len(self.params)} parameters, got {[len(g['params']) for g in groups]}")
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise NotImplementedError("FlatAdam implements torch.optim.Adam's defaults only (no weight decay / amsgrad / maximize)")
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = v
        self.flat_m.zero_()
        self.flat_v.zero_()
        t = 0
        ids = groups[0]["params"]
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(ids[i], sd["state"].get(str(ids[i])))
            if st is None:
                continue
            n = p.numel()
            self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            t = max(t, int(float(st["step"])))
        if self.flat_p.is_cuda:
            self.set_step(t)
        else:
            self.t = t
        self._lr_dev = None              # re-upload the learning rate on the next step

    @torch.no_grad()
    def launch(self):
        """Device half: one fused kernel over the flat buffers (hipGraph-capturable)."""
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        _hip.call("cn_adam_advance", self.hyper, float(b1), float(b2))
        _hip.call("cn_adam_step", self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.numel, float(g["lr"]),
                  float(b1), float(b2), float(g["eps"]), 1.0, 1.0, float(self.grad_scale), self.hyper)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if self.flat_p.is_cuda:
            self.prepare_step()
            self.launch()
        else:  # host logic tests (gloo / CPU): same arithmetic with torch ops
            self.t += 1
            WeightsEpoch.bump()
            gr = self.flat_g * self.grad_scale
            self.flat_m.mul_(b1).add_(gr, alpha=1 - b1)
            self.flat_v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
            denom = self.flat_v.sqrt() / (1.0 - b2 ** self.t) ** 0.5 + g["eps"]
            self.flat_p.addcdiv_(self.flat_m, denom, value=-g["lr"] / (1.0 - b1 ** self.t))


class GradSync:
    """Bucketed SUM all-reduce of FlatAdam.flat_g across ranks (RCCL on GPUs, gloo on CPU), overlapped with backward in BOTH launch
    modes.  A bucket (~25 MB of consecutive parameters, built in reverse parameter order = the order backward produces them) is
    exchanged as soon as its last gradient has been ENQUEUED.  Gradients reach the flat buffer three ways — autograd's accumulation
    (post-accumulate hook), weight-gradient kernels on the side stream, the BN backward on the launch stream — and the last two
    report through `ops.GradReady`.  The collective is issued from a small fork stream that first waits for everything enqueued
    so far on the launch stream AND on the weight-gradient stream (a bucket mixes both), so neither of those two streams ever
    waits for the other or for the network.  Under hipGraph capture the fork, the RCCL kernels and the join are captured into the
    step's graph: the replayed step overlaps its exchange exactly like the eager one."""

    def __init__(self, opt, bucket_bytes=25 << 20, group=None, hooks=True):
        self.opt, self.group = opt, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        opt.grad_scale = 1.0 / self.world
        # CN_FORCE_EXCHANGE: run the collectives on a 1-rank group too (exercises RCCL next to the hipGraphs on a 1-GPU box)
        self.exchange = self.world > 1 or bool(os.environ.get("CN_FORCE_EXCHANGE"))
        self.buckets = []      # (start, end, [param indices]) over the flat buffer, built in REVERSE parameter order
        cur, size, end = [], 0, opt.numel
        for i in reversed(range(len(opt.params))):
            cur.append(i)
            size += opt.params[i].numel() * 4
            if size >= bucket_bytes or i == 0:
                self.buckets.append((opt.offsets[i], end, cur))
                end, cur, size = opt.offsets[i], [], 0
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self.index = {id(p): i for i, p in enumerate(opt.params)}
        self.live = None       # params that receive gradients (learned on the first backward)
        self._seen, self._pending, self._works, self._launched = set(), [], [], set()
        self._fork = self._main = None
        self.launch_log = []   # (bucket, #parameters noted when it was launched) of the last backward: tests read the interleaving
        self._claimed = set()
        if self.exchange and hooks:
            for p in opt.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        """autograd accumulated (or was handed None for) `p`: counts unless a deferred direct deposit has claimed the parameter —
        the hook fires when the autograd NODE returns, which is before a side-stream closure has enqueued its kernels"""
        if self.index.get(id(p)) not in self._claimed:
            self.note(p)

    def claim(self, p):
        i = self.index.get(id(p))
        if i is not None:
            self._claimed.add(i)

    def note(self, p):
        """parameter `p`'s gradient has been enqueued (each parameter is produced once per backward in these networks)"""
        i = self.index.get(id(p))
        if i is None or i in self._seen or not self._pending:
            return
        self._seen.add(i)
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0 and self.live is not None:
            self._launch(b)

    def _launch(self, b):
        if b in self._launched:
            return
        self._launched.add(b)
        self.launch_log.append((b, len(self._seen)))
        s, e, _ = self.buckets[b]
        buf = self.opt.flat_g[s:e]
        if buf.is_cuda:
            if self._fork is None:
                self._fork = torch.cuda.Stream()
            self._fork.wait_stream(self._main)
            for st in SideGrads.all_streams():
                self._fork.wait_stream(st)
            with torch.cuda.stream(self._fork):
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(work)

    def begin(self):
        """Call before backward (on the stream the step is launched on)."""
        self._seen, self._works, self._launched, self.launch_log, self._claimed = set(), [], set(), [], set()
        live = self.live
        self._pending = [sum(1 for i in idx if live is None or i in live) for _, _, idx in self.buckets]
        if self.exchange:
            if self.opt.flat_g.is_cuda:
                self._main = torch.cuda.current_stream()
            GradReady.sink, GradReady.claim_sink = self.note, self.claim

    def finish(self):
        """Call after backward (and after the weight-gradient stream was joined): launches whatever is left (first step / dead
        parameters), then makes the launch stream wait for every bucket."""
        GradReady.sink = GradReady.claim_sink = None
        if not self.exchange:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self._works:
            w.wait()
        if self.live is None:
            self.live = set(self._seen)
            # one-time audit (first step only; one host sync): a parameter nobody reported must really be dead — an un-reported
            # deposit site would let its bucket leave before the gradient is written
            for i, (p, o) in enumerate(zip(self.opt.params, self.opt.offsets)):
                if i not in self.live and bool(self.opt.flat_g[o:o + p.numel()].any()):
                    raise RuntimeError(f"GradSync: parameter #{i} {tuple(p.shape)} received a gradient that no hook / "
                                       "ops.GradReady.note reported")
        self._pending = []

    def abort(self):
        """backward raised: drop this pass's bookkeeping so that nothing installed by begin() outlives it"""
        GradReady.sink = GradReady.claim_sink = None
        self._pending, self._works = [], []
        self._seen, self._launched, self._claimed = set(), set(), set()

    def allreduce_all(self):
        """Non-overlapped variant (used between the two captured graphs): every bucket, then wait."""
        if not self.exchange:
            return
        # nothing to overlap with between the two graphs: ONE collective over the whole flat gradient (79 MB for DLA-34) pays
        # the ring latency once instead of once per bucket
        dist.all_reduce(self.opt.flat_g, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_state(self, module):
        """DDP init: parameters (flat) + buffers from rank 0."""
        if not self.exchange:
            return
        dist.broadcast(self.opt.flat_p, 0, group=self.group)
        for buf in module.buffers():
            dist.broadcast(buf, 0, group=self.group)


def _watchdog_backlog():
    """Collectives the NCCL / RCCL process groups' watchdog threads have not retired yet (enqueued - completed sequence numbers of
    every group, from the flight recorder's status block), or None when this build does not report it."""
    try:
        import pickle
        from torch._C import _distributed_c10d as c10d
        # only the status block is wanted: without the collectives the recorder neither pickles its (up to 2000) entries nor queries
        # the events of unretired ones on every poll (round-5 ADVICE)
        st = pickle.loads(c10d._dump_nccl_trace(includeCollectives=False, includeStackTraces=False, onlyActive=True)).get("pg_status")
    except Exception:
        return None
    if not isinstance(st, dict):
        return None
    if not st:
        # no status block: either no NCCL group exists, or the flight recorder is off (TORCH_NCCL_TRACE_BUFFER_SIZE unset when the
        # group was created — init_distributed sets it) and nothing can be read off it
        nccl = dist.is_initialized() and torch.cuda.is_available() and "nccl" in str(dist.get_backend()).lower()
        return None if nccl else 0
    backlog = 0
    for v in st.values():
        try:
            backlog += max(0, int(v["last_enqueued_collective"]) - int(v["last_completed_collective"]))
        except Exception:
            return None
    return backlog


def drain_watchdog(timeout=5.0):
    """Block until the process groups' watchdog threads have retired every collective issued so far (the caller has already
    synchronised the device, so they are complete: this waits for the watchdog's next pass, which drops them from its poll list).
    Deterministic where the status block is available; otherwise (or on timeout) three watchdog periods of sleep."""
    import time
    if not torch.cuda.is_available():
        return True          # gloo: no device events, nothing polls during a capture
    end = time.monotonic() + timeout
    while True:
        n = _watchdog_backlog()
        if n is None:
            break
        if n == 0:
            return True
        if time.monotonic() > end:
            break
        time.sleep(0.002)
    import sys
    print("[centernet_amd] drain_watchdog: the process group's status block is unavailable or did not drain in time; "
          "sleeping 0.35 s (three watchdog periods) before the capture instead", file=sys.stderr, flush=True)
    time.sleep(0.35)
    return False


def init_distributed():
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and join the RCCL world."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get("CN_FORCE_EXCHANGE")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # the flight recorder's status block is how drain_watchdog learns that the watchdog has retired the warm-up collectives
        # (the group only keeps it when a trace buffer is configured at creation)
        # (CN_NO_FLIGHT_RECORDER=1 leaves the recorder as the user configured it: drain_watchdog then falls back to its timer and says so)
        if not os.environ.get("CN_NO_FLIGHT_RECORDER"):
            os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")
            os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
    return rank, local, world


class HostFeed:
    """Pinned host batches -> HBM one step ahead, on a copy stream of its own (what the reference leaves to Lightning's
    `batch.to(device)`, on the launch stream).  `put(batch)` starts the host->device copy of the NEXT batch while the current
    step computes; `get()` makes the launch stream wait for that copy only.  Batches are `(image, {name: tensor})`; tensors
    that are already on the device pass through."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._q = []

    def _to_dev(self, t):
        if t.device == self.device:
            return t
        assert t.is_pinned(), "HostFeed wants pinned host memory (pageable copies serialise with the launch stream)"
        return t.to(self.device, non_blocking=True)

    def put(self, batch):
        x, tgt = batch
        with torch.cuda.stream(self.stream):
            dev = (self._to_dev(x), {k: self._to_dev(v) for k, v in tgt.items()})
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._q.append((dev, ev))

    def get(self):
        dev, ev = self._q.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in (dev[0], *dev[1].values()):
            t.record_stream(cur)              # allocated on the copy stream, consumed on the launch stream
        return dev


class TrainStep:
    """forward + loss + backward (+ gradient exchange) + Adam, i.e. what Lightning's loop does around
    `CenterNet.training_step` (centernet.py:70-80).

    graph=False: eager launches.
    graph=True : the ~1150 launches of a step are captured once into two hipGraphs (zero_grad+forward+loss+backward incl. the
                 gradient exchange | Adam+post_step) and replayed, which removes the launch-bound gaps.
    In both modes the RCCL buckets are issued as soon as their gradients are enqueued and overlap the rest of backward
    (`GradSync`); weight gradients run on the side stream in both modes.
    `post_step` (optional callable, no grad) runs after the optimizer.  `post_forward` (optional callable, no grad) is
    forked onto its own stream right after the forward pass and joined at the end of the step, so work that only needs
    the forward outputs (the bench's ctdet_decode of the head maps) overlaps backward instead of trailing it.
    """

    def __init__(self, model, lr=None, distributed=None, graph=False, post_step=None, side_grads=True, post_forward=None,
                 adopt_batch=False):
        """adopt_batch (graph mode): use the FIRST batch's own device tensors as the captured graph's static inputs instead of
        private copies.  A caller that feeds the same resident tensors every step (a benchmark on synthetic data) then pays no
        device-to-device copy per step; any other batch is still copied into those tensors — i.e. INTO the first batch's
        storage, which is why this is opt-in."""
        self.model = model
        self.adopt_batch = adopt_batch
        self.post_forward, self._pf_stream = post_forward, None
        self._packs = PackArena()
        lr = lr if lr is not None else getattr(model.hparams, "learning_rate", 1e-4)
        self.opt = FlatAdam(model.parameters(), lr=lr)
        self.graph, self.post_step, self.post_out = graph, post_step, None
        use_dist = dist.is_initialized() if distributed is None else distributed
        self.sync = GradSync(self.opt) if use_dist else None
        # escape hatch: keep the collectives OUT of the captured graph (one all-reduce over the whole flat gradient between the two
        # graphs, round 1's scheme) if a runtime mishandles captured RCCL kernels; only read in graph mode
        self._between = bool(os.environ.get("CN_EXCHANGE_BETWEEN_GRAPHS")) and graph
        # weight gradients on a second stream, deposited straight into the flat gradient buffer (they report to the exchange
        # through ops.GradReady, so data parallelism keeps them)
        self.side = SideGrads.enable(side_grads and self.opt.flat_p.is_cuda and not os.environ.get("CN_NO_SIDE"),
                                     fp32=getattr(model, "compute_dtype", None) == torch.float32)
        if self.sync is not None:
            self.sync.broadcast_state(model)
        self._g1 = self._g2 = None
        TrainStep._ns_next += 1
        self._ns = ("step", TrainStep._ns_next)      # namespace of this step's BatchNorm statistics sinks (ops.BnStats)
        self._inflight = None
        self._bns = [m for m in model.modules() if hasattr(m, "_pending")]

    def _fork_post_forward(self):
        if self.post_forward is None:
            return
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream()
        self._pf_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._pf_stream), torch.no_grad():
            self.post_out = self.post_forward()

    def _join_post_forward(self):
        if self.post_forward is not None:
            torch.cuda.current_stream().wait_stream(self._pf_stream)

    def _begin_packs(self):
        """one launch packs every weight operand of the step (recorded during the first step)"""
        SideGrads.fwd_order = 0
        if not self.opt.flat_p.is_cuda:
            return
        ops.FarFlags.begin(self.opt.flat_p.device, id(self))
        PackArena.current = self._packs
        if self._packs.table is not None:
            self._packs.repack()
        elif not self._packs.slots:
            self._packs.recording = True

    def _end_packs(self):
        if self._packs.recording:
            self._packs.build()
        PackArena.current = None
        ops.FarFlags.end()

    def _eager(self, batch, batch_idx=0):
        # this step's chain of BatchNorm statistics sinks, kept apart from every other chain in the process.  Once a graph has
        # been captured its namespace belongs to the REPLAYS (the graph's baked-in sink pointers assume that nothing else dirties
        # or retires those sinks): eager steps of the same TrainStep (the bench's probe steps, the capture-failure fallback) get a
        # namespace of their own, and the previous namespace comes back on exit, so BN passes outside any step (a train-mode
        # forward under no_grad, recalibration) never touch a step's rings (round-3 ADVICE).
        ns_was, ops.BnStats.ns = ops.BnStats.ns, (self._ns if self._g1 is None else self._ns + ("eager",))
        try:
            return self._eager_body(batch, batch_idx)
        finally:
            ops.BnStats.ns = ns_was

    def _eager_body(self, batch, batch_idx=0):
        self.opt.zero_grad()
        self._begin_packs()
        try:
            loss = self.model.training_step(batch, batch_idx)
        except BaseException:
            # a producer may have filled a BatchNorm statistics sink that its BN never got to consume (the sinks are persistent and
            # must be all-zero when handed out): clear them, or the next training-mode BN of that width normalises with stale sums
            ops.BnStats.reset()
            PackArena.current, self._packs.recording = None, False
            ops.FarFlags.end()
            raise
        self._fork_post_forward()
        if self.sync is not None and not self._between:
            self.sync.begin()
        SideGrads.active = self.side
        try:
            loss.backward()
            SideGrads.join()
        except BaseException:
            # a failed backward (OOM, kernel error) must not leave this step's deposit sinks installed: a later backward in the
            # process would report into this GradSync and could launch collectives the other ranks never match
            self._abort_backward()
            raise
        self._end_packs()
        self._join_post_forward()
        if self.sync is not None:
            self.sync.allreduce_all() if self._between else self.sync.finish()
        self.opt.step()
        if self.post_step is not None:
            with torch.no_grad():
                self.post_out = self.post_step()
        return loss.detach()

    def _abort_backward(self):
        SideGrads.pending, SideGrads.active = [], False
        GradReady.sink = GradReady.claim_sink = None
        PackArena.current, self._packs.recording = None, False
        ops.FarFlags.end()
        ops.BnStats.reset()
        if self.sync is not None:
            self.sync.abort()

    def _capture(self, batch):
        x, tgt = batch
        dev = self.opt.flat_p.device           # the static batch lives in HBM; later batches may arrive in (pinned) host memory
        own = lambda t: t if (self.adopt_batch and t.device == dev and t.is_contiguous()) else t.to(dev, copy=True)
        self._sx, self._st = own(x), {k: own(v) for k, v in tgt.items()}
        static = (self._sx, self._st)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # the two warm-up steps below are real steps on the first batch; one step per batch is the contract (and what eager mode
        # does), so the optimizer state, the parameters and the BN buffers are put back before the capture
        torch.cuda.synchronize()
        opt = self.opt
        snap = [t.clone() for t in (opt.flat_p, opt.flat_m, opt.flat_v, opt.hyper)]
        bufs = [b for b in self.model.buffers()]
        snap_b = [b.clone() for b in bufs]
        t0, pend = opt.t, [m._pending for m in self._bns]
        try:
            with torch.cuda.stream(side):            # warm-up off the capture stream: workspaces, lazy attributes, allocator
                for _ in range(2):
                    self._eager(static)
        finally:
            # ALWAYS put the snapshot back: a warm-up step that raised on this rank only (round-5 ADVICE) would otherwise leave this
            # rank's parameters / Adam state / BN buffers one or two steps ahead of its peers
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for dst, src in zip((opt.flat_p, opt.flat_m, opt.flat_v, opt.hyper), snap):
                    dst.copy_(src)
                for dst, src in zip(bufs, snap_b):
                    dst.copy_(src)
            opt.t = t0
            opt._lr_dev = None               # the restored `hyper` predates the first learning-rate upload
            for m, n in zip(self._bns, pend):
                m._pending = n
            WeightsEpoch.bump()
            torch.cuda.synchronize()
        if dist.is_initialized():
            # The process group's watchdog thread polls the completion events of every collective it has not yet retired (it wakes
            # every 100 ms); an event query landing while this stream (and RCCL's own, which joins the capture) is capturing ends the
            # process (hipErrorCapturedEvent, seen in about one of four runs of test_rccl_exchange_next_to_graphs).  The warm-up
            # steps' collectives have finished on the device (synchronize above); wait until the watchdog has RETIRED them, so that
            # it has nothing left to poll — read off its own bookkeeping, not off a timer.
            self.drained = drain_watchdog()
        fail = os.environ.get("CN_FAIL_CAPTURE")       # test hook for the eager fallback: "1" = every rank, "rank:<r>" = that rank only
        if fail and (not fail.startswith("rank:") or int(fail[5:]) == (dist.get_rank() if dist.is_initialized() else 0)):
            raise RuntimeError("CN_FAIL_CAPTURE set (test hook for the eager fallback)")
        self._g1, self._g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # with a process group alive, its watchdog thread polls events while we capture: only police THIS thread's calls
        mode = "thread_local" if (dist.is_initialized() or os.environ.get("CN_CAPTURE_THREAD_LOCAL")) else "global"
        # capture on the stream the warm-up steps ran on: `_hip.workspace` is keyed by stream, so the capture replays into the
        # buffers the warm-up sized instead of allocating a second set from the graph's private pool
        ns_was, ops.BnStats.ns = ops.BnStats.ns, self._ns
        try:
            with torch.cuda.graph(self._g1, stream=side, capture_error_mode=mode):
                self.opt.zero_grad()
                self._begin_packs()
                loss = self.model.training_step(static, 0)
                self._fork_post_forward()
                if self.sync is not None and not self._between:
                    self.sync.begin()
                SideGrads.active = self.side
                loss.backward()
                SideGrads.join()
                self._end_packs()
                self._join_post_forward()
                if self.sync is not None and not self._between:
                    self.sync.finish()         # the bucketed all-reduces are part of the captured graph
                self._loss = loss.detach()
        finally:
            ops.BnStats.ns = ns_was
        for m in self._bns:
            m._pending -= 1          # the capture pass ran host code only; nothing executed on the device
        with torch.cuda.graph(self._g2, pool=self._g1.pool(), stream=side, capture_error_mode=mode):
            self.opt.launch()
            if self.post_step is not None:
                with torch.no_grad():
                    self.post_out = self.post_step()

    def __call__(self, batch, batch_idx=0):
        if not self.graph:
            loss = self._eager(batch, batch_idx)
            if self.opt.flat_p.is_cuda:
                self._throttle()
            return loss
        if self._g1 is None:
            err = None
            pend = [m._pending for m in self._bns]
            try:
                self._capture(batch)
            except Exception as e:      # e.g. a runtime that refuses capture next to a live process group: keep training
                err = e
                self._g1 = self._g2 = None
                self._abort_backward()
                for m, n in zip(self._bns, pend):      # a capture that aborted midway leaves the BN launch counters where it stopped
                    m._pending = n
                if self.opt.flat_p.is_cuda:
                    torch.cuda.synchronize()
            # The launch mode is a COLLECTIVE decision: a rank replaying a captured graph and a rank issuing eager buckets post
            # different collective sequences (one graph launch vs one all-reduce per bucket) and would hang each other.  Every rank
            # reports whether its capture succeeded; one failure anywhere puts all of them on eager launches.
            if not self._agree(err is None):
                import sys
                why = f"{type(err).__name__}: {err}" if err is not None else "another rank's capture failed"
                print(f"[centernet_amd] hipGraph capture failed ({why}); falling back to eager launches on every rank",
                      file=sys.stderr, flush=True)
                self._g1 = self._g2 = None
                self.graph = False
                loss = self._eager(batch, batch_idx)
                if self.opt.flat_p.is_cuda:
                    self._throttle()
                return loss
        if batch[0] is not self._sx:
            self._sx.copy_(batch[0], non_blocking=True)
        for k, v in batch[1].items():
            if v is not self._st[k]:
                self._st[k].copy_(v, non_blocking=True)
        self.opt.prepare_step()
        self._g1.replay()
        if self.sync is not None and self._between:
            self.sync.allreduce_all()
        self._g2.replay()
        for m in self._bns:
            m._pending += 1
        if self.opt.flat_p.is_cuda:
            self._throttle()
        return self._loss

    def _agree(self, ok):
        """AND of `ok` over the ranks that exchange gradients with this one (no-op without a process group)"""
        if self.sync is None or self.sync.world < 2 or not dist.is_initialized():
            return bool(ok)
        # Every rank reaches this point with NO gradient bucket of its own in flight (a failed capture / warm-up went through
        # _abort_backward, which drops the pass's pending work; a successful one has joined its collectives), and the device is
        # synchronised first, so the 1-element MIN all-reduce is the next collective every rank posts on the group.
        if self.opt.flat_p.is_cuda:
            torch.cuda.synchronize()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.opt.flat_p.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.sync.group)
        return bool(int(flag.item()))

    _ns_next = 0
    MAX_STEPS_AHEAD = int(os.environ.get("CN_MAX_STEPS_AHEAD", 3))

    def _throttle(self):
        """Bound the host's run-ahead: a caller that never synchronises would otherwise enqueue replays hundreds of steps ahead
        of the GPU (each one ~1 150 kernel nodes plus their events).  The host waits for the step that was launched
        MAX_STEPS_AHEAD steps ago — never for the one it has just enqueued, so the GPU queue stays full.  (CN_MAX_STEPS_AHEAD
        overrides it; unbounded run-ahead was measured to work too — the fault first blamed on it was the top-K race.)"""
        if self._inflight is None:
            from collections import deque
            self._inflight = deque()
        ev = torch.cuda.Event()
        ev.record()
        self._inflight.append(ev)
        if len(self._inflight) > self.MAX_STEPS_AHEAD:
            self._inflight.popleft().synchronize()


class Trainer:
    """Minimal stand-in for pl.Trainer.fit on this package's modules (Lightning is not available offline)."""

    def __init__(self, max_epochs=1, limit_train_batches=None, limit_val_batches=None):
        self.max_epochs, self.limit_train_batches, self.limit_val_batches = max_epochs, limit_train_batches, limit_val_batches

    def fit(self, model, train_loader, val_loader=None):
        step = TrainStep(model)
        dev = next(model.parameters()).device
        to = lambda b: (b[0].to(dev), {k: v.to(dev) for k, v in b[1].items()})
        history = []
        # centernet.py:94-105: MultiStepLR(milestones, gamma 0.1) stepped once per epoch, on the optimizer that actually steps
        milestones = [int(m) for m in (getattr(model, "learning_rate_milestones", None) or [])]
        sched = torch.optim.lr_scheduler.MultiStepLR(step.opt, milestones=milestones) if milestones else None
        for _ in range(self.max_epochs):
            model.train()
            for i, batch in enumerate(train_loader):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                history.append(step(to(batch), i))
            if val_loader is not None:
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val_loader):
                        if self.limit_val_batches is not None and i >= self.limit_val_batches:
                            break
                        model.validation_step(to(batch), i)
            if sched is not None:
                sched.step()
        return history
