"""Headline benchmark: images/sec of (train step + decode), DLA-34 ctdet, 512x512, bs=64 per GPU, bf16 compute.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + focal/L1 losses + backward (+ RCCL gradient all-reduce overlapped with backward when N > 1) + Adam
+ ctdet_decode of that step's head maps, on one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.
`roofline` is measured live with HIP events around EVERY C-ABI launch of the step (on the launch stream): the dominant MFMA-bound
kernel template over all entry points (convs, weight gradients, the four DCNv2 kernels), plus `roofline.step` for the whole step
against both roofs.  `inference` is the north-star inference figure of the same config (eval forward + decode, hipGraph).
`cpu_baseline` times the torch-CPU oracle (port of the reference path) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


# Algorithmic work per image: forward GFLOP and the unfused bf16 activation+weight
# traffic of the forward pass; a train step is 3x both (data gradient + weight gradient).
WORK = {"dla_34": (66.20, 289.5), "res_18": (45.43, None), "resdcn_18": (None, None)}
PEAK_HBM_GBPS = 8000.0


def _mfma_flops(name, a):
    """2 * MACs of one launch of an MFMA-bound entry point (None: not an MFMA kernel), from its integer arguments."""
    if name == "cn_conv2d_fwd":
        N, H, W, Ci, x_ld, OH, OW, Co, y_ld, res_ld, KH, KW, stride, pad, transposed = a[:15]
        if transposed and stride > 1:   # only the taps of the output pixel's parity class are visited
            return 2.0 * N * H * W * KH * KW * Ci * Co
        return 2.0 * N * OH * OW * KH * KW * Ci * Co
    if name == "cn_conv2d_wgrad":
        N, H, W, Ci, x_ld, OH, OW, Co, ld, KH, KW = a[:11]
        return 2.0 * N * OH * OW * KH * KW * Ci * Co
    if name == "cn_dcn_fwd":
        N, H, W, Ci, x_ld, Co = a[:6]
        return 2.0 * N * H * W * 9 * Ci * Co
    if name == "cn_dcn_wgrad":
        N, H, W, Ci, x_ld, Co = a[:6]
        return 2.0 * N * H * W * 9 * Ci * Co
    if name == "cn_dcn_bwd_dom":
        slabs, N, H, W, Ci, Co = a[:6]
        return 2.0 * N * H * W * 9 * Ci * Co
    if name == "cn_dcn_bwd_dx":
        N, H, W, Ci, dy_ld = a[:5]
        return 2.0 * N * H * W * 9 * Ci * dy_ld
    if name == "cn_conv1x1_smallk":
        npix, K, k_ld, Co = a[:4]
        return 2.0 * npix * K * Co
    return None


def _algo_bytes(name, a, es):
    """Algorithmic HBM bytes of one launch of an MFMA entry point: every operand tensor read once, the result written once
    (es = bytes per activation element; weights in es, offsets/masks and weight gradients fp32)."""
    if name == "cn_conv2d_fwd":
        N, H, W, Ci, x_ld, OH, OW, Co, y_ld, res_ld, KH, KW, stride, pad, transposed, relu, dt, odt = a[:18]
        osz = 4 if (odt != dt and odt == 0) else es
        return N * H * W * x_ld * es + N * OH * OW * (y_ld * osz + res_ld * es) + Co * Ci * KH * KW * es
    if name == "cn_conv2d_wgrad":
        N, H, W, Ci, x_ld, OH, OW, Co, ld, KH, KW = a[:11]
        return N * H * W * x_ld * es + N * OH * OW * ld * es + Co * Ci * KH * KW * 4
    if name == "cn_dcn_fwd":
        N, H, W, Ci, x_ld, Co, y_ld, om_ld = a[:8]
        return N * H * W * (x_ld * es + om_ld * 4 + y_ld * es) + Co * Ci * 9 * es
    if name == "cn_dcn_wgrad":
        N, H, W, Ci, x_ld, Co, dy_ld, om_ld = a[:8]
        return N * H * W * (x_ld * es + om_ld * 4 + dy_ld * es) + Co * Ci * 9 * 4
    if name == "cn_dcn_bwd_dom":
        slabs, N, H, W, Ci, Co, dy_ld, x_ld, om_ld = a[:9]
        return N * H * W * (dy_ld * es + x_ld * es + om_ld * 4 + om_ld * (4 * slabs if slabs else es)) + Co * Ci * 9 * es
    if name == "cn_dcn_bwd_dx":
        N, H, W, Ci, dy_ld, om_ld = a[:6]
        return N * H * W * (dy_ld * es + om_ld * 4 + Ci * es) + dy_ld * Ci * 9 * es
    if name == "cn_conv1x1_smallk":
        npix, K, k_ld, Co = a[:4]
        return npix * (k_ld + Co) * es
    return None


def _kernel_name(hip, name, a, tn):
    """rocprofv3's name of the kernel template an MFMA entry point dispatches to (mirrors the launch functions in csrc/)."""
    if name == "cn_conv2d_fwd":
        N, H, W, Ci, x_ld, OH, OW, Co, y_ld, res_ld, KH, KW, stride, pad, transposed, relu, dt, odt = a[:18]
        v = hip.lib().cn_conv2d_variant(Ci, Co, KH, KW, stride, pad, dt)
        if (v >= 3000000 and tn == "bf16" and Ci == 64 and H % 16 == 0 and W % 16 == 0 and x_ld % 8 == 0 and y_ld % 8 == 0
                and res_ld % 8 == 0 and not os.environ.get("CN_DISABLE_CONV_WS")):
            # csrc/conv3x3_ws.hip conv3x3_ws_launch(): 64 input channels and at least four tiles per workgroup
            bn = 32 if Co <= 32 else 64
            nblk = -(-Co // bn)
            cus = torch.cuda.get_device_properties(0).multi_processor_count
            if N * (H // 16) * (W // 16) * nblk >= 4 * (cus // (8 * nblk)) * 8 * nblk > 0:
                return f"conv3x3_ws_kernel<{bn}>"
        if (KH == 1 and KW == 1 and stride == 1 and pad == 0 and tn == "bf16" and Ci in (32, 64, 128, 256) and x_ld % 8 == 0
                and y_ld % 8 == 0 and res_ld % 8 == 0 and N * OH * OW >= 65536 and not os.environ.get("CN_DISABLE_CONV1X1_STREAM")):
            nj = (min(Co, y_ld) + 31) // 32          # csrc/conv1x1_stream.hip conv1x1_stream_launch()
            if nj in (1, 2, 3, 4, 8) and not (nj == 8 and Ci > 128) and nj * 32 * (Ci + 8) * 2 <= 72 * 1024 and nj * 16 + Ci // 2 <= 200:
                return f"conv1x1_stream_kernel<{Ci // 16},{nj}>"
        waves8 = os.environ.get("CN_CONV3X3_WAVES", "0") in ("0", "8")      # csrc/conv3x3.hip launch3(): 8 waves on the 128x64 tile
        if v >= 4000000 and not transposed:
            return f"conv3x3s1_kernel<{tn},{(v - 4000000) // 1000},32,S=2>"
        if v >= 4000000:
            bn, bk = 128 if Co > 64 else (64 if Co > 32 else 32), 32       # strided data gradient: implicit GEMM, parity classes
            return f"conv_igemm_kernel<{tn},{bn},{bk},transposed>"
        if v >= 3000000:
            if v == 3128064 and tn == "bf16" and waves8:
                return f"conv3x3s1_kernel<{tn},128,64,8>"
            return f"conv3x3s1_kernel<{tn},{(v - 3000000) // 1000},{v % 1000}>"
        return f"conv_igemm_kernel<{tn},{v // 1000},{v % 1000}>"
    # the four DCNv2 entry points: ask the library which template it dispatches to (cn_dcn_variant mirrors the launch functions), so
    # that the rows group exactly like rocprofv3's kernel names
    dcn = {"cn_dcn_fwd": (0, lambda a: (a[3], a[5])), "cn_dcn_wgrad": (1, lambda a: (a[3], a[5])),
           "cn_dcn_bwd_dom": (2, lambda a: (a[4], a[6])), "cn_dcn_bwd_dx": (3, lambda a: (a[3], a[4]))}
    if name in dcn and tn == "bf16":
        entry, pick = dcn[name]
        Ci, Co = pick(a)
        H, W = (a[2], a[3]) if name == "cn_dcn_bwd_dom" else (a[1], a[2])
        v = hip.lib().cn_dcn_variant_hw(entry, int(Ci), int(Co), int(H), int(W))
        if entry == 0 and v == 5000000:
            return "dcn_fwd_b2_kernel"
        if entry == 0:
            return (f"dcn_fwd_bm_kernel<{v - 1000000}>" if v < 2000000 else f"dcn_fwd_tile_kernel<{v - 2000000}>" if v < 3000000
                    else f"dcn_fwd_kernel<bf16,{(v - 3000000) // 1000},{v % 1000}>")
        if entry == 1:
            return "dcn_wgrad_bm_kernel" if v == 1 else f"dcn_wgrad_kernel<{v // 1000000},{v // 1000 % 1000},{v % 1000}>"
        if entry == 2:
            if v >= 1000000:
                return f"dcn_dom_bm_kernel<{v - 1000000}>"
            return f"dcn_bwd_dom_kernel<{v}>" if v else "conv_igemm_kernel<dom epilogue>"
        return f"dcn_dx_bm_kernel<{v - 1000000}>" if v < 2000000 else f"dcn_bwd_dx_kernel<bf16,{(v - 3000000) // 1000},{v % 1000}>"
    if name in dcn:
        return name + "[f32]"
    if name == "cn_conv2d_wgrad":
        N, H, W, Ci, x_ld, OH, OW, Co, ld, KH, KW, stride = a[:12]
        return f"conv_wgrad[{KH}x{KW}s{stride} {Ci}->{Co}]"
    return name


class ConvProbe:
    """HIP-event timing of EVERY C-ABI launch of a step (events on the launch stream, side streams folded into it so that each launch
    is timed alone), grouped by entry point and, for the MFMA-bound ones, by the kernel template they dispatch to."""

    def __init__(self, hip, tn):
        self.hip, self.ops, self.orig, self.tn = hip, [], hip.call, tn

    def __enter__(self):
        def call(name, *args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(name, *args, **kw)
            e1.record()
            self.ops.append((name, tuple(a for a in args if isinstance(a, (int, float)) and not isinstance(a, bool)), e0, e1))
            return r
        self.hip.call = call
        import centernet_amd.ops as ops
        self._mods = [(ops, ops.call)]
        ops.call = call
        return self

    def __exit__(self, *a):
        self.hip.call = self.orig
        for m, f in self._mods:
            m.call = f

    def detail_all(self, path, steps):
        by = {}
        for name, sig, e0, e1 in self.ops:
            d = by.setdefault((name,) + sig, [0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3; d[1] += 1
        tot = {}
        with open(path, "w") as f:
            for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
                f.write(f"{t / steps * 1e3:9.3f} ms/step {n / steps:5.0f} calls {t / n * 1e6:9.1f} us  {k}\n")
                tot[k[0]] = tot.get(k[0], 0.0) + t
            f.write("---- per entry point ----\n")
            for k, t in sorted(tot.items(), key=lambda kv: -kv[1]):
                f.write(f"{t / steps * 1e3:9.3f} ms/step  {k}\n")

    def mfma_kernels(self):
        """{kernel template: [flops, seconds, launches, algorithmic bytes]} over the launches of the MFMA entry points"""
        by = {}
        es = 2 if self.tn == "bf16" else 4
        for name, sig, e0, e1 in self.ops:
            fl = _mfma_flops(name, sig)
            if fl is None:
                continue
            d = by.setdefault(_kernel_name(self.hip, name, sig, self.tn), [0.0, 0.0, 0, 0.0])
            d[0] += fl; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += float(_algo_bytes(name, sig, es) or 0)
        return by

    def entry_points(self):
        by = {}
        for name, sig, e0, e1 in self.ops:
            d = by.setdefault(name, [0.0, 0.0, 0])
            d[0] += _mfma_flops(name, sig) or 0.0; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1
        return by


def _git_blob_sha(path):
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (FETCH_SIZE and WRITE_SIZE are
    collected in separate rocprofv3 --pmc runs; recipe and the gfx950 correction are in the JSON).  The record is stamped with the
    git blob hash of the kernel's source file: when the source has changed since the measurement the number is stale -> None."""
    root = os.path.dirname(os.path.abspath(__file__))
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(root, "profiles", fn)) as f:
                rec = json.load(f)
            k = rec["kernels"][kernel]
            src = k.get("source", rec.get("source"))
            if src and k.get("source_blob_sha", rec.get("source_blob_sha")) != _git_blob_sha(os.path.join(root, src)):
                continue
            if not src:
                continue                 # unstamped record (round 1): cannot be tied to the current kernel source
            return k["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def _norm_kernel(name):
    """rocprofv3's and _kernel_name()'s spellings of a template on common ground: no spaces, no `void`, bf16 = unsigned short"""
    return name.replace("void ", "").replace(" ", "").replace("bf16", "unsignedshort").replace("f32", "float")


def rocprof_avg_us(kernel, args):
    """Average duration of `kernel` in the committed rocprofv3 kernel-trace summary of THIS bench command on THIS build, or None.
    The summary (profiles/rNN_bench_kernel_stats.txt) is only used when its sidecar (profiles/rNN_bench_trace_meta.json, written by
    tools/trace_meta.py next to the trace) says that (a) the traced command had the same --arch / --batch / --size / --dtype /
    --dcn-offsets, (b) every kernel source under csrc/ still has the git blob hash it had when the trace was taken, and (c) exactly
    ONE instantiation in the summary carries the full template name bench.py derives for the launch (`dcn_dom_bm_kernel<64` never
    borrows the duration of `<128, true>`).  Otherwise the line keeps the live HIP-event number alone (round-5 ADVICE, medium)."""
    import glob
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    want = {"arch": args.arch, "batch": args.batch, "size": args.size, "dtype": args.dtype, "dcn_offsets": args.dcn_offsets}
    for mf in sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_trace_meta.json")), reverse=True):
        try:
            with open(mf) as f:
                meta = json.load(f)
            if meta.get("cmd_args") != want:
                continue
            if any(_git_blob_sha(os.path.join(root, src)) != sha for src, sha in meta["sources"].items()):
                continue
            key = _norm_kernel(kernel).rstrip(">")
            hits = []
            for ln in open(os.path.join(root, meta["stats_file"])):
                m = re.match(r"^(?:void )?(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+%", ln)
                if not m:
                    continue
                nm = _norm_kernel(m.group(1).strip())
                if nm == key + ">" or nm.startswith(key + ",") or (("<" not in key) and nm.startswith(key + "<")) or nm == key:
                    hits.append((m.group(1).strip(), int(m.group(2)), float(m.group(4))))
            if len(hits) == 1:
                return {"file": meta["stats_file"], "kernel": hits[0][0], "calls": hits[0][1], "avg_us": hits[0][2],
                        "meta": os.path.relpath(mf, root)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_sq(kernel):
    """MFMA-pipe busy fraction and VALU instructions per MFMA of `kernel` from the newest committed SQ counter pass
    (profiles/rNN_pmc_sq.txt, tools/pmc_sq.sh): what limits a kernel that is far from both roofs."""
    import glob
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    for fn in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_sq.txt")), reverse=True):
        try:
            lines = open(fn).read().splitlines()
        except OSError:
            continue
        for i, ln in enumerate(lines):
            if ln and not ln.startswith((" ", "#")) and kernel.split("<")[0] in ln and (kernel in ln or "<" not in kernel):
                for l2 in lines[i + 1:i + 3]:
                    m = re.search(r"MFMA pipe busy ([\d.]+);\s+VALU per MFMA ([\d.]+|n/a)", l2)
                    if m:
                        return {"file": os.path.relpath(fn, root), "mfma_pipe_busy": float(m.group(1)),
                                "valu_per_mfma": None if m.group(2) == "n/a" else float(m.group(2))}
    return None


def inference_rate(model, x, steps=20, materialise=False):
    """The north-star inference figure, driver-timed: eval forward (BN folded into the conv epilogues) + sigmoid + ctdet_decode of
    the SAME config (same weights, batch, resolution, bf16), one hipGraph, `steps` replays between two synchronisations.
    materialise: the reference's test_step_end (centernet_detection.py:183-187) leaves `sigmoid_()` in out["heatmap"]; this variant
    writes that map (sigmoid_clamped: one more pass over the 335 MB tensor) and decodes from it instead of from the logits."""
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.utils.decode import sigmoid_clamped
    was_training = model.training
    model.eval()

    def infer():
        with torch.no_grad():
            out = model(x)[-1]
            if materialise:
                out["heatmap"] = sigmoid_clamped(out["heatmap"])
                return ctdet_decode(out["heatmap"], out["width_height"], reg=out["regression"])
            # sigmoid + clamp applied by the top-K kernel on load (cn_ctdet_decode_logits): bit-identical detections to
            # ctdet_decode(sigmoid_clamped(hm), ...), without the separate pass over the 335 MB map
            return ctdet_decode(out["heatmap"], out["width_height"], reg=out["regression"], logits_clamp=1e-4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            det = infer()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):       # same stream as the warm-up: its stream-keyed workspaces are reused
        det = infer()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert det.shape[0] == x.shape[0] and bool(torch.isfinite(det).all())
    model.train(was_training)
    return dt


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(full=False):
    """the oracle (torch-CPU restatement of the reference path: same op sequence and state_dict, pure-torch
    DCNv2) on the host cores of the GPU box, fp32, DLA-34 ctdet, 512x512: (a) train step (fwd+loss+bwd+Adam) + ctdet_decode and
    (b) forward + ctdet_decode, batch 2 and 8, 3 warm-ups, median images/s.  The default run is BOUNDED (about 30 s of CPU work: 5
    timed iterations at batch 2, 2 at batch 8); --cpu-baseline-full runs >= 5 everywhere and adds ResNet-18.  `value` is leg (a) at
    batch 2 (the reference's own CPU-runnable configuration, BASELINE.json configs[0] shape).  Threads are capped at 16: on the
    256-thread GPU host torch's intra-op pool gets SLOWER beyond that (measured: 16 threads 0.5 s, 128 threads 11.9 s per step)."""
    import statistics
    from centernet_amd import rng, synth
    from oracle import models_ref, ops_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    legs, t_all = {}, time.time()

    def run(arch, bs, train, n_timed, warm=3):
        m = models_ref.CenterNetRef(arch)
        rng.fill_state_dict(m, 1234)
        m.train(train)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        x, tgt = synth.ctdet_batch(1234, bs)

        def one():
            if train:
                opt.zero_grad()
                out = m(x)
                loss, _ = m.loss(out, tgt)
                loss.backward()
                opt.step()
            else:
                with torch.no_grad():
                    out = m(x)
            with torch.no_grad():
                ops_ref.ctdet_decode(ops_ref.sigmoid_clamped(out[0]["heatmap"].detach().clone()), out[0]["width_height"].detach(),
                                     out[0]["regression"].detach())
        for _ in range(warm):
            one()
        ts = []
        for _ in range(n_timed):
            t0 = time.time(); one(); ts.append(time.time() - t0)
        legs[f"{arch} {'train step + decode' if train else 'forward + decode'} bs={bs}"] = {
            "images_per_s": round(bs / statistics.median(ts), 3), "timed": n_timed, "warmup": warm}

    n2, n8 = (5, 5) if full else (5, 2)
    w8 = 3 if full else 1                       # batch 8 warm-ups in the bounded run: 1 (a DLA-34 batch-8 train step is ~5 s)
    run("dla_34", 2, True, n2)
    run("dla_34", 2, False, n2)
    run("dla_34", 8, False, n8, warm=w8)
    run("dla_34", 8, True, n8, warm=w8)
    if full:
        for bs in (2, 8):
            run("res_18", bs, True, 5)
            run("res_18", bs, False, 5)
    dt = time.time() - t_all
    head = legs["dla_34 train step + decode bs=2"]
    return {"value": head["images_per_s"], "unit": "images/s", "cores": cores, "cores_available": avail, "kind": "port", "cpu": _cpu_model(),
            "sample": f"median of {head['timed']} timed steps after 3 warm-ups of DLA-34 ctdet train step (fwd+loss+bwd+Adam) + decode, "
                      f"batch 2, 512x512, fp32, torch-CPU oracle (pure-torch DCNv2), {cores} threads; all legs {dt:.0f} s "
                      f"({'full plan' if full else 'bounded: batch-8 legs 1 warm-up + 2 timed'})",
            "legs": legs}


def set_trained_offsets(model, seed=4321):
    """Put every DCN's `conv_offset_mask` into the regime of a trained network: the reference initialises it to zero (every sampling
    offset 0, every mask 0.5 — the cheapest input of the sampling kernels); here weights ~ N(0, 0.5 / sqrt(9 Ci)) and biases ~
    U[-0.25, 0.25], i.e. offsets of roughly N(0, 0.5 px) on unit-variance activations, written IN PLACE (the parameters are views of
    the optimizer's flat buffer, so a captured graph sees them).  Returns the number of layers touched."""
    from centernet_amd import rng
    from centernet_amd.ops import WeightsEpoch
    n = 0
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "conv_offset_mask.weight" in name:
                p.copy_(rng.t_normal(seed, name, tuple(p.shape), 0.0, 0.5 / (9 * p.shape[1]) ** 0.5).to(p.device))
                n += 1
            elif "conv_offset_mask.bias" in name:
                p.copy_(rng.t_uniform(seed, name, tuple(p.shape), -0.25, 0.25).to(p.device))
    WeightsEpoch.bump()
    return n


def _smi_start():
    """rocm-smi clocks + power, started asynchronously (the timed loop does not wait for it); -> Popen | None"""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if not exe:
        return None
    try:
        return subprocess.Popen([exe, "-d", "0", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except OSError:
        return None


def _smi_collect(p):
    """-> {"sclk_mhz": ..., "mclk_mhz": ..., "power_w": ...} (whatever rocm-smi reported) | None"""
    if p is None:
        return None
    try:
        out, _ = p.communicate(timeout=20)
        card = next(iter(json.loads(out).values()))
    except Exception:      # noqa: BLE001 - a box without a working rocm-smi must not lose the headline
        return None
    import re
    rec = {"when": "sampled once during the second timed block"}
    for k, v in card.items():
        kl = k.lower()
        m = re.search(r"([\d.]+)\s*mhz", str(v).lower())
        if "sclk" in kl and m:
            rec["sclk_mhz"] = float(m.group(1))
        elif "mclk" in kl and m:
            rec["mclk_mhz"] = float(m.group(1))
        elif "power" in kl and "(w)" in kl:
            try:
                rec["power_w"] = float(v)
            except ValueError:
                pass
    rec["raw"] = {k: str(v)[:40] for k, v in list(card.items())[:12]}
    return rec


def timed_steps(step, batch, steps, warmup, fence):
    for _ in range(warmup):
        step(batch)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(batch)
    fence()
    return (time.perf_counter() - t0) / steps, loss


def dry_run(args, rank, world):
    """The launch contract without a GPU: RANK / WORLD_SIZE parsing, per-rank batch offsets (`start=rank * nuniq` below is the
    expression the real run uses), the MAX-over-ranks reduction of the timing and the rank-0-only JSON line."""
    from centernet_amd import synth
    nuniq = args.batch
    x, tgt = synth.ctdet_batch(1234, min(nuniq, 8), 64, 64, start=rank * nuniq)
    mine = {"rank": rank, "first_image_index": rank * nuniq, "checksum": round(float(x.double().sum()), 6),
            "objects": int(tgt["regression_mask"].sum())}
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    per_rank = [mine]
    backend = {"backend": None, "backend_world_size": 1}
    if dist.is_initialized():
        backend = {"backend": dist.get_backend(), "backend_world_size": dist.get_world_size()}
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "images/sec (train step + decode) DLA-34 512\u00d7512 bs=64 at 1/2/4/8 MI355X", "value": None,
                          "unit": "images/s", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "weak", "max_over_ranks": float(t.item()),
                          "config": {"global_batch": args.batch * world, "parallelism": f"dp{world}"}, "ranks": per_rank, **backend}), flush=True)


def spawn_ranks(n):
    """Re-run this very command line as `n` ranks of one node (python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    bench.py <same flags>); the children see WORLD_SIZE and take the normal path.  The rendezvous is the launcher's own c10d store on
    127.0.0.1 port 0, i.e. the port is picked by the process that binds it (round-4 ADVICE: probing a 'free' port here and handing
    the number on is a time-of-check / time-of-use race between concurrent benches).  stdout / stderr are inherited, so rank 0's ONE
    JSON line is this process's JSON line; the exit status is the launcher's."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id=bench{os.getpid()}", "--local-addr", "127.0.0.1",
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    if rc:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--arch", default="dla_34")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="BASELINE.md section 3 in full (>= 5 timed iterations per leg, + ResNet-18): minutes")
    ap.add_argument("--dcn-offsets", default="init", choices=["init", "trained"],
                    help="regime of the HEADLINE run: init = the reference's zero-initialised conv_offset_mask (SURVEY 8d; on-spec), trained = "
                         "offsets of ~N(0, 0.5 px).  The default run measures init as `value` and trained as `trained_offsets`.")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: join the process group (gloo on a CPU-only host), build this rank's slice of the synthetic batch at "
                         "64x64, take the max-over-ranks of a per-rank number and let rank 0 print the JSON line — what "
                         "tests/test_host.py drives through torch.distributed.run to cover the launch contract without hardware")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (trained offsets, fp32 rate, exchange overhead)")
    ap.add_argument("--no-probe", action="store_true", help="skip the per-launch HIP-event pass that feeds `roofline`")
    ap.add_argument("--no-inference", action="store_true", help="skip the eval forward + decode sub-measurement (`inference`)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches + backward-overlapped RCCL buckets instead of hipGraph replay")
    ap.add_argument("--host-input", default="none", choices=["none", "image", "full"],
                    help="NOT the headline: hand the step pinned HOST buffers (image, or image + dense targets like the reference's "
                         "dataloader) so that the PCIe copy is inside the timed region; DESIGN.md quotes these rates")
    ap.add_argument("--no-prefetch", action="store_true", help="with --host-input: copy on the launch stream (no HostFeed overlap)")
    ap.add_argument("--stamps", action="store_true",
                    help="capture four device wall-clock stamps (cn_stamp) into the step's graph: where the launch-stream chain and the "
                         "weight-gradient stream end in a REPLAYED step, no profiler attached -> `stream_tail` in the line")
    ap.add_argument("--blocks", type=int, default=3, help="back-to-back timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--probe-steps", type=int, default=2)
    ap.add_argument("--probe-detail", default=None, help="write a per-shape table of every launch of a step to this file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher (the shape of the driver's N = 1 command): become the launcher — one rank per
        # GPU under torch.distributed.run on a free local port, like the reference's one-command DDP launch
        # (centernet_detection.py:405-409, Trainer flags -> DDP).  Rank 0 of the children prints the JSON line; it passes through.
        return spawn_ranks(args.gpus)

    from centernet_amd import _hip, synth
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.engine import TrainStep, init_distributed

    rank, local, world = init_distributed()
    assert args.gpus == world, (f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch N > 1 as `python -m torch.distributed.run "
                                f"--nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}`")
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path for the product)"
    dev = torch.device("cuda", local)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1234 + rank)

    model = CenterNetDetection(args.arch, compute_dtype=dt).to(dev).train()
    if args.dcn_offsets == "trained":
        set_trained_offsets(model)
    # one synthetic COCO-like batch per rank, every image of it different (and different from every other rank's), resident in HBM
    # before timing starts: top-K / decode cost is data dependent, and repeated images could be served from L2 / MALL
    nuniq = args.batch
    x, tgt = synth.ctdet_batch(1234, nuniq, args.size, args.size, start=rank * nuniq)
    x = x.to(dev)
    tgt = {k: v.to(dev) for k, v in tgt.items()}
    batch = (x, tgt)
    if args.host_input != "none":        # PCIe-inclusive variant: the step's input copy becomes host -> device
        batch = (x.cpu().pin_memory(), {k: v.cpu().pin_memory() for k, v in tgt.items()} if args.host_input == "full" else tgt)
        args.no_probe = True             # the probe launches eagerly from device tensors; this variant is not the headline
    captured = {}
    orig_loss = model.loss

    def loss_and_keep(outputs, target):          # keep this step's head maps for the decode half of the metric
        r = orig_loss(outputs, target)
        captured["out"] = outputs[-1]
        return r
    model.loss = loss_and_keep

    def decode():   # heat map is already sigmoid (+clamp) from the loss, like after sigmoid_() in test_step_end
        out = captured["out"]
        return ctdet_decode(out["heatmap"].detach(), out["width_height"].detach(), reg=out["regression"].detach())

    # decode overlaps backward; the resident synthetic batch IS the graph's static input (no per-step device-to-device copy of
    # inputs that are already in HBM — with --host-input the copy is host -> device and stays in the timed region)
    if args.stamps:
        import centernet_amd.ops as _ops
        _ops.SideGrads.stamps = torch.zeros(4, dtype=torch.int64, device=dev)
    step = TrainStep(model, lr=1e-4, graph=not args.no_graph, post_forward=decode, adopt_batch=args.host_input == "none")
    side_grid = int(__import__("centernet_amd.ops", fromlist=["SideGrads"]).SideGrads.thin)      # what the timed steps and the probe run with

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    feed = None
    if args.host_input != "none" and not args.no_prefetch:
        from centernet_amd.engine import HostFeed
        feed = HostFeed(dev)
        feed.put(batch)
        run = step

        def step(b):                        # the next batch crosses PCIe on the copy stream while this one computes
            d = feed.get()
            feed.put(b)
            return run(d)

    for _ in range(args.warmup):
        step(batch)
    # The timed region: `--blocks` (default 3) back-to-back blocks of EXACTLY --steps steps, each bracketed by barrier + synchronize on
    # both sides, each reduced to the slowest rank's time.  `value` / `ms_per_step` are the MEDIAN block (box-to-box and run-to-run
    # spread is +-2 %, larger than most single changes: round-5 VERDICT #4); every block's time is in the line (`blocks_ms_per_step`,
    # `value_min` / `value_max`), so ms_per_step x steps is the duration of one real, contiguous block of the run.
    block_s, smi = [], None
    for b in range(max(1, args.blocks)):
        fence()
        if b == 1 and rank == 0:
            smi = _smi_start()                      # clocks / power while the GPU is under this load (sampled once, asynchronously)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step(batch)
        fence()
        e = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([e], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)     # the slowest rank's time
            block_s.append((float(t.item()), e))
        else:
            block_s.append((e, e))
    if feed is not None:
        step = run
    order = sorted(range(len(block_s)), key=lambda i: block_s[i][0])
    mid = order[(len(order) - 1) // 2]                  # the median block (the lower one of an even count)
    elapsed, mine_s = block_s[mid]
    ranks_info = None
    if dist.is_initialized():
        # value = all ranks' images / the SLOWEST rank's time; each rank's own rate and the world the backend itself reports go
        # into the line next to it (all_gather over RCCL: the collective path is exercised even when a rank's step is local)
        mine = torch.tensor([mine_s], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ranks_info = {"backend": dist.get_backend(), "backend_world_size": dist.get_world_size(),
                      "per_rank_images_per_s": [round(args.batch * args.steps / float(e.item()), 2) for e in every]}
    det = step.post_out
    assert det.shape == (args.batch, 100, 6) and bool(torch.isfinite(det).all())
    stream_tail = None
    if args.stamps:
        st_ = _ops.SideGrads.stamps.cpu().tolist()           # the LAST timed step's stamps (10 ns ticks)
        _ops.SideGrads.stamps = None
        stream_tail = {"launch_stream_done_ms": round((st_[1] - st_[0]) / 1e5, 3), "weight_gradient_stream_done_ms": round((st_[2] - st_[0]) / 1e5, 3),
                       "joined_ms": round((st_[3] - st_[0]) / 1e5, 3),
                       "what": "device wall clock (cn_stamp) inside the last replayed step, from the step's first launch; the optimizer graph follows the join"}

    probe = None
    tn = "bf16" if dt == torch.bfloat16 else "f32"
    if rank != 0 and not args.no_probe and step.sync is not None and not step.graph:
        # eager DP mode: the probe steps hold collectives, so every rank runs them; rank 0 measures.  (Graph mode's probe steps on
        # rank 0 skip the exchange below: nothing for the other ranks to take part in.)
        side_was, step.side = step.side, False
        for _ in range(args.probe_steps):
            step._eager(batch)
        torch.cuda.synchronize()
        step.side = side_was
    if rank == 0 and not args.no_probe:
        # same step, launched eagerly, with a HIP event pair around every launch on the launch stream
        probe = ConvProbe(_hip, tn)
        side_was, step.side = step.side, False      # weight gradients on the launch stream: every launch is timed alone
        sync_was = step.sync
        if step.graph:
            step.sync = None                        # single-rank probe: no collective in the probe steps
        with probe:
            for _ in range(args.probe_steps):
                step._eager(batch)
            torch.cuda.synchronize()
        step.side, step.sync = side_was, sync_was
        if args.probe_detail:
            probe.detail_all(args.probe_detail, args.probe_steps)

    inference = None
    if rank == 0 and world == 1 and not args.no_inference and args.host_input == "none":
        dt_inf = inference_rate(model, x)
        dt_mat = inference_rate(model, x, materialise=True)
        inference = {"metric": "images/sec (eval forward + ctdet_decode), same config", "value": round(args.batch / dt_inf, 1),
                     "unit": "images/s", "ms_per_batch": round(dt_inf * 1e3, 3), "launch": "hipGraph replay", "dtype": args.dtype,
                     "target": 3000.0,
                     "decode": "fused logits decode (cn_ctdet_decode_logits: sigmoid + clamp applied by the top-K kernel on load, detections "
                               "bit-identical to decode/ctdet.py:6-38); the sigmoid heat map itself is NOT materialised in the timed region "
                               "(the reference's test_step leaves it in out['heatmap']) - `materialised_heatmap` is the rate with it written",
                     "materialised_heatmap": {"value": round(args.batch / dt_mat, 1), "unit": "images/s", "ms_per_batch": round(dt_mat * 1e3, 3),
                                              "what": "same graph with out['heatmap'] = sigmoid_clamped(logits) stored (as centernet_detection.py:183-187 "
                                                      "leaves it) and ctdet_decode reading that map: the figure comparable with the reference's test step"}}

    # ---- secondary measurements (single GPU only; never `value`) ----
    extras = {}
    if world == 1 and not args.no_extras and args.host_input == "none" and args.arch.startswith(("dla", "resdcn")):
        n_steps = max(3, args.steps // 2)
        if args.dcn_offsets == "init":
            # (1) the same replayed graph on TRAINED-regime sampling offsets: the DCN kernels' time depends on the offset field
            nl = set_trained_offsets(model)
            dt_tr, _ = timed_steps(step, batch, n_steps, 2, fence)
            extras["trained_offsets"] = {"value": round(args.batch / dt_tr, 2), "unit": "images/s", "ms_per_step": round(dt_tr * 1e3, 3),
                                         "steps": n_steps, "dcn_layers": nl,
                                         "what": "same graph, conv_offset_mask ~ N(0, 0.5/sqrt(9 Ci)) (offsets of about N(0, 0.5 px))"}
            if probe is not None:
                p2 = ConvProbe(_hip, tn)
                side_was, step.side = step.side, False
                sync_was, step.sync = step.sync, None
                with p2:
                    step._eager(batch)
                    torch.cuda.synchronize()
                step.side, step.sync = side_was, sync_was
                extras["trained_offsets"]["dcn_entry_points_ms_per_step"] = {
                    k: round(v[1] * 1e3, 3) for k, v in p2.entry_points().items() if k.startswith("cn_dcn")}
        # (2) gradient-exchange overhead on ONE rank: the same step with the bucketed RCCL all-reduces captured into its graph
        # (CN_FORCE_EXCHANGE: a 1-rank group) against the step above — launch + capture cost of the exchange, not its wire time
        try:
            os.environ["CN_FORCE_EXCHANGE"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if not dist.is_initialized():
                dist.init_process_group("nccl", rank=0, world_size=1)
            base_ms = (extras["trained_offsets"]["ms_per_step"] if "trained_offsets" in extras else elapsed / args.steps * 1e3)
            step_x = TrainStep(model, lr=1e-4, graph=not args.no_graph, post_forward=decode, adopt_batch=True, distributed=True)
            dt_x, _ = timed_steps(step_x, batch, n_steps, 2, fence)
            extras["exchange_overhead_ms"] = round(dt_x * 1e3 - base_ms, 3)
            extras["exchange"] = {"ms_per_step": round(dt_x * 1e3, 3), "baseline_ms_per_step": round(base_ms, 3),
                                  "buckets": len(step_x.sync.buckets), "captured": bool(step_x.graph),
                                  "what": "1-rank RCCL group (CN_FORCE_EXCHANGE=1): bucketed all-reduce captured into the step's hipGraph"}
            del step_x
        except Exception as e:      # noqa: BLE001 - a box without a working RCCL must not lose the headline
            extras["exchange_overhead_ms"] = None
            extras["exchange"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            os.environ.pop("CN_FORCE_EXCHANGE", None)
        # (3) fp32 compute mode (the mode in which the 1e-4 parity of the north star holds), same workload, fewer steps
        if args.dtype == "bf16":
            try:
                torch.cuda.empty_cache()
                m32 = CenterNetDetection(args.arch, compute_dtype=torch.float32).to(dev).train()
                cap32 = {}
                l32 = m32.loss

                def keep32(outputs, target):
                    r = l32(outputs, target)
                    cap32["out"] = outputs[-1]
                    return r
                m32.loss = keep32
                dec32 = lambda: ctdet_decode(cap32["out"]["heatmap"].detach(), cap32["out"]["width_height"].detach(),
                                             reg=cap32["out"]["regression"].detach())
                s32 = TrainStep(m32, lr=1e-4, graph=not args.no_graph, post_forward=dec32, adopt_batch=True, distributed=False)
                dt32, _ = timed_steps(s32, batch, 3, 1, fence)
                extras["fp32"] = {"value": round(args.batch / dt32, 2), "unit": "images/s", "ms_per_step": round(dt32 * 1e3, 3), "steps": 3,
                                  "what": "compute_dtype=float32 (parity mode: heat maps / losses within 1e-4 of the reference), same workload"}
                del s32, m32
            except Exception as e:      # noqa: BLE001
                extras["fp32"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    if rank == 0:
        total_images = args.batch * world * args.steps
        roof = None
        peak = PEAK_BF16_TFLOPS if dt == torch.bfloat16 else PEAK_F32_TFLOPS
        gflop_img, mb_img = WORK.get(args.arch, (None, None)) if args.size == 512 else (None, None)
        step_s = elapsed / args.steps
        step_roof = None
        if gflop_img:
            tfl = 3.0 * gflop_img * args.batch / step_s / 1e3          # per GPU: every rank runs the same step
            step_roof = {"tflops": round(tfl, 1), "frac_mfma": round(tfl / peak, 4),
                         "flop_per_image": f"3 x {gflop_img} GFLOP (SURVEY 8d)"}
            if mb_img:
                gbps = 3.0 * mb_img * args.batch / step_s / 1e3
                step_roof.update({"gbps": round(gbps, 1), "frac_hbm": round(gbps / PEAK_HBM_GBPS, 4),
                                  "bytes_per_image": f"3 x {mb_img} MB unfused bf16 traffic (SURVEY 8d)"})
        if step_roof and mb_img:
            # per-layer max(t_MFMA, t_HBM) bound of the forward = mb_img / 6.3 TB/s per image (46 us for DLA-34 ctdet); x3 for
            # the train step.  frac_of_layerwise_bound = that bound / the measured step
            lb = 3.0 * (mb_img * 1e6 / 6.3e12) * args.batch
            step_roof.update({"layerwise_bound_ms": round(lb * 1e3, 3), "frac_of_layerwise_bound": round(lb / step_s, 4)})
        if probe and probe.ops:
            by = probe.mfma_kernels()
            kern, (fl, tt, n, nbytes) = max(by.items(), key=lambda kv: kv[1][1])
            ach = fl / tt / 1e12
            # which roof binds the kernel: its arithmetic intensity (algorithmic flops / algorithmic bytes of its launch mix)
            # against the ridge peak_flops / peak_bandwidth — below the ridge no implementation can reach the MFMA roof
            ridge = peak * 1e12 / (PEAK_HBM_GBPS * 1e9)
            ai = fl / nbytes if nbytes else float("inf")
            gbps = nbytes / tt / 1e9
            hbm_bound = ai < ridge
            ig = {k: v for k, v in by.items() if k.startswith(("conv3x3s1_kernel", "conv_igemm_kernel", "conv3x3_ws_kernel"))}
            allf = sum(v[0] for v in ig.values()); allt = sum(v[1] for v in ig.values())
            eps = probe.entry_points()

            def row(v):
                r = {"tflops": round(v[0] / v[1] / 1e12, 2), "frac_mfma": round(v[0] / v[1] / 1e12 / peak, 4),
                     "ms_per_step": round(v[1] / args.probe_steps * 1e3, 3), "launches_per_step": v[2] // args.probe_steps}
                if v[3]:
                    r.update({"gbps": round(v[3] / v[1] / 1e9, 1), "frac_hbm": round(v[3] / v[1] / 1e9 / PEAK_HBM_GBPS, 4),
                              "flop_per_byte": round(v[0] / v[3], 1)})
                return r

            # `frac` / `achieved` are measured by THIS run: HIP events around the dominant template's launches of the eagerly launched probe
            # steps (round-5 ADVICE, medium: the committed trace is a cross-check, not the source).  `rocprof` carries the same kernel's
            # average duration in the committed rocprofv3 summary of this command (replayed steps, the other stream's kernels next to
            # it) when that summary provably belongs to this build and these arguments (rocprof_avg_us), with `frac_trace` from it.
            rp = rocprof_avg_us(kern, args)
            t_launch = tt / n
            ach_t, gbps_t = fl / n / t_launch / 1e12, nbytes / n / t_launch / 1e9
            t_trace = rp["avg_us"] * 1e-6 if rp else None
            # which roof is the nearer one is a label, not a finding: when neither is close the kernel is bound by instruction issue
            # around its MFMAs, and the machine-readable field says so (SQ counters of the committed pass next to it)
            far = max(gbps_t / PEAK_HBM_GBPS, ach_t / peak) < 0.25
            sq = pmc_sq(kern)
            limiter = ("neither roof binds (both fractions < 0.25): bound by VALU / LDS instruction issue around the MFMAs (a wave issues one "
                       "VALU instruction per ~4.9 cycles: tools/probe/valu_rate.hip)" if far else ("HBM" if hbm_bound else "MFMA"))
            roof = {"bound": "valu" if far else ("hbm" if hbm_bound else "mfma"), "nearer_roof": "hbm" if hbm_bound else "mfma",
                    "limiter": limiter, "sq": sq,
                    "achieved": round(gbps_t, 1) if hbm_bound else round(ach_t, 2),
                    "peak": PEAK_HBM_GBPS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round(gbps_t / PEAK_HBM_GBPS, 4) if hbm_bound else round(ach_t / peak, 4),
                    "frac_from": "avg_launch_us (HIP events around this run's own launches of the dominant template)",
                    "achieved_trace": (None if not rp else round(nbytes / n / t_trace / 1e9, 1) if hbm_bound else round(fl / n / t_trace / 1e12, 2)),
                    "frac_trace": (None if not rp else round(nbytes / n / t_trace / 1e9 / PEAK_HBM_GBPS, 4) if hbm_bound
                                   else round(fl / n / t_trace / 1e12 / peak, 4)),
                    "traffic": pmc_traffic(kern),
                    "rocprof": rp,
                    "how": f"HIP events around each launch, {args.probe_steps} eagerly launched step(s) of the same workload right after "
                           f"the timed region; dominant = largest total time among the kernel templates of ALL MFMA entry points; bound = "
                           f"hbm when the kernel's algorithmic flop/byte ({ai:.0f}) is below the ridge ({ridge:.0f}), else mfma; both "
                           f"fractions of every template are in mfma_kernels",
                    "kernel": kern,
                    "launches": n, "avg_launch_us": round(tt / n * 1e6, 2), "flop_per_launch": round(fl / n, 1),
                    "bytes_per_launch": round(nbytes / n, 1), "flop_per_byte": round(ai, 1),
                    "tflops": round(ach, 2), "frac_mfma": round(ach / peak, 4), "gbps": round(gbps, 1),
                    "frac_hbm": round(gbps / PEAK_HBM_GBPS, 4),
                    "step": step_roof,
                    # the dominant template may be a weight-gradient kernel: those run on the side stream with a deliberately thin
                    # grid (cn_hooks.wgrad_blocks = 160 workgroups: fewer CUs taken from the critical chain), which is the grid the
                    # probe times them on; the largest template of the LAUNCH-STREAM chain (what decides the step) is named next to it
                    "side_stream_grid": side_grid,
                    "launch_stream_dominant": (lambda kv: dict(kernel=kv[0], **row(kv[1])))(
                        max(((k, v) for k, v in by.items() if "wgrad" not in k), key=lambda kv: kv[1][1])),
                    "mfma_kernels": {k: row(v) for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]},
                    "all_igemm": {"achieved": round(allf / allt / 1e12, 2), "ms_per_step": round(allt / args.probe_steps * 1e3, 3)},
                    "entry_points_ms_per_step": {k: round(v[1] / args.probe_steps * 1e3, 3)
                                                 for k, v in sorted(eps.items(), key=lambda kv: -kv[1][1])[:24]}}
        elif step_roof:
            roof = {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None, "step": step_roof}
        line = {"metric": "images/sec (train step + decode) DLA-34 512\u00d7512 bs=64 at 1/2/4/8 MI355X",
                "value": round(total_images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "blocks": len(block_s), "blocks_ms_per_step": [round(b[0] / args.steps * 1e3, 3) for b in block_s],
                "value_min": round(total_images / max(b[0] for b in block_s), 2), "value_max": round(total_images / min(b[0] for b in block_s), 2),
                "value_is": f"median of {len(block_s)} back-to-back blocks of {args.steps} steps, each bracketed by barrier + synchronize",
                "gpu_state": _smi_collect(smi),
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if args.host_input == "none" else f"synthetic, pinned host input ({args.host_input}) copied inside the timed region",
                "config": {"workload": f"{args.arch} ctdet (80 classes) train step (fwd+loss+bwd+Adam) + ctdet_decode, "
                                       f"{args.size}x{args.size}, batch {args.batch}/GPU, {args.dtype} compute / fp32 master weights",
                           "global_batch": args.batch * world, "parallelism": f"dp{world}",
                           "launch": "hipGraph replay (2 graphs/step)" if step.graph else "eager",
                           "dcn_offsets": ("init: conv_offset_mask zero-initialised like the reference (SURVEY 8d), i.e. sampling offsets are 0 at "
                                           "step 0 and O(1e-3 px) during the timed steps; `trained_offsets` holds the rate on ~N(0, 0.5 px) offsets")
                                          if args.dcn_offsets == "init" else "trained: conv_offset_mask ~ N(0, 0.5/sqrt(9 Ci)), offsets ~ N(0, 0.5 px)",
                           "parity": "bf16 compute / fp32 master weights; decode indices bit-exact vs the reference goldens, fp32 mode 1e-4; DCNv2 "
                                     "arithmetic pinned to the published algorithm only (the extension is not under /root/reference)",
                           "final_loss": round(float(loss.detach()), 4)},
                "roofline": roof,
                "inference": inference,
                "cpu_baseline": None}
        if ranks_info:
            line["ranks"] = ranks_info
        if stream_tail:
            line["stream_tail"] = stream_tail
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(full=args.cpu_baseline_full)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a pipe: push it out now so that
    # the JSON line is the LAST line of stdout, whatever the reader keys on
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
