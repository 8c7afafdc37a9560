"""Round-6 GPU tests: (1) bf16 TRAINING tracks fp32 training (round-5 VERDICT, weak #1 / next #5); (2) multi-rank RCCL tests that fire
the day a box has two GPUs (next #6): they are collected and skipped on the 1-GPU box; their CPU (gloo) variants live in test_host.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from centernet_amd.centernet_detection import CenterNetDetection

pytestmark = pytest.mark.gpu
DEV = "cuda"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recall(det, tgt, thr=0.3):
    """share of the synthetic ground-truth boxes with a detection of the right class whose centre is within 1 output pixel
    (det [B,K,6] = x1,y1,x2,y2,score,class in output pixels, the layout of decode/ctdet.py:36)"""
    hit = tot = 0
    H, W = tgt["heatmap"].shape[-2:]
    ind, msk = tgt["indices"].cpu().numpy(), tgt["regression_mask"].cpu().numpy()
    hm = tgt["heatmap"].cpu().numpy()
    d = det.cpu().numpy()
    for b in range(d.shape[0]):
        for k in np.nonzero(msk[b])[0]:
            cy, cx = divmod(int(ind[b, k]), W)
            cls = int(np.argmax(hm[b, :, cy, cx]))
            tot += 1
            cand = d[b][(d[b][:, 4] > thr) & (d[b][:, 5] == cls)]
            if len(cand):
                ccx, ccy = (cand[:, 0] + cand[:, 2]) / 2, (cand[:, 1] + cand[:, 3]) / 2
                hit += bool(((np.abs(ccx - cx) <= 1.0) & (np.abs(ccy - cy) <= 1.0)).any())
    return hit / max(tot, 1)


def test_bf16_training_tracks_fp32_training():
    """DLA-34 ctdet, 128x128, batch 8, same seed / same batch / same learning rate, 150 optimizer steps through the hipGraph-replayed
    TrainStep in bf16 compute mode (the headline mode) and in fp32 compute mode (the mode the 1e-4 parity holds in).  The per-element
    bounds of bf16 MAPS under random weights (tests/test_gpu_configs.py: 12 % eval / 35 % training-mode worst element) say little about
    what an optimizer sees; the statement that matters is about TRAINING.  150 Adam steps on one small batch are a chaotic trajectory:
    an fp32 run whose initial weights are perturbed by 1e-3 (relative, gaussian) leaves the unperturbed fp32 curve by 20 % (10-step
    means) / 36 % (single steps) — measured, DESIGN.md section 4 — so that CONTROL run is the yardstick: the bf16 curve must stay as
    close to the fp32 curve as a 1e-3-perturbed fp32 run does (within 3x + 0.1, and within 0.40 absolute), reach the same late-phase loss
    (30 %), and end with the same detections on the boxes it was trained on (recall within 0.15)."""
    from centernet_amd.engine import TrainStep
    seed, steps = 611, 150
    x, tgt = synth.ctdet_batch(seed, 8, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    curves, recall = {}, {}
    for name, dt, perturb in (("fp32", torch.float32, 0.0), ("bf16", torch.bfloat16, 0.0), ("ctrl", torch.float32, 1e-3)):
        m = CenterNetDetection("dla_34", compute_dtype=dt)
        rng.fill_state_dict(m, seed)
        m = m.to(DEV).train()
        if perturb:
            g = torch.Generator(device=DEV).manual_seed(seed)
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1 + perturb * torch.randn(p.shape, generator=g, device=DEV))
        step = TrainStep(m, lr=5e-4, distributed=False)
        curves[name] = np.array([float(step(batch)) for _ in range(steps)])
        m.eval()
        with torch.no_grad():
            out = m(batch[0])[-1]
            det = m.decode({k: v.float() for k, v in out.items()})
        recall[name] = _recall(det, tgt)
        del step, m
    f, b, c = curves["fp32"], curves["bf16"], curves["ctrl"]
    assert np.isfinite(f).all() and np.isfinite(b).all() and np.isfinite(c).all()
    sm = lambda v, w=10: np.convolve(v, np.ones(w) / w, mode="valid")      # 10-step moving average: single steps are noisy in every mode
    dev_s = lambda u: float((np.abs(sm(u) - sm(f)) / sm(f))[20:].max())
    dev_1 = lambda u: float((np.abs(u - f) / f)[20:].max())
    late = lambda u: float(u[-20:].mean())
    print(f"loss fp32 {f[0]:.3f} -> {f[-1]:.3f}, bf16 {b[0]:.3f} -> {b[-1]:.3f}, fp32 perturbed 1e-3 {c[0]:.3f} -> {c[-1]:.3f}; "
          f"deviation from the fp32 curve after step 20 (10-step means | single steps): bf16 {dev_s(b):.3f} | {dev_1(b):.3f}, "
          f"control {dev_s(c):.3f} | {dev_1(c):.3f}; last-20-step mean loss {late(f):.3f} / {late(b):.3f} / {late(c):.3f}; "
          f"recall {recall['fp32']:.3f} / {recall['bf16']:.3f} / {recall['ctrl']:.3f}")
    assert f[-1] < 0.05 * f[0] and b[-1] < 0.05 * b[0], "both modes train (loss falls by more than 20x)"
    assert float((np.abs(b - f) / f)[:3].max()) < 0.03, "the first steps (same weights) agree to bf16 accuracy"
    # measured over six runs (the weight-gradient atomics make every run its own trajectory): bf16 0.15-0.27 | 0.27-0.55, control 0.09-0.20 | 0.22-0.36
    assert dev_s(b) < 3.0 * dev_s(c) + 0.10, "bf16 leaves the fp32 curve no further than a 1e-3-perturbed fp32 run does (3x + 0.1)"
    assert dev_s(b) < 0.40 and dev_1(b) < 0.8
    assert late(b) == pytest.approx(late(f), rel=0.30), "same late-phase loss"
    assert abs(recall["bf16"] - recall["fp32"]) <= 0.15 and min(recall.values()) > 0.6


# ------------------------------------------------------------------------------------------------ two real GPUs
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (RCCL refuses two ranks on one device)")

_RCCL2_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CN_REPO"])
from centernet_amd import rng, synth
from centernet_amd.engine import TrainStep, init_distributed
from centernet_amd.centernet_detection import CenterNetDetection
rank, local, world = init_distributed()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 2
def batch_of(r):
    x, tgt = synth.ctdet_batch(97, 2, 128, 128, start=2 * r)
    return x.cuda(), {k: v.cuda() for k, v in tgt.items()}
def make():
    m = CenterNetDetection("res_18", compute_dtype=torch.float32)
    rng.fill_state_dict(m, 97)
    return m.cuda().train()
out = {}
for mode in ("eager", "graph", "fail"):
    if mode == "fail":
        os.environ["CN_FAIL_CAPTURE"] = "rank:1"
    step = TrainStep(make(), lr=2e-4, graph=mode != "eager")
    assert step.sync is not None and step.sync.exchange
    out[mode] = [float(step(batch_of(rank))) for _ in range(4)]
    out["is_graph_" + mode] = bool(step.graph)
    torch.cuda.synchronize()
    p = step.opt.flat_p.clone()
    every = [torch.zeros_like(p) for _ in range(world)]
    dist.all_gather(every, p)
    out["same_" + mode] = bool(torch.equal(every[0], every[1]))
    out["p_" + mode] = p[:4096].cpu().tolist() if rank == 0 else None
    os.environ.pop("CN_FAIL_CAPTURE", None)
if rank == 0:
    # what ONE process computes when it averages the two ranks' locally normalised gradients itself
    step = TrainStep(make(), lr=2e-4, distributed=False, graph=False)
    for it in range(4):
        step.opt.zero_grad()
        for r in range(2):
            loss = step.model.training_step(batch_of(r), 0) / 2
            loss.backward()
        step.opt.step()
    torch.cuda.synchronize()
    out["p_ref"] = step.opt.flat_p[:4096].cpu().tolist()
dist.barrier(); dist.destroy_process_group()
if rank == 0:
    print("RESULT " + json.dumps(out))
"""


def _launch(nproc, script_args, extra_env=None, timeout=900):
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, CN_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port)] + script_args, env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)


@two_gpus
def test_two_rank_rccl_graph_equals_eager_equals_one_process_average(tmp_path):
    """2-rank RCCL TrainStep: graph mode == eager mode == the 1-process average of the per-rank gradients (the assertion of the gloo
    tests in test_host.py, on real devices); with the capture failing on rank 1 only, BOTH ranks end on eager launches and still agree."""
    script = tmp_path / "rccl_two_ranks.py"
    script.write_text(_RCCL2_SCRIPT)
    r = _launch(2, [str(script)])
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["is_graph_graph"] and not res["is_graph_eager"]
    assert not res["is_graph_fail"], "a capture failure on one rank must put EVERY rank on eager launches"
    for mode in ("eager", "graph", "fail"):
        assert res["same_" + mode], f"{mode}: the two ranks' parameters differ"
        np.testing.assert_allclose(res["p_" + mode], res["p_ref"], rtol=2e-4, atol=2e-6, err_msg=mode)
    np.testing.assert_allclose(res["graph"], res["eager"], rtol=5e-3)


@two_gpus
def test_bench_two_gpus_reports_the_backends_world_size():
    r = _launch(2, ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2", "--batch", "4", "--size", "128", "--no-cpu-baseline",
                    "--no-extras", "--no-inference", "--no-probe"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"]["backend"] == "nccl" and line["ranks"]["backend_world_size"] == 2
    assert len(line["ranks"]["per_rank_images_per_s"]) == 2 and line["config"]["parallelism"] == "dp2"
    assert line["blocks"] == 2 and len(line["blocks_ms_per_step"]) == 2
    assert line["value"] == pytest.approx(4 * 2 * 3 / (line["ms_per_step"] * 3 / 1e3), rel=1e-3)
