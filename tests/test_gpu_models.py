"""GPU: whole-network parity (backbone -> heads -> losses -> backward -> decode) of the HIP path in fp32 compute mode
against the golden vectors produced by the reference's own modules, plus bf16 sanity and a short training run."""
import os

import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from oracle import models_ref, ops_ref
from conftest import strided, summary, assert_det_rank_tolerant

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(arch, seed, dtype, task="ctdet", var_scale=1.0):
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.centernet_multi_pose import CenterNetMultiPose
    m = (CenterNetDetection if task == "ctdet" else CenterNetMultiPose)(arch, compute_dtype=dtype)
    rng.fill_state_dict(m, seed, var_scale=var_scale)
    return m.to(DEV)


@pytest.mark.parametrize("arch,size,train", [("res_18", 256, False), ("res_18", 256, True),
                                             ("dla_34", 128, False), ("dla_34", 128, True),
                                             ("resdcn_18", 128, False), ("resdcn_18", 128, True),
                                             ("res_101", 128, False), ("res_101", 256, True),         # Bottleneck (msra_resnet.py:61-100)
                                             ("resdcn_101", 128, False), ("resdcn_101", 256, True)])
def test_network_fp32_vs_reference_golden(golden, arch, size, train):
    name = arch.replace("_", "") + ("_train" if train else "_eval") + ".npz"
    g = golden(name)
    seed = int(g["seed"])
    m = _model(arch, seed, torch.float32, var_scale=float(g["var_scale"]) if "var_scale" in g.files else 1.0)
    m.train(train)
    x, tgt = synth.ctdet_batch(seed, 2, size, size)
    xg, tg = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    with torch.set_grad_enabled(train):
        outs = m(xg)
    out = outs[0]
    # north-star tolerance 1e-4 (relative) for the BASELINE archs.  The 101-layer Bottleneck nets in TRAINING mode are held to
    # 3e-3 / 1e-3: 33 blocks of batch-statistic BN re-normalise fp32 summation-order differences (MFMA 32x32x2 k-order vs ATen's)
    # at every layer — measured 0.8-1.5e-3 of the map's maximum; their eval-mode fixtures stay at 1e-4.
    deep_train = train and arch.endswith("_101")
    tol_map, tol_loss = (3e-3, 1e-3) if deep_train else (1e-4, 1e-4)
    for k in ("heatmap", "width_height", "regression"):
        assert out[k].dtype == torch.float32 and out[k].shape[2:] == (size // 4, size // 4)
        ref_s = g[f"{k}_s"]
        assert np.abs(strided(out[k]).cpu().numpy() - ref_s).max() < tol_map * np.abs(ref_s).max() + 1e-6, k
        # (sum, sum|x|, sum x^2): the signed sum cancels (|sum| << sum|x|), so its 1e-4 is taken relative to sum|x|
        np.testing.assert_allclose(summary(out[k]), g[f"{k}_sum"], rtol=tol_map, atol=tol_map * float(g[f"{k}_sum"][1]), err_msg=k)
    raw = {k: v.detach().clone() for k, v in out.items()}
    with torch.set_grad_enabled(train):
        loss, st = m.loss(outs, tg)
    for k, gk in (("hm_loss", "hm"), ("wh_loss", "wh"), ("off_loss", "off"), ("loss", "loss")):
        assert float(st[k]) == pytest.approx(float(g[gk]), rel=tol_loss), k      # north_star: within 1e-4 relative
    if train:
        loss.backward()
        params = dict(m.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                ref = g[key].astype(np.float64)
                got = strided(params[n].grad, 512).cpu().numpy().astype(np.float64)
                # Deep gradients pass through batch-statistic BN + ReLU / max-pool: a single activation whose sign
                # flips (|y| ~ 1e-7, summation order) perturbs a handful of entries by O(1e-2); the oracle's own
                # fp32-vs-fp64 run shows the same (tools/grad_modes.py).  Layer-exact parity lives in test_gpu_ops.py;
                # here: small relative L2 error and an accurate bulk.
                rel_l2 = np.linalg.norm(got - ref) / max(1e-30, np.linalg.norm(ref))
                med = np.median(np.abs(got - ref)) / max(1e-30, np.abs(ref).max())
                # (101-layer nets: measured rel-L2 0.07-0.08 on the first conv — more ReLU decisions at |y| ~ 1e-7 to flip)
                lim_l2, lim_med = (0.15, 4e-2) if deep_train else (3e-2, 5e-3)
                assert rel_l2 < lim_l2 and med < lim_med, f"grad {n}: rel-L2 {rel_l2:.3e}, median err {med:.3e}"
        # parameters of the reference's dead branches get no gradient (the flat optimizer keeps them at zero grad)
        dead = sorted(n for n, p in params.items() if p.grad is None or float(p.grad.abs().max()) == 0.0)
        assert set(str(s) for s in g["dead_params"]) <= set(dead)
        sd = m.state_dict()
        bn = "backbone.bn1" if arch.startswith("res") else "backbone.base.base_layer.1"
        np.testing.assert_allclose(sd[bn + ".running_mean"].cpu().numpy(), g["bn_running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[bn + ".running_var"].cpu().numpy(), g["bn_running_var"], rtol=1e-4, atol=1e-6)
        if arch == "dla_34":   # dead `project` BN of the outer trees still tracks running statistics like the reference
            assert float(sd["backbone.base.level3.project.1.num_batches_tracked"]) == 1
    else:
        det = m.decode({"heatmap": raw["heatmap"], "width_height": raw["width_height"], "regression": raw["regression"]})
        got, ref = det.cpu().numpy(), g["det"]
        if arch in ("resdcn_18", "res_101", "resdcn_101"):
            assert_det_rank_tolerant(got, ref)      # top-100 scores closer together than the heat-map tolerance
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-3)
            assert np.array_equal(got[..., 5], ref[..., 5]), "decoded classes identical"


POSE_HEADS = ("heatmap", "width_height", "regression", "heatmap_keypoints", "keypoints", "heatmap_keypoints_offset")
POSE_STATS = ("loss", "hm_loss", "kp_loss", "hm_kp_loss", "hm_offset_loss", "wh_loss", "off_loss")


@pytest.mark.parametrize("train", [False, True])
def test_pose_network_fp32_vs_reference_golden(golden, train):
    """C5 pinned to the REFERENCE (not the oracle): DLA-34 + the six multi_pose heads, the reference's own loss body and
    `multi_pose_decode` (fixture: oracle/gen_golden.py gen_pose_models); HIP fp32 within the north-star 1e-4."""
    g = golden("dla34_pose_train.npz" if train else "dla34_pose_eval.npz")
    seed, size = int(g["seed"]), int(g["size"])
    m = _model("dla_34", seed, torch.float32, task="pose")
    m.train(train)
    x, tgt = synth.pose_batch(seed, 2, size, size)
    xg, tg = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    with torch.set_grad_enabled(train):
        outs = m(xg)
    out = outs[0]
    for k in POSE_HEADS:
        ref_s = g[f"{k}_s"]
        assert np.abs(strided(out[k]).cpu().numpy() - ref_s).max() < 1e-4 * np.abs(ref_s).max() + 1e-6, k
        np.testing.assert_allclose(summary(out[k]), g[f"{k}_sum"], rtol=1e-4, atol=1e-4 * float(g[f"{k}_sum"][1]), err_msg=k)
    raw = {k: v.detach().clone() for k, v in out.items()}
    with torch.set_grad_enabled(train):
        loss, st = m.loss(outs, tg)
    for k in POSE_STATS:
        assert float(st[k]) == pytest.approx(float(g["stat:" + k]), rel=1e-4), k
    if train:
        loss.backward()
        params = dict(m.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                ref = g[key].astype(np.float64)
                got = strided(params[n].grad, 512).cpu().numpy().astype(np.float64)
                rel_l2 = np.linalg.norm(got - ref) / max(1e-30, np.linalg.norm(ref))
                med = np.median(np.abs(got - ref)) / max(1e-30, np.abs(ref).max())
                assert rel_l2 < 3e-2 and med < 5e-3, f"grad {n}: rel-L2 {rel_l2:.3e}, median err {med:.3e}"   # same rule as the ctdet nets
    else:
        # centre scores of the reference's decode of ITS heat map (near-flat, 130 of 200 scores exactly tied: only the sorted score
        # column is order-proof; full rows are pinned bit-exactly by pose_decode.npz), through the HIP decode on the same map
        from centernet_amd.decode.multi_pose import multi_pose_decode
        from centernet_amd.utils.decode import sigmoid_clamped
        # (the sigmoid of the reference's map is taken with the reference's arithmetic — ATen on the host, centernet.py's
        # `clamp(x.sigmoid_(), 1e-4, 1 - 1e-4)` — so that the comparison pins the DECODE bit-exactly; the device sigmoid differs from
        # ATen's by 1 ulp on 3 % of the elements and is held to its own tolerance in test_gpu_decode_loss.py)
        heat = torch.clamp(torch.from_numpy(g["map:heatmap"]).sigmoid(), 1e-4, 1 - 1e-4).to(DEV)
        det = multi_pose_decode(heat, raw["width_height"], raw["keypoints"], reg=raw["regression"],
                                hm_hp=sigmoid_clamped(raw["heatmap_keypoints"]), hp_offset=raw["heatmap_keypoints_offset"], K=100)
        np.testing.assert_array_equal(det[..., 4].cpu().numpy(), g["det_scores"])


@pytest.mark.parametrize("train", [False, True])
def test_hourglass_fp32_vs_reference_golden(golden, train):
    """2-stack Hourglass-104 + one CenterHead per stack against the reference's own modules (fixture from
    oracle/gen_golden.py gen_hourglass); the loss averages the stacks (centernet_detection.py:99-123)."""
    g = golden("hourglass_train.npz" if train else "hourglass_eval.npz")
    seed, size = int(g["seed"]), int(g["size"])
    from centernet_amd.centernet_detection import CenterNetDetection
    m = CenterNetDetection("hourglass", compute_dtype=torch.float32)
    rng.fill_state_dict(m, seed, var_scale=float(g["var_scale"]))
    m = m.to(DEV).train(train)
    x, tgt = synth.ctdet_batch(seed, 2, size, size)
    xg, tg = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    with torch.set_grad_enabled(train):
        outs = m(xg)
    assert len(outs) == 2
    for s_, out in enumerate(outs):
        for k in ("heatmap", "width_height", "regression"):
            ref_s = g[f"{k}{s_}_s"]
            got_s = strided(out[k]).cpu().numpy()
            # 1e-4 of the map's range (north_star) against the reference's fp32 values.  In train mode the second stack
            # sits behind ~100 convs with batch statistics over as few as 8 samples (2x2 maps, B=2) and the reference's
            # OWN fp32 run is ~1e-4 away from exact arithmetic: there the yardstick is the reference run in fp64
            # (`*_s64`), and the HIP path may be at most 4x as far from it as the reference's fp32 run is.
            tol = 1e-4
            if train:
                ref64 = g[f"{k}{s_}_s64"]
                ref_noise = np.abs(ref_s - ref64).max() / np.abs(ref64).max()
                assert np.abs(got_s - ref64).max() < max(1e-4, 4 * ref_noise) * np.abs(ref64).max() + 1e-6, (k, s_, ref_noise)
                tol = max(1e-4, 5 * ref_noise)
            assert np.abs(got_s - ref_s).max() < tol * np.abs(ref_s).max() + 1e-6, (k, s_)
            np.testing.assert_allclose(summary(out[k]), g[f"{k}{s_}_sum"], rtol=tol, atol=tol * float(g[f"{k}{s_}_sum"][1]))
    raw = {k: v.detach().clone() for k, v in outs[-1].items()}
    with torch.set_grad_enabled(train):
        loss, st = m.loss(outs, tg)
    for k, gk in (("hm_loss", "hm"), ("wh_loss", "wh"), ("off_loss", "off"), ("loss", "loss")):
        assert float(st[k]) == pytest.approx(float(g[gk]), rel=1e-4), k
    if train:
        loss.backward()
        params = dict(m.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                ref = g[key].astype(np.float64)
                got = strided(params[n].grad, 512).cpu().numpy().astype(np.float64)
                # yardstick: the reference's gradients in fp64 (`g64:`); its own fp32 run is up to 3.6e-2 (rel. L2) away
                # from them on the deepest layers (ReLU sign flips behind batch-statistic BN).  Over ALL 600+ parameter
                # tensors the HIP fp32 path is a uniform 2.2x (median; max 4.6x) as far as the reference's fp32 run
                # (sequential-K MFMA accumulation vs oneDNN's blocked sums); the 512-entry samples of the fixture
                # scatter around that, so: 6x the reference's own distance, or 0.1 where that distance is tiny.
                ref64 = g["g64:" + n + ":s"]
                ref_noise = np.linalg.norm(ref - ref64) / np.linalg.norm(ref64)
                mine = np.linalg.norm(got - ref64) / np.linalg.norm(ref64)
                bound = 6 * ref_noise + 1e-5 if n.startswith("heads") else max(6 * ref_noise, 0.1)
                assert mine < bound, f"grad {n}: rel-L2 to fp64 {mine:.3e}, reference fp32 {ref_noise:.3e}"
        sd = m.state_dict()
        np.testing.assert_allclose(sd["backbone.pre.0.bn.running_mean"].cpu().numpy(), g["bn_running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd["backbone.pre.0.bn.running_var"].cpu().numpy(), g["bn_running_var"], rtol=1e-4, atol=1e-6)
    else:
        det = m.decode({"heatmap": raw["heatmap"], "width_height": raw["width_height"], "regression": raw["regression"]})
        assert_det_rank_tolerant(det.cpu().numpy(), g["det"])


def test_hourglass_bf16_trains():
    from centernet_amd.engine import TrainStep
    from centernet_amd.centernet_detection import CenterNetDetection
    m = CenterNetDetection("hourglass", compute_dtype=torch.bfloat16)
    rng.fill_state_dict(m, 96)
    m = m.to(DEV).train()
    x, tgt = synth.ctdet_batch(96, 2, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    step = TrainStep(m, lr=2.5e-4, distributed=False)
    hist = [float(step(batch)) for _ in range(6)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist


@pytest.mark.parametrize("arch,size", [("res_18", 128), ("dla_34", 128)])
def test_network_bf16_tracks_fp32(arch, size):
    seed = 91
    x, tgt = synth.ctdet_batch(seed, 2, size, size)
    xg, tg = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _model(arch, seed, dt).train()
        loss, st = m.loss(m(xg), tg)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        losses[dt] = {k: float(v) for k, v in st.items()}
    print("fp32", losses[torch.float32], "bf16", losses[torch.bfloat16])
    assert losses[torch.bfloat16]["loss"] == pytest.approx(losses[torch.float32]["loss"], rel=5e-2)
    for k in losses[torch.float32]:
        assert losses[torch.bfloat16][k] == pytest.approx(losses[torch.float32][k], rel=0.3), k
    # eval (BN folded into the conv epilogue) agrees with the unfused training-mode graph fed the same statistics
    m = _model(arch, seed, torch.float32).eval()
    with torch.no_grad():
        fused = m(xg)[0]["heatmap"].clone()
    with torch.enable_grad():
        for p in m.parameters():
            p.requires_grad_(True)
        unfused = m(xg)[0]["heatmap"]
    assert torch.allclose(fused, unfused.detach(), rtol=1e-3, atol=1e-4)


def test_multi_pose_loss_and_decode_vs_oracle():
    seed, size = 92, 128
    ref = models_ref.CenterNetRef("dla_34", task="pose")
    rng.fill_state_dict(ref, seed)
    ref.train()
    x, tgt = synth.pose_batch(seed, 2, size, size)
    out_ref = ref(x)
    loss_ref, st_ref = ref.loss(out_ref, tgt)
    loss_ref.backward()
    m = _model("dla_34", seed, torch.float32, task="pose").train()
    m.load_state_dict(ref.state_dict())
    xg, tg = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    outs = m(xg)
    raw = {k: v.detach().clone() for k, v in outs[0].items()}
    for k in raw:
        assert torch.allclose(raw[k].cpu(), out_ref[0][k].detach(), rtol=1e-3, atol=2e-4), k
    loss, st = m.loss(outs, tg)
    for k in st_ref:
        assert float(st[k]) == pytest.approx(float(st_ref[k]), rel=1e-4), k
    loss.backward()
    pr, pg = dict(ref.named_parameters()), dict(m.named_parameters())
    for n in ("heads.0.keypoints.fc.2.weight", "heads.0.heatmap_keypoints.fc.2.weight", "heads.0.heatmap_keypoints_offset.fc.0.weight",
              "backbone.ida_up.node_1.conv.weight"):
        a, b = pg[n].grad.cpu().double(), pr[n].grad.double()
        assert float((a - b).norm() / b.norm()) < 3e-2, n        # see the note on ReLU sign flips in the golden test
    det = m.decode(raw).cpu()
    ro = {k: v.detach() for k, v in out_ref[0].items()}
    det_ref = ops_ref.multi_pose_decode(torch.sigmoid(ro["heatmap"]), ro["width_height"], ro["keypoints"], reg=ro["regression"],
                                        hm_hp=torch.sigmoid(ro["heatmap_keypoints"]), hp_offset=ro["heatmap_keypoints_offset"])
    assert det.shape == det_ref.shape == (2, 100, 57)
    assert torch.allclose(det[..., 4], det_ref[..., 4], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_training_reduces_loss(dt):
    from centernet_amd.engine import TrainStep
    m = _model("res_18", 93, dt).train()
    x, tgt = synth.ctdet_batch(93, 4, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    step = TrainStep(m, lr=5e-4, distributed=False)
    hist = [float(step(batch)) for _ in range(8)]
    assert all(np.isfinite(hist)), hist
    assert hist[-1] < 0.7 * hist[0], hist


def test_graph_replay_matches_eager():
    """hipGraph-captured step (2 graphs) == eager step: same losses over several optimizer steps, BN counters advance."""
    from centernet_amd.engine import TrainStep
    x, tgt = synth.ctdet_batch(95, 2, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    hist = {}
    for graph in (False, True):
        m = _model("dla_34", 95, torch.float32).train()
        step = TrainStep(m, lr=2e-4, distributed=False, graph=graph)
        hist[graph] = [float(step(batch)) for _ in range(5)]
        nb = int(m.state_dict()["backbone.base.level2.root.bn.num_batches_tracked"])
        assert nb == 5, nb      # the 2 eager warm-up steps before the capture are rolled back (one step per batch)
    # graph mode's first call = 2 warm-up steps, state restored, capture, 1 replay: the first loss IS eager's first loss
    assert hist[True][0] == pytest.approx(hist[False][0], rel=1e-4)
    assert hist[True][1] == pytest.approx(hist[False][1], rel=2e-3)
    # later steps only qualitatively: Adam's first updates are ~lr*sign(g), so parameters whose gradient is at the
    # summation-noise level take opposite steps in two runs and the trajectories drift apart by a few % within 5 steps
    assert hist[True][4] == pytest.approx(hist[False][4], rel=0.15)
    assert hist[False][4] < hist[False][0] and hist[True][4] < hist[True][0]


def _bf16_losses(seed, graph, n, between=None, arch="res_18"):
    from centernet_amd.engine import TrainStep
    x, tgt = synth.ctdet_batch(seed, 4, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    m = _model(arch, seed, torch.bfloat16).train()
    step = TrainStep(m, lr=2e-4, distributed=False, graph=graph)
    out = []
    for i in range(n):
        out.append(float(step(batch)))
        if between is not None:
            between(m, batch, i)
    return out, m, step


def test_train_mode_forward_between_graph_replays_leaves_the_steps_bn_sinks_alone():
    """Round-3 ADVICE (medium): a captured step's BatchNorm statistics sinks (ops.BnStats: persistent, all-zero when armed, cleared
    by the NEXT consumer) belong to its replays.  A training-mode forward under no_grad between two replays used to run in the
    step's namespace: it left a forward sink dirty on the device, the next replay added a second batch's sums to it (wrong mean /
    variance, polluted running statistics) and the host's bookkeeping drifted.  Now `TrainStep` restores the previous namespace on
    exit, so the stray forward has its own rings: the losses of the following replays are those of an undisturbed run, and the
    namespace is back to what it was."""
    from centernet_amd import ops

    def stray(m, batch, i):
        assert ops.BnStats.ns is None            # restored by TrainStep (eager warm-up, capture and replays alike)
        if i in (1, 2):
            with torch.no_grad():
                m(batch[0])                      # train mode: every BN takes batch statistics through the sinks

    ref, _, _ = _bf16_losses(97, True, 5)
    got, m, step = _bf16_losses(97, True, 5, between=stray)
    assert step._g1 is not None
    assert all(np.isfinite(got)), got
    for a, b in zip(ref, got):
        assert b == pytest.approx(a, rel=2e-2), (ref, got)
    # an EAGER step of the same TrainStep after the capture (the bench's probe steps) must not touch the graph's rings either
    x, tgt = synth.ctdet_batch(97, 4, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    l_eager = float(step._eager(batch))
    l_next = float(step(batch))
    assert np.isfinite(l_eager) and np.isfinite(l_next) and l_next < ref[0]
    assert ops.BnStats.ns is None


def test_two_models_stepping_alternately_in_one_process():
    """Round-3 VERDICT, weak 12: the side channels next to autograd (ops.SparseRows / DualLayout / BnStats / GradCell) are process
    wide.  Two TrainSteps on two different models, stepped alternately in one process (one eager, one graph-replayed), must each
    follow the trajectory they follow alone; and no gradient map stays pinned in a registry after a backward pass."""
    from centernet_amd import ops
    from centernet_amd.engine import TrainStep
    alone_a, _, _ = _bf16_losses(101, False, 4, arch="resdcn_18")
    alone_b, _, _ = _bf16_losses(103, True, 4)
    xa, ta = synth.ctdet_batch(101, 4, 128, 128)
    xb, tb = synth.ctdet_batch(103, 4, 128, 128)
    ba = (xa.to(DEV), {k: v.to(DEV) for k, v in ta.items()})
    bb = (xb.to(DEV), {k: v.to(DEV) for k, v in tb.items()})
    ma, mb = _model("resdcn_18", 101, torch.bfloat16).train(), _model("res_18", 103, torch.bfloat16).train()
    sa, sb = TrainStep(ma, lr=2e-4, distributed=False, graph=False), TrainStep(mb, lr=2e-4, distributed=False, graph=True)
    both_a, both_b = [], []
    for _ in range(4):
        both_a.append(float(sa(ba)))
        both_b.append(float(sb(bb)))
        assert not ops.SparseRows.entries and not ops.DualLayout.entries
    # (a bf16 DLA-34 at this batch size is not reproducible enough for this comparison: the fp32 atomics of its BN sums differ in the
    # last bits from run to run and 34 batch-statistic layers amplify that to 3-4 % of the loss by the third step, DESIGN section 3;
    # the ResNets are: first two steps to 2 %, the later ones to 5 %)
    for alone, both in ((alone_a, both_a), (alone_b, both_b)):
        for k, (a, b) in enumerate(zip(alone, both)):
            assert b == pytest.approx(a, rel=2e-2 if k < 2 else 5e-2), (alone, both)


def test_state_dict_round_trip_with_oracle():
    """Drop-in claim: the HIP model's state_dict loads into the reference-shaped (oracle) modules and back."""
    m = _model("dla_34", 94, torch.bfloat16)
    ref = models_ref.CenterNetRef("dla_34")
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    m.load_state_dict(ref.state_dict())
    assert len(m.state_dict()) == 398


def test_extension_is_loaded_and_required():
    from centernet_amd import _hip
    assert _hip.lib().cn_version() >= 100
    with pytest.raises(RuntimeError):
        _hip.call("cn_add", torch.zeros(8), torch.zeros(8), torch.zeros(8), 8, 0)   # CPU tensors: no fallback path


_RCCL_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CN_REPO"])
from centernet_amd import rng, synth
from centernet_amd.engine import TrainStep, init_distributed
from centernet_amd.centernet_detection import CenterNetDetection
rank, local, world = init_distributed()
assert dist.is_initialized() and dist.get_backend() == "nccl"
x, tgt = synth.ctdet_batch(97, 2, 128, 128)
batch = (x.cuda(), {k: v.cuda() for k, v in tgt.items()})
out = {}
for graph in (False, True, "between"):
    m = CenterNetDetection("res_18", compute_dtype=torch.float32)
    rng.fill_state_dict(m, 97)
    m = m.cuda().train()
    if graph == "between":                       # escape hatch: collectives outside the captured graph
        os.environ["CN_EXCHANGE_BETWEEN_GRAPHS"] = "1"
        step = TrainStep(m, lr=2e-4, graph=True)
        out["between"] = [float(step(batch)) for _ in range(4)]
        out["drained_between"] = bool(getattr(step, "drained", False))
        del os.environ["CN_EXCHANGE_BETWEEN_GRAPHS"]
        continue
    step = TrainStep(m, lr=2e-4, graph=graph)
    assert step.sync is not None and step.sync.exchange and step.side, "weight gradients stay on the side stream under DP"
    out["graph" if graph else "eager"] = [float(step(batch)) for _ in range(4)]
    out["is_graph_" + str(graph)] = bool(step.graph)
    out["drained_" + str(graph)] = bool(getattr(step, "drained", False))
    # buckets leave DURING backward (for the graph: during the captured backward): (bucket, #parameters produced at launch)
    out["log_" + ("graph" if graph else "eager")] = step.sync.launch_log
    out["live"] = len(step.sync.live)
dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_rccl_exchange_next_to_graphs(tmp_path):
    """The gradient exchange through RCCL (backend "nccl") on a 1-rank group: bucketed hooks in eager mode, the
    all-reduce between the two hipGraphs in graph mode (capture with a live process group + watchdog thread)."""
    import json, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CN_FORCE_EXCHANGE="1", CN_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_SCRIPT)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["is_graph_True"], "capture fell back to eager next to the process group:\n" + r.stderr[-2000:]
    assert res["drained_True"], "the watchdog was not drained by its own bookkeeping (timer fallback taken)"
    assert res["drained_between"], "second capture in the same process (earlier captured collectives in the recorder): timer fallback taken"
    assert all(np.isfinite(res["eager"])) and all(np.isfinite(res["graph"]))
    for mode in ("eager", "graph"):
        log = res["log_" + mode]
        assert len(log) >= 2 and log[0][1] < res["live"], f"{mode}: the first bucket must leave before backward is over: {log}"
        assert [b for b, _ in log] == sorted(b for b, _ in log), f"{mode}: buckets leave in reverse-layer order: {log}"
    assert res["graph"][0] == pytest.approx(res["eager"][0], rel=1e-4)     # the warm-up steps before the capture are rolled back
    assert res["graph"][1] == pytest.approx(res["eager"][1], rel=5e-3)
    assert res["eager"][3] < res["eager"][0]
    assert res["between"][0] == pytest.approx(res["eager"][0], rel=1e-4) and res["between"][1] == pytest.approx(res["eager"][1], rel=5e-3)


_GLOO2_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.environ["CN_REPO"])
from centernet_amd import rng, synth
from centernet_amd.engine import TrainStep
from centernet_amd.centernet_detection import CenterNetDetection

def make(arch):
    m = CenterNetDetection(arch, compute_dtype=torch.float32)
    rng.fill_state_dict(m, 97)
    return m.cuda().train()

def batch_of(rank):
    x, tgt = synth.ctdet_batch(97, 2, 128, 128, start=2 * rank)
    return x.cuda(), {k: v.cuda() for k, v in tgt.items()}

def worker(rank, world, port, arch, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # two ranks on ONE GPU: RCCL refuses that, gloo does not
    step = TrainStep(make(arch), lr=0.0, graph=False)
    assert step.sync is not None and step.sync.exchange and step.side
    b = batch_of(rank)
    step(b)                                   # learns which parameters are live; every bucket leaves in finish()
    step(b)                                   # buckets leave while backward is still producing gradients
    torch.cuda.synchronize()
    out[rank] = (step.opt.flat_g.cpu(), list(step.sync.launch_log), len(step.sync.live))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    arch = sys.argv[1]
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(worker, args=(2, int(os.environ["CN_PORT"]), arch, out), nprocs=2, join=True)
    ref = 0
    for r in range(2):                         # what each rank computes on its own, no exchange
        step = TrainStep(make(arch), lr=0.0, distributed=False, graph=False)
        step(batch_of(r)); step(batch_of(r))
        torch.cuda.synchronize()
        ref = ref + step.opt.flat_g.cpu()
    g0, log, live = out[0]
    g1 = out[1][0]
    err = float((g0 - ref).abs().max() / ref.abs().max())
    print("RESULT " + json.dumps({"same": bool(torch.equal(g0, g1)), "err": err, "log": log, "live": live,
                                  "nonzero": float(ref.abs().max())}))
"""


@pytest.mark.parametrize("arch", ["res_18", "dla_34"])
def test_overlapped_exchange_sums_complete_gradients_two_ranks_one_gpu(tmp_path, arch):
    """The backward-overlapped bucket exchange with REAL data movement between two ranks (gloo on CUDA tensors, both ranks on
    this GPU): weight gradients arrive from the side stream and BN gradients from the launch stream through ops.GradReady, the
    buckets leave mid-backward — and every rank must still end up with exactly g_rank0 + g_rank1.  A bucket that left before
    its last deposit landed would miss that deposit."""
    import json, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "gloo_two_ranks.py"
    script.write_text(_GLOO2_SCRIPT)
    env = dict(os.environ, CN_REPO=repo, CN_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run([sys.executable, str(script), arch], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["same"], "both ranks hold the same summed gradient"
    assert res["nonzero"] > 0 and res["err"] < 2e-5, res      # fp32 atomics in the weight-gradient kernels: order noise only
    assert len(res["log"]) >= 2 and res["log"][0][1] < res["live"], res["log"]


def test_graph_replay_host_never_runs_far_ahead():
    """TrainStep bounds the host's run-ahead to MAX_STEPS_AHEAD graph-replayed steps; a long un-synchronised run completes,
    keeps its pace and leaves every parameter finite."""
    import time
    from centernet_amd.engine import TrainStep
    m = _model("res_18", 92, torch.bfloat16).train()
    x, tgt = synth.ctdet_batch(92, 4, 128, 128)
    batch = (x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()})
    step = TrainStep(m, lr=1e-4, distributed=False, graph=True)
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step(batch)
    torch.cuda.synchronize()
    t40 = time.perf_counter() - t0
    assert len(step._inflight) <= step.MAX_STEPS_AHEAD
    t0 = time.perf_counter()
    for _ in range(400):
        loss = step(batch)
    torch.cuda.synchronize()
    t400 = time.perf_counter() - t0
    assert t400 < 10 * t40 * 1.5, (t40, t400)
    assert np.isfinite(float(loss)) and all(bool(torch.isfinite(p).all()) for p in m.parameters())


def test_host_feed_prefetch_trains_like_resident_batches():
    """HostFeed: pinned host batches copied one step ahead on the copy stream give the same losses, step by step, as the same
    batches resident in HBM (different batch every step, so a late or reordered copy would show)."""
    from centernet_amd.engine import TrainStep, HostFeed
    batches = [synth.ctdet_batch(300 + i, 4, 128, 128) for i in range(6)]
    losses = []
    for fed in (False, True):
        m = _model("res_18", 93, torch.float32).train()
        step = TrainStep(m, lr=1e-6, distributed=False, graph=True)     # tiny steps: each loss is a fingerprint of its batch
        out = []
        if fed:
            feed = HostFeed("cuda:0")
            host = [(x.pin_memory(), {k: v.pin_memory() for k, v in t.items()}) for x, t in batches]
            feed.put(host[0])
            for i in range(len(host)):
                b = feed.get()
                if i + 1 < len(host):
                    feed.put(host[i + 1])
                out.append(step(b).clone())
            assert not feed._q
        else:
            for x, t in batches:
                out.append(step((x.to(DEV), {k: v.to(DEV) for k, v in t.items()})).clone())
        losses.append(torch.stack(out).cpu())
    # weight gradients are accumulated with fp32 atomics: two runs agree to rounding, not bit for bit
    assert torch.allclose(losses[0], losses[1], rtol=1e-3), losses
    assert (losses[0][1:] - losses[0][:-1]).abs().min() > 0.01 * losses[0].mean(), losses[0]    # a swapped batch would show
    feed = HostFeed("cuda:0")                  # and the bytes themselves, two copies in flight
    feed.put(host[3]); feed.put(host[4])
    for i in (3, 4):
        x, t = feed.get()
        assert torch.equal(x.cpu(), host[i][0]) and all(torch.equal(t[k].cpu(), host[i][1][k]) for k in t)


def test_training_next_to_overlapped_decode_survives(tmp_path):
    """The scenario that exposed the top-K race (see test_decode_is_stable_next_to_other_streams): eager DLA-34 training at the
    bench's size with ctdet_decode forked onto its own stream after every forward pass and a garbage collection per step — it
    used to die of a GPU memory fault after 60-180 steps.  Runs in a subprocess with a timeout: a GPU fault must fail this
    test, not hang the session."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "long_run.py"), "--steps", "120", "--no-graph", "--decode",
                        "--gc", "every", "--every", "40"], env=dict(os.environ, CN_NO_SIDE="1"), capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    last = [l for l in r.stdout.splitlines() if l.startswith("step 120:")]
    assert last and "params finite True" in last[0], r.stdout[-800:]


def _dla_grads(dtype, cells, seed=11, size=128, batch=2):
    from centernet_amd import ops
    old = ops.GradCell.enabled, ops.BnStats.enabled, ops.BnStats.fused
    ops.GradCell.enabled = cells
    ops.GradCell.adds = 0
    # BatchNorm statistics through the deterministic two-level reduction (no fp32 atomics): a bf16 DLA-34 at batch 2 x 128^2 (32
    # samples per channel in the deepest BNs) turns one flipped rounding into percent-level changes of single gradients, and the
    # sinks' atomics do flip roundings from run to run; this test compares two ACCUMULATION schemes, so everything else is pinned
    ops.BnStats.enabled = ops.BnStats.fused = False
    try:
        m = _model("dla_34", seed, dtype)
        m.train()
        x, tgt = synth.ctdet_batch(seed, batch, size, size)
        outs = m(x.to(DEV))
        loss, _ = m.loss(outs, {k: v.to(DEV) for k, v in tgt.items()})
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}, ops.GradCell.adds
    finally:
        ops.GradCell.enabled, ops.BnStats.enabled, ops.BnStats.fused = old


def test_shared_tensor_gradients_summed_in_epilogues_match_autograd_sums():
    """ops.GradCell (gradients of multiply-used activations summed in their consumers' kernel epilogues) against autograd's own
    accumulation (CN_DISABLE_GRAD_CELLS path) on the whole DLA-34 graph: pose_dla_dcn.py:245-262 (Tree fan-outs), :482-488 (IDAUp),
    heads.py:38-43 (sibling heads).  fp32: the two differ by summation order only; bf16: by the rounding of partial sums."""
    l0, g0, a0 = _dla_grads(torch.float32, False)
    l1, g1, a1 = _dla_grads(torch.float32, True)
    assert a0 == 0 and l0 == pytest.approx(l1, rel=1e-6)
    assert set(g0) == set(g1)
    # 25 autograd adds per step before; with cells only a consumer without an epilogue slot that finds the cell occupied pays one
    assert 0 < a1 <= 6, a1
    # (a bias in front of a batch-statistic BN has a zero gradient up to rounding noise: absolute floor from the largest gradient)
    floor = 1e-6 * max(float(v.norm()) for v in g0.values())
    for n in g0:
        den = float(g0[n].norm()) + 1e-20
        assert float((g0[n] - g1[n]).norm()) <= 2e-3 * den + floor, (n, float((g0[n] - g1[n]).norm()) / den)
    # bf16: same graph, same kernels, partial sums rounded at different points
    _, h0, _ = _dla_grads(torch.bfloat16, False)
    _, h1, _ = _dla_grads(torch.bfloat16, True)
    floor = 1e-3 * max(float(v.norm()) for v in h0.values())
    for n in h0:
        den = float(h0[n].norm()) + 1e-20
        assert float((h0[n] - h1[n]).norm()) <= 0.12 * den + floor, (n, float((h0[n] - h1[n]).norm()) / den)


def test_backbone_output_gradient_is_visible_to_autograd():
    """The NHWC handle a backbone returns is what a caller may hold on to (retain_grad, torch.autograd.grad): it must not be one of
    the shared aliases whose consumers route their gradients past autograd (ops.GradCell), although three sibling heads read it."""
    from centernet_amd import ops
    m = _model("dla_34", 5, torch.bfloat16).train()
    x, tgt = synth.ctdet_batch(5, 2, 128, 128)
    feats = m.backbone(x.to(DEV))
    assert all(ops.cell_of(f) is None for f in feats)
    feats[-1].retain_grad()
    outs = [head(f) for head, f in zip(m.heads, feats)]                 # CenterNetDetection.forward (centernet_detection.py:40-41)
    loss, _ = m.loss(outs, {k: v.to(DEV) for k, v in tgt.items()})
    loss.backward()
    g = feats[-1].grad
    assert g is not None and g.shape == feats[-1].shape and float(g.float().abs().sum()) > 0


@pytest.mark.parametrize("task,arch", [("ctdet", "dla_34"), ("multi_pose", "res_18")])
def test_row_path_head_backward_equals_dense_on_the_whole_network(task, arch):
    """Every parameter gradient of a network (fp32 compute) with the heads' backward on the gathered rows (ops.HeadFn + SparseRows:
    width_height / regression, and keypoints / keypoint offsets for multi_pose) against the same network with the dense head
    backward: 2e-5 of each gradient's largest element — the sums are the same, taken in a different order."""
    from centernet_amd import ops
    m = _model(arch, 71, torch.float32, task).train()
    if task == "ctdet":
        x, tgt = synth.ctdet_batch(71, 2, 256, 256)
    else:
        x, tgt = synth.pose_batch(71, 2, 256, 256)
    x, tgt = x.to(DEV), {k: v.to(DEV) for k, v in tgt.items()}
    grads, runs = {}, {}
    for rows in (True, False):
        ops.SparseRows.enabled = rows
        try:
            m.zero_grad(set_to_none=True)
            before = ops.HeadFn.sparse_runs
            loss, _ = m.loss(m(x), tgt)
            loss.backward()
            runs[rows] = ops.HeadFn.sparse_runs - before
            grads[rows] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.SparseRows.enabled = True
    # (the keypoint-offset head gathers 128 x 17 rows: more than a 64x64 map has room for under the row path's 4 M <= H W rule)
    assert runs[False] == 0 and runs[True] == (2 if task == "ctdet" else 3), runs
    assert grads[True].keys() == grads[False].keys()
    # floor of the scale: the bias of a DCN conv in front of a batch-statistic BN has an identically-zero gradient (both runs hold
    # rounding noise there, a few 1e-9 of the network's largest gradient element)
    top = max(float(g.abs().max()) for g in grads[False].values())
    worst = {n: float((grads[True][n] - grads[False][n]).abs().max()) / max(float(grads[False][n].abs().max()), 1e-3 * top) for n in grads[True]}
    bad = {n: round(v, 8) for n, v in worst.items() if v > 2e-5}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:8])


@pytest.mark.gpu
def test_bare_dla_returns_normalised_level0_in_training_mode():
    """round-4 ADVICE: with the BN of the 16-channel 512^2 layers deferred to the consuming conv, a bare DLA handed out the RAW
    (pre-BN, pre-ReLU) level0 tensor as y[0].  The reference returns six normalised levels (pose_dla_dcn.py:372-378): by default
    level0 is materialised; only a caller that declares it never reads y[0] (DLASeg) gets the deferred form, and then y[0] is None."""
    from centernet_amd.models.backbones import pose_dla_dcn as P
    torch.manual_seed(0)
    m = P.dla34(compute_dtype=torch.bfloat16).cuda().train()
    rng.fill_state_dict(m, 11)
    img = synth.ctdet_batch(5, 2, 64, 64)[0].cuda()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        y = m(img)
    assert len(y) == 6 and all(t is not None for t in y) and getattr(y[0], "_cn_pre", None) is None
    assert float(y[0].float().min()) >= 0.0                      # post-ReLU
    m.load_state_dict(sd)                                          # same BN running statistics for the second pass
    m.expose_level0 = False
    with torch.no_grad():
        z = m(img)
    assert z[0] is None
    for a, b in zip(y[1:], z[1:]):
        assert torch.equal(a, b)                                   # applying the BN on load is bit-identical to storing it
    # ... and the eval-mode network (folded BN) returns the same normalised map either way
    m.eval()
    with torch.no_grad():
        e = m(img)
    assert e[0] is not None and float(e[0].float().min()) >= 0.0
