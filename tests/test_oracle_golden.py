"""CPU: the oracle (torch restatement) against golden vectors produced by the imported reference
(oracle/gen_golden.py) and against the reference's own known-answer test."""
import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from oracle import ops_ref, models_ref
from conftest import strided, summary, assert_det_rank_tolerant

torch.set_num_threads(8)


def _decode_inputs(seed, B, C, realistic):
    z = rng.t_normal(seed, "heat", (B, C, 128, 128))
    if realistic:
        z = 0.5 * z - 2.19
    return (torch.sigmoid(z), rng.t_uniform(seed, "wh", (B, 2, 128, 128), 1.0, 40.0),
            rng.t_uniform(seed, "reg", (B, 2, 128, 128), 0.0, 1.0))


def test_encoder_matches_reference_fixture(golden):
    g = golden("encode_fixture.npz")
    boxes = [(list(b), int(c)) for b, c in zip(g["boxes"], g["cls"])]
    e = synth.encode_ctdet(boxes)
    hm = e["heatmap"].reshape(-1)
    nz = np.nonzero(hm)[0]
    assert np.array_equal(nz, g["heatmap_nz_idx"])
    np.testing.assert_allclose(hm[nz], g["heatmap_nz_val"], rtol=1e-6)   # exp() differs in the last ulp (numpy vs torch)
    assert np.array_equal(hm[nz] == 1, g["heatmap_nz_val"] == 1)        # peaks exactly 1 (focal-loss gt==1 test)
    for k in ("indices", "width_height", "regression", "regression_mask"):
        assert np.array_equal(e[k], g[k]), k


def test_msra_encoder_matches_reference_fixture(golden):
    """sample/ctdet.py:53-55 with gaussian_type="msra": the host restatement against the reference's outputs, NaN pixels included."""
    g = golden("encode_msra_fixture.npz")
    for i in range(int(g["n"])):
        boxes = [(list(map(float, b)), int(c)) for b, c in zip(g[f"boxes{i}"], g[f"cls{i}"])]
        e = synth.encode_ctdet(boxes, gaussian="msra")
        ref = np.zeros(e["heatmap"].size, np.float32)
        ref[g[f"hm_nz_idx{i}"]] = g[f"hm_nz_val{i}"]
        hm = e["heatmap"].reshape(-1)
        assert np.array_equal(np.isnan(hm), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(hm), np.nan_to_num(ref), rtol=1e-6, atol=0)
        for k in ("indices", "width_height", "regression", "regression_mask"):
            assert np.array_equal(e[k], g[f"{k}{i}"]), k


def test_known_answer_encode_decode(golden):
    """reference tests/test_sample_encode_decode.py:35-56 restated on the oracle."""
    g = golden("known_answer.npz")
    e = synth.encode_ctdet(synth.FIXTURE_BOXES)
    hm = torch.from_numpy(e["heatmap"]).unsqueeze(0)
    wh = torch.zeros(1, 2, 128, 128)
    reg = torch.zeros(1, 2, 128, 128)
    for k in range(2):
        y, x = divmod(int(e["indices"][k]), 128)
        wh[0, :, y, x] = torch.from_numpy(e["width_height"][k])
        reg[0, :, y, x] = torch.from_numpy(e["regression"][k])
    det = ops_ref.ctdet_decode(hm, wh, reg)[0].numpy()
    det = 4 * det[det[:, 4] > 0.5]
    assert len(det) == int(g["n_det"]) == 2
    centers = (det[:, :2] + det[:, 2:4]) / 2
    ann = sum(b[0] + b[2] / 2 + b[1] + b[3] / 2 for b, _ in synth.FIXTURE_BOXES)
    assert abs(centers.sum() - ann) < 1e-3
    assert np.array_equal(det[np.argsort(det[:, 0])], g["det_sorted"])


@pytest.mark.parametrize("tag", ["rand", "real", "small"])
def test_decode_bit_exact(golden, tag):
    g = golden(f"decode_{tag}.npz")
    heat, wh, reg = _decode_inputs(int(g["seed"]), int(g["B"]), int(g["C"]), bool(g["realistic"]))
    det, inds, clses = ops_ref.ctdet_decode(heat, wh, reg, K=int(g["K"]), return_aux=True)
    assert np.array_equal(inds.numpy(), g["inds"])
    assert np.array_equal(clses.numpy(), g["clses"])
    assert np.array_equal(det.numpy(), g["det"])
    assert np.array_equal(ops_ref.ctdet_decode(heat, wh, None, K=int(g["K"])).numpy(), g["det_noreg"])
    keep = ops_ref.peak_mask(heat) & (heat != 0)
    assert np.array_equal(keep.flatten(2).sum(-1).numpy(), g["peak_popcount"])
    assert np.array_equal(np.packbits(keep[0, 0].numpy()), g["peak_mask_b0c0"])
    s, i, _, _ = ops_ref.topk_channel(ops_ref.nms(heat), int(g["K"]))
    assert np.array_equal(s[:, :3].numpy(), g["chan_scores"]) and np.array_equal(i[:, :3].numpy(), g["chan_inds"])


def test_unused_reference_options(golden):
    """Two options the reference defines and never uses, pinned by its own outputs: NormRegL1Loss (utils/losses.py:66-78) and the
    pseudo-NMS with a 5x5 window (utils/decode.py:5)."""
    g = golden("unused_options.npz")
    seed = int(g["seed"])
    _, tgt = synth.ctdet_batch(seed, 2)
    whn = rng.t_normal(seed, "whn", (2, 2, 128, 128), 0, 5).requires_grad_(True)
    nrm = ops_ref.norm_reg_l1_loss(whn, tgt["regression_mask"], tgt["indices"], tgt["width_height"])
    nrm.backward()
    assert nrm.item() == pytest.approx(float(g["nrm"]), rel=1e-6)
    np.testing.assert_allclose(summary(whn.grad), g["dwhn_sum"], rtol=1e-6)
    heat5 = torch.sigmoid(rng.t_normal(seed, "heat5", (2, 3, 32, 40)))
    heat5[0, 1, 4:7, 10:14] = 0.75
    keep = ops_ref.nms(heat5, 5)
    assert np.array_equal(torch.nonzero(keep.flatten()).flatten().numpy(), g["nms5_nz_idx"])


def test_losses(golden):
    g = golden("losses.npz")
    seed = int(g["seed"])
    _, tgt = synth.ctdet_batch(seed, 2)
    logits = (rng.t_normal(seed, "logit", (2, 80, 128, 128)) * 1.5 - 2.19).requires_grad_(True)
    whp = rng.t_normal(seed, "whp", (2, 2, 128, 128), 0, 5).requires_grad_(True)
    regp = rng.t_normal(seed, "regp", (2, 2, 128, 128)).requires_grad_(True)
    loss, st = ops_ref.ctdet_loss({"heatmap": logits, "width_height": whp, "regression": regp}, tgt)
    loss.backward()
    for k, gk in (("hm_loss", "hm"), ("wh_loss", "wh"), ("off_loss", "off"), ("loss", "loss")):
        assert st[k].item() == pytest.approx(float(g[gk]), rel=1e-6), k
    np.testing.assert_allclose(strided(logits.grad).numpy(), g["dlogits_s"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(summary(whp.grad), g["dwh_sum"], rtol=1e-6)
    np.testing.assert_allclose(summary(regp.grad), g["dreg_sum"], rtol=1e-6)
    kpp = rng.t_normal(seed, "kpp", (2, 34, 128, 128), 0, 3)
    kmask = rng.t_uniform(seed, "kmask", (2, 128, 34)) > 0.5
    ktgt = rng.t_normal(seed, "ktgt", (2, 128, 34), 0, 3)
    assert ops_ref.reg_weighted_l1_loss(kpp, kmask, tgt["indices"], ktgt).item() == pytest.approx(float(g["kp"]), rel=1e-6)
    gt0 = tgt["heatmap"].clone()
    gt0[gt0 == 1] = 0.99
    assert ops_ref.focal_loss(ops_ref.sigmoid_clamped(logits.detach()), gt0).item() == pytest.approx(float(g["hm_nopos"]), rel=1e-6)


def test_pose_decode(golden):
    g = golden("pose_decode.npz")
    seed, B, K = int(g["seed"]), int(g["B"]), int(g["K"])
    heat = torch.sigmoid(rng.t_normal(seed, "heat", (B, 1, 128, 128)))
    hm_hp = torch.sigmoid(rng.t_normal(seed, "hmhp", (B, 17, 128, 128)) * 0.7 - 1.0)
    wh = rng.t_uniform(seed, "wh", (B, 2, 128, 128), 4.0, 60.0)
    reg = rng.t_uniform(seed, "reg", (B, 2, 128, 128))
    kps = rng.t_normal(seed, "kps", (B, 34, 128, 128), 0, 6.0)
    hpo = rng.t_uniform(seed, "hpo", (B, 2, 128, 128))
    det = ops_ref.multi_pose_decode(heat, wh, kps, reg=reg, hm_hp=hm_hp, hp_offset=hpo, K=K)
    np.testing.assert_allclose(det.numpy(), g["det"], rtol=0, atol=0)
    det2 = ops_ref.multi_pose_decode(heat, wh, kps, reg=None, hm_hp=hm_hp, hp_offset=None, K=K)
    np.testing.assert_allclose(det2.numpy(), g["det_nooff"], rtol=0, atol=0)


@pytest.mark.parametrize("arch,size,train", [("res_18", 256, False), ("res_18", 256, True),
                                             ("dla_34", 128, False), ("dla_34", 128, True),
                                             ("resdcn_18", 128, False), ("resdcn_18", 128, True),
                                             ("res_101", 128, False), ("res_101", 256, True),         # Bottleneck (msra_resnet.py:61-100)
                                             ("resdcn_101", 128, False), ("resdcn_101", 256, True)])
def test_model_matches_reference_graph(golden, arch, size, train):
    name = arch.replace("_", "") + ("_train" if train else "_eval") + ".npz"
    g = golden(name)
    seed = int(g["seed"])
    net = models_ref.CenterNetRef(arch)
    rng.fill_state_dict(net, seed, var_scale=float(g["var_scale"]) if "var_scale" in g.files else 1.0)
    net.train(train)
    x, tgt = synth.ctdet_batch(seed, 2, size, size)
    feat = net.backbone(x)[0]
    out = net.heads[0](feat)
    np.testing.assert_allclose(strided(feat).numpy(), g["feat_s"], rtol=1e-4, atol=1e-5)
    for k in ("heatmap", "width_height", "regression"):
        np.testing.assert_allclose(strided(out[k]).numpy(), g[f"{k}_s"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(summary(out[k]), g[f"{k}_sum"], rtol=1e-4)
    loss, st = net.loss([out], tgt)
    for k, gk in (("hm_loss", "hm"), ("wh_loss", "wh"), ("off_loss", "off"), ("loss", "loss")):
        assert st[k].item() == pytest.approx(float(g[gk]), rel=1e-4), k
    if train:
        loss.backward()
        params = dict(net.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                np.testing.assert_allclose(strided(params[n].grad, 512).numpy(), g[key], rtol=2e-3, atol=1e-6, err_msg=n)
        dead = sorted(n for n, p in params.items() if p.grad is None)
        assert dead == sorted(str(s) for s in g["dead_params"])
    else:
        det = ops_ref.ctdet_decode(ops_ref.sigmoid_clamped(out["heatmap"]), out["width_height"], out["regression"])
        np.testing.assert_allclose(det.detach().numpy(), g["det"], rtol=1e-4, atol=1e-4)


POSE_HEADS = ("heatmap", "width_height", "regression", "heatmap_keypoints", "keypoints", "heatmap_keypoints_offset")
POSE_STATS = ("loss", "hm_loss", "kp_loss", "hm_kp_loss", "hm_offset_loss", "wh_loss", "off_loss")


@pytest.mark.parametrize("train", [False, True])
def test_pose_model_matches_reference_graph(golden, train):
    """C5 (DLA-34 multi_pose, 6 heads) pinned to the reference: its DLASeg + CenterHead and its own `CenterNetMultiPose.loss`
    body (centernet_multi_pose.py:97-155, compiled from the source by oracle/gen_golden.py gen_pose_models)."""
    g = golden("dla34_pose_train.npz" if train else "dla34_pose_eval.npz")
    seed, size = int(g["seed"]), int(g["size"])
    net = models_ref.CenterNetRef("dla_34", task="pose")
    rng.fill_state_dict(net, seed)
    net.train(train)
    x, tgt = synth.pose_batch(seed, 2, size, size)
    feat = net.backbone(x)[0]
    out = net.heads[0](feat)
    np.testing.assert_allclose(strided(feat).numpy(), g["feat_s"], rtol=1e-4, atol=1e-5)
    for k in POSE_HEADS:
        np.testing.assert_allclose(strided(out[k]).numpy(), g[f"{k}_s"], rtol=1e-4, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(summary(out[k]), g[f"{k}_sum"], rtol=1e-4, err_msg=k)
    raw = {k: v.detach().clone() for k, v in out.items()}
    loss, st = net.loss([out], tgt)
    for k in POSE_STATS:
        assert float(st[k]) == pytest.approx(float(g["stat:" + k]), rel=1e-4), k
    if train:
        loss.backward()
        params = dict(net.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                np.testing.assert_allclose(strided(params[n].grad, 512).numpy(), g[key], rtol=2e-3, atol=1e-6, err_msg=n)
    else:
        # the centre scores of the reference's decode of its own (near-flat, heavily tied) heat map: sorted values are tie-proof
        heat = ops_ref.sigmoid_clamped(torch.from_numpy(g["map:heatmap"]))
        sc = ops_ref.topk(ops_ref.nms(heat), 100)[0]
        np.testing.assert_array_equal(sc.reshape(2, 100).numpy(), g["det_scores"])


@pytest.mark.parametrize("train", [False, True])
def test_hourglass_matches_reference(golden, train):
    """the oracle's Hourglass restatement against the reference's own HourglassNet + 2 CenterHeads."""
    g = golden("hourglass_train.npz" if train else "hourglass_eval.npz")
    seed, size = int(g["seed"]), int(g["size"])
    net = models_ref.CenterNetRef("hourglass")
    rng.fill_state_dict(net, seed, var_scale=float(g["var_scale"]))
    net.train(train)
    x, tgt = synth.ctdet_batch(seed, 2, size, size)
    feats = net.backbone(x)
    outs = [h(f) for h, f in zip(net.heads, feats)]
    assert len(outs) == 2
    for s_, (f, out) in enumerate(zip(feats, outs)):
        np.testing.assert_allclose(strided(f).numpy(), g[f"feat{s_}_s"], rtol=1e-4, atol=1e-5)
        for k in ("heatmap", "width_height", "regression"):
            np.testing.assert_allclose(strided(out[k]).numpy(), g[f"{k}{s_}_s"], rtol=1e-4, atol=1e-5)
    loss, st = net.loss(outs, tgt)
    for k, gk in (("hm_loss", "hm"), ("wh_loss", "wh"), ("off_loss", "off"), ("loss", "loss")):
        assert st[k].item() == pytest.approx(float(g[gk]), rel=1e-4), k
    if train:
        loss.backward()
        params = dict(net.named_parameters())
        for key in g.files:
            if key.startswith("g:") and key.endswith(":s"):
                n = key[2:-2]
                np.testing.assert_allclose(strided(params[n].grad, 512).numpy(), g[key], rtol=2e-3, atol=1e-6, err_msg=n)
        assert [n for n, p in params.items() if p.grad is None] == []
    else:
        o = outs[-1]
        det = ops_ref.ctdet_decode(ops_ref.sigmoid_clamped(o["heatmap"]), o["width_height"], o["regression"])
        assert_det_rank_tolerant(det.detach().numpy(), g["det"])      # random heads: a flat map, top-100 within 1e-6


def test_pose_encoder_matches_reference_fixture(golden):
    """SURVEY 8 f-3 (multi_pose): the numpy restatement of sample/multi_pose.py (synth.encode_multi_pose) is BIT-EXACT against
    the reference's MultiPoseSample on 4 annotation sets (gaussians dropped at the border, invisible joints, a clipped and a
    zero-width box)."""
    g = golden("encode_pose_fixture.npz")
    for i in range(int(g["n"])):
        anns = [(list(b), list(k)) for b, k in zip(g[f"boxes{i}"], g[f"kps{i}"])]
        T = synth.encode_multi_pose(anns)
        hk = T["heatmap_keypoints"].flatten()
        nz = np.nonzero(hk)[0]
        assert np.array_equal(nz, g[f"hm_nz_idx{i}"])
        assert np.array_equal(hk[nz], g[f"hm_nz_val{i}"])
        for k in ("keypoints", "keypoints_mask", "heatmap_keypoints_offset", "heatmap_keypoints_indices", "heatmap_keypoints_mask"):
            assert np.array_equal(T[k], g[f"{k}{i}"]), (i, k)


def test_soft_nms_restatement_known_answers():
    """utils/nms.py has no golden here (numba absent): hand-checkable cases of the restatement the HIP kernel is held to."""
    from oracle import post_ref
    # identical boxes: the second is decayed by exp(-1^2 / 0.5); a disjoint box is untouched; output sorted by final score
    b = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [100, 100, 109, 109, 0.85]], np.float32)
    n = post_ref.soft_nms(b, Nt=0.5, method=2)
    assert n == 3 and np.array_equal(b[:, 4] > 0, [True, True, True])
    assert b[0, 4] == np.float32(0.9) and b[1, 4] == np.float32(0.85)
    assert b[2, 4] == pytest.approx(0.8 * np.exp(-2.0), rel=1e-6)
    # hard NMS (method 0) zeroes an overlapping box, which then falls below the threshold and is dropped
    b = np.array([[0, 0, 9, 9, 0.9], [1, 1, 10, 10, 0.8], [50, 50, 59, 59, 0.3]], np.float32)
    n = post_ref.soft_nms(b, Nt=0.3, method=0)
    assert n == 2 and sorted(b[:n, 4].tolist()) == [pytest.approx(0.3), pytest.approx(0.9)]
    # IoU of the +1-pixel convention: boxes [0,0,9,9] and [5,0,14,9] overlap 5x10 of 150
    b = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float32)
    post_ref.soft_nms(b, Nt=0.5, method=1)
    assert b[1, 4] == pytest.approx(0.8)             # ov = 1/3 <= Nt: weight 1
    b = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float32)
    post_ref.soft_nms(b, method=2)
    assert b[1, 4] == pytest.approx(0.8 * np.exp(-(1 / 3) ** 2 / 0.5), rel=1e-6)


NMS_CASES = [("gauss", dict(Nt=0.5, method=2)), ("linear", dict(Nt=0.5, method=1)), ("hard", dict(Nt=0.3, method=0)),
             ("gauss39", dict(Nt=0.5, method=2))]


@pytest.mark.parametrize("name,args", NMS_CASES)
def test_soft_nms_restatement_matches_reference_source_run(golden, name, args):
    """tests/golden/soft_nms.npz: utils/nms.py:5-206 executed from the reference's own source with `numba.jit` shimmed to the
    identity (oracle/gen_golden.py:gen_soft_nms).  The whole array is compared, the discarded tail included: boxes and
    travelling columns exactly (they are only moved), scores to 1e-6 (float32 storage vs the shim run's float64)."""
    from oracle import post_ref
    g = golden("soft_nms.npz")
    b = g[name + "_in"].copy()
    n = (post_ref.soft_nms_39 if b.shape[1] > 5 else post_ref.soft_nms)(b, **args)
    ref = g[name + "_out"]
    assert n == int(g[name + "_n"]) and n < len(b)
    assert np.array_equal(b[:, :4], ref[:, :4].astype(np.float32)) and np.array_equal(b[:, 5:], ref[:, 5:].astype(np.float32))
    np.testing.assert_allclose(b[:, 4], ref[:, 4], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag,S", [("ms", 2), ("ss", 1)])
def test_test_step_end_restatement_matches_reference_source_run(golden, tag, S):
    """tests/golden/test_step_end.npz: the reference's own test_step_end bodies (centernet_detection.py:175-225,
    centernet_multi_pose.py:215-264) executed from source on seeded head maps of two test scales
    (oracle/gen_golden.py:gen_test_step_end).  Same rows, same order; scores to 1e-6 (soft-NMS decay)."""
    from oracle import post_ref
    g = golden("test_step_end.npz")
    outs = synth.tta_head_maps(int(g["det_seed"]), synth.DET_MAPS, synth.TTA_SIZES)[:S]
    dets = [ops_ref.ctdet_decode(o["heatmap"].sigmoid(), o["width_height"], reg=o["regression"])[0].numpy() for o in outs]
    res = post_ref.test_step_end(dets, synth.TTA_METAS[:S], 3)
    rows = np.concatenate([np.concatenate([res[j], np.full((len(res[j]), 1), j, np.float32)], 1) for j in sorted(res)])
    ref = g[f"det_{tag}_rows"]
    assert rows.shape == ref.shape and np.array_equal(rows[:, 5], ref[:, 5])
    np.testing.assert_allclose(rows[:, :4], ref[:, :4], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(rows[:, 4], ref[:, 4], rtol=1e-6, atol=1e-7)
    outs = synth.tta_head_maps(int(g["pose_seed"]), synth.POSE_MAPS, synth.TTA_SIZES)[:S]
    dets = [ops_ref.multi_pose_decode(o["heatmap"].sigmoid(), o["width_height"], o["keypoints"], reg=o["regression"],
                                      hm_hp=o["heatmap_keypoints"].sigmoid(), hp_offset=o["heatmap_keypoints_offset"])[0].numpy()
            for o in outs]
    rows = post_ref.pose_test_step_end(dets, synth.TTA_METAS[:S])
    ref = g[f"pose_{tag}_rows"]
    assert rows.shape == ref.shape
    np.testing.assert_allclose(np.delete(rows, 4, 1), np.delete(ref, 4, 1), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(rows[:, 4], ref[:, 4], rtol=1e-6, atol=1e-7)


def test_test_step_end_restatement_single_scale_is_a_regrouping():
    """Single scale: no NMS, K = 100 = test_max_per_image, so test_step_end only rescales and regroups (centernet_detection.py:189-223)."""
    from oracle import post_ref
    det = np.zeros((100, 6), np.float32)
    det[:, :4] = rng.uniform(3, "boxes", (100, 4)) * 128
    det[:, 4] = np.sort(rng.uniform(3, "sc", (100,)))[::-1]
    det[:, 5] = np.floor(rng.uniform(3, "cls", (100,)) * 80)
    res = post_ref.test_step_end([det], [{"scale": [1.0, 1.0], "padding": [16, 16]}], 80)
    assert sum(len(v) for v in res.values()) == 100
    j = int(det[0, 5]) + 1
    np.testing.assert_allclose(res[j][0, :4], det[0, :4] * 4 - 16, rtol=1e-6)
    assert post_ref.tta_pad(512, 31) == 16 and post_ref.tta_pad(512, 127) == 64 and post_ref.tta_pad(500, 31) == 6
