"""GPU parity of the decode and loss kernels: bit-exact indices / peak masks / detections against the golden vectors
made by the imported reference and against the oracle; losses within 1e-4 relative (north_star tolerance)."""
import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from oracle import ops_ref
from conftest import strided, summary

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _decode_inputs(seed, B, C, realistic, H=128, W=128):
    z = rng.t_normal(seed, "heat", (B, C, H, W))
    if realistic:
        z = 0.5 * z - 2.19
    return (torch.sigmoid(z), rng.t_uniform(seed, "wh", (B, 2, H, W), 1.0, 40.0), rng.t_uniform(seed, "reg", (B, 2, H, W)))


@pytest.mark.parametrize("tag", ["rand", "real", "small"])
def test_ctdet_decode_golden_bit_exact(golden, tag):
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.utils.decode import _nms, _topk, _topk_channel
    g = golden(f"decode_{tag}.npz")
    heat, wh, reg = _decode_inputs(int(g["seed"]), int(g["B"]), int(g["C"]), bool(g["realistic"]))
    K = int(g["K"])
    det, inds, clses = ctdet_decode(heat.to(DEV), wh.to(DEV), reg.to(DEV), K=K, return_aux=True)
    assert np.array_equal(inds.cpu().numpy(), g["inds"]), "top-k indices must be bit-identical"
    assert np.array_equal(clses.cpu().numpy(), g["clses"])
    assert np.array_equal(det.cpu().numpy(), g["det"])
    assert np.array_equal(ctdet_decode(heat.to(DEV), wh.to(DEV), None, K=K).cpu().numpy(), g["det_noreg"])
    # peak mask (max-pool pseudo NMS) bit-exact
    nm = _nms(heat.to(DEV)).cpu()
    keep = (nm == heat) & (heat != 0)
    assert np.array_equal(keep.flatten(2).sum(-1).numpy(), g["peak_popcount"])
    assert np.array_equal(np.packbits(keep[0, 0].numpy()), g["peak_mask_b0c0"])
    assert torch.equal(nm, ops_ref.nms(heat))
    # reference-named primitives
    s, i, ys, xs = _topk_channel(nm.to(DEV), K)
    assert np.array_equal(s[:, :3].cpu().numpy(), g["chan_scores"]) and np.array_equal(i[:, :3].cpu().numpy(), g["chan_inds"])
    s2, i2, c2, y2, x2 = _topk(nm.to(DEV), K)
    assert np.array_equal(s2.cpu().numpy(), g["scores"]) and np.array_equal(i2.cpu().numpy(), g["inds"])
    assert np.array_equal(c2.cpu().numpy(), g["clses"])


def test_known_answer_encode_decode(golden):
    """The reference's own known-answer test (tests/test_sample_encode_decode.py:35-56) through the HIP decode."""
    from centernet_amd.decode.ctdet import ctdet_decode
    g = golden("known_answer.npz")
    e = synth.encode_ctdet(synth.FIXTURE_BOXES)
    hm = torch.from_numpy(e["heatmap"]).unsqueeze(0)
    wh = torch.zeros(1, 2, 128, 128); reg = torch.zeros(1, 2, 128, 128)
    for k in range(2):
        y, x = divmod(int(e["indices"][k]), 128)
        wh[0, :, y, x] = torch.from_numpy(e["width_height"][k]); reg[0, :, y, x] = torch.from_numpy(e["regression"][k])
    det = ctdet_decode(hm.to(DEV), wh.to(DEV), reg.to(DEV))[0].cpu().numpy()
    assert np.array_equal(det, ops_ref.ctdet_decode(hm, wh, reg)[0].numpy()), "tie rule (2 peaks at 1.0, zeros elsewhere)"
    det = 4 * det[det[:, 4] > 0.5]
    assert len(det) == 2
    centers = (det[:, :2] + det[:, 2:4]) / 2
    assert abs(centers.sum() - float(g["ann_center_sum"])) < 1e-3
    assert np.array_equal(det[np.argsort(det[:, 0])], g["det_sorted"])


@pytest.mark.parametrize("case", ["plateau", "all_equal", "few_peaks", "negative", "odd_size", "k1", "k256"])
def test_topk_edge_cases_match_oracle(case):
    """Ties, plateaus (max-pool keeps ALL equal maxima), fewer than K peaks, non-square / non-multiple-of-4 maps."""
    from centernet_amd.decode.ctdet import ctdet_decode
    B, C, H, W, K = 2, 3, 32, 32, 40
    heat = torch.sigmoid(rng.t_normal(50, case, (B, C, H, W)))
    if case == "plateau":
        heat = (heat * 8).round() / 8          # heavy ties + plateaus
    elif case == "all_equal":
        heat = torch.full_like(heat, 0.25)
    elif case == "few_peaks":
        heat = torch.zeros_like(heat); heat[:, :, 5, 7] = 0.9; heat[:, 1, 20, 3] = 0.9; heat[0, 2, 31, 31] = 0.4
    elif case == "negative":
        heat = rng.t_normal(50, "neg", (B, C, H, W))   # raw (un-sigmoided) maps: negative peaks lose to masked zeros
    elif case == "odd_size":
        H, W = 34, 27
        heat = torch.sigmoid(rng.t_normal(50, "odd", (B, C, H, W)))
    elif case == "k1":
        K = 1
    elif case == "k256":
        K = 256
    wh = rng.t_uniform(51, case, (B, 2, H, W), 1, 9); reg = rng.t_uniform(52, case, (B, 2, H, W))
    det, inds, clses = ctdet_decode(heat.to(DEV), wh.to(DEV), reg.to(DEV), K=K, return_aux=True)
    rdet, rinds, rcls = ops_ref.ctdet_decode(heat, wh, reg, K=K, return_aux=True)
    assert torch.equal(inds.cpu(), rinds) and torch.equal(clses.cpu(), rcls)
    assert torch.equal(det.cpu(), rdet)


@pytest.mark.parametrize("case", ["plateau", "all_equal", "few_peaks", "negative", "k1", "k256", "bf16_logits", "two_values",
                                  "ties_cross_rows", "one_bin", "no_nms_random", "no_nms_flat"])
def test_topk_stream_kernel_edge_cases_128(case):
    """The streaming top-K (csrc/topk_stream.h: register-resident keys, 12-bit histogram select with early exit, index
    histograms for ties) only serves 128x128 maps: every tie / plateau / degenerate situation at THAT size, against the oracle."""
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.utils.decode import _topk_channel
    B, C, H, W, K = 2, 3, 128, 128, 100
    heat = torch.sigmoid(rng.t_normal(60, case, (B, C, H, W)))
    if case == "plateau":
        heat = (heat * 8).round() / 8
    elif case == "all_equal":                       # every element ties: the K lowest flat indices win (index histograms)
        heat = torch.full_like(heat, 0.25)
    elif case == "few_peaks":                       # fewer than K non-zero peaks: zeros (lowest indices) fill the tail
        heat = torch.zeros_like(heat); heat[:, :, 5, 7] = 0.9; heat[:, 1, 120, 3] = 0.9; heat[0, 2, 127, 127] = 0.4
    elif case == "negative":
        heat = rng.t_normal(60, "neg", (B, C, H, W))
    elif case == "k1":
        K = 1
    elif case == "k256":
        K = 256
    elif case == "bf16_logits":                     # head maps of an untrained bf16 network: a handful of distinct values per map
        heat = torch.sigmoid((rng.t_normal(60, "bf", (B, C, H, W)) * 0.01 - 2.19).to(torch.bfloat16).float())
    elif case == "two_values":                      # K-th value shared by thousands, a few hundred strictly above it
        z = rng.t_uniform(60, "tv", (B, C, H, W))
        heat = torch.where(z > 0.995, torch.full_like(z, 0.75), torch.full_like(z, 0.5))
    elif case == "ties_cross_rows":                 # the index cut falls in the middle of a row, 3 keys resolved to the last bit
        heat = torch.zeros_like(heat)
        heat[:, :, ::3, ::3] = 0.3                  # isolated equal peaks: 43*43 = 1849 of them
        heat[:, 0, 3, 6] = float(np.nextafter(np.float32(0.3), np.float32(1)))      # one ulp above
    elif case == "one_bin":                         # distinct values that share key bits [31:8]: resolved by the third pass
        base = np.float32(0.3).view(np.uint32) & np.uint32(0xffffff00)
        u = (rng.t_uniform(60, "ob", (B, C, H, W)) * 255).to(torch.int32).numpy().astype(np.uint32)
        heat = torch.from_numpy((base | u).view(np.float32))
    if case.startswith("no_nms"):                   # _topk_channel on raw maps (apply_nms = 0)
        if case == "no_nms_flat":
            heat = torch.full_like(heat, -1.5)
        s, i, ys, xs = _topk_channel(heat.to(DEV), K)
        rs, ri, rys, rxs = ops_ref.topk_channel(heat, K)
        assert torch.equal(s.cpu(), rs) and torch.equal(i.cpu(), ri) and torch.equal(ys.cpu(), rys) and torch.equal(xs.cpu(), rxs)
        return
    wh = rng.t_uniform(61, case, (B, 2, H, W), 1, 9); reg = rng.t_uniform(62, case, (B, 2, H, W))
    det, inds, clses = ctdet_decode(heat.to(DEV), wh.to(DEV), reg.to(DEV), K=K, return_aux=True)
    rdet, rinds, rcls = ops_ref.ctdet_decode(heat, wh, reg, K=K, return_aux=True)
    assert torch.equal(inds.cpu(), rinds) and torch.equal(clses.cpu(), rcls)
    assert torch.equal(det.cpu(), rdet)
    s, i, _, _ = _topk_channel(ops_ref.nms(heat).to(DEV), min(K, 128))       # per-class lists (stage 1 alone)
    rs, ri, _, _ = ops_ref.topk_channel(ops_ref.nms(heat), min(K, 128))
    assert torch.equal(s.cpu(), rs) and torch.equal(i.cpu(), ri)


@pytest.mark.parametrize("seed", range(8))
def test_topk_stream_kernel_fuzz_128(seed):
    """Random quantisation levels (ties of every multiplicity), random K, 1..5 classes."""
    from centernet_amd.decode.ctdet import ctdet_decode
    g = torch.Generator().manual_seed(1000 + seed)
    B, C = 2, int(torch.randint(1, 6, (1,), generator=g))
    K = int(torch.randint(1, 257, (1,), generator=g))
    levels = int(2 ** torch.randint(1, 16, (1,), generator=g))
    heat = torch.sigmoid(torch.randn(B, C, 128, 128, generator=g) * 1.5 - 1.0)
    heat = (heat * levels).round() / levels
    wh = torch.rand(B, 2, 128, 128, generator=g) * 20; reg = torch.rand(B, 2, 128, 128, generator=g)
    det, inds, clses = ctdet_decode(heat.to(DEV), wh.to(DEV), reg.to(DEV), K=K, return_aux=True)
    rdet, rinds, rcls = ops_ref.ctdet_decode(heat, wh, reg, K=K, return_aux=True)
    assert torch.equal(inds.cpu(), rinds) and torch.equal(clses.cpu(), rcls) and torch.equal(det.cpu(), rdet)


def test_decode_full_size_properties():
    """BASELINE config size (B=64, C=80, 128x128): size-independent properties instead of a CPU oracle run."""
    from centernet_amd.decode.ctdet import ctdet_decode
    B, C, H, W, K = 64, 80, 128, 128, 100
    gen = torch.Generator(device="cpu").manual_seed(7)
    heat = torch.sigmoid(torch.randn(B, C, H, W, generator=gen) * 0.5 - 2.19).to(DEV)
    wh = torch.rand(B, 2, H, W, generator=gen).to(DEV) * 30; reg = torch.rand(B, 2, H, W, generator=gen).to(DEV)
    det, inds, clses = ctdet_decode(heat, wh, reg, K=K, return_aux=True)
    sc = det[..., 4]
    assert bool((sc[:, :-1] >= sc[:, 1:]).all()), "scores sorted descending"
    flat = heat.flatten(2)
    picked = torch.gather(flat.view(B, -1), 1, clses.long() * H * W + inds)
    assert torch.equal(picked, sc), "score == heat[class, index]"
    # every pick is a 3x3 local maximum, and nothing outside the picks beats the K-th score as a peak
    pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
    peaks = torch.where(pooled == heat, heat, torch.zeros_like(heat)).view(B, -1)
    assert torch.equal(torch.gather(peaks, 1, clses.long() * H * W + inds), sc)
    kth = sc[:, -1:]
    assert bool(((peaks > kth).sum(1) <= K - 1).all())
    # idempotence / determinism
    det2 = ctdet_decode(heat, wh, reg, K=K)
    assert torch.equal(det, det2)
    # sub-batch consistency with the oracle on 2 images
    r = ops_ref.ctdet_decode(heat[:2].cpu(), wh[:2].cpu(), reg[:2].cpu(), K=K)
    assert torch.equal(det[:2].cpu(), r)


def test_multi_pose_decode_golden(golden):
    from centernet_amd.decode.multi_pose import multi_pose_decode
    g = golden("pose_decode.npz")
    seed, B, K = int(g["seed"]), int(g["B"]), int(g["K"])
    heat = torch.sigmoid(rng.t_normal(seed, "heat", (B, 1, 128, 128)))
    hm_hp = torch.sigmoid(rng.t_normal(seed, "hmhp", (B, 17, 128, 128)) * 0.7 - 1.0)
    wh = rng.t_uniform(seed, "wh", (B, 2, 128, 128), 4.0, 60.0); reg = rng.t_uniform(seed, "reg", (B, 2, 128, 128))
    kps = rng.t_normal(seed, "kps", (B, 34, 128, 128), 0, 6.0); hpo = rng.t_uniform(seed, "hpo", (B, 2, 128, 128))
    d = lambda t: t.to(DEV)
    det = multi_pose_decode(d(heat), d(wh), d(kps), reg=d(reg), hm_hp=d(hm_hp), hp_offset=d(hpo), K=K).cpu().numpy()
    np.testing.assert_array_equal(det, g["det"])
    det2 = multi_pose_decode(d(heat), d(wh), d(kps), reg=None, hm_hp=d(hm_hp), hp_offset=None, K=K).cpu().numpy()
    np.testing.assert_array_equal(det2, g["det_nooff"])


def test_unused_reference_options_on_device(golden):
    """NormRegL1Loss (utils/losses.py:66-78) and `_nms(heat, kernel=5)` (utils/decode.py:5) against the reference's outputs."""
    from centernet_amd.utils.decode import _nms
    from centernet_amd.utils.losses import NormRegL1Loss
    g = golden("unused_options.npz")
    seed = int(g["seed"])
    _, tgt = synth.ctdet_batch(seed, 2)
    whn = rng.t_normal(seed, "whn", (2, 2, 128, 128), 0, 5).to(DEV).requires_grad_(True)
    nrm = NormRegL1Loss()(whn, tgt["regression_mask"].to(DEV), tgt["indices"].to(DEV), tgt["width_height"].to(DEV))
    nrm.backward()
    assert nrm.item() == pytest.approx(float(g["nrm"]), rel=1e-5)
    np.testing.assert_allclose(summary(whn.grad), g["dwhn_sum"], rtol=1e-5)
    gnz = whn.grad.flatten()
    np.testing.assert_allclose(gnz[gnz != 0][:64].cpu().numpy(), g["dwhn_nz"], rtol=1e-5)
    heat5 = torch.sigmoid(rng.t_normal(seed, "heat5", (2, 3, 32, 40)))
    heat5[0, 1, 4:7, 10:14] = 0.75
    keep = _nms(heat5.to(DEV), 5)
    assert np.array_equal(torch.nonzero(keep.flatten()).flatten().cpu().numpy(), g["nms5_nz_idx"])
    assert torch.equal(keep.cpu()[keep.cpu() != 0], heat5[keep.cpu() != 0])
    assert torch.equal(_nms(heat5.to(DEV), 3).cpu(), heat5 * (torch.nn.functional.max_pool2d(heat5, 3, 1, 1) == heat5).float())
    with pytest.raises(ValueError):
        _nms(heat5.to(DEV), 4)


def test_losses_golden_and_grads(golden):
    from centernet_amd.utils.decode import sigmoid_clamped
    from centernet_amd.utils.losses import FocalLoss, RegL1Loss, RegWeightedL1Loss
    g = golden("losses.npz")
    seed = int(g["seed"])
    _, tgt = synth.ctdet_batch(seed, 2)
    tg = {k: v.to(DEV) for k, v in tgt.items()}
    logits = (rng.t_normal(seed, "logit", (2, 80, 128, 128)) * 1.5 - 2.19)
    whp = rng.t_normal(seed, "whp", (2, 2, 128, 128), 0, 5); regp = rng.t_normal(seed, "regp", (2, 2, 128, 128))
    lg = logits.to(DEV).requires_grad_(True); wg = whp.to(DEV).requires_grad_(True); rg = regp.to(DEV).requires_grad_(True)
    x = lg.clone()                                   # sigmoid_clamped works in place on its argument (like the reference)
    pred = sigmoid_clamped(x)
    assert torch.allclose(x.detach().cpu(), torch.sigmoid(logits), rtol=1e-6, atol=1e-7), "in-place sigmoid must be observable"
    hm = FocalLoss()(pred, tg["heatmap"])
    wh = RegL1Loss()(wg, tg["regression_mask"], tg["indices"], tg["width_height"])
    off = RegL1Loss()(rg, tg["regression_mask"], tg["indices"], tg["regression"])
    loss = hm + 0.1 * wh + off
    loss.backward()
    for v, k in ((hm, "hm"), (wh, "wh"), (off, "off"), (loss, "loss")):
        assert v.item() == pytest.approx(float(g[k]), rel=1e-4), k
    np.testing.assert_allclose(strided(lg.grad).cpu().numpy(), g["dlogits_s"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(summary(lg.grad), g["dlogits_sum"], rtol=1e-4)
    np.testing.assert_allclose(summary(wg.grad), g["dwh_sum"], rtol=1e-5)
    np.testing.assert_allclose(summary(rg.grad), g["dreg_sum"], rtol=1e-5)
    kpp = rng.t_normal(seed, "kpp", (2, 34, 128, 128), 0, 3).to(DEV)
    kmask = (rng.t_uniform(seed, "kmask", (2, 128, 34)) > 0.5).to(DEV)
    ktgt = rng.t_normal(seed, "ktgt", (2, 128, 34), 0, 3).to(DEV)
    assert RegWeightedL1Loss()(kpp, kmask, tg["indices"], ktgt).item() == pytest.approx(float(g["kp"]), rel=1e-4)
    gt0 = tg["heatmap"].clone(); gt0[gt0 == 1] = 0.99
    p0 = sigmoid_clamped(logits.to(DEV).clone())
    assert FocalLoss()(p0, gt0).item() == pytest.approx(float(g["hm_nopos"]), rel=1e-4), "num_pos == 0 branch"


def test_focal_broadcast_gt():
    """tests/test_train_multi_pose.py relies on _neg_loss broadcasting an 80-channel gt against a 1-channel prediction
    the other way round is what the kernel supports: gt with a size-1 batch/channel broadcast onto pred."""
    from centernet_amd.utils.losses import FocalLoss
    pred = torch.sigmoid(rng.t_normal(60, "p", (2, 5, 16, 16))).clamp(1e-4, 1 - 1e-4)
    gt = rng.t_uniform(60, "g", (1, 1, 16, 16)); gt[0, 0, 3, 3] = 1.0
    ref = ops_ref.focal_loss(pred, gt.expand_as(pred))
    out = FocalLoss()(pred.to(DEV), gt.to(DEV))
    assert out.item() == pytest.approx(ref.item(), rel=1e-5)


def test_encode_ctdet_on_device(golden):
    """cn_encode_ctdet vs the host restatement of sample/ctdet.py (synth.encode_ctdet, itself pinned to the
    reference by encode_fixture.npz): indices / masks bit-exact, fp32 targets exact, gaussians within one ulp of expf."""
    from centernet_amd import synth
    from centernet_amd.sample import CenterDetectionSample, encode_ctdet_batch
    B, M = 6, 128
    boxes = np.zeros((B, M, 4), np.float32); cls = np.zeros((B, M), np.int32); cnt = np.zeros((B,), np.int32)
    ref = []
    for b in range(B):
        bl = list(synth.FIXTURE_BOXES) if b == 0 else synth.random_boxes(77, b)
        if b == 1:                                   # degenerate boxes (clipped to zero extent) and a box hanging over the border
            bl = bl + [([600.0, 10.0, 20.0, 20.0], 3), ([500.0, 500.0, 40.0, 40.0], 5), ([-30.0, -30.0, 60.0, 90.0], 7)]
        bl = [([float(np.float32(v)) for v in bb], c) for bb, c in bl]
        for k, (bb, c) in enumerate(bl):
            boxes[b, k], cls[b, k] = bb, c
        cnt[b] = len(bl)
        ref.append(synth.encode_ctdet(bl))
    t = encode_ctdet_batch(torch.from_numpy(boxes).to(DEV), torch.from_numpy(cls).to(DEV), torch.from_numpy(cnt).to(DEV), 512, 512)
    for b in range(B):
        r = ref[b]
        assert np.array_equal(t["regression_mask"][b].cpu().numpy(), r["regression_mask"])
        assert np.array_equal(t["indices"][b].cpu().numpy(), r["indices"])
        assert np.array_equal(t["width_height"][b].cpu().numpy(), r["width_height"])
        assert np.array_equal(t["regression"][b].cpu().numpy(), r["regression"])
        hm = t["heatmap"][b].cpu().numpy()
        assert np.array_equal(hm != 0, r["heatmap"] != 0), "same support (gaussian truncation)"
        np.testing.assert_allclose(hm, r["heatmap"], rtol=3e-7, atol=1e-7)
        assert np.array_equal(hm == 1.0, r["heatmap"] == 1.0), "peaks are exactly 1"
    # the reference's transform signature, on the reference's own annotation fixture
    g = golden("encode_fixture.npz")
    ann = [{"bbox": [float(v) for v in bb], "class_id": int(c)} for bb, c in zip(g["boxes"], g["cls"])]
    _, one = CenterDetectionSample()(torch.zeros(3, 512, 512, device=DEV), ann)
    assert np.array_equal(one["indices"].cpu().numpy(), g["indices"])
    flat = one["heatmap"].flatten()
    assert np.array_equal(torch.nonzero(flat).flatten().cpu().numpy(), g["heatmap_nz_idx"])
    np.testing.assert_allclose(flat[flat != 0].cpu().numpy(), g["heatmap_nz_val"], rtol=3e-7, atol=1e-7)
    np.testing.assert_allclose(one["width_height"].cpu().numpy(), g["width_height"], rtol=1e-6)
    np.testing.assert_allclose(one["regression"].cpu().numpy(), g["regression"], rtol=1e-5, atol=1e-6)


def test_encode_ctdet_msra_on_device(golden):
    """The reference's other gaussian option for detection targets (sample/ctdet.py:53-55, gaussian_type="msra") through
    cn_encode_ctdet(gaussian_type=1), against the reference's own outputs (encode_msra_fixture.npz): same support, values within
    one ulp of expf, the 0/0 NaN pixels of radius-0 objects in the same places, patches touching the border dropped."""
    from centernet_amd.sample import CenterDetectionSample
    g = golden("encode_msra_fixture.npz")
    for i in range(int(g["n"])):
        ann = [{"bbox": [float(v) for v in bb], "class_id": int(c)} for bb, c in zip(g[f"boxes{i}"], g[f"cls{i}"])]
        _, t = CenterDetectionSample(gaussian_type="msra")(torch.zeros(3, 512, 512, device=DEV), ann)
        ref = np.zeros(80 * 128 * 128, np.float32)
        ref[g[f"hm_nz_idx{i}"]] = g[f"hm_nz_val{i}"]
        hm = t["heatmap"].flatten().cpu().numpy()
        assert np.array_equal(np.isnan(hm), np.isnan(ref)), f"set {i}: NaN pixels"
        ok = ~np.isnan(ref)
        assert np.array_equal(hm[ok] != 0, ref[ok] != 0), f"set {i}: support"
        np.testing.assert_allclose(hm[ok], ref[ok], rtol=3e-7, atol=1e-37)
        assert np.array_equal(t["indices"].cpu().numpy(), g[f"indices{i}"])
        assert np.array_equal(t["regression_mask"].cpu().numpy(), g[f"regression_mask{i}"])
        np.testing.assert_allclose(t["width_height"].cpu().numpy(), g[f"width_height{i}"], rtol=1e-6)
        np.testing.assert_allclose(t["regression"].cpu().numpy(), g[f"regression{i}"], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        CenterDetectionSample(gaussian_type="other")


def test_encode_multi_pose_on_device(golden):
    """SURVEY 8 f-3 (multi_pose): cn_encode_multi_pose against the reference's MultiPoseSample outputs (encode_pose_fixture.npz)
    and, on more annotation sets, against the host restatement that fixture pins: indices / masks / fp32 targets bit-exact,
    gaussians on the same support and within one ulp of expf."""
    from centernet_amd import synth
    from centernet_amd.sample import MultiPoseSample, encode_multi_pose_batch
    g = golden("encode_pose_fixture.npz")
    sets = [[(list(b), list(k)) for b, k in zip(g[f"boxes{i}"], g[f"kps{i}"])] for i in range(int(g["n"]))]
    sets += [synth.random_pose_anns(78, i) for i in range(4)]
    B, M, J = len(sets), 128, 17
    boxes = np.zeros((B, M, 4), np.float32); kps = np.zeros((B, M, J, 3), np.float32); cnt = np.zeros((B,), np.int32)
    for b, anns in enumerate(sets):
        for k, (bb, kp) in enumerate(anns):
            boxes[b, k], kps[b, k] = bb, np.array(kp, np.float32).reshape(J, 3)
        cnt[b] = len(anns)
    t = encode_multi_pose_batch(torch.from_numpy(boxes).to(DEV), torch.from_numpy(kps).to(DEV), torch.from_numpy(cnt).to(DEV), 512, 512)
    for b, anns in enumerate(sets):
        r = synth.encode_multi_pose(anns)
        for k in ("keypoints", "keypoints_mask", "heatmap_keypoints_offset", "heatmap_keypoints_indices", "heatmap_keypoints_mask"):
            assert np.array_equal(t[k][b].cpu().numpy(), r[k]), (b, k)
        hm = t["heatmap_keypoints"][b].cpu().numpy()
        assert np.array_equal(hm != 0, r["heatmap_keypoints"] != 0), "same support"
        np.testing.assert_allclose(hm, r["heatmap_keypoints"], rtol=3e-7, atol=1e-7)
        e = synth.encode_ctdet([(bb, 0) for bb, _ in anns], num_classes=1)
        assert np.array_equal(t["indices"][b].cpu().numpy(), e["indices"]) and np.array_equal(t["regression_mask"][b].cpu().numpy(), e["regression_mask"])
        np.testing.assert_allclose(t["heatmap"][b].cpu().numpy(), e["heatmap"], rtol=3e-7, atol=1e-7)
    for i in range(int(g["n"])):                       # straight against the reference's outputs
        flat = t["heatmap_keypoints"][i].flatten()
        assert np.array_equal(torch.nonzero(flat).flatten().cpu().numpy(), g[f"hm_nz_idx{i}"])
        np.testing.assert_allclose(flat[flat != 0].cpu().numpy(), g[f"hm_nz_val{i}"], rtol=3e-7, atol=1e-7)
        assert np.array_equal(t["keypoints"][i].cpu().numpy(), g[f"keypoints{i}"])
        assert np.array_equal(t["heatmap_keypoints_indices"][i].cpu().numpy(), g[f"heatmap_keypoints_indices{i}"])
    # the reference's transform signature
    ann = [{"bbox": bb, "keypoints": kp} for bb, kp in sets[1]]
    _, one = MultiPoseSample()(torch.zeros(3, 512, 512, device=DEV), ann)
    assert np.array_equal(one["keypoints_mask"].cpu().numpy(), g["keypoints_mask1"])
    assert np.array_equal(one["heatmap_keypoints_offset"].cpu().numpy(), g["heatmap_keypoints_offset1"])


def test_empty_inputs_losses_and_encoders():
    """Images without objects: all-false regression masks give an exactly-zero L1 loss (0 / (0 + 1e-4), utils/losses.py:62)
    with zero gradients, the focal loss takes its num_pos == 0 branch, and the encoders emit empty targets (nobj = 0)."""
    from centernet_amd.sample import encode_ctdet_batch, encode_multi_pose_batch
    from centernet_amd.utils.decode import sigmoid_clamped
    from centernet_amd.utils.losses import FocalLoss, RegL1Loss, RegWeightedL1Loss
    B, M = 2, 128
    t = encode_ctdet_batch(torch.zeros(B, M, 4, device=DEV), torch.zeros(B, M, dtype=torch.int32, device=DEV),
                           torch.zeros(B, dtype=torch.int32, device=DEV), 512, 512)
    assert float(t["heatmap"].abs().max()) == 0 and not bool(t["regression_mask"].any()) and int(t["indices"].abs().max()) == 0
    p = encode_multi_pose_batch(torch.zeros(B, M, 4, device=DEV), torch.zeros(B, M, 17, 3, device=DEV),
                                torch.zeros(B, dtype=torch.int32, device=DEV), 512, 512)
    assert float(p["heatmap_keypoints"].abs().max()) == 0 and not bool(p["keypoints_mask"].any()) and not bool(p["heatmap_keypoints_mask"].any())
    wh = rng.t_normal(61, "wh", (B, 2, 128, 128)).to(DEV).requires_grad_(True)
    l1 = RegL1Loss()(wh, t["regression_mask"], t["indices"], t["width_height"])
    l1.backward()
    assert l1.item() == 0.0 and float(wh.grad.abs().max()) == 0.0
    kp = rng.t_normal(61, "kp", (B, 34, 128, 128)).to(DEV).requires_grad_(True)
    lk = RegWeightedL1Loss()(kp, p["keypoints_mask"], t["indices"], p["keypoints"])
    lk.backward()
    assert lk.item() == 0.0 and float(kp.grad.abs().max()) == 0.0
    logits = (rng.t_normal(61, "lg", (B, 80, 128, 128)) - 2.19)
    lg = logits.to(DEV).requires_grad_(True)
    hm = FocalLoss()(sigmoid_clamped(lg.clone()), t["heatmap"])
    ref = ops_ref.focal_loss(ops_ref.sigmoid_clamped(logits.clone()), torch.zeros(B, 80, 128, 128))
    hm.backward()
    assert hm.item() == pytest.approx(ref.item(), rel=1e-4) and bool(torch.isfinite(lg.grad).all())


def test_fused_sigmoid_focal_matches_two_step():
    """FocalLoss.on_logits (one autograd node, single-pass backward) == sigmoid_clamped followed by FocalLoss: same clamped map,
    same in-place sigmoid, same loss, same d loss / d logits (incl. saturated logits where the clamp blocks the gradient)."""
    from centernet_amd.utils.decode import sigmoid_clamped
    from centernet_amd.utils.losses import FocalLoss
    _, tgt = synth.ctdet_batch(62, 2)
    gt = tgt["heatmap"].to(DEV)
    logits = rng.t_normal(62, "lg", (2, 80, 128, 128)) * 6.0 - 2.19          # |z| up to ~30: both clamp bounds are hit
    a = logits.to(DEV).requires_grad_(True)
    xa = a.clone()
    ya = sigmoid_clamped(xa)
    la = FocalLoss()(ya, gt)
    (3.0 * la).backward()
    from centernet_amd import ops
    b = logits.to(DEV).requires_grad_(True)
    xb = b.clone()
    xb._cn_head_dtype = torch.bfloat16       # what models/heads.py tags a bf16 HeadFn's map with: only then is the second layout written
    noted = []
    ops.DualLayout.note = classmethod(lambda cls, t, alt: noted.append((t, alt)))
    try:
        yb, lb = FocalLoss().on_logits(xb, gt)
        (3.0 * lb).backward()
    finally:
        del ops.DualLayout.note              # back to the inherited SparseRows.note
    assert torch.equal(ya, yb) and torch.equal(xa, xb), "clamped copy and in-place sigmoid"
    assert la.item() == lb.item()
    assert float((yb == 1e-4).float().mean()) > 0.01 and float((yb == 1 - 1e-4).float().mean()) > 1e-4
    assert torch.equal(a.grad, b.grad), "single-pass backward is bit-identical to the two kernels"
    # the backward also left d loss / d logits as NHWC bf16 for a head's backward (ops.DualLayout): exactly the layout change of dz
    from centernet_amd._hip import call, CN_BF16
    (dz, alt), = noted
    assert not ops.DualLayout.entries and not ops.SparseRows.entries      # nothing stays pinned after a backward pass
    ref = torch.empty_like(alt)
    call("cn_nchw_to_nhwc", dz, ref, 2, 80, 128, 128, 80, CN_BF16)
    assert alt.dtype == torch.bfloat16 and tuple(alt.shape) == (2, 128, 128, 80) and torch.equal(alt, ref)


@pytest.mark.parametrize("cfg", [(2, 80, 128, 128, 100, 0), (3, 5, 128, 128, 40, 1), (1, 17, 128, 128, 7, 2), (2, 3, 64, 64, 20, 0)])
def test_ctdet_decode_from_logits_equals_sigmoid_then_decode(cfg):
    """cn_ctdet_decode_logits: the top-K kernel applies clamp(sigmoid(x), lo, 1 - lo) to the logits it loads — the arithmetic of
    cn_sigmoid_clamp_fwd — so detections, indices and classes are BIT-identical to sigmoid_clamped + ctdet_decode
    (centernet_detection.py:183-187) and the logits stay untouched.  Quantised logits (many ties), saturating logits (sigmoid = clamp
    bound: plateaus of equal scores) included; a 64x64 map takes the two-step fallback."""
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.utils.decode import sigmoid_clamped
    B, C, H, W, K, kind = cfg
    z = rng.t_normal(77, f"z{cfg}", (B, C, H, W))
    z = z * 0.5 - 2.19 if kind == 0 else ((z * 4).round() / 4 if kind == 1 else z * 12.0)
    wh = rng.t_uniform(77, f"wh{cfg}", (B, 2, H, W), 1, 30).to(DEV)
    reg = rng.t_uniform(77, f"reg{cfg}", (B, 2, H, W)).to(DEV)
    zl = z.to(DEV).contiguous()
    keep = zl.clone()
    got = ctdet_decode(zl, wh, reg, K=K, return_aux=True, logits_clamp=1e-4)
    assert torch.equal(zl, keep), "the logits are left untouched"
    ref = ctdet_decode(sigmoid_clamped(zl.clone(), 1e-4), wh, reg, K=K, return_aux=True)
    for a, b, nm in zip(got, ref, ("det", "inds", "classes")):
        assert torch.equal(a, b), nm


@pytest.mark.parametrize("seed", list(range(12)))
def test_ctdet_decode_fuzz_bit_exact(seed):
    """Random map sizes (both NMS code paths: rows that divide the 1024-thread strip layout and rows that do not), class
    counts, K and heat statistics (smooth, quantised = many ties / plateaus, sparse): indices, classes and boxes bit-exact
    against the oracle's ctdet_decode with the (score desc, index asc) tie rule."""
    from centernet_amd.decode.ctdet import ctdet_decode
    u = rng.uniform(900 + seed, "cfg", (8,))
    B = 1 + int(u[0] * 3)
    C = [1, 2, 5, 17, 80][int(u[1] * 5)]
    H, W = [(128, 128), (64, 64), (32, 64), (48, 40), (17, 23), (96, 128), (128, 32), (8, 8)][int(u[2] * 8)]
    K = min([1, 7, 40, 100][int(u[3] * 4)], H * W)
    z = rng.t_normal(900 + seed, "heat", (B, C, H, W))
    kind = int(u[4] * 3)
    heat = torch.sigmoid(z * 0.5 - 2.19) if kind == 0 else (torch.sigmoid(z) * 16).round() / 16 if kind == 1 else torch.where(z > 1.5, torch.sigmoid(z), torch.zeros_like(z))
    wh = rng.t_uniform(900 + seed, "wh", (B, 2, H, W), 1, 30)
    reg = rng.t_uniform(900 + seed, "reg", (B, 2, H, W)) if u[5] > 0.3 else None
    det, inds, clses = ctdet_decode(heat.to(DEV), wh.to(DEV), None if reg is None else reg.to(DEV), K=K, return_aux=True)
    rdet, rinds, rcls = ops_ref.ctdet_decode(heat, wh, reg, K=K, return_aux=True)
    assert torch.equal(inds.cpu(), rinds) and torch.equal(clses.cpu(), rcls), (B, C, H, W, K, kind)
    assert torch.equal(det.cpu(), rdet), (B, C, H, W, K, kind)


@pytest.mark.parametrize("seed", list(range(6)))
def test_multi_pose_decode_fuzz(seed):
    """Random map sizes / K / offset options against the oracle's multi_pose_decode (continuous inputs: no ties)."""
    from centernet_amd.decode.multi_pose import multi_pose_decode
    u = rng.uniform(950 + seed, "cfg", (6,))
    B = 1 + int(u[0] * 2)
    H, W = [(128, 128), (64, 96), (40, 48), (32, 32)][int(u[1] * 4)]
    K = [20, 50, 100][int(u[2] * 3)]
    heat = torch.sigmoid(rng.t_normal(950 + seed, "heat", (B, 1, H, W)))
    hm_hp = torch.sigmoid(rng.t_normal(950 + seed, "hmhp", (B, 17, H, W)) * 0.7 - 1.0)
    wh = rng.t_uniform(950 + seed, "wh", (B, 2, H, W), 4.0, 0.4 * min(H, W))
    kps = rng.t_normal(950 + seed, "kps", (B, 34, H, W), 0, 5.0)
    reg = rng.t_uniform(950 + seed, "reg", (B, 2, H, W)) if u[3] > 0.4 else None
    hpo = rng.t_uniform(950 + seed, "hpo", (B, 2, H, W)) if u[4] > 0.4 else None
    d = lambda t: None if t is None else t.to(DEV)
    det = multi_pose_decode(d(heat), d(wh), d(kps), reg=d(reg), hm_hp=d(hm_hp), hp_offset=d(hpo), K=K).cpu()
    ref = ops_ref.multi_pose_decode(heat, wh, kps, reg=reg, hm_hp=hm_hp, hp_offset=hpo, K=K)
    assert det.shape == ref.shape == (B, K, 57)
    np.testing.assert_array_equal(det.numpy(), ref.numpy())


def test_decode_is_stable_next_to_other_streams():
    """Regression for a read/write race on the radix select's shared `need` counter: a late wave saw the already-reduced count,
    picked a second pivot bin and overran the survivor list (garbage indices -> GPU memory fault in the gather).  It needed the
    decode to share the GPU with another stream's kernels (about one map in 10^6), exactly what TrainStep's post_forward does.
    Here: 300 full-size decodes on a side stream while the main stream runs conv + BN forward/backward; every result must be
    bit-identical (a stress test: the original fault needed ~4e5 maps under a real backward pass; tools/long_run.py reproduces that)."""
    from centernet_amd.decode.ctdet import ctdet_decode
    B, C, H, W = 64, 80, 128, 128
    heat = torch.clamp(torch.sigmoid(rng.t_normal(70, "heat", (B, C, H, W)) * 2.0 - 6.0), 1e-4, 1 - 1e-4).to(DEV)
    wh = rng.t_uniform(70, "wh", (B, 2, H, W), 1, 30).to(DEV)
    reg = rng.t_uniform(70, "reg", (B, 2, H, W)).to(DEV)
    ref = ctdet_decode(heat, wh, reg)
    torch.cuda.synchronize()
    from centernet_amd import nn as hnn
    conv = hnn.Conv2d(64, 64, 3, 1, 1).to(DEV)
    bnm = hnn.BatchNorm2d(64).to(DEV).train()
    xin = torch.randn(32, 128, 128, 64, device=DEV).to(torch.bfloat16).requires_grad_(True)
    side = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int32, device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    for it in range(300):
        with torch.cuda.stream(side):
            det = ctdet_decode(heat, wh, reg)
            bad += (det != ref).any().int()
        y = bnm(conv(xin), None, True)         # main-stream work shaped like the backward pass the decode overlaps in a
        y.backward(y.detach())                 # TrainStep: LDS-heavy conv / weight-gradient kernels and streaming BN passes
        xin.grad = None
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(bad) == 0
