"""GPU: test-time augmentation + detection post-processing (soft-NMS of f-4) through the C ABI against the
oracle restatement of centernet_detection.py:132-225 / utils/nms.py (oracle/post_ref.py — soft-NMS and
test_step_end pinned by tests/golden/{soft_nms,test_step_end}.npz, the test_step half unpinned: see its header)."""
import numpy as np
import pytest
import torch

from centernet_amd import rng, synth
from oracle import post_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
MEAN, STD = [0.408, 0.447, 0.470], [0.289, 0.274, 0.278]


@pytest.mark.parametrize("cfg", [(2, 64, 96, 16, 0, True), (1, 50, 38, 7, 13, True), (3, 32, 32, 0, 0, False)])
def test_tta_prepare_bit_exact(cfg):
    from centernet_amd.utils import post
    B, H, W, px, py, flip = cfg
    img = rng.t_uniform(7, f"img{cfg}", (B, 3, H, W))
    ref = post_ref.tta_prepare(img, MEAN, STD, px, py, flip)
    got = post.tta_prepare(img.to(DEV), MEAN, STD, px, py, flip).cpu()
    assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.parametrize("cfg", [(2, 64, 96, 48, 72, 12, 8, True), (1, 50, 38, 62, 47, 0, 0, True), (2, 40, 56, 20, 28, 6, 6, False),
                                 (1, 33, 65, 99, 195, 14, 2, True), (2, 32, 32, 32, 32, 0, 0, True)])
def test_tta_prepare_scaled(cfg):
    """The multi-scale resize of test_step folded into the prepare launch: bilinear (half-pixel centres, no antialias) against
    ATen's CPU interpolate + the restated pad / normalise / mirror; 2e-6 covers the different summation order of the four
    corner products divided by std (fp32); equal sizes take the copy path and are bit-exact."""
    from centernet_amd.utils import post
    B, H, W, nh, nw, px, py, flip = cfg
    img = rng.t_uniform(9, f"img{cfg}", (B, 3, H, W))
    ref = post_ref.tta_prepare(post_ref.resize(img, nh, nw), MEAN, STD, px, py, flip)
    got = post.tta_prepare_scaled(img.to(DEV), nh, nw, MEAN, STD, px, py, flip).cpu()
    assert got.shape == ref.shape
    if (nh, nw) == (H, W):
        assert torch.equal(got, ref)
    else:
        assert torch.allclose(got, ref, rtol=0, atol=2e-6 / min(STD)), float((got - ref).abs().max())


def test_flip_merge_bit_exact():
    from centernet_amd.utils import post
    x = rng.t_normal(8, "maps", (6, 5, 12, 20))
    assert torch.equal(post.flip_merge(x.to(DEV)).cpu(), post_ref.flip_merge(x))


def _random_dets(seed, S, B, K, C, clustered):
    """decode-like outputs: scores descending per (scale, image); `clustered` puts boxes of a class on top of each other."""
    d = np.zeros((S, B, K, 6), np.float32)
    for s in range(S):
        for b in range(B):
            u = rng.uniform(seed, f"d{s}_{b}", (K, 6))
            cls = np.floor(u[:, 5] * C)
            if clustered:
                cx, cy = 20 + 90 * (cls % 5) / 5 + 3 * u[:, 0], 20 + 90 * (cls // 5 % 5) / 5 + 3 * u[:, 1]
            else:
                cx, cy = 128 * u[:, 0], 128 * u[:, 1]
            w, h = 4 + 30 * u[:, 2], 4 + 30 * u[:, 3]
            d[s, b, :, 0], d[s, b, :, 1], d[s, b, :, 2], d[s, b, :, 3] = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
            d[s, b, :, 4] = np.sort(u[:, 4])[::-1] * (0.2 if clustered and s == 1 else 1.0)
            d[s, b, :, 5] = cls
    return d


@pytest.mark.parametrize("S,C,clustered", [(1, 80, False), (3, 80, True), (2, 3, True), (5, 80, False)])
def test_ctdet_merge_matches_restatement(S, C, clustered):
    """boxes back to image coordinates, class grouping, multi-scale soft-NMS (gaussian, Nt 0.5), max-per-image cut."""
    from centernet_amd.utils import post
    B, K = 3, 100
    d = _random_dets(11 + S, S, B, K, C, clustered)
    scales = [1.0, 0.75, 1.25, 0.5, 1.5][:S]
    metas = [{"scale": [int(512 * sc) / 512, int(500 * sc) / 500], "padding": [post_ref.tta_pad(int(512 * sc), 31), post_ref.tta_pad(int(500 * sc), 31)]}
             for sc in scales]
    rows, counts = post.ctdet_merge([torch.from_numpy(d[s]).to(DEV) for s in range(S)], metas, C)
    got = post.results_by_class(rows, counts, C)
    for b in range(B):
        ref = post_ref.test_step_end([d[s, b] for s in range(S)], metas, C)
        assert set(got[b]) == set(ref)
        n_ref = sum(len(v) for v in ref.values())
        assert int(counts[b]) == n_ref
        for j in ref:
            assert got[b][j].shape == ref[j].shape, (b, j)
            assert np.array_equal(got[b][j][:, :4], ref[j][:, :4]), "same boxes in the same (soft-NMS selection) order"
            np.testing.assert_allclose(got[b][j][:, 4], ref[j][:, 4], rtol=2e-7, atol=0)     # exp() in double, stored fp32
    if S > 1 and clustered:
        raw = set(np.float32(v) for v in d[:, 0, :, 4].flatten())
        decayed = sum(int(np.float32(v) not in raw) for r in got[0].values() for v in r[:, 4])
        assert decayed > 10, "soft-NMS decayed the scores of overlapping boxes"


def test_test_step_flip_tta_end_to_end():
    """CenterNetDetection.test_step + test_step_end on a batch (flip TTA, 2 scales) against the restated pipeline run on the
    same network: prepared images, merged head maps, and the per-class results."""
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.utils import post
    m = CenterNetDetection("res_18", compute_dtype=torch.float32, test_flip=True, test_scales=[1, 0.75])
    rng.fill_state_dict(m, 98)
    m = m.to(DEV).eval()
    img = rng.t_uniform(98, "img", (2, 3, 128, 160)).to(DEV)
    ids, outs, metas = m.test_step((img, None), 3)
    assert ids == [6, 7] and len(outs) == 2
    assert metas[0] == {"scale": [1.0, 1.0], "padding": [16, 16]} and metas[1]["padding"] == [4, 16]
    # the same pipeline from the restatement's pieces (forward on the device network)
    for s_, scale in enumerate([1, 0.75]):
        nh, nw = int(128 * scale), int(160 * scale)
        x = post_ref.resize(img.cpu(), nh, nw)
        px, py = post_ref.tta_pad(nw, 31), post_ref.tta_pad(nh, 31)
        xin = post_ref.tta_prepare(x, MEAN, STD, px, py, True)
        if scale != 1:      # the resized image differs from ATen's CPU bilinear in the last bits (test_tta_prepare_scaled); the
            xdev = post.tta_prepare_scaled(img, nh, nw, MEAN, STD, px, py, True)      # network below amplifies that, so it is fed
            assert torch.allclose(xdev.cpu(), xin, rtol=0, atol=1e-5)                 # the device's own prepared input
            xin = xdev
        with torch.no_grad():
            o = m(xin.to(DEV))[-1]
        hm = post_ref.flip_merge(o["heatmap"].cpu())
        assert torch.allclose(outs[s_]["heatmap"].cpu(), hm, rtol=1e-4, atol=1e-5)
        assert torch.allclose(outs[s_]["regression"].cpu(), o["regression"][:2].cpu(), rtol=1e-4, atol=1e-5)
    dets = [m.decode({k: v.clone() for k, v in o.items()}).cpu().numpy() for o in outs]
    res = m.test_step_end((ids, outs, metas))
    assert [r[0] for r in res] == ids
    for b in range(2):
        ref = post_ref.test_step_end([d[b] for d in dets], metas, 80)
        for j in ref:
            assert res[b][1][j].shape == ref[j].shape
            np.testing.assert_allclose(res[b][1][j], ref[j], rtol=1e-5, atol=1e-5)
        assert sum(len(v) for v in res[b][1].values()) >= 100


FLIP_IDX = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]


@pytest.mark.parametrize("S", [1, 3])
def test_pose_merge_matches_restatement(S):
    """multi_pose rows [K,57]: boxes + keypoints to image coordinates, soft_nms_39 over the scales (columns 0-38 travel,
    39-56 stay: the reference's behaviour), top-20 cut."""
    from centernet_amd.utils import post
    B, K = 2, 100
    d = np.zeros((S, B, K, 57), np.float32)
    base = _random_dets(21 + S, S, B, K, 1, True)
    d[..., :5] = base[..., :5]
    d[..., 5:39] = rng.uniform(22, f"kps{S}", (S, B, K, 34)) * 128
    d[..., 40:] = rng.uniform(23, f"hs{S}", (S, B, K, 17))
    metas = [{"scale": [int(512 * sc) / 512, int(384 * sc) / 384], "padding": [post_ref.tta_pad(int(512 * sc), 31), post_ref.tta_pad(int(384 * sc), 31)]}
             for sc in [1.0, 0.75, 1.25][:S]]
    rows, counts = post.pose_merge([torch.from_numpy(d[s]).to(DEV) for s in range(S)], metas)
    for b in range(B):
        ref = post_ref.pose_test_step_end([d[s, b] for s in range(S)], metas)
        got = rows[b, :int(counts[b])].cpu().numpy()
        assert got.shape == ref.shape and ref.shape[0] >= 20
        assert np.array_equal(got[:, :4], ref[:, :4]) and np.array_equal(got[:, 5:], ref[:, 5:])
        np.testing.assert_allclose(got[:, 4], ref[:, 4], rtol=2e-7, atol=0)


def test_pose_test_step_flip_end_to_end():
    """CenterNetMultiPose.test_step (pose-aware flip merge) + test_step_end against the restated pipeline on the same network."""
    from centernet_amd.centernet_multi_pose import CenterNetMultiPose
    m = CenterNetMultiPose("res_18", compute_dtype=torch.float32, test_flip=True, test_scales=[1, 1.25])
    rng.fill_state_dict(m, 99)
    m = m.to(DEV).eval()
    img = rng.t_uniform(99, "img", (2, 3, 128, 128)).to(DEV)
    ids, outs, metas = m.test_step((img, None), 0)
    x = post_ref.tta_prepare(img.cpu(), MEAN, STD, 16, 16, True)
    with torch.no_grad():
        o = {k: v.cpu() for k, v in m(x.to(DEV))[-1].items()}
    ref = post_ref.flip_merge_pose(o, FLIP_IDX)
    for k in ref:
        assert torch.allclose(outs[0][k].cpu(), ref[k], rtol=1e-4, atol=1e-5), k
    dets = [m.decode({k: v.clone() for k, v in o_.items()}).cpu().numpy() for o_ in outs]
    res = m.test_step_end((ids, outs, metas))
    for b in range(2):
        r = post_ref.pose_test_step_end([d[b] for d in dets], metas)
        got = np.array(res[b][1], np.float32)
        assert res[b][0] == ids[b] and got.shape == r.shape
        np.testing.assert_allclose(got, r, rtol=1e-5, atol=1e-5)


IDENT = {"scale": [1.0, 1.0], "padding": [0, 0]}


@pytest.mark.parametrize("name,method,nt", [("gauss", 2, 0.5), ("linear", 1, 0.5), ("hard", 0, 0.3)])
def test_soft_nms_kernel_matches_reference_source_run(golden, name, method, nt):
    """The HIP soft-NMS held to tests/golden/soft_nms.npz directly (utils/nms.py:5-107 run from the reference's source, see
    oracle/gen_golden.py:gen_soft_nms): the golden boxes as two scales of one class with an identity transform."""
    from centernet_amd.utils import post
    g = golden("soft_nms.npz")
    b, ref, n = g[name + "_in"], g[name + "_out"], int(g[name + "_n"])
    K = len(b) // 2
    d = np.zeros((2, 1, K, 6), np.float32)
    d[0, 0, :, :5], d[1, 0, :, :5] = b[:K], b[K:]
    rows, counts = post.ctdet_merge([torch.from_numpy(d[s]).to(DEV) for s in range(2)], [IDENT, IDENT], 1, down_ratio=1,
                                    max_per_image=2 * K, nms_method=method, nms_nt=nt)
    got = rows[0, :int(counts[0])].cpu().numpy()
    assert int(counts[0]) == n and n < 2 * K
    assert np.array_equal(got[:, :4], ref[:n, :4].astype(np.float32)) and not got[:, 5].any()
    np.testing.assert_allclose(got[:, 4], ref[:n, 4], rtol=1e-6, atol=1e-7)


def test_soft_nms_39_kernel_matches_reference_source_run(golden):
    """utils/nms.py:109-206 (columns 5-38 travel with the box, 39-56 stay in place) against the same golden file."""
    from centernet_amd.utils import post
    g = golden("soft_nms.npz")
    b, ref, n = g["gauss39_in"], g["gauss39_out"], int(g["gauss39_n"])
    K = len(b) // 2
    d = np.stack([b[:K], b[K:]])[:, None]
    rows, counts = post.pose_merge([torch.from_numpy(d[s].copy()).to(DEV) for s in range(2)], [IDENT, IDENT], down_ratio=1,
                                   max_per_image=2 * K)
    got = rows[0, :int(counts[0])].cpu().numpy()
    assert int(counts[0]) == n and n < 2 * K
    assert np.array_equal(got[:, :4], ref[:n, :4].astype(np.float32)) and np.array_equal(got[:, 5:], ref[:n, 5:].astype(np.float32))
    np.testing.assert_allclose(got[:, 4], ref[:n, 4], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag,S", [("ms", 2), ("ss", 1)])
def test_test_step_end_matches_reference_source_run(golden, tag, S):
    """CenterNetDetection.test_step_end / CenterNetMultiPose.test_step_end (decode + merge kernels through the C ABI) against
    tests/golden/test_step_end.npz — the reference's own method bodies (centernet_detection.py:175-225,
    centernet_multi_pose.py:215-264) run from source on the same seeded two-scale head maps."""
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.centernet_multi_pose import CenterNetMultiPose
    g = golden("test_step_end.npz")
    m = CenterNetDetection("res_18", num_classes=3, test_scales=[1, 0.75][:S], compute_dtype=torch.float32)
    outs = [{k: v.to(DEV) for k, v in o.items()} for o in synth.tta_head_maps(int(g["det_seed"]), synth.DET_MAPS, synth.TTA_SIZES)[:S]]
    (iid, res), = m.test_step_end(([5], outs, synth.TTA_METAS[:S]))
    rows = np.concatenate([np.concatenate([res[j], np.full((len(res[j]), 1), j, np.float32)], 1) for j in sorted(res)])
    ref = g[f"det_{tag}_rows"]
    assert iid == 5 and rows.shape == ref.shape and np.array_equal(rows[:, 5], ref[:, 5])
    np.testing.assert_allclose(rows[:, :4], ref[:, :4], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(rows[:, 4], ref[:, 4], rtol=1e-6, atol=1e-7)
    m = CenterNetMultiPose("res_18", test_scales=[1, 0.75][:S], compute_dtype=torch.float32)
    outs = [{k: v.to(DEV) for k, v in o.items()} for o in synth.tta_head_maps(int(g["pose_seed"]), synth.POSE_MAPS, synth.TTA_SIZES)[:S]]
    (iid, res), = m.test_step_end(([5], outs, synth.TTA_METAS[:S]))
    rows, ref = np.asarray(res, np.float32), g[f"pose_{tag}_rows"]
    assert rows.shape == ref.shape
    np.testing.assert_allclose(np.delete(rows, 4, 1), np.delete(ref, 4, 1), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(rows[:, 4], ref[:, 4], rtol=1e-6, atol=1e-7)
