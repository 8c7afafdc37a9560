"""CPU: known-answer tests that pin the (unpinned-by-reference) DCNv2 restatement — SURVEY Appendix A."""
import torch
import torch.nn.functional as F

from centernet_amd import rng
from oracle.dcn_ref import DCN, dcn_v2_conv, bilinear_zero


def _xw(ci=6, co=5, h=9, w=11, b=2):
    return (rng.t_normal(5, "x", (b, ci, h, w)).double(), rng.t_normal(5, "w", (co, ci, 3, 3)).double(),
            rng.t_normal(5, "b", (co,)).double())


def test_zero_init_is_half_conv():
    x, w, b = _xw()
    m = DCN(6, 5).double()
    with torch.no_grad():
        m.weight.copy_(w); m.bias.copy_(b)
    torch.testing.assert_close(m(x), 0.5 * F.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1))


def test_unit_mask_zero_offset_is_conv():
    x, w, b = _xw()
    B, _, H, W = x.shape
    y = dcn_v2_conv(x, torch.zeros(B, 18, H, W).double(), torch.ones(B, 9, H, W).double(), w, b)
    torch.testing.assert_close(y, F.conv2d(x, w, b, 1, 1))


def test_integer_offsets_shift_taps():
    x, w, b = _xw()
    B, _, H, W = x.shape
    off = torch.zeros(B, 18, H, W).double()
    off[:, 0::2] = 1.0    # every tap samples one row lower
    off[:, 1::2] = -2.0   # and two columns to the left
    y = dcn_v2_conv(x, off, torch.ones(B, 9, H, W).double(), w, b)
    # expected[h,w] = sum_ij W[i,j] * X0(h+i, w+j-3), X0 = zero-extended x
    full = F.conv2d(F.pad(x, (3, 3, 3, 3)), w, b)       # full[h',w'] = sum W[i,j] X0(h'+i-3, w'+j-3)
    torch.testing.assert_close(y, full[:, :, 3:3 + H, 0:W])


def test_fractional_offsets_on_linear_ramp():
    H, W = 8, 10
    yy, xx = torch.meshgrid(torch.arange(H).double(), torch.arange(W).double(), indexing="ij")
    x = (2.0 * yy + 3.0 * xx + 1.0).view(1, 1, H, W)
    py = torch.full((1, H - 2, W - 2), 0.0).double() + yy[1:-1, 1:-1] + 0.25
    px = xx[1:-1, 1:-1].unsqueeze(0) - 0.5
    torch.testing.assert_close(bilinear_zero(x, py, px)[0, 0], 2.0 * py[0] + 3.0 * px[0] + 1.0)
    # fully outside -> 0 ; half outside -> only in-bounds corners contribute
    assert bilinear_zero(x, torch.tensor([[[-1.0]]]).double(), torch.tensor([[[3.0]]]).double()).item() == 0.0
    v = bilinear_zero(x, torch.tensor([[[-0.5]]]).double(), torch.tensor([[[3.0]]]).double()).item()
    assert abs(v - 0.5 * x[0, 0, 0, 3].item()) < 1e-12


def test_gradcheck_fp64():
    x = rng.t_normal(6, "x", (1, 2, 5, 5)).double().requires_grad_(True)
    off = (rng.t_normal(6, "o", (1, 18, 5, 5)).double() * 0.7 + 0.13).requires_grad_(True)   # keep away from integers
    msk = torch.sigmoid(rng.t_normal(6, "m", (1, 9, 5, 5)).double()).requires_grad_(True)
    w = rng.t_normal(6, "w", (3, 2, 3, 3)).double().requires_grad_(True)
    b = rng.t_normal(6, "b", (3,)).double().requires_grad_(True)
    assert torch.autograd.gradcheck(dcn_v2_conv, (x, off, msk, w, b), eps=1e-6, atol=1e-5)


def test_border_known_answers_hand_computed():
    """1x1x4x4 image, fractional sample points at all four borders and corners, py = -1 / py = H / px = -1 / px = W exactly
    (the `<= -1 or >= H` rule of SURVEY Appendix A): values computed by hand in tests/conftest.py:dcn_border_vectors."""
    from conftest import dcn_border_vectors
    x, off, mask, weight, want = dcn_border_vectors()
    y = dcn_v2_conv(x, off, mask, weight, None)
    torch.testing.assert_close(y[0, 0], want, rtol=0, atol=1e-12)
    y32 = dcn_v2_conv(x.float(), off.float(), mask.float(), weight.float(), None)
    torch.testing.assert_close(y32[0, 0].double(), want, rtol=0, atol=1e-6)
    # modulation: the mask scales the sample, the bias is added after it
    y2 = dcn_v2_conv(x, off, 0.25 * mask, weight, torch.tensor([2.0]).double())
    torch.testing.assert_close(y2[0, 0], 0.25 * want + 2.0, rtol=0, atol=1e-12)
    # the other taps see their own (zero-offset) sample points: moving the weight to tap 0 gives the plain shifted image
    w0 = torch.zeros_like(weight); w0[0, 0, 0, 0] = 1.0
    y3 = dcn_v2_conv(x, off, mask, w0, None)
    torch.testing.assert_close(y3[0, 0], F.pad(x, (1, 1, 1, 1))[0, 0, 0:4, 0:4])
