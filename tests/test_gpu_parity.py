"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference goldens.
Everything here needs a real MI355X:  python -m pytest tests -m gpu"""
import hashlib

import os

import numpy as np
import pytest
import torch

from conftest import golden, golden_calib, small_dhds_cfg
from dhd_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def grid_cfgs(cfg):
    from dhd_amd.lss_heightmap import _FULL_GRID
    return [dict(_FULL_GRID), cfg['mask_1_grid'], cfg['mask_2_grid'], cfg['mask_3_grid']]


def make_plan(cfg, batch, n_cams, channels=None, n_grids=4):
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    fh, fw = cfg['input_size'][0] // cfg['downsample'], cfg['input_size'][1] // cfg['downsample']
    grids = [mghs_op.grid_from_cfg(g) for g in grid_cfgs(cfg)[:n_grids]]
    plan = mghs_op.Plan(batch, n_cams, len(axes[2]), fh, fw, channels or cfg['out_channels'], grids)
    return plan, axes


def device_calib(calib, axes, dev, inv_post_rot=None, combine=None):
    from dhd_amd import mghs_op
    s2e, _, intrin, post_rot, post_tran, bda = [T(a, dev) for a in calib]
    return mghs_op.make_calib(s2e, intrin, post_rot, post_tran, bda, tuple(T(a, dev) for a in axes),
                              None if inv_post_rot is None else T(inv_post_rot, dev),
                              None if combine is None else T(combine, dev))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------- operator seam

def test_reference_known_answer_test(gpu):
    """ops/bev_pool_v2/bev_pool.py:163-194, verbatim values, through the reference's signature."""
    from dhd_amd import bev_pool_v2
    depth = torch.tensor([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], device=gpu).view(1, 1, 2, 2, 2).requires_grad_()
    feat = torch.ones(1, 1, 2, 2, 2, device=gpu).requires_grad_()
    rd = torch.tensor([0, 4, 1, 6], device=gpu).int()
    rf = torch.tensor([0, 0, 1, 2], device=gpu).int()
    rb = torch.tensor([0, 0, 1, 1], device=gpu).int()
    st = torch.tensor([0, 2], device=gpu).int()
    ln = torch.tensor([2, 2], device=gpu).int()
    out = bev_pool_v2(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    assert out.shape == (1, 2, 1, 2, 2)
    loss = out.sum()
    loss.backward()
    assert abs(loss.item() - 4.4) < 1e-6
    assert torch.allclose(depth.grad.flatten().cpu(), torch.tensor([2., 2., 0., 0., 2., 0., 2., 0.]))
    assert torch.allclose(feat.grad.flatten().cpu(), torch.tensor([1., 1., .4, .4, .8, .8, 0., 0.]))


@pytest.mark.parametrize('channels', [1, 8, 24, 64, 80, 128])
def test_bev_pool_v2_operator_vs_oracle(gpu, channels):
    """Operator seam with the golden's canonical index lists (full grid of g2b), any channel count."""
    from dhd_amd import bev_pool_v2
    from oracle import mghs_oracle as O
    g = golden('g2b_small_dhds')
    rb, rd, rf, st, ln = (g[n + '0'] for n in ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths'))
    B, N, D, fh, fw = 2, 2, 44, 4, 11
    depth = g['depth'].reshape(B, N, D, fh, fw)
    feat = syn.hash_signed(900 + channels, (B, N, fh, fw, channels))
    shape = (B, 1, 200, 200, channels)
    ref = O.bev_pool_v2(depth, feat, rd, rf, rb, shape, st, ln)
    dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    out = bev_pool_v2(dt, ft, T(rd, gpu), T(rf, gpu), T(rb, gpu), shape, T(st, gpu), T(ln, gpu))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-6, rtol=1e-6)
    w = syn.hash_signed(901, ref.shape)
    (out * T(w, gpu)).sum().backward()
    og = np.ascontiguousarray(w.transpose(0, 2, 3, 4, 1))  # (B,Dz,Dy,Dx,C)
    dg, fg = O.bev_pool_v2_backward(og, depth, feat, rd, rf, rb)
    np.testing.assert_allclose(dt.grad.cpu().numpy(), dg, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), fg, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('channels', [4, 8, 64, 128, 256, 20])
def test_bev_pool_v2_operator_ragged_intervals(gpu, channels):
    """Interval lengths around every branch of the sub-wave kernels (C = 4 L: a group's own walk covers 4 L points, longer
    intervals are taken by the whole wave in 64-point batches, 256 per chunk): 1, L, 4L, 4L+1, 63..65, 255..257, 1000, in
    shuffled order, points in scattered pixels; forward and both gradients against the oracle."""
    from dhd_amd import bev_pool_v2
    from oracle import mghs_oracle as O
    rng = np.random.default_rng(channels)
    lens = [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000] + [1] * 40 + [7] * 40
    lens = np.array(lens, dtype=np.int32)[rng.permutation(len(lens) + 0)]
    B, N, D, fh, fw = 1, 2, 44, 8, 22
    n_pts = int(lens.sum())
    assert n_pts <= B * N * D * fh * fw
    rd = rng.permutation(B * N * D * fh * fw)[:n_pts].astype(np.int32)          # distinct points
    rf = (rd // (D * fh * fw)) * (fh * fw) + rd % (fh * fw)                      # the point's pixel (b, n, h, w)
    st = (np.cumsum(lens) - lens).astype(np.int32)
    vox = rng.permutation(40 * 40)[:len(lens)].astype(np.int32)
    rb = np.repeat(vox, lens).astype(np.int32)
    depth = syn.hash_signed(70 + channels, (B, N, D, fh, fw))
    feat = syn.hash_signed(71 + channels, (B, N, fh, fw, channels))
    shape = (B, 1, 40, 40, channels)
    ref = O.bev_pool_v2(depth, feat, rd, rf.astype(np.int32), rb, shape, st, lens)
    dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    out = bev_pool_v2(dt, ft, T(rd, gpu), T(rf.astype(np.int32), gpu), T(rb, gpu), shape, T(st, gpu), T(lens, gpu))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=3e-5, rtol=1e-5)
    w = syn.hash_signed(72, ref.shape)
    (out * T(w, gpu)).sum().backward()
    og = np.ascontiguousarray(w.transpose(0, 2, 3, 4, 1))
    dg, fg = O.bev_pool_v2_backward(og, depth, feat, rd, rf.astype(np.int32), rb)
    np.testing.assert_allclose(dt.grad.cpu().numpy(), dg, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), fg, atol=3e-5, rtol=1e-5)


def _ragged_lists(seed, B, N, D, fh, fw, nz, ny, nx, extra_empty=False):
    rng = np.random.default_rng(seed)
    lens = [1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000] + [1] * 40 + [7] * 40
    if extra_empty:
        lens = lens + [0] * 5
    lens = np.array(lens, dtype=np.int32)[rng.permutation(len(lens))]
    n_pts = int(lens.sum())
    assert n_pts <= B * N * D * fh * fw
    rd = rng.permutation(B * N * D * fh * fw)[:n_pts].astype(np.int32)
    rf = ((rd // (D * fh * fw)) * (fh * fw) + rd % (fh * fw)).astype(np.int32)
    st = np.minimum(np.cumsum(lens) - lens, max(n_pts - 1, 0)).astype(np.int32)
    vox = rng.permutation(B * nz * ny * nx)[:len(lens)].astype(np.int32)        # shuffled voxel order: not sorted by ranks_bev
    rb = np.repeat(vox, lens).astype(np.int32)
    return rd, rf, rb, st, lens


@pytest.mark.parametrize('case', ['g2b', 'ragged', 'ragged_nz3_empty'])
def test_bev_pool_v2_fused_equals_the_three_step_operator_bit_for_bit(gpu, case):
    """bev_pool_v2(..., fused=True): the (B, C, Dz, Dy, Dx) tensor written once by the segment writer and
    its gradient read once in that layout, against zero-fill + kernel + permute copy (bev_pool.py:27,86-106) -- the same sums
    in the same order, so every value and both gradients must be IDENTICAL; and against the oracle."""
    from dhd_amd import bev_pool_v2
    from dhd_amd.bev_pool_v2 import fused_supported
    from oracle import mghs_oracle as O
    C = 64
    if case == 'g2b':
        g = golden('g2b_small_dhds')
        rb, rd, rf, st, ln = (g[n + '0'] for n in ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths'))
        B, N, D, fh, fw = 2, 2, 44, 4, 11
        depth = g['depth'].reshape(B, N, D, fh, fw)
        shape = (B, 1, 200, 200, C)
    else:
        B, N, D, fh, fw = 2, 2, 44, 8, 22
        nz = 3 if case.endswith('empty') else 1
        shape = (B, nz, 40, 48, C)
        rd, rf, rb, st, ln = _ragged_lists(5, B, N, D, fh, fw, nz, 40, 48, extra_empty=case.endswith('empty'))
        depth = syn.hash_signed(170, (B, N, D, fh, fw))
    assert fused_supported(shape)
    feat = syn.hash_signed(171, (B, N, fh, fw, C))
    idx = [T(a, gpu) for a in (rd, rf, rb)]
    w = T(syn.hash_signed(172, (shape[0], C) + tuple(shape[1:4])), gpu)
    res = []
    for fused in (False, True):
        dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
        out = bev_pool_v2(dt, ft, *idx, shape, T(st, gpu), T(ln, gpu), fused=fused)
        assert out.shape == (shape[0], C) + tuple(shape[1:4]) and out.is_contiguous()
        (out * w).sum().backward()
        res.append((out.detach(), dt.grad, ft.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # the voxel -> row map is kept while the same index tensors come back (static rig) and rebuilt when one was written to
    import importlib
    bm = importlib.import_module('dhd_amd.bev_pool_v2')
    stc, lnc = T(st, gpu), T(ln, gpu)
    dt, ft = T(depth, gpu), T(feat, gpu)
    first = bev_pool_v2(dt, ft, *idx, shape, stc, lnc, fused=True)
    kept = bm._state_cache[first.device.index][2]
    again = bev_pool_v2(dt, ft, *idx, shape, stc, lnc, fused=True)
    assert bm._state_cache[first.device.index][2] is kept and torch.equal(first, again) and torch.equal(first, res[1][0])
    idx[2].add_(0)                                            # in-place write: version bump, same values
    third = bev_pool_v2(dt, ft, *idx, shape, stc, lnc, fused=True)
    assert bm._state_cache[first.device.index][2] is not kept and torch.equal(third, first)
    keep = ln > 0
    ref = O.bev_pool_v2(depth, feat, rd, rf, rb, shape, st[keep], ln[keep])
    np.testing.assert_allclose(res[1][0].cpu().numpy(), ref, atol=3e-5, rtol=1e-5)
    og = np.ascontiguousarray(w.cpu().numpy().transpose(0, 2, 3, 4, 1))
    dg, fg = O.bev_pool_v2_backward(og, depth, feat, rd, rf, rb)
    np.testing.assert_allclose(res[1][1].cpu().numpy(), dg, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(res[1][2].cpu().numpy(), fg, atol=3e-5, rtol=1e-5)


def test_bev_pool_v2_fused_edge_cases(gpu):
    """Shapes the fused entry points do not take fall back to the three steps (the reference's KAT has C = 2); no interval at
    all gives zeros; voxels outside the grid are dropped (the reference kernel would write out of bounds: bev_pool_cuda.cu:47)."""
    from dhd_amd import bev_pool_v2
    from dhd_amd.bev_pool_v2 import fused_supported
    depth = torch.tensor([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], device=gpu).view(1, 1, 2, 2, 2).requires_grad_()
    feat = torch.ones(1, 1, 2, 2, 2, device=gpu).requires_grad_()
    I = lambda v: torch.tensor(v, device=gpu).int()
    assert not fused_supported((1, 1, 2, 2, 2))
    out = bev_pool_v2(depth, feat, I([0, 4, 1, 6]), I([0, 0, 1, 2]), I([0, 0, 1, 1]), (1, 1, 2, 2, 2), I([0, 2]), I([2, 2]), fused=True)
    out.sum().backward()
    assert abs(out.sum().item() - 4.4) < 1e-6
    assert torch.allclose(depth.grad.flatten().cpu(), torch.tensor([2., 2., 0., 0., 2., 0., 2., 0.]))
    # C = 64, 4 x 8 grid: two real intervals + one whose voxel is outside the grid
    B, N, D, fh, fw, C = 1, 1, 2, 2, 2, 64
    d2 = torch.rand(B, N, D, fh, fw, device=gpu).requires_grad_()
    f2 = torch.randn(B, N, fh, fw, C, device=gpu).requires_grad_()
    shape = (1, 1, 4, 8, C)
    rd, rf, rb = I([0, 4, 1, 6, 3]), I([0, 0, 1, 2, 3]), I([5, 5, 31, 31, 32])
    out = bev_pool_v2(d2, f2, rd, rf, rb, shape, I([0, 2, 4]), I([2, 2, 1]), fused=True)
    want = torch.zeros(1, 4 * 8, C, device=gpu)
    df, ff = d2.detach().flatten(), f2.detach().view(-1, C)
    want[0, 5] = df[0] * ff[0] + df[4] * ff[0]
    want[0, 31] = df[1] * ff[1] + df[6] * ff[2]
    assert torch.allclose(out.view(1, C, 32).transpose(1, 2), want, atol=1e-6)
    out.sum().backward()
    assert d2.grad.flatten()[3].item() == 0 and float(f2.grad.view(-1, C)[3].abs().max()) == 0   # the dropped point
    assert torch.allclose(d2.grad.flatten()[0], ff[0].sum(), atol=1e-5)
    # no interval at all
    e = torch.empty(0, device=gpu).int()
    z = bev_pool_v2(d2.detach(), f2.detach(), e, e, e, shape, e, e, fused=True)
    assert z.shape == (1, C, 1, 4, 8) and float(z.abs().max()) == 0
    # cache=False: the voxel -> row map is rebuilt per call and nothing is pinned; an index list rewritten behind torch's version
    # counter (the documented hole of the cached form) is then seen
    import sys
    bp = sys.modules['dhd_amd.bev_pool_v2']      # (the package re-exports the function under the module's name)
    bp.clear_caches()
    iv_s, iv_l = I([0, 2, 4]), I([2, 2, 1])
    a = bev_pool_v2(d2.detach(), f2.detach(), rd, rf, rb, shape, iv_s, iv_l, fused=True, cache=False)
    assert not bp._state_cache and torch.equal(a, out.detach())
    rb.data.copy_(I([6, 6, 30, 30, 32]))          # no version bump
    b = bev_pool_v2(d2.detach(), f2.detach(), rd, rf, rb, shape, iv_s, iv_l, fused=True, cache=False)
    assert torch.allclose(b.view(1, C, 32)[0, :, 6], want[0, 5], atol=1e-6) and float(b.view(1, C, 32)[0, :, 5].abs().max()) == 0
    # the gate is the library's: more voxels than the fused entry points index fall back (here only asked, not run)
    assert not fused_supported((2, 8192, 512, 256, 64)) and fused_supported((1, 8192, 512, 256, 64))


def test_bev_pool_v2_regroup_vs_argsort(gpu):
    """dhd_bev_pool_v2_regroup (device counting sort by feature pixel, ascending ranks_depth inside a pixel, one interval per
    pixel incl. empty ones) against the reference's formulation (bev_pool.py:47-57: argsort by ranks_feat, run-length scan) on
    ragged lists: many empty pixels, one pixel holding most points, out-of-range pixel ids (dropped), and the empty list."""
    import ctypes as C
    from dhd_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for n_points, n_pixels in ((5000, 700), (3, 64), (0, 10), (40000, 9000)):
        rf = torch.randint(0, n_pixels, (n_points,), generator=g)
        if n_points > 100:
            rf[::3] = 17                     # a heavy pixel
            rf[5] = n_pixels + 4             # out of range: dropped
            rf[6] = -2
        rd = torch.randperm(max(n_points, 1), generator=g)[:n_points]
        rb = torch.randint(0, 1000, (n_points,), generator=g)
        d = lambda t: t.int().to(gpu).contiguous()
        rfd, rdd, rbd = d(rf), d(rd), d(rb)
        out = [torch.full((max(n_points, 1),), -7, dtype=torch.int32, device=gpu) for _ in range(3)]
        starts = torch.empty(n_pixels, dtype=torch.int32, device=gpu)
        lengths = torch.empty(n_pixels, dtype=torch.int32, device=gpu)
        nb = int(lib.dhd_bev_pool_v2_regroup_scratch_bytes(n_points, n_pixels))
        scratch = torch.empty(nb, dtype=torch.uint8, device=gpu)
        assert lib.dhd_bev_pool_v2_regroup(_lib.ptr(rdd), _lib.ptr(rfd), _lib.ptr(rbd), n_points, n_pixels, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                           _lib.ptr(out[2]), _lib.ptr(starts), _lib.ptr(lengths), _lib.ptr(scratch), nb - 1, None) == -2
        _lib.check(lib.dhd_bev_pool_v2_regroup(_lib.ptr(rdd), _lib.ptr(rfd), _lib.ptr(rbd), n_points, n_pixels, _lib.ptr(out[0]),
                                               _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(starts), _lib.ptr(lengths), _lib.ptr(scratch), nb,
                                               _lib.stream_ptr(gpu)), 'regroup')
        keep = (rf >= 0) & (rf < n_pixels)
        rf_k, rd_k, rb_k = rf[keep], rd[keep], rb[keep]
        order = torch.argsort(rf_k * (max(n_points, 1) + 1) + rd_k)          # by pixel, then by ranks_depth
        n_kept = int(keep.sum())
        assert torch.equal(out[1][:n_kept].cpu().long(), rf_k[order]) and torch.equal(out[0][:n_kept].cpu().long(), rd_k[order])
        assert torch.equal(out[2][:n_kept].cpu().long(), rb_k[order])
        cnt = torch.bincount(rf_k, minlength=n_pixels)
        assert torch.equal(lengths.cpu().long(), cnt) and torch.equal(starts.cpu().long(), torch.cumsum(cnt, 0) - cnt)


def test_bev_pool_v2_empty_and_cpu_tensor_errors(gpu):
    from dhd_amd import bev_pool_v2, _lib
    z = torch.zeros(0, dtype=torch.int32, device=gpu)
    depth = torch.rand(1, 1, 2, 2, 2, device=gpu)
    feat = torch.rand(1, 1, 2, 2, 4, device=gpu)
    out = bev_pool_v2(depth, feat, z, z, z, (1, 1, 3, 3, 4), z, z)
    assert out.shape == (1, 4, 1, 3, 3) and float(out.abs().sum()) == 0.0
    with pytest.raises(_lib.DhdError):
        bev_pool_v2(depth.cpu(), feat.cpu(), z.cpu(), z.cpu(), z.cpu(), (1, 1, 3, 3, 4), z.cpu(), z.cpu())


# --------------------------------------------------------------------------- geometry / indices

@pytest.mark.parametrize('name', ['g2_smoke', 'g2b_small_dhds', 'g2c_no_band', 'g2d_out_of_grid'])
def test_voxel_index_bit_exact_vs_reference(gpu, name):
    """Given the reference's own small matrices, ego coordinates and every grid's point->voxel map
    are bit-identical to what the reference's Python computed (lss_heightmap.py:206-230,331-354)."""
    from dhd_amd import mghs_op
    g = golden(name)
    cfg = syn.smoke_config() if name == 'g2_smoke' else small_dhds_cfg()
    B, N = g['sensor2ego'].shape[:2]
    plan, axes = make_plan(cfg, B, N)
    calib, keep = device_calib(golden_calib(g), axes, gpu, g['ref_inv_post_rot'], g['ref_combine'])
    for k in range(4):
        rank, ego = mghs_op.voxel_index(plan, calib, k, want_ego=True)
        assert np.array_equal(ego.cpu().numpy(), g['coor']), k
        assert np.array_equal(rank.cpu().numpy(), g[f'rank_map{k}']), k


@pytest.mark.parametrize('name', ['g2_smoke', 'g2b_small_dhds'])
def test_voxel_index_raw_calibration_vs_oracle(gpu, name):
    """From raw calibration the device's own 3x3 inverse is the oracle's algorithm: bit-exact too."""
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    g = golden(name)
    cfg = syn.smoke_config() if name == 'g2_smoke' else small_dhds_cfg()
    calib_np = golden_calib(g)
    B, N = calib_np[0].shape[:2]
    plan, axes = make_plan(cfg, B, N)
    calib, keep = device_calib(calib_np, axes, gpu)
    coor = O.ego_coor(axes, calib_np[0], calib_np[2], calib_np[3], calib_np[4], calib_np[5])
    grids = [O.FULL_GRID] + [{a: cfg[k][a] for a in 'xyz'} for k in ('mask_1_grid', 'mask_2_grid', 'mask_3_grid')]
    for k in range(4):
        rank, ego = mghs_op.voxel_index(plan, calib, k, want_ego=True)
        assert np.array_equal(ego.cpu().numpy(), coor), k
        assert np.array_equal(rank.cpu().numpy(), O.voxel_rank(coor, grids[k])), k


def test_height_band_vs_oracle_with_ties(gpu):
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    cfg = syn.dhd_s_config()
    idx = syn.height_index(5, (12, 16, 44), 65)
    idx[0, 0, :5] = [0, 15, 16, 63, 64]
    h = syn.height_probs_from_index(idx, 65)
    h[1, 10, 3, 3] = 0.5  # tie with the hashed maximum: first index must win
    h[1, 50, 3, 3] = 0.5
    h[1, idx[1, 3, 3], 3, 3] = 0.25
    band = mghs_op.height_band(T(h, gpu), cfg['height_range'], cfg['mask_range']).cpu().numpy()
    ref_idx = h.argmax(1)
    assert ref_idx[1, 3, 3] == 10
    assert np.array_equal(band, O.band_index(ref_idx, cfg['height_range'], cfg['mask_range']))
    assert list(band[0, 0, :5]) == [0, 0, 1, 2, 255]


def test_feature_relayout_round_trip(gpu):
    from dhd_amd import mghs_op
    x = T(syn.hash_signed(3, (5, 70, 7, 13)), gpu)
    y = mghs_op._nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(mghs_op._nhwc_to_nchw(y), x)


# --------------------------------------------------------------------------- fused view transform

def run_fused(gpu, cfg, calib_np, depth, feat, hidx, inv=None, comb=None, weights_seed=None, separate=False):
    from dhd_amd import mghs_op
    B, N = calib_np[0].shape[:2]
    plan, axes = make_plan(cfg, B, N, channels=feat.shape[1])
    calib, keep = device_calib(calib_np, axes, gpu, inv, comb)
    height = T(syn.height_probs_from_index(hidx, len(cfg['height_range'])), gpu)
    dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    ws = plan.new_workspace(gpu)
    if separate:   # the stand-alone entry points: dhd_height_band, dhd_feat_nchw_to_nhwc, dhd_mghs_prepare
        band = mghs_op.height_band(height, cfg['height_range'], cfg['mask_range'])
        outs = mghs_op.mghs_pool(plan, calib, band, dt, ft, ws)
    else:          # dhd_mghs_lift: the same three steps as four launches
        outs = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], dt, ft, ws)
    grads = None
    if weights_seed is not None:
        loss = sum((o * T(syn.hash_signed(weights_seed + k, tuple(o.shape)), gpu)).sum() for k, o in enumerate(outs))
        loss.backward()
        grads = (dt.grad.cpu().numpy(), ft.grad.cpu().numpy())
    return [o.detach().cpu().numpy() for o in outs], grads, (plan, ws)


@pytest.mark.parametrize('name', ['g2_smoke', 'g2b_small_dhds', 'g2c_no_band', 'g2d_out_of_grid'])
def test_view_transform_small_vs_reference(gpu, name):
    """The 4 outputs of MGHS.view_transform and the gradients of a weighted sum, against the
    reference's own results (ragged cases: a band with no pixel, a rig outside every grid)."""
    g = golden(name)
    cfg = syn.smoke_config() if name == 'g2_smoke' else small_dhds_cfg()
    outs, grads, _ = run_fused(gpu, cfg, golden_calib(g), g['depth'], g['tran_feat'], g['height_idx'],
                               g['ref_inv_post_rot'], g['ref_combine'], int(g['seed_w']))
    for k, o in enumerate(outs):
        assert o.shape == g[f'out{k}'].shape
        np.testing.assert_allclose(o, g[f'out{k}'], atol=1e-5, rtol=1e-5)
        assert np.array_equal(o != 0, g[f'out{k}'] != 0) or np.abs(o - g[f'out{k}']).max() < 1e-6
    np.testing.assert_allclose(grads[0], g['depth_grad'], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(grads[1], g['feat_grad'], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('channels', [3, 24, 64, 96, 128])
def test_view_transform_channel_counts_vs_oracle(gpu, channels):
    from oracle import mghs_oracle as O
    cfg = small_dhds_cfg()
    calib_np = syn.make_calibration(70 + channels, 1, 3, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(80 + channels, 1, 3, 44, 4, 11, channels, 65)
    outs, grads, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=500)
    ref = O.view_transform(cfg, calib_np, depth, feat, hidx)
    for o, r in zip(outs, ref):
        np.testing.assert_allclose(o, r, atol=1e-5, rtol=1e-5)
    ws = [syn.hash_signed(500 + k, r.shape) for k, r in enumerate(ref)]
    dg, fg = O.view_transform_backward(cfg, calib_np, depth, feat, hidx, ws)
    np.testing.assert_allclose(grads[0], dg, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(grads[1], fg, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('channels', [24, 64])
def test_backward_after_another_prepare_on_the_shared_scratch(gpu, channels):
    """fwd(plan A), fwd(plan B), bwd(plan A) with both plans on the stream's shared scratch: the backward of A may only
    depend on A's STATE (include/dhd_amd.h: the scratch is valid from a prepare to the forward after it).  channels = 24
    takes the generic (non-compact) path, whose backward walks the grouped entry lists again; 64 the compact one."""
    from oracle import mghs_oracle as O
    from dhd_amd import mghs_op
    cfg = small_dhds_cfg()
    runs = []
    for k in range(2):   # A, then B with another calibration and other inputs (same sizes: the same shared scratch)
        calib_np = syn.make_calibration(170 + k, 1, 3, cfg['input_size'])
        depth, feat, hidx = syn.lift_inputs(180 + k, 1, 3, 44, 4, 11, channels, 65)
        plan, axes = make_plan(cfg, 1, 3, channels=channels)
        calib, keep = device_calib(calib_np, axes, gpu)
        height = T(syn.height_probs_from_index(hidx, len(cfg['height_range'])), gpu)
        dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
        outs = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], dt, ft)   # shared scratch
        runs.append((calib_np, depth, feat, hidx, dt, ft, outs, keep))
    assert runs[0][6][0].grad_fn is not None
    for k in (0, 1):     # A's backward comes after B's prepare + forward
        calib_np, depth, feat, hidx, dt, ft, outs, _ = runs[k]
        ws = [syn.hash_signed(700 + 10 * k + i, tuple(o.shape)) for i, o in enumerate(outs)]
        sum((o * T(w, gpu)).sum() for o, w in zip(outs, ws)).backward()
        dg, fg = O.view_transform_backward(cfg, calib_np, depth, feat, hidx, ws)
        np.testing.assert_allclose(dt.grad.cpu().numpy(), dg, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(ft.grad.cpu().numpy(), fg, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('name,input_size', [('DHD-M', (256, 704)), ('DHD-L', (512, 1408))])
def test_dhd_m_and_l_geometry_vs_oracle(gpu, name, input_size):
    """BASELINE configs 4/5 geometry (DHD-M.py / DHD-L.py: depth bins 0.5 m -> D = 88; L: 512x1408 images ->
    32x88 feature maps, 1.49 M frustum points per sample): voxel indices bit-exact, outputs and gradients
    against the oracle."""
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    cfg = syn.dhd_s_config()
    cfg['input_size'] = input_size
    cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])
    fh, fw = input_size[0] // 16, input_size[1] // 16
    calib_np = syn.make_calibration(41, 1, 6, input_size)
    depth, feat, hidx = syn.lift_inputs(42, 1, 6, 88, fh, fw, 64, 65)
    plan, axes = make_plan(cfg, 1, 6)
    assert plan.n_depth == 88 if hasattr(plan, 'n_depth') else True
    calib, keep = device_calib(calib_np, axes, gpu)
    coor = O.ego_coor(axes, calib_np[0], calib_np[2], calib_np[3], calib_np[4], calib_np[5])
    for k, gcfg in enumerate(grid_cfgs(cfg)):
        rank, _ = mghs_op.voxel_index(plan, calib, k)
        assert np.array_equal(rank.cpu().numpy(), O.voxel_rank(coor, {a: gcfg[a] for a in 'xyz'})), k
    outs, grads, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=600)
    ref = O.view_transform(cfg, calib_np, depth, feat, hidx)
    for o, r in zip(outs, ref):
        np.testing.assert_allclose(o, r, atol=5e-5, rtol=1e-5)
    ws = [syn.hash_signed(600 + k, r.shape) for k, r in enumerate(ref)]
    dg, fg = O.view_transform_backward(cfg, calib_np, depth, feat, hidx, ws)
    np.testing.assert_allclose(grads[0], dg, atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(grads[1], fg, atol=5e-4, rtol=1e-5)


def assert_product_keys_equal_maps(gpu, cfg, plan, ws, maps, hidx):
    """The keys the PRODUCT's counting kernel wrote in the last prepare on `ws` (dhd_mghs_debug_keys) against per-grid point ->
    voxel maps that have been verified against the reference: grid 0 for every point, the band grid of the point's pixel for
    the second key (-1 where the pixel has no band or the point falls outside)."""
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    d = plan.desc
    keys = mghs_op.debug_keys(plan, ws)
    P = keys.shape[1]
    base = np.cumsum([0] + [d.batch * g_.n[0] * g_.n[1] * g_.n[2] for g_ in plan.grids])
    m0 = maps[0].to(torch.int64)
    want0 = torch.where(m0 >= 0, m0 + int(base[0]), m0)
    assert torch.equal(keys[0].to(torch.int64), want0)
    band = torch.from_numpy(O.band_index(hidx, cfg['height_range'], cfg['mask_range']).astype(np.int64)).to(gpu)   # (B*N, fH, fW): 0/1/2, 255 = none
    band_pt = band[:, None].expand(-1, d.n_depth, -1, -1).reshape(-1)
    assert band_pt.numel() == P
    want1 = torch.full((P,), -1, dtype=torch.int64, device=gpu)
    for b in range(3):
        mb = maps[b + 1].to(torch.int64)
        sel = (band_pt == b) & (mb >= 0)
        want1[sel] = mb[sel] + int(base[b + 1])
    assert torch.equal(keys[1].to(torch.int64), want1)
    assert int((keys[1] >= 0).sum()) > 0


@pytest.mark.parametrize('batch', [1, 2, 4])
def test_full_dhds_size_vs_reference(gpu, batch):
    """DHD-S, 6 cameras, 200x200x{1,4,4,8}: index hashes, sampled voxels, sums and gradients of the
    reference run (fixtures hold hashes/samples; inputs are regenerated from integer hashes)."""
    from dhd_amd import mghs_op
    g = golden(f'g3_dhds_b{batch}')
    cfg = syn.dhd_s_config()
    calib_np = golden_calib(g)
    _, s_in, s_w = (int(v) for v in g['seeds'])
    depth, feat, hidx = syn.lift_inputs(s_in, batch, 6, 44, 16, 44, 64, 65)
    plan, axes = make_plan(cfg, batch, 6)
    calib, keep = device_calib(calib_np, axes, gpu, g['ref_inv_post_rot'], g['ref_combine'])
    for k in range(4):
        rank, ego = mghs_op.voxel_index(plan, calib, k, want_ego=(k == 0))
        if k == 0:
            assert sha(ego.cpu().numpy()) == str(g['coor_sha'])
        assert sha(rank.cpu().numpy()) == str(g[f'rank_map_sha{k}']), k
    maps = [mghs_op.voxel_index(plan, calib, k)[0] for k in range(4)]      # verified against the reference's hashes above
    outs, grads, (plan, ws) = run_fused(gpu, cfg, calib_np, depth, feat, hidx, g['ref_inv_post_rot'], g['ref_combine'], s_w)
    assert_product_keys_equal_maps(gpu, cfg, plan, ws, maps, hidx)
    kept, ivs = mghs_op.stats(plan, ws)
    assert kept[0] == int(g['n_kept0']) and ivs[0] == int(g['n_intervals0'])
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.reshape(-1)[g[f'out_pos{k}']], g[f'out_val{k}'], atol=3e-5, rtol=1e-5)
        s = g[f'out_sum{k}']
        assert abs(o.astype(np.float64).sum() - s[0]) < 1e-6 * s[1] + 1e-3
        assert abs(np.abs(o).astype(np.float64).sum() - s[1]) < 1e-6 * s[1] + 1e-3
        assert int(np.count_nonzero(o)) == int(s[2])
    np.testing.assert_allclose(grads[0].reshape(-1)[g['depth_grad_pos']], g['depth_grad_val'], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(grads[1].reshape(-1)[g['feat_grad_pos']], g['feat_grad_val'], atol=2e-4, rtol=1e-5)
    for a, key in ((grads[0], 'depth_grad_sum'), (grads[1], 'feat_grad_sum')):
        assert abs(a.astype(np.float64).sum() - g[key][0]) < 1e-5 * g[key][1] + 1e-2


@pytest.mark.parametrize('geometry,batch', [('dhd-s', 4), ('dhd-l', 2)])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_half_outputs_equal_the_cast_of_the_float32_path_bit_for_bit(gpu, geometry, batch, dtype):
    """dhd_tensor_view.dtype = DHD_F16 / DHD_BF16 (the autocast configurations, DHD-S.py:281; the reference's operator returns
    float32, bev_pool.py:20-21, which the next convolution casts at once): at the full G3 (B = 4) and G15 (DHD-L, B = 2) sizes
    the writer's half tensors are bit-identical to `float32 outputs -> .to(dtype)`, in the collapsed and the stacked layout, and
    the backward fed with half gradients returns bit-identical depth / context gradients to the float32 backward fed with the
    same gradients widened (deterministic grouping: both runs sum in the same order)."""
    from dhd_amd import mghs_op
    cfg = syn.dhd_s_config()
    if geometry == 'dhd-l':
        cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])
        cfg['input_size'] = (512, 1408)
    N = 6
    fh, fw = cfg['input_size'][0] // 16, cfg['input_size'][1] // 16
    calib_np = syn.make_calibration(611, batch, N, cfg['input_size'])
    from oracle import mghs_oracle as O
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    D = len(axes[2])
    depth, feat, hidx = syn.lift_inputs(612, batch, N, D, fh, fw, 64, 65)
    grids = [mghs_op.grid_from_cfg(g_) for g_ in grid_cfgs(cfg)]
    plan = mghs_op.Plan(batch, N, D, fh, fw, 64, grids, deterministic=True)
    assert plan.half_outputs_supported
    calib, keep = device_calib(calib_np, axes, gpu)
    height = T(syn.height_probs_from_index(hidx, 65), gpu)
    for layout in ('collapsed', 'stacked'):
        res = {}
        for odt in (torch.float32, dtype):
            dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
            outs = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], dt, ft, layout=layout, out_dtype=odt)
            assert all(o.dtype == odt for o in outs)
            gs = [T(syn.hash_signed(620 + k, tuple(o.shape)), gpu).to(dtype) for k, o in enumerate(outs)]   # half-representable gradients
            torch.autograd.backward(outs, [g_.to(odt) for g_ in gs])
            res[odt] = ([o.detach() for o in outs], dt.grad, ft.grad)
        for a, b in zip(res[torch.float32][0], res[dtype][0]):
            assert torch.equal(a.to(dtype), b) and int((b != 0).sum()) > 1000
        assert torch.equal(res[torch.float32][1], res[dtype][1]) and torch.equal(res[torch.float32][2], res[dtype][2])
        assert res[dtype][1].abs().sum() > 0
        del res


def test_half_outputs_are_refused_off_the_compact_path(gpu):
    from dhd_amd import _lib, mghs_op
    cfg = small_dhds_cfg()
    plan, axes = make_plan(cfg, 1, 3, channels=24)          # C != 64: the generic row kernels, float32 only
    assert not plan.half_outputs_supported
    calib, keep = device_calib(syn.make_calibration(5, 1, 3, cfg['input_size']), axes, gpu)
    depth, feat, hidx = syn.lift_inputs(6, 1, 3, 44, 4, 11, 24, 65)
    with pytest.raises(_lib.DhdError, match='UNSUPPORTED'):
        mghs_op.mghs_lift_pool(plan, calib, T(syn.height_probs_from_index(hidx, 65), gpu), cfg['height_range'], cfg['mask_range'],
                               T(depth, gpu), T(feat, gpu), out_dtype=torch.float16)


def test_full_size_properties_batch4(gpu):
    """BASELINE workload size (B=4): size-independent checks.  (1) checksum: the sum over every
    voxel and channel equals sum over kept points of depth * sum_c feat, computed from the per-point
    maps; (2) linearity in depth; (3) two runs agree to rounding (within-voxel order is free);
    (4) raw-calibration path keeps the same number of points as the oracle-free count."""
    from dhd_amd import mghs_op
    cfg = syn.dhd_s_config()
    B = 4
    calib_np = syn.make_calibration(77, B, 6, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(78, B, 6, 44, 16, 44, 64, 65)
    outs, grads, (plan, ws) = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=600)
    outs2, _, _ = run_fused(gpu, cfg, calib_np, (2 * depth).astype(np.float32), feat, hidx)
    _, axes = make_plan(cfg, B, 6)
    calib, keep = device_calib(calib_np, axes, gpu)
    from oracle import mghs_oracle as O
    band = O.band_index(hidx, cfg['height_range'], cfg['mask_range'])
    fsum = feat.astype(np.float64).sum(1)  # (BN, fH, fW)
    d64 = depth.astype(np.float64)
    kept, ivs = mghs_op.stats(plan, ws)
    for k in range(4):
        rank, _ = mghs_op.voxel_index(plan, calib, k)
        m = (rank.cpu().numpy() >= 0).reshape(B * 6, 44, 16, 44)
        if k > 0:
            m = m & (band == k - 1)[:, None]
        assert int(m.sum()) == kept[k]
        expect = (d64 * m * fsum[:, None]).sum()
        got = outs[k].astype(np.float64).sum()
        assert abs(got - expect) < 1e-4 * np.abs(d64 * m * np.abs(feat).astype(np.float64).sum(1)[:, None]).sum() + 1e-3
        # not bit-linear: the within-voxel summation order differs from run to run
        np.testing.assert_allclose(outs2[k], 2 * outs[k], atol=1e-4, rtol=1e-4)
        assert int(np.count_nonzero(outs[k].reshape(B, -1, 64, *outs[k].shape[2:]).any(2))) <= ivs[k]
    outs3, grads3, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=600)
    for a, b in zip(outs, outs3):
        np.testing.assert_allclose(a, b, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(grads[0], grads3[0], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(grads[1], grads3[1], atol=1e-3, rtol=1e-4)


def test_deterministic_mode_makes_the_forward_bit_reproducible(gpu):
    """DHD_MGHS_DETERMINISTIC (dhd_mghs_desc.flags): the entries of a voxel are ordered by point id instead of by the arrival order of
    the counting atomics, so the per-voxel float32 sums -- hence every output -- are bit-identical from run to run, at
    the full DHD-S size and B = 4; the results still agree with the default mode to rounding; gradients too."""
    from dhd_amd import mghs_op
    cfg = syn.dhd_s_config()
    B = 4
    calib_np = syn.make_calibration(91, B, 6, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(92, B, 6, 44, 16, 44, 64, 65)
    ref_outs, ref_grads, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=700)
    assert not mghs_op.is_deterministic()
    mghs_op.set_deterministic(True)
    try:
        runs = [run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=700) for _ in range(3)]
    finally:
        mghs_op.set_deterministic(False)
    for outs, grads, _ in runs[1:]:
        for a, b in zip(runs[0][0], outs):
            assert np.array_equal(a, b)
        assert np.array_equal(runs[0][1][0], grads[0]) and np.array_equal(runs[0][1][1], grads[1])
    # the stand-alone entry points (dhd_height_band + dhd_feat_nchw_to_nhwc + dhd_mghs_prepare) and the fused dhd_mghs_lift
    # produce the same grouping: in deterministic mode the results are bit-identical
    mghs_op.set_deterministic(True)
    try:
        sep_outs, sep_grads, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=700, separate=True)
    finally:
        mghs_op.set_deterministic(False)
    for a, b in zip(runs[0][0], sep_outs):
        assert np.array_equal(a, b)
    assert np.array_equal(runs[0][1][0], sep_grads[0]) and np.array_equal(runs[0][1][1], sep_grads[1])
    for a, b in zip(runs[0][0], ref_outs):
        assert np.array_equal(a != 0, b != 0)
        np.testing.assert_allclose(a, b, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(runs[0][1][0], ref_grads[0], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(runs[0][1][1], ref_grads[1], atol=1e-3, rtol=1e-4)
    # and against the oracle on a small case, where the ascending-point-order sum can be restated exactly
    from oracle import mghs_oracle as O
    cfg2 = small_dhds_cfg()
    calib2 = syn.make_calibration(93, 1, 3, cfg2['input_size'])
    d2, f2, h2 = syn.lift_inputs(94, 1, 3, 44, 4, 11, 8, 65)
    mghs_op.set_deterministic(True)
    try:
        outs2, _, _ = run_fused(gpu, cfg2, calib2, d2, f2, h2)
    finally:
        mghs_op.set_deterministic(False)
    for o, r in zip(outs2, O.view_transform(cfg2, calib2, d2, f2, h2)):
        np.testing.assert_allclose(o, r, atol=1e-5, rtol=1e-5)


# --------------------------------------------------------------------------- module level

def test_mghs_module_matches_reference_outputs(gpu):
    """MGHS.view_transform through the registry-built module (raw calibration -> device inverse)."""
    from dhd_amd import build_neck
    g = golden('g2b_small_dhds')
    cfg = dict(small_dhds_cfg(), type='MGHS', heightnet_cfg=dict(use_dcn=False, use_aspp=False))
    m = build_neck(cfg).to(gpu)
    calib = [T(a, gpu) for a in golden_calib(g)]
    B, N = calib[0].shape[:2]
    x = torch.zeros(B, N, 1, 4, 11, device=gpu)
    height = T(syn.height_probs_from_index(g['height_idx'], 65), gpu)
    dt, ft = T(g['depth'], gpu).requires_grad_(), T(g['tran_feat'], gpu).requires_grad_()
    bev, depth, h, lo, mid, hi = m.view_transform([x] + calib, dt, ft, height)
    assert depth is dt and h is height
    for k, o in enumerate((bev, lo, mid, hi)):
        ref = g[f'out{k}']
        diff = np.abs(o.detach().cpu().numpy() - ref)
        # own inverse: a boundary point may land in the neighbouring voxel; everything else is equal
        assert (diff > 1e-5).sum() <= 4 * 8 * 3, k
    assert m.grid_config is m.mask_3_grid  # state the reference leaves behind
    ego = m.get_ego_coor(*calib)
    assert np.abs(ego.cpu().numpy() - g['coor']).max() < 1e-4
    # operator-seam route through the API-parity methods gives the same tensors
    m.create_grid_infos(**{a: cfg['mask_2_grid'][a] for a in 'xyz'})
    via_seam = m.voxel_pooling_v2(ego, dt.view(B, N, 44, 4, 11), (ft * (T(g['band'], gpu) == 1)[:, None]).view(B, N, 8, 4, 11))
    np.testing.assert_allclose(via_seam.detach().cpu().numpy(), mid.detach().cpu().numpy(), atol=1e-5)


def test_mghs_forward_backward_end_to_end(gpu):
    from dhd_amd import MGHS
    cfg = small_dhds_cfg()
    cfg['in_channels'] = 32
    torch.manual_seed(0)
    m = MGHS(**cfg).to(gpu)
    B, N = 2, 2
    calib = [T(a, gpu) for a in syn.make_calibration(5, B, N, cfg['input_size'])]
    x = torch.randn(B, N, 32, 4, 11, device=gpu)
    mlp = m.get_mlp_input(*calib)
    bev, depth, height, lo, mid, hi = m([x] + calib + [mlp])
    assert bev.shape == (B, 8, 200, 200) and lo.shape == (B, 32, 200, 200) and hi.shape == (B, 64, 200, 200)
    assert depth.shape == (B * N, 44, 4, 11) and height.shape == (B * N, 65, 4, 11)
    gt_d = T(np.where(syn.hash_uniform(1, (B, N, 64, 176)) < 0.05, 1 + 40 * syn.hash_uniform(2, (B, N, 64, 176)), 0).astype(np.float32), gpu)
    gt_h = T(np.where(syn.hash_uniform(1, (B, N, 64, 176)) < 0.05, -1 + 6 * syn.hash_uniform(3, (B, N, 64, 176)), 0).astype(np.float32), gpu)
    loss = m.get_height_loss(gt_d, gt_h, height) + bev.square().mean() + lo.mean() + mid.mean() + hi.mean()
    loss.backward()
    assert m.depth_net.weight.grad is not None and torch.isfinite(m.depth_net.weight.grad).all()
    assert m.height_net.depth_conv[-1].weight.grad is not None


# --------------------------------------------------------------------------- SFA

@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_sfa_vs_reference(gpu, mode):
    from dhd_amd import SFA
    g = golden('g5_sfa')
    sfa = SFA(in_channels=32, out_channels=16)
    sfa.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    sfa = sfa.to(gpu).train(mode == 'train')
    x = T(g['x'], gpu).requires_grad_()
    sd0 = {k: v.clone() for k, v in sfa.state_dict().items()}
    stage = sfa.mysk_7(x)
    np.testing.assert_allclose(stage.detach().cpu().numpy(), g[f'{mode}.stage'], atol=2e-5, rtol=1e-4)
    sfa.load_state_dict(sd0)
    out = sfa(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f'{mode}.out'], atol=5e-5, rtol=1e-4)
    (out * T(g['w'], gpu)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'{mode}.xgrad'], atol=2e-4, rtol=1e-3)
    for k, p in sfa.named_parameters():
        ref = g[f'{mode}.pgrad.{k}']
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-4 * max(1.0, np.abs(ref).max()), rtol=1e-3)


class _ReluMargin(torch.autograd.Function):
    """relu(z) whose backward passes the gradient where z > margin."""

    @staticmethod
    def forward(ctx, z, margin):
        ctx.save_for_backward(z)
        ctx.margin = margin
        return z.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return g * (z > ctx.margin), None


def _plain_stage(st, x, margin=0.0):
    """mix.py:37-59 in plain PyTorch on the module's own layers."""
    c = st.channels
    xb, xv = torch.split(x, c, dim=1)
    a1 = st.fc(x.mean(-1).mean(-1))[:, :, None, None]
    xb1, xv1 = a1 * xb, (1 - a1) * xv
    sp = st.spacial_leanring
    z = _ReluMargin.apply(sp[1](sp[0](xb1 + xv1)), margin)
    a2 = torch.sigmoid(sp[4](sp[3](z)))
    return a2 * xb1 + (1 - a2) * xv1


_TIE = 2e-6

# GEMM precisions of the stage operator (include/dhd_amd.h: dhd_sfa_weights.gemm, per call) and the tolerance factor the
# tests grant them relative to the float32-level one: bf16x3 (the default) drops product terms of relative size
# <= 3 * 2^-18, the path's bar on outputs is 1e-3 (BASELINE.json north_star).  The factor is what the gradients measure, not slack:
# round 4 reran the bf16x3 cases with DHD_TEST_BF16X3_FACTOR = 3 / 5 / 8 (the full-size comparison with PyTorch and the float64
# oracle's training case fail: parameter gradients of sums over 10^5 products) and 12 (two C = 128 cases fail); 20 is the first
# round value that holds everywhere.
import os as _os
GEMM_MODES = {'bf16x3': float(_os.environ.get('DHD_TEST_BF16X3_FACTOR', 20.0)), 'bf16x6': 1.0}


class gemm_mode:
    """with gemm_mode('bf16x6') as tol_factor: ...  -- every channel_spatial_stage called inside uses that precision
    (class-level default of the `gemm` attribute; an instance can still set its own)."""

    def __init__(self, name):
        self.name, self.factor = name, GEMM_MODES.get(name, 1.0)

    def __enter__(self):
        from dhd_amd.mix import channel_spatial_stage
        channel_spatial_stage.gemm = self.name
        return self.factor

    def __exit__(self, *exc):
        from dhd_amd.mix import channel_spatial_stage
        channel_spatial_stage.gemm = None
        return False


def _check_stage_against_torch(st, x, tol_x=1e-4, tol_p=5e-4, tol_out=1e-4, tie=_TIE, rel_l2=None):
    """Forward + backward of `st` (HIP) on x against plain PyTorch fp32 on a copy of the module.

    A pre-ReLU activation within rounding of zero (measured: 1.4e-7 at the one or two pixels that
    differ at full size) can fall on either side of the ReLU in two correct float32 implementations
    (BatchNorm folded into scale/shift here, (y - mean) * rstd in PyTorch); the gradient through that
    element is then passed by one and blocked by the other.  The reference is therefore evaluated with
    the ReLU gradient cut at +tie and at -tie: every gradient must agree with the first up to
    tol * scale plus the (element-wise) difference between the two -- zero unless a tie touches it.
    (bf16x3 mode: the conv output carries an error of ~2e-5, so `tie` widens by the same factor as the tolerances.)"""
    import copy
    refs = [copy.deepcopy(st) for _ in range(2)]
    out = st(x)
    g = torch.randn_like(out)
    out.backward(g)
    res = []
    for ref, margin in zip(refs, (tie, -tie)):
        x2 = x.detach().clone().requires_grad_()
        o2 = _plain_stage(ref, x2, margin)
        o2.backward(g)
        res.append((o2.detach(), x2.grad, [q.grad for q in ref.parameters()]))
    (o_a, gx_a, gp_a), (_, gx_b, gp_b) = res

    def close(mine, ra, rb, tol, what):
        scale = max(1.0, ra.abs().max().item())
        over = (mine - ra).abs() - 1.01 * (ra - rb).abs()
        excess = over.max().item()
        assert excess <= tol * scale, (what, excess, scale)
        # next to the element-wise maximum (x 20 for bf16x3, see GEMM_MODES): the relative L2 error of the whole tensor, which is
        # what DESIGN.md section 6 quotes (1.5e-3 for dW1 at full size in bf16x3) -- tie elements excluded as above
        if rel_l2 is not None and ra.norm().item() > 1e-3 * max(1.0, float(ra.numel()) ** 0.5 * 1e-3):
            rel = (over.clamp_min(0).norm() / ra.norm()).item()
            assert rel <= rel_l2, (what, 'relative L2', rel)
    close(out.detach(), o_a, o_a, tol_out, 'out')
    close(x.grad, gx_a, gx_b, tol_x, 'gx')
    for (k, p), qa, qb in zip(st.named_parameters(), gp_a, gp_b):
        close(p.grad, qa, qb, tol_p, k)
    for (k, u), v in zip(st.named_buffers(), refs[0].buffers()):
        close(u.float(), v.float(), v.float(), 1e-5, k)


@pytest.mark.parametrize('gemm', list(GEMM_MODES))
def test_sfa_stage_full_size_vs_torch(gpu, gemm):
    """(1,512,200,200): the fused stage against the same math in plain PyTorch fp32."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(1)
    st = channel_spatial_stage(512).to(gpu)
    x = torch.randn(1, 512, 200, 200, device=gpu, requires_grad=True)
    with gemm_mode(gemm) as f:
        _check_stage_against_torch(st, x, tol_x=1e-4 * f, tol_p=5e-4 * f, tol_out=1e-4 * min(f, 3.0), tie=_TIE * f,
                                   rel_l2=3e-3 if f > 1.0 else 1e-4)


@pytest.mark.parametrize('gemm', list(GEMM_MODES))
@pytest.mark.parametrize('c,b,h,w,train', [(128, 3, 36, 40, True), (128, 2, 20, 28, False), (256, 2, 52, 60, True),
                                           (512, 1, 24, 40, True), (64, 2, 20, 20, True),
                                           # more samples than coefficient tables fit beside the resident weights in LDS:
                                           # the GEMM launcher splits the batch (4 + 2 at C = 256, 2 + 1 at C = 512 in bf16x3)
                                           (256, 6, 12, 20, True), (512, 3, 12, 20, True)])
def test_sfa_stage_vs_torch(gpu, c, b, h, w, train, gemm):
    """The stage operator (dhd_sfa_stage_forward/backward: f32-MFMA 1x1 convs with fused blends /
    BatchNorm / ReLU; C = 64 takes the generic path) against plain PyTorch fp32 on the same parameters:
    output, input gradient, all 12 parameter gradients and the running statistics."""
    from dhd_amd.mix import channel_spatial_stage, fused_stage_supported
    torch.manual_seed(c + h)
    st = channel_spatial_stage(2 * c).to(gpu)
    with torch.no_grad():  # non-trivial BatchNorm state
        for bn in (st.spacial_leanring[1], st.spacial_leanring[4]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    st.train(train)
    x = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).requires_grad_()
    assert fused_stage_supported(st, x) == (c != 64)
    with gemm_mode(gemm) as f:
        _check_stage_against_torch(st, x, tol_x=1e-4 * f, tol_p=5e-4 * f, tol_out=1e-4 * min(f, 3.0), tie=_TIE * f,
                                   rel_l2=3e-3 if f > 1.0 else 1e-4)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('c,b,h,w', [(256, 2, 52, 60), (128, 3, 36, 40), (512, 1, 24, 40)])
def test_sfa_stage_half_io_equals_the_float32_operator_on_the_widened_input(gpu, c, b, h, w, dtype):
    """dhd_sfa_weights.io_dtype (a caller inside an autocast region hands a half x): the operator widens x once, computes in
    float32, and returns `out` / `gx` in x's dtype.  Bit for bit: out == float32 operator(x.float()).to(dtype), and with the same
    half output gradient gx == float32 gx .to(dtype) and the 12 parameter gradients are identical."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(c + h)
    st = channel_spatial_stage(2 * c).to(gpu).train()
    st.half_storage = False   # this test pins the ABI 3 form (half edges only); half storage: the tests below
    xh = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).to(dtype)
    gh = torch.randn(b, c, h, w, device=gpu).to(dtype)
    sd0 = {k: v.clone() for k, v in st.state_dict().items()}
    res = []
    for x_in, g_in in ((xh.float(), gh.float()), (xh, gh)):
        st.load_state_dict(sd0)
        st.zero_grad()
        x_in = x_in.clone().requires_grad_()
        out = st(x_in)
        assert out.dtype == x_in.dtype
        out.backward(g_in)
        assert x_in.grad.dtype == x_in.dtype
        res.append((out.detach(), x_in.grad, [p.grad.clone() for p in st.parameters()], [v.clone() for v in st.state_dict().values()]))
    (o32, gx32, gp32, sd32), (oh, gxh, gph, sdh) = res
    assert torch.equal(o32.to(dtype), oh) and torch.equal(gx32.to(dtype), gxh)
    for a, b_ in zip(gp32, gph):
        assert torch.equal(a, b_)
    for a, b_ in zip(sd32, sdh):
        assert torch.equal(a, b_)


def _stage_errors_against_float64(st, xh, gh, run):
    """Relative L2 errors of `run(stage copy, x, g) -> (out, gx, [parameter grads], [buffers])` against the same mathematics
    (mix.py:37-59, _plain_stage) in float64 on the same half inputs."""
    import copy
    ref = copy.deepcopy(st).double()
    xd = xh.double().requires_grad_()
    od = _plain_stage(ref, xd)
    od.backward(gh.double())
    want = [od.detach(), xd.grad] + [p.grad for p in ref.parameters()] + [v for v in ref.buffers() if v.dtype.is_floating_point]
    mine = copy.deepcopy(st)
    out, gx, gp, bufs = run(mine, xh, gh)
    got = [out, gx] + list(gp) + [v for v in bufs if v.dtype.is_floating_point]
    names = ['out', 'gx'] + [k for k, _ in st.named_parameters()] + [k for k, v in st.named_buffers() if v.dtype.is_floating_point]
    errs = {}
    for k, a, r in zip(names, got, want):
        assert a.shape == r.shape, k
        errs[k] = ((a.double() - r).norm() / r.norm().clamp_min(1e-30)).item(), r.norm().item()
    return errs


def _run_ours(st, xh, gh):
    x = xh.clone().requires_grad_()
    out = st(x)
    assert out.dtype == xh.dtype
    out.backward(gh)
    assert x.grad.dtype == xh.dtype
    return out.detach(), x.grad, [p.grad for p in st.parameters()], list(st.buffers())


def _run_autocast(st, xh, gh):
    """torch.autocast of the plain formulation on the module's own layers: what the reference does under
    `fp16 = dict(loss_scale='dynamic')` (DHD-S.py:281) with mix.py:37-59."""
    x = xh.clone().requires_grad_()
    with torch.autocast('cuda', dtype=xh.dtype):
        out = _plain_stage(st, x)
    out.backward(gh.to(out.dtype))
    return out.detach(), x.grad, [p.grad for p in st.parameters()], list(st.buffers())


def _assert_no_worse_than_autocast(mine, auto, slack=1.05):
    """Every quantity's relative L2 error against float64: ours <= autocast's (x slack for the noise of two roundings of the
    same size).  Quantities that vanish identically (the convolution biases' gradients in front of a train-mode BatchNorm) are
    compared on their absolute scale instead."""
    worse = []
    for k, (e, nrm) in mine.items():
        ea, _ = auto[k]
        if nrm < 1e-6:
            e, ea = e * nrm, ea * nrm     # absolute errors of a quantity whose reference is zero
            if e > max(slack * ea, 1e-3):
                worse.append((k, e, ea))
        elif e > slack * ea + 1e-7:
            worse.append((k, e, ea))
    assert not worse, worse


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('c,b,h,w', [(256, 2, 52, 60), (128, 3, 36, 40), (256, 2, 18, 28), (128, 1, 8, 8), (256, 9, 16, 24)])
def test_sfa_stage_half_storage_is_no_less_accurate_than_autocast(gpu, c, b, h, w, train, dtype):
    """dhd_sfa_weights.storage_dtype (ABI 4): x read in half, y1 / y2 / g2 / g1 / du kept in half, single-product half GEMMs
    with float32 accumulation, float32 statistics and parameter gradients.  The bar: against float64 on the
    same inputs, no larger an error than torch.autocast of the reference formulation on the same GPU -- output, input gradient,
    the 12 parameter gradients and the BatchNorm running statistics.  Shapes: several tiles, a ragged last tile (hw % 64 != 0),
    a single partial tile, and more samples than one GEMM launch holds coefficient tables for.

    At sizes like these the gradient errors of BOTH implementations are dominated by the few pre-ReLU activations that the half
    rounding of y1 pushes across zero (each contributes its whole gradient: 1.2e-2 .. 1.6e-2 relative error of dW1 for either side,
    against 5e-4 for quantities behind no ReLU), and WHICH elements flip differs between two roundings: one draw's ratio ours /
    autocast scatters between 0.1 and 3.3 at 160 pixels (experiments/half_vs_autocast_stats.py, profiles/r5/half_vs_autocast_stats.txt:
    geometric means over 12 seeds 0.24 .. 0.85 for every quantity, the output 0.39 .. 0.50).  So: the GEOMETRIC MEAN of the ratio
    over eight seeds must be <= 1 for every quantity -- within the noise of that mean itself: with single draws scattering by a factor
    e^0.4 the mean of eight has a standard deviation of 15 %, the assertion is at 1.25 -- the forward output (no flips involved) is no
    worse in any single draw, and the full-size test below compares single draws."""
    import math
    from dhd_amd.mix import channel_spatial_stage
    assert bool(__import__('dhd_amd')._lib.load().dhd_sfa_stage_half_storage_supported(c, h * w))
    logs = {}
    for seed in range(8):
        torch.manual_seed(1000 * seed + c + h + b)
        st = channel_spatial_stage(2 * c).to(gpu)
        with torch.no_grad():
            for bn in (st.spacial_leanring[1], st.spacial_leanring[4]):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.5, 0.5)
                bn.running_mean.uniform_(-0.2, 0.2)
                bn.running_var.uniform_(0.5, 1.5)
        st.train(train)
        xh = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).to(dtype)
        gh = torch.randn(b, c, h, w, device=gpu).to(dtype)
        mine = _stage_errors_against_float64(st, xh, gh, _run_ours)
        auto = _stage_errors_against_float64(st, xh, gh, _run_autocast)
        eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
        assert mine['out'][0] < 2 * eps and mine['gx'][0] < 32 * eps, mine   # absolute sanity bounds
        assert mine['out'][0] <= auto['out'][0], (seed, mine['out'], auto['out'])
        for k, (e, nrm) in mine.items():
            if nrm > 1e-6 and auto[k][0] > 0 and e > 0:
                logs.setdefault(k, []).append(math.log(e / auto[k][0]))
            elif nrm <= 1e-6:     # vanishes identically (conv bias in front of a train-mode BatchNorm): absolute scale
                assert e * nrm <= max(1.05 * auto[k][0] * auto[k][1], 1e-3), (k, e * nrm)
    worse = {k: math.exp(sum(v) / len(v)) for k, v in logs.items() if sum(v) / len(v) > math.log(1.25)}
    assert not worse, worse


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_sfa_stage_half_storage_full_size_vs_autocast(gpu, dtype):
    """The benchmark's shape (4,512,200,200), train mode: same bar."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(3)
    st = channel_spatial_stage(512).to(gpu).train()
    xh = torch.randn(4, 512, 200, 200, device=gpu).to(dtype)
    gh = torch.randn(4, 256, 200, 200, device=gpu).to(dtype)
    mine = _stage_errors_against_float64(st, xh, gh, _run_ours)
    auto = _stage_errors_against_float64(st, xh, gh, _run_autocast)
    _assert_no_worse_than_autocast(mine, auto)


def test_sfa_stage_half_storage_falls_back_where_unsupported(gpu):
    """hw % 8 != 0 or C = 512: the operator keeps float32 storage with half edges (ABI 3 form) -- same dtypes out."""
    from dhd_amd.mix import channel_spatial_stage
    import dhd_amd
    lib = dhd_amd._lib.load()
    assert not lib.dhd_sfa_stage_half_storage_supported(256, 18 * 22) and not lib.dhd_sfa_stage_half_storage_supported(512, 64)
    st = channel_spatial_stage(512).to(gpu).train()
    x = torch.randn(2, 512, 18, 22, device=gpu).half().requires_grad_()
    out = st(x)
    out.sum().backward()
    assert out.dtype == torch.float16 and x.grad.dtype == torch.float16
    # and the generic path (C = 64: blend kernels around library convolutions) returns x's dtype too (it returned float32)
    from dhd_amd.mix import fused_stage_supported
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        st64 = channel_spatial_stage(128).to(gpu).train()
        x64 = torch.randn(2, 128, 12, 12, device=gpu).to(dt).requires_grad_()
        assert not fused_stage_supported(st64, x64)
        o64 = st64(x64)
        o64.float().sum().backward()
        assert o64.dtype == dt and x64.grad.dtype == dt


@pytest.mark.parametrize('c,b,h,w', [(128, 2, 20, 28), (256, 2, 36, 40), (512, 1, 24, 40)])
def test_sfa_stage_f32_mfma_mode_vs_torch(gpu, c, b, h, w):
    """gemm = 'f32' (DHD_SFA_GEMM_F32: plain float32 MFMA kernels) through the same comparison."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(c + w)
    st = channel_spatial_stage(2 * c).to(gpu).train()
    st.gemm = 'f32'          # per instance: other stages of the process keep their own precision
    x = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).requires_grad_()
    _check_stage_against_torch(st, x)


@pytest.mark.parametrize('c,b,h,w', [(256, 2, 200, 200), (512, 1, 200, 200), (256, 1, 64, 72), (128, 3, 36, 40)])
def test_sfa_stage_is_bit_reproducible_and_precision_is_per_instance(gpu, c, b, h, w):
    """Every precision accumulates each output element in a fixed order (per-worker partial matrices reduced by a tree, no
    float atomics): two runs are bit-identical, forward and all gradients.  And the precision is a property of the call:
    two stages of one process with different `gemm` interleave without influencing each other."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(c + h)
    st = channel_spatial_stage(2 * c).to(gpu).train()
    other = channel_spatial_stage(2 * c).to(gpu).train()
    other.load_state_dict(st.state_dict())
    st.gemm, other.gemm = 'bf16x6', 'bf16x3'
    x = (torch.randn(b, 2 * c, h, w, device=gpu) * 0.7 + 0.1).requires_grad_()
    g = torch.randn(b, c, h, w, device=gpu)
    res = []
    for rep in range(3):
        for m in (st, other):
            for p in m.parameters():
                p.grad = None
            x.grad = None
            out = m(x)
            out.backward(g)
            if m is st:
                res.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()])
            elif rep == 0:
                x3 = out.detach().clone()
    assert not torch.equal(x3, res[0][0]) and torch.allclose(x3, res[0][0], atol=2e-4)   # different precision, same result
    for other in res[1:]:
        for a, bb in zip(res[0], other):
            assert torch.equal(a, bb)


def test_fused_sfa_stage_matches_generic_path(gpu):
    """Same module, fused operator vs the blend kernels around library convolutions."""
    from dhd_amd.mix import channel_spatial_stage
    torch.manual_seed(5)
    st = channel_spatial_stage(256).to(gpu).eval()
    x = torch.randn(2, 256, 40, 40, device=gpu)
    with torch.no_grad():
        a = st(x)
        st.fused = False
        b = st(x)
    assert (a - b).abs().max().item() < 1e-4


# --------------------------------------------------------------------------- layouts / API variants

def _small64(seed, n_cams=3):
    cfg = small_dhds_cfg()
    cfg['out_channels'] = 64
    calib_np = syn.make_calibration(seed, 1, n_cams, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(seed + 1, 1, n_cams, 44, 4, 11, 64, 65)
    return cfg, calib_np, depth, feat, hidx


def test_single_grid_view_transform_core_compact_path(gpu):
    """view_transform_core (reference :380-405) = one grid, C = 64: the compact path with G = 1."""
    from dhd_amd import MGHS
    from oracle import mghs_oracle as O
    cfg, calib_np, depth, feat, hidx = _small64(300)
    m = MGHS(**dict(cfg, heightnet_cfg=dict(use_dcn=False, use_aspp=False))).to(gpu)
    calib = [T(a, gpu) for a in calib_np]
    x = torch.zeros(1, 3, 1, 4, 11, device=gpu)
    dt, ft = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    m._set_grid(m.mask_3_grid)
    out, d2 = m.view_transform_core([x] + calib, dt, ft)
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], 16)
    coor = O.ego_coor(axes, calib_np[0], calib_np[2], calib_np[3], calib_np[4], calib_np[5])
    grid = {a: cfg['mask_3_grid'][a] for a in 'xyz'}
    ref = O.voxel_pooling_v2(coor, depth.reshape(1, 3, 44, 4, 11), feat.reshape(1, 3, 64, 4, 11), grid)
    assert d2 is dt and out.shape == ref.shape == (1, 8 * 64, 200, 200)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=1e-5)
    out.sum().backward()
    assert torch.isfinite(dt.grad).all() and ft.grad.abs().sum() > 0
    # the parity hook on a single-grid plan: row 0 = the keys, row 1 (no band grids) = -1, not stale scratch
    from dhd_amd import mghs_op
    plan1, axes1 = make_plan(cfg, 1, 3, n_grids=1)
    calib1, _ = device_calib(calib_np, axes1, gpu)
    ws1 = plan1.new_workspace(gpu, private_scratch=True)
    ws1.scratch.fill_(7)
    mghs_op.prepare(plan1, calib1, None, ws1)
    keys = mghs_op.debug_keys(plan1, ws1)
    assert bool((keys[1] == -1).all()) and int((keys[0] >= 0).sum()) > 0


def test_uncollapsed_layouts_match_collapsed(gpu):
    """collapse_z=False: base MGHS returns (B,C,nz,ny,nx) per grid, MGHS_Depth one (B,C,16,ny,nx)
    tensor for the three bands (lss_heightmap.py:845) -- written in place through strided views."""
    from dhd_amd import MGHS, MGHS_Depth
    cfg, calib_np, depth, feat, hidx = _small64(310)
    calib = [T(a, gpu) for a in calib_np]
    x = torch.zeros(1, 3, 1, 4, 11, device=gpu)
    height = T(syn.height_probs_from_index(hidx, 65), gpu)
    hn = dict(use_dcn=False, use_aspp=False)
    ref_m = MGHS(**dict(cfg, heightnet_cfg=hn)).to(gpu)
    d0, f0 = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    bev, _, _, lo, mid, hi = ref_m.view_transform([x] + calib, d0, f0, height)
    ws = [T(syn.hash_signed(320 + k, tuple(o.shape)), gpu) for k, o in enumerate((bev, lo, mid, hi))]
    sum((o * w).sum() for o, w in zip((bev, lo, mid, hi), ws)).backward()

    def split(o):  # (B, nz*C, ny, nx) -> (B, C, nz, ny, nx)
        return o.view(1, -1, 64, 200, 200).transpose(1, 2)

    m1 = MGHS(**dict(cfg, heightnet_cfg=hn, collapse_z=False)).to(gpu)
    d1, f1 = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    outs = m1.view_transform([x] + calib, d1, f1, height)
    for o, r in zip((outs[0], outs[3], outs[4], outs[5]), (bev, lo, mid, hi)):
        assert o.dim() == 5 and o.is_contiguous()
        assert torch.allclose(o, split(r), atol=1e-5)
    sum((o * split(w)).sum() for o, w in zip((outs[0], outs[3], outs[4], outs[5]), ws)).backward()
    assert torch.allclose(d1.grad, d0.grad, atol=1e-4) and torch.allclose(f1.grad, f0.grad, atol=1e-4)

    m2 = MGHS_Depth(**dict(cfg, heightnet_cfg=hn, depthnet_cfg=hn, collapse_z=False)).to(gpu)
    d2, f2 = T(depth, gpu).requires_grad_(), T(feat, gpu).requires_grad_()
    bev2, bev_w_z, dd, hh = m2.view_transform([x] + calib, d2, f2, height)
    assert bev2.shape == (1, 64, 1, 200, 200) and bev_w_z.shape == (1, 64, 16, 200, 200) and dd is d2 and hh is height
    stacked = torch.cat([split(lo), split(mid), split(hi)], dim=2)
    assert torch.allclose(bev_w_z, stacked, atol=1e-5) and torch.allclose(bev2, split(bev), atol=1e-5)
    wz = torch.cat([split(w) for w in ws[1:]], dim=2)
    ((bev2 * split(ws[0])).sum() + (bev_w_z * wz).sum()).backward()
    assert torch.allclose(d2.grad, d0.grad, atol=1e-4) and torch.allclose(f2.grad, f0.grad, atol=1e-4)
    assert m2.grid_config['z'] == [-1, 5.4, 6.4]  # MGHS_Depth resets the grid (:848-854)


def test_accelerate_caches_only_what_is_static(gpu):
    """accelerate=True at inference (the reference's dormant pre_compute idea, lss_heightmap.py:234-258,374-378):
    while the same calibration tensors are passed unmodified, the four-grid view_transform reuses camera matrices and the
    whole grouping of the full-height grid and redoes only the band grids' part per frame (dhd_mghs_lift_static); the
    single-grid call reuses everything.  A frame with a changed height map, a frame with a changed calibration (new tensors
    or an in-place edit) and a return to the first calibration must each give exactly what the uncached module gives."""
    from dhd_amd import MGHS, mghs_op
    cfg, calib_np, depth, feat, hidx = _small64(330)
    hn = dict(use_dcn=False, use_aspp=False)
    mghs_op.set_deterministic(True)     # bit-comparable sums
    try:
        m = MGHS(**dict(cfg, heightnet_cfg=hn, accelerate=True)).to(gpu).eval()
        plain = MGHS(**dict(cfg, heightnet_cfg=hn, accelerate=False)).to(gpu).eval()
        calib = [T(a, gpu) for a in calib_np]
        x = torch.zeros(1, 3, 1, 4, 11, device=gpu)
        height = T(syn.height_probs_from_index(hidx, 65), gpu)
        calib2 = [T(a, gpu) for a in syn.make_calibration(331, 1, 3, cfg['input_size'])]
        height2 = T(syn.height_probs_from_index(syn.height_index(332, hidx.shape, 65), 65), gpu)
        height3 = T(syn.height_probs_from_index(syn.height_index(333, hidx.shape, 65), 65), gpu)
        lifts = []
        real = mghs_op.lift
        mghs_op.lift = lambda *a, **k: (lifts.append(bool(k.get('static'))), real(*a, **k))[1]
        try:
            with torch.no_grad():
                for cal, hgt in ((calib, height), (calib, height2), (calib, height3), (calib2, height2), (calib2, height), (calib, height2)):
                    a = m.view_transform([x] + cal, T(depth, gpu), T(feat, gpu), hgt)
                    b = plain.view_transform([x] + cal, T(depth, gpu), T(feat, gpu), hgt)
                    for k in (0, 3, 4, 5):
                        assert torch.equal(a[k], b[k]), k
                # accelerated module: full lift, static, static, full (new calibration), static, full; the plain one never static
                assert lifts[0::2] == [False, True, True, False, True, False] and not any(lifts[1::2])
                calib2[0][:, :, 0, 3] += 0.37            # in-place edit of a cached calibration tensor -> full lift
                n0 = len(lifts)
                a = m.view_transform([x] + calib2, T(depth, gpu), T(feat, gpu), height)
                b = plain.view_transform([x] + calib2, T(depth, gpu), T(feat, gpu), height)
                assert lifts[n0] is False and all(torch.equal(a[k], b[k]) for k in (0, 3, 4, 5))
                # single grid: everything is reused while the calibration tensors are the same objects, unmodified
                m._set_grid(cfg['mask_2_grid'])
                plain._set_grid(cfg['mask_2_grid'])
                n0 = len(lifts)
                o1, _ = m.view_transform_core([x] + calib, T(depth, gpu), T(feat, gpu))
                o2, _ = m.view_transform_core([x] + calib, T(2 * depth, gpu), T(feat, gpu))
                assert len(lifts) == n0 + 1 and torch.allclose(o2, 2 * o1, atol=1e-4)      # one lift for the two calls
                r1, _ = plain.view_transform_core([x] + calib, T(depth, gpu), T(feat, gpu))
                assert torch.equal(o1, r1)
                o3, _ = m.view_transform_core([x] + calib2, T(depth, gpu), T(feat, gpu))
                r3, _ = plain.view_transform_core([x] + calib2, T(depth, gpu), T(feat, gpu))
                assert torch.equal(o3, r3) and not torch.allclose(o3, o1, atol=1e-5)
        finally:
            mghs_op.lift = real
    finally:
        mghs_op.set_deterministic(False)


def test_scan_without_waiting_for_other_workgroups_gives_the_same_grouping(gpu):
    """mghs_scan waits for lower-numbered workgroups' chunk aggregates with a bounded spin; past the bound a thread sums the
    predecessor's chunk itself (exact: the counters are final).  DHD_MGHS_DEBUG_SCAN_SELF_SERVE sets the bound to zero, so EVERY
    aggregate takes the fallback: at the full DHD-S size with B = 4 (1 328 chunks) the slots, the per-point keys and (deterministic
    mode) the pooled tensors are identical to the normal run."""
    from dhd_amd import _lib, mghs_op
    cfg = syn.dhd_s_config()
    B = 4
    calib_np = syn.make_calibration(901, B, 6, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(902, B, 6, 44, 16, 44, 64, 65)
    plan, axes = make_plan(cfg, B, 6)
    calib, keep = device_calib(calib_np, axes, gpu)
    height = T(syn.height_probs_from_index(hidx, 65), gpu)
    res = []
    for self_serve in (False, True):
        p = mghs_op.Plan(B, 6, 44, 16, 44, 64, plan.grids, deterministic=True)
        if self_serve:
            p.desc.flags |= _lib.MGHS_DEBUG_SCAN_SELF_SERVE
        ws = p.new_workspace(gpu, private_scratch=True)
        with torch.no_grad():
            outs = mghs_op.mghs_lift_pool(p, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu), ws)
        res.append((outs, mghs_op.debug_keys(p, ws), mghs_op.stats(p, ws)))
    (o0, k0, s0), (o1, k1, s1) = res
    assert s0 == s1 and s0[0][0] > 400000 and torch.equal(k0, k1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)


def test_batch_of_eight_equals_two_batches_of_four(gpu):
    """B = 8 at the full DHD-S size: 5.44 M voxel counters = 2 656 scan chunks, more workgroups than the chip holds at once
    (the single-pass scan waits only for lower-numbered chunks, which are dispatched first).  Samples are independent, so
    in deterministic mode the pooled tensors and both gradients of the batch of eight are bit-identical to those of its
    two halves run as batches of four."""
    from dhd_amd import mghs_op
    cfg = syn.dhd_s_config()
    calib_np = syn.make_calibration(501, 8, 6, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(502, 8, 6, 44, 16, 44, 64, 65)
    mghs_op.set_deterministic(True)
    try:
        outs8, grads8, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=None)
        halves = []
        for h in range(2):
            sl, bn = slice(4 * h, 4 * h + 4), slice(24 * h, 24 * h + 24)
            o, _, _ = run_fused(gpu, cfg, [a[sl] for a in calib_np], depth[bn], feat[bn], hidx[bn])
            halves.append(o)
    finally:
        mghs_op.set_deterministic(False)
    for k in range(4):
        assert np.array_equal(outs8[k][:4], halves[0][k]) and np.array_equal(outs8[k][4:], halves[1][k]), k
    assert np.count_nonzero(outs8[0]) > 1000000


@pytest.mark.parametrize('batch', [3, 5])
def test_writer_channel_parts_batches_equal_single_samples(gpu, batch):
    """The streaming writer gives every segment to several workgroups, a contiguous part of the channels each
    (mghs_stream_fwd: 4 parts at the DHD-S geometry, 2 at D = 88).  In deterministic mode a batch's pooled tensors are
    bit-identical to those of its samples run one by one (odd batch sizes: the last round of workgroups is partly filled)."""
    from dhd_amd import mghs_op
    cfg = syn.dhd_s_config()
    calib_np = syn.make_calibration(511, batch, 6, cfg['input_size'])
    depth, feat, hidx = syn.lift_inputs(512, batch, 6, 44, 16, 44, 64, 65)
    mghs_op.set_deterministic(True)
    try:
        outs, _, _ = run_fused(gpu, cfg, calib_np, depth, feat, hidx, weights_seed=None)
        for b in range(batch):
            bn = slice(6 * b, 6 * b + 6)
            o, _, _ = run_fused(gpu, cfg, [a[b:b + 1] for a in calib_np], depth[bn], feat[bn], hidx[bn])
            for k in range(4):
                assert np.array_equal(outs[k][b:b + 1], o[k]), (b, k)
    finally:
        mghs_op.set_deterministic(False)


def test_static_lift_full_size_is_bit_identical_to_a_full_lift(gpu):
    """dhd_mghs_lift_static at the full DHD-S size, B = 4: after one full lift, three frames with new height maps (new
    bands) and new depth / context values each redo only the band grids' grouping; in deterministic mode the four pooled
    tensors are bit-identical to those of a full dhd_mghs_lift on a fresh workspace, frame by frame."""
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    cfg = syn.dhd_s_config()
    B, N = 4, 6
    calib_np = syn.make_calibration(401, B, N, cfg['input_size'])
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    grids = [mghs_op.grid_from_cfg(g) for g in grid_cfgs(cfg)]
    plan = mghs_op.Plan(B, N, 44, 16, 44, 64, grids, deterministic=True)
    calib, keep = device_calib(calib_np, axes, gpu)
    ws_static = plan.new_workspace(gpu, private_scratch=True)
    kept = None
    with torch.no_grad():
        for frame in range(4):
            depth, feat, hidx = syn.lift_inputs(410 + 7 * frame, B, N, 44, 16, 44, 64, 65)
            height = T(syn.height_probs_from_index(hidx, 65), gpu)
            a = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu),
                                       ws_static, static=frame > 0)
            b = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu),
                                       plan.new_workspace(gpu, private_scratch=True))
            for k, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x, y), (frame, k)
            k_now, _ = mghs_op.stats(plan, ws_static)
            if frame == 0:
                kept = k_now
                assert kept[0] > 400000
            else:
                assert k_now[0] == kept[0] and k_now[1:] != kept[1:]      # grid 0's grouping is the static part, the bands moved


def test_static_lift_column_form_equals_a_full_lift(gpu):
    """dhd_mghs_lift_static at the DHD-L geometry (32 x 88 maps, D = 88, B = 2) under DEFAULT flags: fH = 32 takes the column
    form of the grid-0 sums (mghs_col_sums; grid 0's sorted entry list is not built), which round 4 left without a static-lift
    test.  After one full lift, two frames with new height maps / depth / context redo only the band grids' grouping: the keys
    of the product's counting kernel and the whole `state` block (voxel -> slot maps, per-point slots) are identical to those of a
    full lift on a fresh workspace, the pooled tensors equal up to the summation order of the atomics."""
    from dhd_amd import mghs_op
    from oracle import mghs_oracle as O
    cfg = syn.dhd_s_config()
    cfg['grid_config'] = dict(cfg['grid_config'], depth=[1.0, 45.0, 0.5])
    cfg['input_size'] = (512, 1408)
    B, N, D, fh, fw = 2, 6, 88, 32, 88
    calib_np = syn.make_calibration(431, B, N, cfg['input_size'])
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    grids = [mghs_op.grid_from_cfg(g) for g in grid_cfgs(cfg)]
    plan = mghs_op.Plan(B, N, D, fh, fw, 64, grids)
    calib, keep = device_calib(calib_np, axes, gpu)
    ws_static = plan.new_workspace(gpu, private_scratch=True)
    kept = None
    with torch.no_grad():
        for frame in range(3):
            depth, feat, hidx = syn.lift_inputs(440 + 5 * frame, B, N, D, fh, fw, 64, 65)
            height = T(syn.height_probs_from_index(hidx, 65), gpu)
            ws_full = plan.new_workspace(gpu, private_scratch=True)
            ws_full.state.zero_()
            if frame == 0:
                ws_static.state.zero_()      # (alignment gaps of the block are never written: make them comparable)
            a = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu),
                                       ws_static, static=frame > 0)
            b = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu), ws_full)
            assert torch.equal(mghs_op.debug_keys(plan, ws_static), mghs_op.debug_keys(plan, ws_full)), frame
            # the state block (csrc/mghs_layout.h: nzoff[V + 1] | nzvox[max_slots] | p_slot[2P], each carved at 256 bytes): every
            # defined word identical -- nzvox beyond the last slot is never written (the static lift leaves frame 0's there)
            V = sum(B * g_.n[0] * g_.n[1] * g_.n[2] for g_ in plan.grids)
            P = B * N * D * fh * fw
            v0 = B * plan.grids[0].n[0] * plan.grids[0].n[1] * plan.grids[0].n[2]
            max_slots = min(v0, P) + min(V - v0, P)
            al = lambda nbytes: (nbytes + 255) // 256 * 256
            o_vox = al((V + 1) * 4)
            o_slot = o_vox + al(max_slots * 4)
            sa, sb = ws_static.state.view(torch.int32), ws_full.state.view(torch.int32)
            assert torch.equal(sa[:V + 1], sb[:V + 1]), frame
            n_slots = int(sa[V])
            assert 0 < n_slots <= max_slots and torch.equal(sa[o_vox // 4:o_vox // 4 + n_slots], sb[o_vox // 4:o_vox // 4 + n_slots]), frame
            assert torch.equal(sa[o_slot // 4:o_slot // 4 + 2 * P], sb[o_slot // 4:o_slot // 4 + 2 * P]), frame
            for k, (x, y) in enumerate(zip(a, b)):
                assert torch.allclose(x, y, atol=2e-4, rtol=1e-5), (frame, k, (x - y).abs().max().item())
                assert int((x != 0).sum()) == int((y != 0).sum())
            k_now, _ = mghs_op.stats(plan, ws_static)
            if frame == 0:
                kept = k_now
                assert kept[0] > 2000000
            else:
                assert k_now[0] == kept[0] and k_now[1:] != kept[1:]


def test_static_lift_with_unaligned_grid0_counters_equals_a_full_lift(gpu):
    """SURVEY 8(d) config 1 (50 x 50 x 1 full-height grid, B = 1): grid 0's 2 500 counters do not end on a 256-byte
    boundary, which the static lift's block-wise zero-fill needs -- dhd_mghs_lift_static must then run as a full lift
    (identical results by contract) instead of refusing the second frame of an accelerate=True module."""
    from dhd_amd import mghs_op
    cfg = syn.smoke_config()
    calib_np = syn.make_calibration(421, 1, 1, cfg['input_size'])
    from oracle import mghs_oracle as O
    axes = O.frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    full = {a: cfg['grid_config'][a] for a in 'xyz'}            # the smoke configuration's own 50 x 50 x 1 full-height grid
    grids = [mghs_op.grid_from_cfg(g) for g in (full, cfg['mask_1_grid'], cfg['mask_2_grid'], cfg['mask_3_grid'])]
    plan = mghs_op.Plan(1, 1, len(axes[2]), 4, 11, 16, grids)
    assert (plan.grids[0].n[0] * plan.grids[0].n[1] * plan.grids[0].n[2] * 4) % 256 != 0
    calib, keep = device_calib(calib_np, axes, gpu)
    ws_static = plan.new_workspace(gpu, private_scratch=True)
    nh = len(cfg['height_range'])
    with torch.no_grad():
        for frame in range(3):
            depth, feat, hidx = syn.lift_inputs(430 + 5 * frame, 1, 1, 44, 4, 11, 16, nh)
            height = T(syn.height_probs_from_index(hidx, nh), gpu)
            a = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu),
                                       ws_static, static=frame > 0)
            b = mghs_op.mghs_lift_pool(plan, calib, height, cfg['height_range'], cfg['mask_range'], T(depth, gpu), T(feat, gpu),
                                       plan.new_workspace(gpu, private_scratch=True))
            for k, (x, y) in enumerate(zip(a, b)):
                np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), atol=1e-6, rtol=1e-6, err_msg=str((frame, k)))
            assert a[0].abs().sum() > 0


def test_mghs_step_is_graph_capturable(gpu):
    """prepare + forward + backward are plain launches / async memsets on the caller's stream: the
    whole view transform can be captured into a HIP graph and replayed on new inputs."""
    from dhd_amd import mghs_op
    cfg, calib_np, depth, feat, hidx = _small64(340)
    plan, axes = make_plan(cfg, 1, 3, channels=64)
    calib, keep = device_calib(calib_np, axes, gpu)
    d_in, f_in = T(depth, gpu), T(feat, gpu)
    height = T(syn.height_probs_from_index(hidx, 65), gpu)
    ws = plan.new_workspace(gpu)
    gouts = [T(syn.hash_signed(350 + k, s), gpu) for k, s in enumerate(plan.out_shapes())]

    def step():
        band = mghs_op.height_band(height, cfg['height_range'], cfg['mask_range'])
        fn = mghs_op._nchw_to_nhwc(f_in)
        mghs_op.prepare(plan, calib, band, ws)
        outs = mghs_op.pool_forward(plan, d_in, fn, ws)
        dg, fg = mghs_op.pool_backward(plan, d_in, fn, gouts, ws)
        return outs, dg, mghs_op._nhwc_to_nchw(fg)

    ref = step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = step()
    d_in.mul_(2.0)  # new input values in the captured buffers
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(cap[0], ref[0]):
        assert torch.allclose(a, 2 * b, atol=1e-4, rtol=1e-4)
    assert torch.allclose(cap[1], ref[1], atol=1e-4) and torch.allclose(cap[2], 2 * ref[2], atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_sfa_stage_and_losses_are_graph_capturable(gpu, dtype):
    """The stage operator (forward + backward; float32, and the half-storage form a half x selects) and the occupancy losses inside a
    HIP graph: every launch is asynchronous on the capture stream (kernels + hipMemsetAsync only), replay with new input values."""
    from dhd_amd.mix import channel_spatial_stage
    from dhd_amd.occ_loss import occ_losses
    torch.manual_seed(11)
    st = channel_spatial_stage(256).to(gpu).train()
    x = torch.randn(2, 256, 20, 24, device=gpu).to(dtype).requires_grad_()
    gy = torch.randn(2, 128, 20, 24, device=gpu).to(dtype)
    z = torch.randn(3000, 18, device=gpu, requires_grad=True)
    t = torch.randint(0, 18, (3000,), device=gpu).to(torch.uint8)
    cam = (torch.rand(3000, device=gpu) < 0.5).to(torch.uint8)
    cw = T(_class_weights(), gpu)

    def step():
        for p in list(st.parameters()) + [x, z]:
            p.grad = None
        y = st(x)
        y.backward(gy)
        sum(occ_losses(z, t, cam, cw)).backward()
        return y, x.grad, z.grad, st.spacial_leanring[0].weight.grad

    sd = {k: v.clone() for k, v in st.state_dict().items()}
    # detached copies: holding a grad-tracking output of an earlier eager step while capturing forward+backward
    # segfaults in PyTorch-ROCm 2.10 even with plain conv layers (experiments/dbg_graph2.py)
    ref = [a.detach().clone() for a in step()]
    st.load_state_dict(sd)  # running statistics back to their initial values
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = step()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(cap, ref):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-4)      # (the operator is bit-reproducible in either storage)
    with torch.no_grad():
        x.mul_(0.5)
    g.replay()
    torch.cuda.synchronize()
    assert not torch.allclose(cap[0].float(), ref[0].float(), atol=1e-3)


# --------------------------------------------------------------------------- occupancy-head losses (8f-2)

def _class_weights():
    from dhd_amd.detector import NUSC_CLASS_FREQUENCIES
    return (1 / np.log(NUSC_CLASS_FREQUENCIES + 0.001)).astype(np.float32)


def test_occ_losses_vs_reference_golden(gpu):
    """dhd_occ_loss_forward/backward against golden G6 (the reference's own sem/geo scal code: values and
    the gradient of sem + 2*geo) and the oracle's float64 cross entropy."""
    from dhd_amd.occ_loss import occ_losses
    from oracle import mghs_oracle as O
    g = golden('g6_occ_losses')
    cw = _class_weights()
    logits = T(g['logits'], gpu).requires_grad_()
    l_ce, l_sem, l_geo = occ_losses(logits, T(g['labels'], gpu), T(g['mask_camera'], gpu), T(cw, gpu))
    assert abs(l_sem.item() - float(g['sem_scal'])) < 1e-5 and abs(l_geo.item() - float(g['geo_scal'])) < 1e-5
    assert abs(l_ce.item() - O.occ_losses(g['logits'], g['labels'], g['mask_camera'], cw)[0]) < 1e-5
    (l_sem + 2.0 * l_geo).backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g['grad'], atol=2e-7, rtol=1e-4)


@pytest.mark.parametrize('m', [1, 255, 4097, 2 * 200 * 200 * 16])
def test_occ_losses_vs_torch_autograd(gpu, m):
    """All three losses and their joint gradient against the vectorised PyTorch formulation (itself pinned
    to the reference by G6 on CPU), including ragged tile sizes, ignored labels and an absent class."""
    from dhd_amd.detector import CrossEntropyLoss, geo_scal_loss_with_mask, sem_scal_loss_with_mask
    from dhd_amd.occ_loss import occ_losses
    gen = torch.Generator().manual_seed(m)
    z = (3.0 * torch.randn(m, 18, generator=gen)).to(gpu)
    t = torch.randint(0, 18, (m,), generator=gen)
    t[t == 5] = 4
    t[::97] = 255
    t[0] = 3
    if m > 1:
        t[1] = 17
    t = t.to(gpu)
    cam = (torch.rand(m, generator=gen) < 0.4).to(gpu)
    cam[:2] = True
    cw = T(_class_weights(), gpu)
    w = [0.7, 1.3, 2.0]
    a = z.clone().requires_grad_()
    la = occ_losses(a, t, cam, cw)
    sum(wi * li for wi, li in zip(w, la)).backward()
    b = z.clone().requires_grad_()
    counts = torch.bincount(t[cam], minlength=256)[:18]
    avg = (counts.double() * cw.double()).sum().float()
    lb = (CrossEntropyLoss(class_weight=cw)(b, t, weight=cam.int(), avg_factor=avg),
          sem_scal_loss_with_mask(b, t, cam.int()), geo_scal_loss_with_mask(b, t, cam.int(), non_empty_idx=17))
    sum(wi * li for wi, li in zip(w, lb)).backward()
    for x, y in zip(la, lb):
        assert abs(x.item() - y.item()) <= 2e-5 * max(1.0, abs(y.item())), (x.item(), y.item())
    scale = b.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 1e-4 * scale + 1e-9


def test_occ_argmax_and_confusion_histogram_vs_oracle(gpu):
    """dhd_occ_argmax_hist: predictions and the 18x18 confusion counts exactly equal to the oracle's
    (get_occ + Metric_mIoU.hist_info); accumulation over two calls; mIoU from the histogram."""
    from dhd_amd.occ_loss import miou_from_hist, occ_argmax_hist
    from oracle import mghs_oracle as O
    m = 200 * 200 * 16 + 37
    z = syn.hash_signed(71, (m, 18)) * 3.0
    t = (syn.hash_u32(72, m) % 18).astype(np.int64)
    t[::101] = 255
    cam = (syn.hash_uniform(73, (m,)) < 0.35)
    pred, hist = occ_argmax_hist(T(z, gpu), T(t, gpu), T(cam, gpu))
    p_ref, h_ref, iu_ref = O.occ_confusion(z, t, cam)
    assert np.array_equal(pred.cpu().numpy(), p_ref)
    assert np.array_equal(hist.cpu().numpy(), h_ref)
    _, hist = occ_argmax_hist(T(z, gpu), T(t, gpu), T(cam, gpu), hist=hist)
    assert np.array_equal(hist.cpu().numpy(), 2 * h_ref)
    miou, iu = miou_from_hist(hist)
    assert abs(miou - float(np.nanmean(iu_ref[:17]) * 100)) < 1e-9


def test_occ_histogram_and_miou_vs_reference_metric(gpu):
    """Golden G14, recorded from the reference's Metric_mIoU (core/evaluation/occ_metrics.py:78-169) over three samples:
    dhd_occ_argmax_hist accumulating into one device histogram must reproduce its 18 x 18 counts exactly, and
    miou_from_hist its rounded mIoU."""
    from dhd_amd.occ_loss import miou_from_hist, occ_argmax_hist
    from test_oracle_golden import g14_inputs
    g = golden('g14_miou')
    shape = tuple(int(v) for v in g['shape'])
    hist = None
    for k in range(3):
        logits, gt, cam = g14_inputs(k, shape)
        _, hist = occ_argmax_hist(T(logits.reshape(-1, 18), gpu), T(gt.reshape(-1), gpu), T(cam.reshape(-1), gpu), hist=hist)
    assert np.array_equal(hist.cpu().numpy(), g['hist'])
    miou, iu = miou_from_hist(hist)
    np.testing.assert_allclose(iu.cpu().numpy(), g['per_class_iou'], rtol=1e-12)
    assert round(miou, 2) == float(g['miou'])


# --------------------------------------------------------------------------- height / depth supervision (a16)

def test_height_loss_labels_and_value_vs_reference_golden(gpu):
    """dhd_sparse_bin_labels + dhd_bin_bce_* against golden G4 (the reference's get_downsampled_gt_* one-hots and
    get_height_loss, in both grid_config states of the quirk) -- labels exact, loss 1e-6."""
    import dhd_amd
    from dhd_amd import label_loss
    g = golden('g4_loss')
    cfg = syn.dhd_s_config()
    cfg['input_size'] = (64, 176)
    m = dhd_amd.MGHS(**cfg).to(gpu)
    gd, gh, hp = T(g['gt_depth'], gpu), T(g['gt_height'], gpu), T(g['height_prob'], gpu)

    def onehot(bins, n):
        b = bins.cpu().numpy().astype(np.int64)
        return (b[:, None] == np.arange(1, n + 1)[None, :]).astype(np.float32)
    for state, key_d, key_l in ((None, 'gt_depth_onehot_init', 'loss_height_init'),
                                (m.mask_3_grid, 'gt_depth_onehot_after_forward', 'loss_height_after_forward')):
        if state is not None:
            m._set_grid(state)
        dbin, hbin = m._hip_labels(gd, gh)
        np.testing.assert_array_equal(onehot(dbin, m.D), g[key_d])
        np.testing.assert_array_equal(onehot(hbin, m.H), g['gt_height_onehot'])
        assert abs(float(m.get_height_loss(gd, gh, hp)) - float(g[key_l])) < 1e-6


def test_height_and_depth_loss_vs_torch_mirror_full_size(gpu):
    """Full DHD-S size (B=2, 6 cameras, 256x704): loss values and gradients of the HIP operators against the
    PyTorch mirror of the reference code (pinned to G4 on CPU); covers MGHS_Depth's two losses as well."""
    import dhd_amd
    cfg = syn.dhd_s_config()
    md = dhd_amd.build_neck(dict(cfg, type='MGHS_Depth', collapse_z=False, depthnet_cfg=dict(use_dcn=False),
                                 heightnet_cfg=dict(use_dcn=False, use_aspp=False))).to(gpu)
    gen = torch.Generator().manual_seed(3)
    B, N = 2, 6
    sel = torch.rand(B, N, 256, 704, generator=gen) < 0.02
    gd = (torch.rand(B, N, 256, 704, generator=gen) * 50.0 * sel).to(gpu)       # some beyond the 45 m range
    gh = ((torch.rand(B, N, 256, 704, generator=gen) * 7.4 - 1.5) * sel).to(gpu)  # some outside [-1, 5.4]
    hp = torch.softmax(torch.randn(B * N, md.H, 16, 44, generator=gen), 1).to(gpu)
    dp = torch.softmax(torch.randn(B * N, md.D, 16, 44, generator=gen), 1).to(gpu)
    a_h, a_d = hp.clone().requires_grad_(), dp.clone().requires_grad_()
    ld, lh = md.get_depth_and_height_loss(gd, gh, a_d, a_h)
    (0.7 * ld + 1.3 * lh).backward()
    b_h, b_d = hp.clone().requires_grad_(), dp.clone().requires_grad_()
    hl, dl = md.get_downsampled_gt_height(gh), md.get_downsampled_gt_depth(gd)
    fg = dl.max(dim=1).values > 0.0
    rd = md.loss_depth_weight * md._fg_bce(b_d.permute(0, 2, 3, 1).reshape(-1, md.D), dl, fg)
    rh = md.loss_height_weight * md._fg_bce(b_h.permute(0, 2, 3, 1).reshape(-1, md.H), hl, fg)
    (0.7 * rd + 1.3 * rh).backward()
    assert fg.sum().item() > 1000
    assert abs(ld.item() - rd.item()) < 1e-5 * max(1.0, abs(rd.item())) and abs(lh.item() - rh.item()) < 1e-5 * max(1.0, abs(rh.item()))
    for a, b in ((a_h, b_h), (a_d, b_d)):
        scale = b.grad.abs().max().item()
        assert (a.grad - b.grad).abs().max().item() <= 1e-5 * scale


def test_sid_depth_labels_on_the_gpu_vs_the_reference_formulation(gpu):
    """MGHS(sid=True): spacing-increasing depth bins (lss_heightmap.py:655-660) are binned by dhd_sparse_bin_labels_sid
    instead of a PyTorch branch.  Labels against the reference's own formulation evaluated on the CPU (float32 torch.log);
    the device logarithm may differ in the last bit, so a value within an ulp of a bin boundary may move by one bin: at
    most a handful of the ~7 000 labelled pixels; the loss agrees to 1e-4."""
    import dhd_amd
    from dhd_amd import label_loss
    cfg = syn.dhd_s_config()
    m = dhd_amd.build_neck(dict(cfg, type='MGHS', sid=True, heightnet_cfg=dict(use_dcn=False, use_aspp=False)))
    assert m.sid and m.D == 44
    gen = torch.Generator().manual_seed(8)
    B, N = 2, 6
    sel = torch.rand(B, N, 256, 704, generator=gen) < 0.02
    gd = torch.rand(B, N, 256, 704, generator=gen) * 50.0 * sel
    gh = (torch.rand(B, N, 256, 704, generator=gen) * 7.4 - 1.5) * sel
    ref_onehot = m.get_downsampled_gt_depth(gd)                        # the reference's expression, CPU
    ref_bin = torch.where(ref_onehot.sum(1) > 0, ref_onehot.argmax(1) + 1, torch.zeros(1, dtype=torch.long))
    dbin, hbin = label_loss.bin_labels(gd.to(gpu), gh.to(gpu), m.downsample, m.grid_config['depth'], m.D, m.height_range[0],
                                       m.height_interval, m.H, sid=True)
    got = dbin.cpu().long()
    assert (ref_bin > 0).sum() > 5000
    diff = got != ref_bin
    assert int(diff.sum()) <= 4 and int((got - ref_bin).abs().max()) <= 1, (int(diff.sum()), int((got - ref_bin).abs().max()))
    lin, _ = label_loss.bin_labels(gd.to(gpu), gh.to(gpu), m.downsample, m.grid_config['depth'], m.D, m.height_range[0],
                                   m.height_interval, m.H, sid=False)
    assert int((lin.cpu().long() != got).sum()) > 1000                  # really another binning
    hp = torch.softmax(torch.randn(B * N, m.H, 16, 44, generator=gen), 1)
    mg = m.to(gpu)
    loss_gpu = mg.get_height_loss(gd.to(gpu), gh.to(gpu), hp.to(gpu))
    loss_ref = dhd_amd.build_neck(dict(cfg, type='MGHS', sid=True, heightnet_cfg=dict(use_dcn=False, use_aspp=False))).get_height_loss(gd, gh, hp)
    assert abs(float(loss_gpu) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))


def test_rasterise_points_vs_oracle_and_reference_golden(gpu):
    """dhd_points_to_maps: bit-identical to the oracle (stable-sort semantics) on the golden G7 points and on a
    6-camera, 34 k-point, 256x704 case; against the reference's own maps everywhere except the tie pixels."""
    from dhd_amd.label_loss import points_to_maps
    from oracle import mghs_oracle as O
    g = golden('g7_rasterise')
    h, w = (int(v) for v in g['size'])
    dm, hm = points_to_maps(T(g['points'][None], gpu), h, w, 1, tuple(g['depth_range']))
    odm, ohm, omk, ties = O.points_to_maps(g['points'], h, w, 1, tuple(g['depth_range']), return_ties=True)
    assert np.array_equal(dm[0].cpu().numpy(), odm) and np.array_equal(hm[0].cpu().numpy(), ohm)
    ok = ~ties
    assert np.array_equal(dm[0].cpu().numpy()[ok], g['depth_map'][ok]) and np.array_equal(hm[0].cpu().numpy()[ok], g['height_map'][ok])
    n = 34000
    pts = np.stack([syn.hash_uniform(81, (6, n)) * 720 - 8, syn.hash_uniform(82, (6, n)) * 270 - 7,
                    syn.hash_uniform(83, (6, n)) * 60.0, syn.hash_signed(84, (6, n)) * 5.0], -1).astype(np.float32)
    for ds in (1, 2):
        dm, hm = points_to_maps(T(pts, gpu), 256, 704, ds)
        for c in range(6):
            odm, ohm, _ = O.points_to_maps(pts[c], 256, 704, ds)
            assert np.array_equal(dm[c].cpu().numpy(), odm) and np.array_equal(hm[c].cpu().numpy(), ohm), (ds, c)
    dm, hm = points_to_maps(T(pts[:, :0], gpu), 256, 704, 1)   # no points: all-zero maps
    assert float(dm.abs().sum()) == 0.0 and float(hm.abs().sum()) == 0.0


# --------------------------------------------------------------------------- DCN sampling (HeightNet / DepthNet, a12)

@pytest.mark.parametrize('c,groups,h,w,dil,scale', [(16, 4, 16, 44, 1, 0.5), (8, 1, 7, 9, 2, 3.0), (64, 4, 32, 88, 1, 8.0)])
def test_dcn_hip_sampling_vs_grid_sample_formulation(gpu, c, groups, h, w, dil, scale):
    """dhd_deform_im2col / dhd_deform_col2im against the grid_sample formulation of the same layer: output,
    input gradient, offset-branch gradients and weight gradient; large offsets put samples outside the image."""
    import copy
    from dhd_amd.depthnet import DCN
    torch.manual_seed(c + h)
    a = DCN(c, 2 * c, kernel_size=3, padding=dil, dilation=dil, groups=groups).to(gpu)
    with torch.no_grad():
        a.conv_offset.weight.normal_(0, 0.05 * scale)
        a.conv_offset.bias.normal_(0, scale)
    b = copy.deepcopy(a)
    b.use_hip = False
    x = torch.randn(3, c, h, w, device=gpu)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = a(xa), b(xb)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert (ya - yb).abs().max().item() < 1e-4 * max(1.0, yb.abs().max().item())
    assert (xa.grad - xb.grad).abs().max().item() < 2e-4 * max(1.0, xb.grad.abs().max().item())
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert (p.grad - q.grad).abs().max().item() < 5e-4 * max(1.0, q.grad.abs().max().item()), k


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('b,c,h,w,dil,scale', [(3, 16, 16, 44, 1, 0.5), (2, 10, 7, 9, 2, 3.0), (2, 12, 32, 88, 1, 8.0), (2, 8, 6, 10, 1, 40.0)])
def test_dcn_gather_col2im_vs_atomic_form_and_typed_columns(gpu, dtype, b, c, h, w, dil, scale):
    """ABI 5: dhd_deform_im2col_t / dhd_deform_col2im_t (gather form, columns in float32 / half) against the float32 LDS-atomic
    entry points of ABI 4 on the same inputs.  float32 columns: the same values up to summation order.  Half columns: im2col
    equals the rounded float32 columns bit for bit; the gradients equal those of the float32 path fed the same (half-valued)
    dcol.  Cases: the HeightNet map, dilation 2 with odd channel counts (tail chunk), the DHD-L map (one channel per block,
    101 KB of LDS for float32), offsets of +-40 pixels (most taps outside, many in the same border cells)."""
    from dhd_amd import _lib
    lib = _lib.load()
    k = 3
    torch.manual_seed(b * 100 + c)
    x = torch.randn(b, c, h, w, device=gpu)
    off = (scale * torch.randn(b, 2 * k * k, h, w, device=gpu)).contiguous()
    st = _lib.stream_ptr(gpu)
    code = _lib.dtype_code(dtype)
    col32 = torch.empty(b, c * k * k, h * w, device=gpu)
    _lib.check(lib.dhd_deform_im2col(_lib.ptr(x), _lib.ptr(off), _lib.ptr(col32), b, c, h, w, k, dil, dil, st), 'im2col')
    col = torch.empty(b, c * k * k, h * w, device=gpu, dtype=dtype)
    _lib.check(lib.dhd_deform_im2col_t(_lib.ptr(x), 0, _lib.ptr(off), _lib.ptr(col), code, b, c, h, w, k, dil, dil, st), 'im2col_t')
    assert torch.equal(col, col32.to(dtype))
    # x in the column type: the columns of the rounded x
    xh = x.to(dtype)
    colh = torch.empty_like(col)
    _lib.check(lib.dhd_deform_im2col_t(_lib.ptr(xh), code, _lib.ptr(off), _lib.ptr(colh), code, b, c, h, w, k, dil, dil, st), 'im2col_t nhwc')
    ref = torch.empty_like(col32)
    xh32 = xh.float()
    _lib.check(lib.dhd_deform_im2col(_lib.ptr(xh32), _lib.ptr(off), _lib.ptr(ref), b, c, h, w, k, dil, dil, st), 'im2col')
    assert torch.equal(colh, ref.to(dtype))
    dcol = torch.randn(b, c * k * k, h * w, device=gpu).to(dtype)
    dx0, doff0 = torch.empty_like(x), torch.empty_like(off)
    dcol32 = dcol.float().contiguous()          # (kept in variables: a temporary freed before the launch could be handed out again)
    _lib.check(lib.dhd_deform_col2im(_lib.ptr(dcol32), _lib.ptr(x), _lib.ptr(off), _lib.ptr(dx0), _lib.ptr(doff0), b, c, h, w,
                                     k, dil, dil, st), 'col2im')
    assert lib.dhd_deform_col2im_gather_supported(code, h, w, k)
    ws = torch.empty(lib.dhd_deform_col2im_workspace_bytes(b, h, w, k), dtype=torch.uint8, device=gpu)
    dx1, doff1 = torch.full_like(x, float('nan')), torch.full_like(off, float('nan'))
    _lib.check(lib.dhd_deform_col2im_t(_lib.ptr(dcol), code, _lib.ptr(x), 0, _lib.ptr(off), _lib.ptr(dx1), _lib.ptr(doff1), b, c, h, w, k, dil, dil,
                                       _lib.ptr(ws), ws.numel(), st), 'col2im_t')
    assert torch.isfinite(dx1).all() and torch.isfinite(doff1).all()
    assert (dx1 - dx0).abs().max().item() <= 2e-5 * max(1.0, dx0.abs().max().item())
    assert torch.equal(doff1, doff0)                 # same kernel, same order of operations
    # x / dx in the column type: dx = the float32 result rounded once, doffset from the rounded x
    dxh = torch.full_like(xh, float('nan'))
    doffh = torch.empty_like(off)
    _lib.check(lib.dhd_deform_col2im_t(_lib.ptr(dcol), code, _lib.ptr(xh), code, _lib.ptr(off), _lib.ptr(dxh), _lib.ptr(doffh), b, c, h, w, k, dil,
                                       dil, _lib.ptr(ws), ws.numel(), st), 'col2im_t nhwc')
    ulp = {torch.float32: 2e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert (dxh.float() - dx0).abs().max().item() <= ulp * max(1.0, dx0.abs().max().item())
    doff2 = torch.empty_like(off)
    _lib.check(lib.dhd_deform_col2im(_lib.ptr(dcol32), _lib.ptr(xh32), _lib.ptr(off), _lib.ptr(dx0), _lib.ptr(doff2),
                                     b, c, h, w, k, dil, dil, st), 'col2im')
    assert torch.equal(doffh, doff2)
    # too small a workspace is refused, nothing is launched
    assert lib.dhd_deform_col2im_t(_lib.ptr(dcol), code, _lib.ptr(x), 0, _lib.ptr(off), _lib.ptr(dx1), _lib.ptr(doff1), b, c, h, w, k, dil, dil,
                                   _lib.ptr(ws), ws.numel() - 1, st) == -1
    # a float16 x with bfloat16 columns (neither float32 nor the column type) is refused
    if dtype != torch.float32:
        other = 3 - code
        assert lib.dhd_deform_col2im_t(_lib.ptr(dcol), code, _lib.ptr(xh), other, _lib.ptr(off), _lib.ptr(dxh), _lib.ptr(doffh), b, c, h, w, k, dil,
                                       dil, _lib.ptr(ws), ws.numel(), st) == -1


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_dcn_module_under_autocast_uses_half_columns(gpu, dtype):
    """DCN.forward under autocast: the column matrix and its gradient are half (no float32 155 MB tensors), results within half
    precision of the float32 layer."""
    import copy
    from dhd_amd.depthnet import DCN
    torch.manual_seed(5)
    a = DCN(32, 32, kernel_size=3, padding=1, groups=4).to(gpu)
    with torch.no_grad():
        a.conv_offset.weight.normal_(0, 0.05)
        a.conv_offset.bias.normal_(0, 1.0)
    b = copy.deepcopy(a)
    x = torch.randn(4, 32, 16, 44, device=gpu)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    seen = []
    import dhd_amd.depthnet as dn
    orig = dn._DeformIm2col.forward

    def spy(ctx, *args):
        out = orig(ctx, *args)
        seen.append(out.dtype)
        return out
    dn._DeformIm2col.forward = staticmethod(spy)
    try:
        with torch.autocast('cuda', dtype=dtype):
            ya = a(xa)
    finally:
        dn._DeformIm2col.forward = staticmethod(orig)
    assert seen == [dtype] and ya.dtype == dtype
    yb = b(xb)
    g = torch.randn_like(yb)
    ya.backward(g.to(dtype))
    yb.backward(g)
    # the output is a plain half GEMM of half columns (2e-3 / 1.6e-2); the gradients also pass the offset branch, where a half
    # offset moves a sampling point by ~1e-3 px and the position derivative multiplies that by feature differences of order 1
    # (measured for fp16: output 1.9e-3, input gradient 4.5e-2)
    tol = 1e-2 if dtype == torch.float16 else 5e-2
    rel = lambda p, q: ((p.float() - q).norm() / q.norm()).item()
    errs = dict(y=rel(ya, yb), x_grad=rel(xa.grad, xb.grad), **{k: rel(p.grad, q.grad) for (k, p), q in zip(a.named_parameters(), b.parameters())})
    print(dtype, {k: round(v, 4) for k, v in errs.items()})
    assert errs['y'] < tol and all(v < 12 * tol for v in errs.values()), errs


def test_aspp_half_channels_last_two_images_does_not_crash(gpu):
    """MIOpen's half-precision training BatchNorm segfaults on an (N, C, 1, 1) map with channels_last strides at N = 2
    (experiments/bn_1x1_crash_probe.py): ASPP's global-pool branch hands its BatchNorm plain strides.  In a subprocess, so that a
    regression fails the test instead of killing the session."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\nfrom dhd_amd.depthnet import ASPP\n"
            "m = ASPP(64, 64).cuda().train().to(memory_format=torch.channels_last)\n"
            "x = torch.randn(2, 64, 4, 11, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_()\n"
            "with torch.autocast('cuda', dtype=torch.float16):\n    y = m(x)\n"
            "y.float().sum().backward(); torch.cuda.synchronize(); print('ok', tuple(y.shape))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok (2, 64, 4, 11)' in out.stdout, (out.returncode, out.stderr[-400:])


# --------------------------------------------------------------------------- softmax(depth), context, softmax(height), band: one launch (a11)

@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('bn,d,c,hb,extra,fh,fw', [(24, 44, 64, 65, 0, 16, 44), (5, 88, 32, 17, 3, 7, 9), (2, 9, 8, 65, 0, 1, 3)])
def test_depth_height_head_one_launch_vs_torch(gpu, dtype, layout, bn, d, c, hb, extra, fh, fw):
    """dhd_mghs_softmax_forward / _backward against the torch formulation of lss_heightmap.py:484-489 on the same tensors:
    the probabilities are torch's own softmax BIT FOR BIT (same operation order as aten's kernel for this shape), tran_feat is
    the float32 slice, band = dhd_height_band of those probabilities; gradients of x_d / the height logits within float32
    rounding of autograd's, in the inputs' dtype and layout."""
    from dhd_amd import mghs_op
    torch.manual_seed(bn + d)
    fmt = torch.channels_last if layout == 'channels_last' else torch.contiguous_format
    hr = [round(-1.0 + 0.1 * i, 1) for i in range(hb)]
    mr = [-1.0, hr[hb // 4], hr[hb // 2], hr[-1]]
    x_d = (3 * torch.randn(bn, d + c + extra, fh, fw, device=gpu)).to(dtype).contiguous(memory_format=fmt).requires_grad_()
    hl = (3 * torch.randn(bn, hb + extra, fh, fw, device=gpu)).to(dtype).contiguous(memory_format=fmt)
    hl[:, 3] = hl[:, 7]                       # exact ties between two bins
    hl = hl.requires_grad_()
    depth, feat, height, band = mghs_op.depth_height_head(x_d, hl, d, c, hr, mr)
    assert depth.dtype == feat.dtype == height.dtype == torch.float32 and band.dtype == torch.uint8
    assert depth.is_contiguous() and feat.is_contiguous() and height.is_contiguous()
    xr, hlr = x_d.detach().clone().requires_grad_(), hl.detach().clone().requires_grad_()
    depth_r = xr[:, :d].float().softmax(dim=1)
    feat_r = xr[:, d:d + c].float()
    height_r = hlr[:, :hb].float().softmax(dim=1)
    bit_equal = torch.equal(depth, depth_r) and torch.equal(height, height_r)
    print('softmax bit-identical to torch:', bit_equal)
    assert (depth - depth_r).abs().max().item() <= 1e-6 and (height - height_r).abs().max().item() <= 1e-6
    if fh * fw > 64:       # aten's one-thread-per-pixel form (every DHD configuration: 16x44, 32x88); smaller maps reduce in a block
        assert bit_equal
    assert torch.equal(feat, feat_r)
    assert torch.equal(band, mghs_op.height_band(height, hr, mr))
    gd, gf, gh = torch.randn_like(depth), torch.randn_like(feat), torch.randn_like(height)
    torch.autograd.backward([depth, feat, height], [gd, gf, gh])
    torch.autograd.backward([depth_r, feat_r, height_r], [gd, gf, gh])
    assert x_d.grad.dtype == dtype and x_d.grad.is_contiguous(memory_format=fmt) and hl.grad.is_contiguous(memory_format=fmt)
    eps = {torch.float32: 1e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    for a_, r_ in ((x_d.grad, xr.grad), (hl.grad, hlr.grad)):
        assert (a_.float() - r_.float()).abs().max().item() <= eps * max(1.0, r_.float().abs().max().item())
    if extra:
        assert float(x_d.grad[:, d + c:].abs().max()) == 0.0 and float(hl.grad[:, hb:].abs().max()) == 0.0
    # no height branch; only one of the gradients requested
    d2, f2, h2, b2 = mghs_op.depth_height_head(x_d.detach().requires_grad_(), None, d, c)
    assert h2 is None and b2 is None and torch.equal(d2, depth) and torch.equal(f2, feat)


# --------------------------------------------------------------------------- stereo cost volume (DHD-M / DHD-L DepthNet)

@pytest.mark.parametrize('bn,c,h,w,d,bias', [(2, 16, 6, 10, 8, 5.0), (3, 256, 16, 44, 88, 5.0), (1, 64, 9, 13, 70, 0.0),
                                             (2, 128, 9, 13, 71, 0.7), (1, 132, 6, 9, 65, 0.3), (1, 520, 5, 6, 9, 0.5)])
@pytest.mark.parametrize('walk', [False, True])
def test_stereo_cost_volume_vs_grid_sample_formulation(gpu, bn, c, h, w, d, bias, walk):
    """dhd_stereo_cost_volume against the reference's formulation (C/4 grid_sample calls + |diff| sums + bias where the
    last group's first channel sampled 0 + softmax over depth, depthnet.py:307-361), incl. samples outside the image.
    Random sampling positions (every tap a fresh load) and `walk`: hypothesis k sits 0.37 k pixels along a line from the pixel
    itself, leaving the image for the later ones (the kernel keeps the previous hypothesis' taps in registers and reuses them
    by index).  Channel counts: the pair mode (c <= 128: two hypotheses per step) and 1 / 3 channel groups per lane."""
    from dhd_amd.depthnet import DepthNet
    torch.manual_seed(bn + c)
    dn = DepthNet(32, 32, 16, d, use_dcn=False, aspp_mid_channels=16, stereo=True, bias=bias).to(gpu)
    prev, curr = torch.randn(bn, c, h, w, device=gpu), torch.randn(bn, c, h, w, device=gpu)
    if walk:
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, h, device=gpu), torch.linspace(-1, 1, w, device=gpu), indexing='ij')
        k = torch.arange(d, device=gpu, dtype=torch.float32).view(1, d, 1, 1)
        sign = torch.tensor([1.0, -1.0, 0.5], device=gpu)[:bn].view(bn, 1, 1, 1) if bn <= 3 else 1.0
        gx = xs.view(1, 1, h, w) + sign * k * 0.37 * 2 / (w - 1)
        gy = ys.view(1, 1, h, w) - k * 0.21 * 2 / (h - 1)
        grid = torch.stack([gx.expand(bn, d, h, w), gy.expand(bn, d, h, w)], -1).reshape(bn, d * h, w, 2).contiguous()
    else:
        grid = torch.rand(bn, d * h, w, 2, device=gpu) * 2.6 - 1.3   # ~20 % of the samples fall outside
    grid[0, :w] = -2.0                                               # the "behind the camera" marker of gen_grid
    got = dn._hip_cost_volume(prev, curr, grid, d, (c // 4 - 1) * 4)
    cost, warped = 0, None
    for f in range(c // 4):
        warped = torch.nn.functional.grid_sample(prev[:, f * 4:(f + 1) * 4], grid, align_corners=True, padding_mode='zeros')
        cost = cost + (curr[:, f * 4:(f + 1) * 4].unsqueeze(2) - warped.view(bn, -1, d, h, w)).abs().sum(dim=1)
    if bias != 0:
        cost = torch.where(warped[:, 0].view(bn, d, h, w) == 0, cost + bias, cost)
    ref = (-cost).softmax(dim=1)
    assert got.shape == ref.shape and abs(float(got.sum()) - bn * h * w) < 1e-2 * bn * h * w * 1e-2 + 1e-1
    assert (got - ref).abs().max().item() < 1e-4  # softmax of sums of up to 256 |diff| terms (cost ~ 300) in another order


def test_stereo_cost_volume_vs_reference_golden(gpu):
    """gen_grid on the GPU + dhd_stereo_cost_volume against golden G8 (the reference's DepthNet.gen_grid /
    calculate_cost_volumn run on CPU): 70 % of the samples inside the adjacent image, the rest zero-padded and biased."""
    from dhd_amd.depthnet import DepthNet
    g = golden('g8_stereo')
    d, h, w = g['frustum'].shape[:3]
    bn = g['curr'].shape[0]
    dn = DepthNet(32, 32, 16, d, use_dcn=False, aspp_mid_channels=16, stereo=True, bias=float(g['bias'])).to(gpu)
    metas = dict(k2s_sensor=T(g['k2s_sensor'], gpu), intrins=T(g['intrins'], gpu), post_rots=T(g['post_rots'], gpu),
                 post_trans=T(g['post_trans'], gpu), frustum=T(g['frustum'], gpu), cv_feat_list=[T(g['prev'], gpu), T(g['curr'], gpu)])
    assert dn.use_hip_cost_volume
    cv = dn.calculate_cost_volumn(metas)
    np.testing.assert_allclose(cv.cpu().numpy(), g['cost_volume'], rtol=2e-4, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# weight EMA (csrc/ema.hip) -- bit-exact against the reference fixture and the oracle
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_ema_hook_on_gpu_is_bit_identical_to_reference_golden(gpu):
    import dhd_amd
    from test_host_logic import _Runner, ema_fixture_net, ema_fixture_step
    g = golden('g9_ema')
    runner = _Runner(ema_fixture_net(g, gpu))
    hook = dhd_amd.build_hook(dict(type='MEGVIIEMAHook', init_updates=10560, priority='NORMAL'))
    hook.before_run(runner)
    for it in range(3):
        ema_fixture_step(runner.model.module, it)
        hook.after_train_iter(runner)
        for k, v in runner.ema_model.ema.state_dict().items():
            assert np.array_equal(v.cpu().numpy(), g[f'ema{it}.{k}']), (it, k)


@pytest.mark.gpu
def test_ema_update_ragged_state_vs_oracle(gpu):
    """Tensors of 1 ... 200 001 values (several chunks, ragged tails), an integer buffer and an empty
    parameter; a second update after load_state_dict; then an unaligned chunk through the C ABI."""
    from oracle import mghs_oracle as O
    from dhd_amd import _lib
    from dhd_amd.ema import CHUNK, ModelEMA
    sizes = [1, 3, 64, 1023, CHUNK, CHUNK + 1, 200001, 0]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(syn.hash_signed(300 + i, (n,)))) for i, n in enumerate(sizes)])
            self.register_buffer('steps', torch.zeros((), dtype=torch.long))
            self.register_buffer('running', torch.from_numpy(syn.hash_signed(299, (5, 7))))
    net = Net().to(gpu)
    ema = ModelEMA(net, decay=0.999, updates=100)
    want = {k: v.cpu().numpy().copy() for k, v in ema.ema.state_dict().items()}
    for it in range(2):
        with torch.no_grad():
            for i, p in enumerate(net.p):
                p.add_(torch.from_numpy(syn.hash_signed(400 + 10 * it + i, (sizes[i],))).to(gpu))
            net.steps += 1
        if it == 1:
            ema.ema.load_state_dict(ema.ema.state_dict())   # in-place copy: the chunk table stays valid
        ema.update(None, net)
        d = O.ema_decay(0.999, 100 + it + 1)
        msd = net.state_dict()
        for k, v in ema.ema.state_dict().items():
            if v.dtype.is_floating_point:
                want[k] = O.ema_update(want[k], msd[k].cpu().numpy(), d)
            assert np.array_equal(v.cpu().numpy(), want[k]), (it, k)
    assert int(ema.ema.steps) == 0
    # chunk starts 4 bytes past a 16-byte boundary take the scalar path: same values
    e = torch.from_numpy(syn.hash_signed(500, (1001,))).to(gpu)
    m = torch.from_numpy(syn.hash_signed(501, (1001,))).to(gpu)
    ref = O.ema_update(e.cpu().numpy()[1:], m.cpu().numpy()[1:], 0.75)
    first = float(e[0])
    ea = torch.tensor([e.data_ptr() + 4], dtype=torch.int64, device=gpu)
    ma = torch.tensor([m.data_ptr() + 4], dtype=torch.int64, device=gpu)
    ln = torch.tensor([1000], dtype=torch.int32, device=gpu)
    _lib.check(_lib.load().dhd_ema_update(_lib.ptr(ea), _lib.ptr(ma), _lib.ptr(ln), 1, 0.75, 0.25, _lib.stream_ptr(gpu)), 'ema')
    assert np.array_equal(e.cpu().numpy()[1:], ref) and float(e[0]) == first
    with pytest.raises(_lib.DhdError):
        half = Net().to(gpu).half()
        ModelEMA(half).update(None, half)


# ---------------------------------------------------------------------------------------------
# training-mode BatchNorm2d of the dense callers (csrc/batchnorm.hip) vs torch.nn.BatchNorm2d
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize('shape', [(6, 64, 32, 88), (3, 130, 16, 44), (2, 8, 200, 200), (1, 5, 4, 4), (24, 16, 8, 22)])
def test_batchnorm2d_training_vs_torch(gpu, dtype, tol, shape):
    """y, running statistics, num_batches_tracked, and the gradients of x / weight / bias against torch's own
    BatchNorm2d on float32 copies of the same (rounded) inputs; second step with momentum=None (cumulative average)."""
    from dhd_amd.batchnorm import BatchNorm2d
    BatchNorm2d = type('AlwaysHipBN', (BatchNorm2d,), dict(_routing=(0, 1 << 30, 0)))  # no size threshold
    torch.manual_seed(sum(shape))
    n, c, h, w = shape
    for momentum in (0.1, None):
        ours = BatchNorm2d(c, momentum=momentum).to(gpu).train()
        ref = torch.nn.BatchNorm2d(c, momentum=momentum).to(gpu).train()
        with torch.no_grad():
            ours.weight.copy_(torch.rand(c) + 0.5); ours.bias.copy_(torch.randn(c))
            ours.running_mean.copy_(torch.randn(c)); ours.running_var.copy_(torch.rand(c) + 0.5)
        ref.load_state_dict(ours.state_dict())
        for step in range(2):
            x = (torch.randn(shape, device=gpu) * 1.7 + 3.0).to(dtype).requires_grad_()
            g = torch.randn(shape, device=gpu).to(dtype)
            xr = x.detach().float().requires_grad_()
            assert ours._hip_ok(x) == ((h * w) % (4 if dtype == torch.float32 else 8) == 0)
            y = ours(x)
            y.backward(g)
            yr = ref(xr)
            yr.backward(g.float())
            assert y.dtype == dtype and x.grad.dtype == dtype
            assert (y.float() - yr).abs().max() <= tol * max(1.0, float(yr.abs().max()))
            assert (x.grad.float() - xr.grad).abs().max() <= tol * max(1.0, float(xr.grad.abs().max()))
            for a, b in ((ours.weight.grad, ref.weight.grad), (ours.bias.grad, ref.bias.grad)):
                assert (a - b).abs().max() <= tol * max(1.0, float(b.abs().max())) * (1 if dtype == torch.float32 else 4)
            assert torch.allclose(ours.running_mean, ref.running_mean, atol=1e-5, rtol=1e-5)
            assert torch.allclose(ours.running_var, ref.running_var, atol=1e-5, rtol=1e-4)
            assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
            ours.zero_grad(); ref.zero_grad()
    ours.eval()
    with torch.no_grad():     # eval mode is the parent's path
        x = torch.randn(shape, device=gpu).to(dtype)
        assert torch.equal(ours(x), torch.nn.functional.batch_norm(x, ours.running_mean, ours.running_var, ours.weight, ours.bias, False, 0.0, ours.eps))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize('mode', ['plain', 'relu', 'add'])
@pytest.mark.parametrize('shape', [(6, 64, 32, 88), (3, 8, 5, 7), (2, 2048, 4, 6), (2, 4096, 3, 3), (2, 24, 6, 10), (1, 136, 40, 40)])
def test_batchnorm2d_channels_last_fused_vs_torch(gpu, dtype, tol, mode, shape):
    """The channels_last kernels (include/dhd_amd.h 9b) with the ReLU / residual + ReLU that follows the normalisation fused in,
    against torch's BatchNorm2d + add + relu on float32 copies of the same (rounded) inputs: output, running statistics, and the
    gradients of x, the residual, weight and bias.  Shapes: 8 vectors per row (32 row lanes), one vector (odd row count), 256 and
    512 vectors (two column groups), 3 vectors (row lanes do not fill the workgroup), 17 vectors."""
    from dhd_amd.batchnorm import BatchNorm2d
    torch.manual_seed(sum(shape))
    n, c, h, w = shape
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    ours = BatchNorm2d(c).to(gpu).train()
    ref = torch.nn.BatchNorm2d(c).to(gpu).train()
    with torch.no_grad():
        ours.weight.copy_(torch.rand(c) + 0.5); ours.bias.copy_(torch.randn(c) * 0.5)
        ours.running_mean.copy_(torch.randn(c)); ours.running_var.copy_(torch.rand(c) + 0.5)
    ref.load_state_dict(ours.state_dict())
    for step in range(2):
        x = cl((torch.randn(shape, device=gpu) * 1.7 + 0.4).to(dtype)).requires_grad_()
        res = cl(torch.randn(shape, device=gpu).to(dtype)).requires_grad_() if mode == 'add' else None
        g = cl(torch.randn(shape, device=gpu).to(dtype))
        assert ours._nhwc_ok(x) == (c % (4 if dtype == torch.float32 else 8) == 0)
        y = ours(x, relu=mode == 'relu', residual=res)
        assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
        y.backward(g)
        xr = x.detach().float().requires_grad_()
        rr = res.detach().float().requires_grad_() if res is not None else None
        pre = ref(xr) if rr is None else ref(xr) + rr
        scale = max(1.0, float(pre.abs().max()))
        if mode == 'plain':
            yr = pre
        else:
            # an element whose pre-activation is within rounding of zero may land on either side; the reference takes the
            # operator's own decision there (as autograd would from the stored half output) and checks it everywhere else
            mask = y.detach().float() > 0
            sure = pre.detach().abs() > 2 * tol * scale
            assert (mask == (pre.detach() > 0))[sure].all()
            yr = pre * mask
        yr.backward(g.float())
        assert (y.float() - yr).abs().max() <= tol * scale
        assert (x.grad.float() - xr.grad).abs().max() <= tol * max(1.0, float(xr.grad.abs().max())) * 2
        if res is not None:
            assert (res.grad.float() - rr.grad).abs().max() <= tol * max(1.0, float(rr.grad.abs().max()))
        for a, b in ((ours.weight.grad, ref.weight.grad), (ours.bias.grad, ref.bias.grad)):
            assert (a - b).abs().max() <= tol * max(1.0, float(b.abs().max())) * (1 if dtype == torch.float32 else 4)
        assert torch.allclose(ours.running_mean, ref.running_mean, atol=1e-5, rtol=1e-5)
        assert torch.allclose(ours.running_var, ref.running_var, atol=1e-5, rtol=1e-4)
        assert int(ours.num_batches_tracked) == step + 1
        ours.zero_grad(); ref.zero_grad()


@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
@pytest.mark.parametrize('mode', ['relu', 'add'])
def test_batchnorm2d_fused_relu_propagates_nan_like_torch(gpu, mode, layout):
    """A NaN input makes the channel's batch statistics NaN; torch.relu (and so the reference's BN -> ReLU) hands the NaN on, so
    it surfaces as a NaN loss.  The fused epilogue must do the same (fmaxf(NaN, 0) = 0 would turn the channel into
    zeros)."""
    from dhd_amd.batchnorm import BatchNorm2d
    torch.manual_seed(3)
    shape = (2, 16, 6, 10)
    fmt = torch.channels_last if layout == 'channels_last' else torch.contiguous_format
    ours = BatchNorm2d(16).to(gpu).train()
    ref = torch.nn.BatchNorm2d(16).to(gpu).train()
    x = torch.randn(shape, device=gpu)
    x[1, 5, 2, 3] = float('nan')
    x[0, 9, 0, 0] = float('inf')
    x = x.contiguous(memory_format=fmt)
    res = torch.randn(shape, device=gpu).contiguous(memory_format=fmt) if mode == 'add' else None
    y = ours(x, relu=mode == 'relu', residual=res)
    pre = ref(x) if res is None else ref(x) + res
    yr = torch.relu(pre)
    # the NaN channel is NaN everywhere, as in torch.  The Inf channel: torch's kernels give NaN at the Inf and 0 elsewhere
    # ((x - inf) * rstd = -inf -> relu -> 0); here the shifted sums (x - x[first]) make the channel's statistics NaN and the whole
    # channel comes out NaN -- a superset of torch's NaNs, never a finite value that torch does not give
    nan_y, nan_r = torch.isnan(y), torch.isnan(yr)
    assert bool((nan_r <= nan_y).all()) and torch.isnan(y[:, 5]).all() and torch.isnan(y[0, 9, 0, 0])
    keep = [ch for ch in range(16) if ch not in (5, 9)]
    assert not nan_y[:, keep].any() and torch.equal(nan_y[:, 5], nan_r[:, 5])
    assert (y[:, keep] - yr[:, keep]).abs().max() < 1e-5


# ---------------------------------------------------------------------------------------------
# bilinear up-sampling, align_corners=True (csrc/upsample.hip) vs torch's float64 interpolate
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('dtype,eps', [(torch.float32, 2.0 ** -23), (torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
@pytest.mark.parametrize('shape,size', [((2, 16, 25, 25), (100, 100)), ((2, 8, 9, 13), (18, 26)), ((1, 8, 7, 5), (15, 16)), ((3, 24, 1, 6), (4, 6)),
                                        ((2, 8, 6, 6), (6, 6)), ((1, 40, 50, 50), (100, 100)), ((1, 8, 3, 4), (24, 32))])
def test_bilinear_upsample_forward_backward_vs_float64(gpu, dtype, eps, layout, shape, size):
    """Forward: every output within one rounding of the float64 interpolation of the same (rounded) input.  Backward: the gather
    equals the float64 transpose of that interpolation (autograd through torch's float64 kernel) -- float32 accumulation, one
    rounding: a few units of the type's epsilon relative to the largest gradient -- and two runs give identical bits (no atomics).
    Shapes: x4 (FPN_LSS), x2 (UNet, FPN_LSS.up2), a non-integer ratio, one input row (scale 0), identity, x8."""
    from dhd_amd.detector import Upsample
    torch.manual_seed(sum(shape) + sum(size))
    x = torch.randn(shape, device=gpu).to(dtype)
    g = torch.randn(shape[:2] + size, device=gpu).to(dtype)
    if layout == 'channels_last':
        x, g = x.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last)
    x.requires_grad_()
    up = Upsample(size=size, mode='bilinear', align_corners=True)
    y = up(x)
    assert y.dtype == dtype and tuple(y.shape) == shape[:2] + size
    assert y.is_contiguous(memory_format=torch.channels_last if layout == 'channels_last' else torch.contiguous_format)
    y.backward(g)
    xr = x.detach().double().requires_grad_()
    yr = torch.nn.functional.interpolate(xr, size=size, mode='bilinear', align_corners=True)
    yr.backward(g.double())
    # the source index is computed in float32 as torch's kernels do: lambda carries ~6e-8 * size of error against float64
    assert ((y.double() - yr).abs() <= (0.5 * eps + 2e-5) * yr.abs().clamp_min(1.0) + 1e-30).all()
    gscale = float(xr.grad.abs().max())
    assert float((x.grad.double() - xr.grad).abs().max()) <= (eps + 4e-5) * gscale
    assert x.grad.is_contiguous(memory_format=torch.channels_last if layout == 'channels_last' else torch.contiguous_format)
    g1 = x.grad.clone()
    x.grad = None
    up(x).backward(g)
    assert torch.equal(g1, x.grad)
    if dtype == torch.float32 and layout == 'nchw':      # and against torch's own float32 kernel: the same index arithmetic
        yt = torch.nn.functional.interpolate(x.detach(), size=size, mode='bilinear', align_corners=True)
        assert torch.allclose(y, yt, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16, torch.int32])
@pytest.mark.parametrize('shape', [(2, 64, 20, 28), (3, 7, 5, 9), (1, 130, 65, 3), (2, 1, 8, 8), (4, 256, 1, 1), (2, 512, 100, 100)])
def test_layout_conversion_is_a_pure_copy_in_both_directions(gpu, dtype, shape):
    """dhd_amd.layout.to_layout (csrc/layout.hip): the same values as torch's `.contiguous(memory_format=...)` bit for bit, in the
    requested format, NCHW -> channels_last -> NCHW; the gradient comes back in the producer's layout; partial tiles, one channel,
    one pixel (both formats at once: returned as is)."""
    from dhd_amd.layout import to_layout
    torch.manual_seed(sum(shape))
    x = (torch.randn(shape, device=gpu) * 100).to(dtype)
    a = to_layout(x, torch.channels_last)
    assert a.is_contiguous(memory_format=torch.channels_last) and torch.equal(a, x)
    assert a.contiguous(memory_format=torch.channels_last).data_ptr() == a.data_ptr()
    b = to_layout(a, torch.contiguous_format)
    assert b.is_contiguous() and torch.equal(b, x)
    if shape[1] == 1 or shape[2:] == (1, 1):
        assert a is x and b is x
    if dtype.is_floating_point:
        for src, dst in ((torch.contiguous_format, torch.channels_last), (torch.channels_last, torch.contiguous_format)):
            t = x.detach().clone(memory_format=torch.contiguous_format).contiguous(memory_format=src).detach().requires_grad_()
            y = to_layout(t, dst)
            g = torch.randn_like(x).contiguous(memory_format=dst)
            (y * g).sum().backward()
            assert torch.equal(t.grad, g)
            if not (shape[1] == 1 or shape[2:] == (1, 1)):
                assert t.grad.is_contiguous(memory_format=src)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,W,C,ws,sh', [(2, 14, 21, 32, 7, 3), (1, 16, 23, 64, 7, 0), (3, 7, 7, 8, 7, 3), (2, 32, 88, 128, 7, 3), (1, 5, 9, 16, 4, 2)])
def test_swin_window_rows_equal_pad_roll_partition(gpu, dtype, B, H, W, C, ws, sh):
    """csrc/window.hip against the reference's sequence (swin.py:448-513): F.pad -> torch.roll(-shift) -> window partition, and
    window reverse -> torch.roll(+shift) -> crop; values bit for bit, gradients too (each direction is the other's transpose);
    with a float32 input and a bfloat16 output the values are the cast of the same rows."""
    import torch.nn.functional as F
    from dhd_amd.swin import _WindowRows
    torch.manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, device=gpu).to(dtype).requires_grad_()
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    Hp, Wp = H + pad_b, W + pad_r
    nh, nw = Hp // ws, Wp // ws

    def ref_partition(t):
        t = F.pad(t, (0, 0, 0, pad_r, 0, pad_b))
        if sh:
            t = torch.roll(t, shifts=(-sh, -sh), dims=(1, 2))
        return t.view(B, nh, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nh * nw, ws * ws, C)

    def ref_reverse(win):
        t = win.view(B, nh, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if sh:
            t = torch.roll(t, shifts=(sh, sh), dims=(1, 2))
        return t[:, :H, :W].contiguous()

    win = _WindowRows.apply(x, H, W, ws, sh, False, dtype)
    xr = x.detach().clone().requires_grad_()
    win_r = ref_partition(xr)
    assert torch.equal(win, win_r)
    gw = torch.randn_like(win)
    win.backward(gw)
    win_r.backward(gw)
    assert torch.equal(x.grad, xr.grad)
    w2 = torch.randn(B, nh * nw, ws * ws, C, device=gpu).to(dtype).requires_grad_()
    w2r = w2.detach().clone().requires_grad_()
    y, yr = _WindowRows.apply(w2, H, W, ws, sh, True, dtype), ref_reverse(w2r)
    assert torch.equal(y, yr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yr.backward(gy)
    assert torch.equal(w2.grad, w2r.grad)          # the padding rows get zeros
    if dtype == torch.float32:                      # the autocast form: float32 rows out in bfloat16, the gradient back in float32
        x2 = x.detach().clone().requires_grad_()
        wb = _WindowRows.apply(x2, H, W, ws, sh, False, torch.bfloat16)
        assert wb.dtype == torch.bfloat16 and torch.equal(wb, win_r.detach().bfloat16())
        wb.backward(gw.bfloat16())
        assert x2.grad.dtype == torch.float32 and torch.equal(x2.grad, ref_reverse(gw.bfloat16().float()))


# ---------------------------------------------------------------------------------------------
# SFA stage under nn.SyncBatchNorm (DHD-L.py:308-311 SyncbnControlHook): the phased operator, two ranks sharing cuda:0 over gloo
# ---------------------------------------------------------------------------------------------

def _syncbn_stage_worker(rank, world, port, q, gemm, sizes=(2, 2), half=False):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from dhd_amd.mix import channel_spatial_stage, fused_stage_supported, needs_cross_rank_statistics
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    c, h, w = 128, 20, 28
    lo, hi = sum(sizes[:rank]), sum(sizes[:rank + 1])        # this rank's samples (the ranks may hold different numbers)
    torch.manual_seed(5)
    st = channel_spatial_stage(2 * c)
    with torch.no_grad():
        for bn in (st.spacial_leanring[1], st.spacial_leanring[4]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    st = torch.nn.SyncBatchNorm.convert_sync_batchnorm(st).to(dev).train()
    st.gemm = gemm
    g = torch.Generator().manual_seed(77)
    x_all = torch.randn(sum(sizes), 2 * c, h, w, generator=g) * 0.7 + 0.1
    w_all = torch.randn(sum(sizes), c, h, w, generator=g)
    x = x_all[lo:hi].to(dev)
    if half:                                       # a caller inside an autocast region: half x, half out / gradients (io_dtype)
        x = x.half()
        st.half_storage = half == 'storage'        # True: ABI 3 form (half edges, float32 inside); 'storage': half storage (ABI 4)
    x.requires_grad_()
    assert needs_cross_rank_statistics(st) and fused_stage_supported(st, x)
    out = st(x)                                    # dhd_sfa_stage_forward_phase x 3, two all-reduces of 2C + 1 doubles
    assert out.dtype == x.dtype
    out.backward(w_all[lo:hi].to(dev).to(out.dtype))
    assert x.grad.dtype == x.dtype
    # plain numpy data: tensors would travel as file descriptors of this process, which is gone by the time the parent reads
    res = dict(out=out.detach().float().cpu().numpy(), gx=x.grad.float().cpu().numpy(), grads={k: p.grad.cpu().numpy() for k, p in st.named_parameters()},
               buffers={k: v.cpu().numpy() for k, v in st.named_buffers()})
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('gemm,sizes,half', [('bf16x6', (2, 2), False), ('bf16x3', (2, 2), False), ('bf16x6', (3, 1), False),
                                             ('bf16x6', (2, 2), True), ('bf16x6', (2, 2), 'storage'), ('bf16x6', (3, 1), 'storage')])
def test_fused_sfa_stage_under_syncbatchnorm_two_ranks(gpu, gemm, sizes, half):
    """core/hook/syncbncontrol.py:18-32 converts every BatchNorm at epoch 0 of DHD-L.py (:308-311), the stage's two included.
    The fused operator then runs cut at its statistics points (dhd_sfa_stage_forward_phase / backward_phase) with the
    (2C + 1) float64 sums all-reduced in between.  Two ranks (sharing the GPU over gloo) with two samples each -- or
    three and one: the counts travel with the sums -- must reproduce plain PyTorch with ordinary BatchNorm on the four samples in one process -- which is what SyncBatchNorm
    means: outputs and input gradients per rank, parameter gradients as the sum of the ranks' contributions (the loss is a
    sum over samples), running statistics identical on both ranks and equal to the full-batch ones."""
    import socket
    import torch.multiprocessing as mp
    from dhd_amd.mix import channel_spatial_stage
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_stage_worker, args=(r, world, port, q, gemm, sizes, half)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the reference: ordinary modules, the whole batch, float64
    c, h, w = 128, 20, 28
    torch.manual_seed(5)
    ref = channel_spatial_stage(2 * c)
    with torch.no_grad():
        for bn in (ref.spacial_leanring[1], ref.spacial_leanring[4]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    sp = ref.spacial_leanring
    plain = torch.nn.Sequential(sp[0], torch.nn.BatchNorm2d(c), torch.nn.ReLU(), sp[3], torch.nn.BatchNorm2d(c))
    for i in (1, 4):
        plain[i].load_state_dict(sp[i].state_dict())
    ref.spacial_leanring = plain
    ref = ref.double().train()
    g = torch.Generator().manual_seed(77)
    x_all = torch.randn(sum(sizes), 2 * c, h, w, generator=g) * 0.7 + 0.1
    w_all = torch.randn(sum(sizes), c, h, w, generator=g)
    if half:    # the operator sees the rounded input and output gradient; its own results are rounded once more on the way out
        x_all, w_all = x_all.half(), w_all.half()
    x_all, w_all = x_all.double().requires_grad_(), w_all.double()
    hr = 2.0 ** -10 if half else 0.0          # half rounding of the returned tensors (relative)
    a = ref.fc(x_all.mean(dim=(2, 3)))[..., None, None]
    xb, xv = x_all[:, :c], x_all[:, c:]
    gate = torch.sigmoid(ref.spacial_leanring(a * xb + (1 - a) * xv))
    out = gate * (a * xb) + (1 - gate) * ((1 - a) * xv)
    (out * w_all).sum().backward()
    f = GEMM_MODES[gemm]
    if half == 'storage':
        # half storage: every tensor of the stage rounded to fp16 once -- relative L2 bounds at that level (the gradients behind
        # the ReLU carry the flipped pre-activations, see test_sfa_stage_half_storage_is_no_less_accurate_than_autocast)
        rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)
        for r in range(world):
            sl = slice(sum(sizes[:r]), sum(sizes[:r + 1]))
            assert rel(res[r]['out'], out[sl].detach().numpy()) < 2.0 ** -9
            assert rel(res[r]['gx'], x_all.grad[sl].numpy()) < 16 * 2.0 ** -10
        for k, p in ref.named_parameters():
            got = sum(res[r]['grads'][k].astype(np.float64) for r in range(world))
            want = p.grad.numpy()
            if np.linalg.norm(want) > 1e-6:
                assert rel(got, want) < 6e-2, (k, rel(got, want))
            else:
                assert np.abs(got).max() < 2e-2, k      # the convolution biases in front of a BatchNorm: zero as a sum over ranks
        for k, v in ref.named_buffers():
            if 'running' in k:
                for r in range(world):
                    np.testing.assert_allclose(res[r]['buffers'][k], v.numpy(), atol=2e-4, rtol=2e-3, err_msg=k)
                assert np.array_equal(res[0]['buffers'][k], res[1]['buffers'][k])
        assert np.abs(res[0]['grads']['spacial_leanring.0.bias']).max() > 1e-4
        return
    for r in range(world):
        sl = slice(sum(sizes[:r]), sum(sizes[:r + 1]))
        np.testing.assert_allclose(res[r]['out'], out[sl].detach().numpy(), atol=2e-5 * min(f, 5.0), rtol=1e-4 + hr)
        gref = x_all.grad[sl].numpy()
        np.testing.assert_allclose(res[r]['gx'], gref, atol=1e-4 * f * np.abs(gref).max(), rtol=1e-3 + hr)
    names = {k: k for k, _ in ref.named_parameters()}
    for k, p in ref.named_parameters():
        got = sum(res[r]['grads'][k].astype(np.float64) for r in range(world))
        want = p.grad.numpy()
        np.testing.assert_allclose(got, want, atol=3e-4 * f * max(1.0, np.abs(want).max()), rtol=1e-3, err_msg=k)
    for k, v in ref.named_buffers():
        if 'running' in k:
            for r in range(world):
                np.testing.assert_allclose(res[r]['buffers'][k], v.numpy(), atol=1e-5, rtol=1e-5, err_msg=k)
            assert np.array_equal(res[0]['buffers'][k], res[1]['buffers'][k])
    # the convolution-bias gradients vanish only as a SUM over ranks: each rank's own contribution is non-zero
    assert np.abs(res[0]['grads']['spacial_leanring.0.bias']).max() > 1e-4


def test_trace_ranges_on_the_gpu(gpu):
    """DHD_AMD_TRACE / dhd_amd.trace.enable(): the operators run inside roctx ranges (torch.cuda.nvtx on ROCm) and give the same
    results as without."""
    from dhd_amd import bev_pool_v2, trace
    depth = torch.rand(1, 1, 2, 2, 2, device=gpu).requires_grad_()
    feat = torch.randn(1, 1, 2, 2, 64, device=gpu).requires_grad_()
    I = lambda v: torch.tensor(v, device=gpu).int()
    args = (I([0, 4, 1, 6]), I([0, 0, 1, 2]), I([5, 5, 31, 31]), (1, 1, 4, 8, 64), I([0, 2]), I([2, 2]))
    was = trace.enabled()
    res = []
    try:
        for on in (False, True):
            trace.enable(on)
            depth.grad = feat.grad = None
            out = bev_pool_v2(depth, feat, *args, fused=True)
            out.sum().backward()
            res.append((out.detach().clone(), depth.grad.clone(), feat.grad.clone()))
    finally:
        trace.enable(was)
    for a, b in zip(*res):
        assert torch.equal(a, b)
