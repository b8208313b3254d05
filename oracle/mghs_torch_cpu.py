"""torch-CPU twin of the reference's view-transform op sequence -- TEST INFRASTRUCTURE / CPU BASELINE ONLY
(same rule as oracle/mghs_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it).

Why it exists.  The reference has no CPU implementation of `bev_pool_v2` (CUDA only: ops/bev_pool_v2/src/bev_pool.cpp:7-14,
53-56), so there is nothing of the reference's to time on host cores.  SURVEY.md 8(d) / BASELINE.md 2.3 therefore define
the CPU baseline as "the build's own torch-CPU restatement of the reference's op sequence", multi-threaded
(`torch.set_num_threads(os.cpu_count())`).  This file is that restatement: per forward it runs, like the reference,

    4 x get_ego_coor                      models/necks/lss_heightmap.py:179-231   (batched 3x3 matmuls over every point)
    4 x voxel_pooling_prepare_v2          :303-371   (truncation, bounds filter, rank, argsort, run lengths)
    3 x masked copies of tran_feat        :436-442
    4 x bev_pool_v2 + permute + cat       :261-300, ops/bev_pool_v2/bev_pool.py:86-106 (kernel = segmented sum of
                                          depth x feat per voxel, bev_pool_cuda.cu:21-50; here `index_add_`)

and the backward pass is autograd's through the same graph (gather of the output gradient per point: what
bev_pool_cuda.cu:69-123 computes).  `sfa_stage` is mix.py:37-59 on the given torch modules.  The numpy oracle
(mghs_oracle.py) stays the bit-level checker; this twin is held to it and to the reference fixtures in
tests/test_oracle_golden.py (indices equal, values to float32 rounding).  cpu_baseline.kind = "port".
"""
import torch

FULL_GRID = {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4]}  # hard-coded in view_transform, :425-430


def frustum(depth_cfg, input_size, downsample):
    """create_frustum (:105-134) -> (D, fH, fW, 3) of (u, v, d)."""
    h_in, w_in = input_size
    fh, fw = h_in // downsample, w_in // downsample
    d = torch.arange(*depth_cfg, dtype=torch.float).view(-1, 1, 1).expand(-1, fh, fw)
    n_d = d.shape[0]
    u = torch.linspace(0, w_in - 1, fw, dtype=torch.float).view(1, 1, fw).expand(n_d, fh, fw)
    v = torch.linspace(0, h_in - 1, fh, dtype=torch.float).view(1, fh, 1).expand(n_d, fh, fw)
    return torch.stack((u, v, d), -1)


def get_ego_coor(fr, sensor2ego, intrin, post_rots, post_trans, bda, inv_post_rot=None, combine=None):
    """(:206-230).  The two optional matrices replace torch.inverse(post_rots) and rot @ inverse(intrin) (used by the
    tests to inject the reference's own matrices; the timed path computes them)."""
    b, n = sensor2ego.shape[:2]
    pts = fr.to(sensor2ego) - post_trans.view(b, n, 1, 1, 1, 3)
    ipr = torch.inverse(post_rots) if inv_post_rot is None else inv_post_rot
    pts = ipr.view(b, n, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
    pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
    comb = sensor2ego[:, :, :3, :3].matmul(torch.inverse(intrin)) if combine is None else combine
    pts = comb.view(b, n, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1)
    pts = pts + sensor2ego[:, :, :3, 3].view(b, n, 1, 1, 1, 3)
    return bda.view(b, 1, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1)).squeeze(-1)


def grid_infos(x, y, z, **_):
    """create_grid_infos (:86-102): float tensors."""
    return (torch.Tensor([c[0] for c in (x, y, z)]), torch.Tensor([c[2] for c in (x, y, z)]),
            torch.Tensor([(c[1] - c[0]) / c[2] for c in (x, y, z)]))


def prepare_v2(coor, grid):
    """voxel_pooling_prepare_v2 (:303-371) -> ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths
    (None x 5 when nothing is kept)."""
    lower, interval, size = grid_infos(**grid)
    b, n, d, h, w, _ = coor.shape
    num = b * n * d * h * w
    ranks_depth = torch.arange(num, dtype=torch.int)
    ranks_feat = torch.arange(num // d, dtype=torch.int).reshape(b, n, 1, h, w).expand(b, n, d, h, w).flatten()
    idx = ((coor - lower) / interval).long().view(num, 3)
    batch_idx = torch.arange(b).reshape(b, 1).expand(b, num // b).reshape(num, 1)
    idx = torch.cat((idx, batch_idx), 1)
    kept = ((idx[:, 0] >= 0) & (idx[:, 0] < size[0]) & (idx[:, 1] >= 0) & (idx[:, 1] < size[1])
            & (idx[:, 2] >= 0) & (idx[:, 2] < size[2]))
    idx, ranks_depth, ranks_feat = idx[kept], ranks_depth[kept], ranks_feat[kept]
    if idx.shape[0] == 0:
        return None, None, None, None, None
    nx, ny, nz = int(size[0]), int(size[1]), int(size[2])
    ranks_bev = idx[:, 3] * (nz * ny * nx) + idx[:, 2] * (ny * nx) + idx[:, 1] * nx + idx[:, 0]
    order = ranks_bev.argsort()
    ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
    first = torch.ones(ranks_bev.shape[0], dtype=torch.bool)
    first[1:] = ranks_bev[1:] != ranks_bev[:-1]
    starts = torch.where(first)[0].int()
    lengths = torch.zeros_like(starts)
    lengths[:-1] = starts[1:] - starts[:-1]
    lengths[-1] = ranks_bev.shape[0] - starts[-1]
    return ranks_bev.int(), ranks_depth.int(), ranks_feat.int(), starts, lengths


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, shape):
    """bev_pool.py:86-106 with the kernel's arithmetic (bev_pool_cuda.cu:21-50) as gather, multiply, index_add_:
    out (B,Dz,Dy,Dx,C) zero-initialised, then permuted to (B,C,Dz,Dy,Dx) and made contiguous (:105)."""
    b, dz, dy, dx, c = shape
    val = depth.reshape(-1)[ranks_depth.long()][:, None] * feat.reshape(-1, c)[ranks_feat.long()]
    out = feat.new_zeros(b * dz * dy * dx, c).index_add(0, ranks_bev.long(), val)
    return out.view(b, dz, dy, dx, c).permute(0, 4, 1, 2, 3).contiguous()


def voxel_pooling_v2(coor, depth, feat, grid, collapse_z=True):
    """(:261-300).  depth (B,N,D,fH,fW), feat (B,N,C,fH,fW)."""
    _, _, size = grid_infos(**grid)
    nx, ny, nz = int(size[0]), int(size[1]), int(size[2])
    rb, rd, rf, _, _ = prepare_v2(coor, grid)
    if rb is None:
        out = torch.zeros(feat.shape[0], feat.shape[2], nz, ny, nx)
    else:
        out = bev_pool_v2(depth, feat.permute(0, 1, 3, 4, 2), rd, rf, rb, (depth.shape[0], nz, ny, nx, feat.shape[2]))
    return torch.cat(out.unbind(dim=2), 1) if collapse_z else out


def view_transform(cfg, fr, calib, depth, tran_feat, height, inv_post_rot=None, combine=None):
    """MGHS.view_transform (:407-459) -> [bev, low, mid, high]; geometry and index preparation run once per grid,
    as in the reference.  calib = (sensor2ego, ego2global, intrin, post_rot, post_tran, bda) tensors; depth
    (B*N,D,fH,fW), tran_feat (B*N,C,fH,fW), height (B*N,H,fH,fW)."""
    s2e, _, intrin, post_rot, post_tran, bda = calib
    b, n = s2e.shape[:2]
    d, fh, fw = depth.shape[1:]
    c = tran_feat.shape[1]
    hmap = torch.tensor(cfg['height_range'])[torch.argmax(height, dim=1)]               # :528-543
    h_min, t1, t2, h_max = cfg['mask_range']
    masks = ((hmap >= h_min) & (hmap < t1), (hmap >= t1) & (hmap < t2), (hmap >= t2) & (hmap < h_max))   # :545-564
    outs = []
    for k, grid in enumerate((FULL_GRID, cfg['mask_1_grid'], cfg['mask_2_grid'], cfg['mask_3_grid'])):
        feat = tran_feat if k == 0 else tran_feat * masks[k - 1].unsqueeze(1).expand_as(tran_feat)
        coor = get_ego_coor(fr, s2e, intrin, post_rot, post_tran, bda, inv_post_rot, combine)
        outs.append(voxel_pooling_v2(coor, depth.view(b, n, d, fh, fw), feat.view(b, n, c, fh, fw),
                                     {a: grid[a] for a in 'xyz'}, cfg.get('collapse_z', True)))
    return outs


def sfa_stage(stage, x):
    """channel_spatial_stage.forward (mix.py:37-59) on the module's own torch layers (fc, spacial_leanring)."""
    c = x.shape[1] // 2
    xb, xv = torch.split(x, c, dim=1)
    a1 = stage.fc(x.mean(-1).mean(-1))[:, :, None, None]
    xb1, xv1 = a1 * xb, (1 - a1) * xv
    a2 = torch.sigmoid(stage.spacial_leanring(xb1 + xv1))
    return a2 * xb1 + (1 - a2) * xv1
