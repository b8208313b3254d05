"""CPU oracle for the DHD height-decoupled view transform (MGHS) -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm.  It is the checker
for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Nothing under dhd_amd/ imports it, and the
product path has no CPU fallback.

Parity status: PINNED.  tests/golden/make_golden.py imports the reference's own
lss_heightmap.py on CPU (under small stubs for mmcv/mmdet3d) and records its
outputs; tests/test_oracle_golden.py checks every function here against those
fixtures (indices bit-exact, values <= 1e-6) and against the reference's only
known-answer test (ops/bev_pool_v2/bev_pool.py:163-194).

One boundary is *not* bit-pinned: torch.inverse on CPU is MKL sgetrf/sgetrs,
whose small-matrix kernels use approximate reciprocals and cannot be restated.
`inv3x3` below is the published LAPACK algorithm (sgetf2 partial pivoting +
strsm substitution) with IEEE division.  Fixtures therefore also store the
reference's own inverse/"combine" matrices so that the per-point arithmetic is
pinned bit-for-bit given identical matrices, and the end-to-end difference from
the inverse is measured (a handful of boundary points per 185 856).

Also restated here, each pinned the same way: the SFA attention stage (`sfa_stage`, golden G5 from the
reference's mix.py), the height loss and its label builders (G4), the occupancy-head losses (`occ_losses`,
G6 from models/losses/semkitti_loss.py), the evaluation histogram (`occ_confusion`, G14 from the reference's Metric_mIoU) and the
LiDAR rasteriser (`points_to_maps`, G7 from datasets/pipelines/loading_new.py; equal keys of the reference's
unstable argsort are identified as ties) and the weight EMA (`ema_decay` / `ema_update`, G9 from
core/hook/ema.py's ModelEMA, bit-exact).  The stereo sampling grid and cost volume are pinned directly against golden G8 (the reference's
DepthNet.gen_grid / calculate_cost_volumn) without a numpy restatement, and so is the Swin backbone mirror (golden G10,
models/backbones/swin.py).  Unpinned (no reference fixture can be
produced here): mmcv's DCN -- oracle/dcn_oracle.py restates its published algorithm (pins the algorithm, not mmcv's bits).
Round 2 fixtures: G5b (SFA at C = 128, the fused operator's shape: `sfa_stage` forward in eval and train mode and its
float64 backward), G11 (`mghs_depth_view_transform`, the z-stacked DHD-M/L variant), G3 at B = 4.

All file:line citations are into /root/reference/projects/mmdet3d_plugin/.
"""
import numpy as np

f32 = np.float32
f64 = np.float64


# --------------------------------------------------------------------------- #
# small-matrix helpers (arithmetic order pinned against torch CPU bmm)
# --------------------------------------------------------------------------- #

def inv3x3(a):
    """Inverse of one 3x3 float32 matrix: LU with partial pivoting, then forward/back
    substitution on the permuted identity (torch.inverse -> linalg.solve(A, I) ->
    LAPACK getrf + getrs; call sites models/necks/lss_heightmap.py:209,220).
    Every operation is a separately rounded float32 op; division is IEEE."""
    a = np.array(a, dtype=f32).copy()
    n = 3
    perm = [0, 1, 2]
    for j in range(n):
        p = j
        for i in range(j + 1, n):
            if abs(a[i, j]) > abs(a[p, j]):
                p = i
        if p != j:
            a[[j, p]] = a[[p, j]]
            perm[j], perm[p] = perm[p], perm[j]
        for i in range(j + 1, n):
            a[i, j] = f32(a[i, j] / a[j, j])
            for k in range(j + 1, n):
                a[i, k] = f32(a[i, k] - f32(a[i, j] * a[j, k]))
    x = np.zeros((n, n), f32)
    for c in range(n):
        b = np.array([1.0 if perm[i] == c else 0.0 for i in range(n)], f32)
        for i in range(n):
            for k in range(i):
                b[i] = f32(b[i] - f32(a[i, k] * b[k]))
        for i in reversed(range(n)):
            for k in range(i + 1, n):
                b[i] = f32(b[i] - f32(a[i, k] * b[k]))
            b[i] = f32(b[i] / a[i, i])
        x[:, c] = b
    return x


def matmul3(a, b):
    """3x3 @ 3x3 the way torch CPU bmm does it for tiny matrices: acc = 0; acc += a_ik*b_kj
    for k = 0,1,2, each product and each sum rounded to float32 (no FMA)."""
    out = np.zeros((3, 3), f32)
    for i in range(3):
        for j in range(3):
            acc = f32(0)
            for k in range(3):
                acc = f32(acc + f32(a[i, k] * b[k, j]))
            out[i, j] = acc
    return out


def _matvec(m, p):
    """(3,3) @ (...,3) with the same sequential, unfused float32 accumulation."""
    cols = []
    for i in range(3):
        acc = np.zeros(p.shape[:-1], f32)
        for k in range(3):
            acc = (acc + (m[i, k] * p[..., k]).astype(f32)).astype(f32)
        cols.append(acc)
    return np.stack(cols, -1)


# --------------------------------------------------------------------------- #
# a1 / a2: frustum template and grid infos
# --------------------------------------------------------------------------- #

def _linspace_f32(start, end, steps):
    """torch.linspace(float32) on CPU: step=(end-start)/(steps-1); first half start+step*i,
    second half end-step*(steps-1-i), each a fused multiply-add (pinned in golden G0)."""
    start, end = f32(start), f32(end)
    if steps == 1:
        return np.array([start], f32)
    step = f32((end - start) / f32(steps - 1))
    out = np.zeros(steps, f32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = f32(f64(start) + f64(step) * f64(i))
        else:
            out[i] = f32(f64(end) - f64(step) * f64(steps - 1 - i))
    return out


def frustum_axes(depth_cfg, input_size, downsample):
    """MGHS.create_frustum (lss_heightmap.py:105-134), sid=False: the (D,fH,fW,3) template is
    the outer product of three axes, returned separately: u (fW,), v (fH,), d (D,)."""
    h_in, w_in = input_size
    fh, fw = h_in // downsample, w_in // downsample
    d = np.arange(depth_cfg[0], depth_cfg[1], depth_cfg[2], dtype=f64).astype(f32)
    # torch.arange(*cfg, dtype=float) computes start + i*step in double then casts
    n = int(np.ceil((depth_cfg[1] - depth_cfg[0]) / depth_cfg[2]))
    d = np.array([f32(depth_cfg[0] + i * depth_cfg[2]) for i in range(n)], f32)
    u = _linspace_f32(0, w_in - 1, fw)
    v = _linspace_f32(0, h_in - 1, fh)
    return u, v, d


def grid_infos(x, y, z, **_):
    """MGHS.create_grid_infos (lss_heightmap.py:86-102): python-double arithmetic, then float32."""
    lower = np.array([c[0] for c in (x, y, z)], f64).astype(f32)
    interval = np.array([c[2] for c in (x, y, z)], f64).astype(f32)
    size = np.array([(c[1] - c[0]) / c[2] for c in (x, y, z)], f64).astype(f32)
    return lower, interval, size


FULL_GRID = {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4]}  # hard-coded, :425-430


# --------------------------------------------------------------------------- #
# a3: frustum -> ego coordinates
# --------------------------------------------------------------------------- #

def camera_matrices(sensor2ego, intrin, post_rot):
    """Per-camera inv(post_rot) and combine = R_s2e @ inv(K) (lss_heightmap.py:209,220)."""
    b, n = sensor2ego.shape[:2]
    ipr = np.zeros((b, n, 3, 3), f32)
    comb = np.zeros((b, n, 3, 3), f32)
    for i in range(b):
        for j in range(n):
            ipr[i, j] = inv3x3(post_rot[i, j])
            comb[i, j] = matmul3(sensor2ego[i, j, :3, :3].astype(f32), inv3x3(intrin[i, j]))
    return ipr, comb


def ego_coor(axes, sensor2ego, intrin, post_rot, post_tran, bda, inv_post_rot=None, combine=None):
    """MGHS.get_ego_coor (lss_heightmap.py:179-231) -> (B,N,D,fH,fW,3) float32.
    Op order: p = frustum - post_tran; p = inv(post_rot) p; p = (px*pz, py*pz, pz);
    p = combine p; p += t; p = bda p.  No FMA anywhere (pinned against torch CPU)."""
    u, v, d = axes
    b, n = sensor2ego.shape[:2]
    dd, fh, fw = len(d), len(v), len(u)
    if inv_post_rot is None or combine is None:
        ipr, comb = camera_matrices(sensor2ego, intrin, post_rot)
        inv_post_rot = ipr if inv_post_rot is None else inv_post_rot
        combine = comb if combine is None else combine
    fr = np.zeros((dd, fh, fw, 3), f32)
    fr[..., 0] = u[None, None, :]
    fr[..., 1] = v[None, :, None]
    fr[..., 2] = d[:, None, None]
    out = np.zeros((b, n, dd, fh, fw, 3), f32)
    for i in range(b):
        for j in range(n):
            p = (fr - post_tran[i, j].astype(f32)).astype(f32)
            p = _matvec(inv_post_rot[i, j].astype(f32), p)
            p = np.stack([(p[..., 0] * p[..., 2]).astype(f32), (p[..., 1] * p[..., 2]).astype(f32), p[..., 2]], -1)
            p = _matvec(combine[i, j].astype(f32), p)
            p = (p + sensor2ego[i, j, :3, 3].astype(f32)).astype(f32)
            out[i, j] = _matvec(bda[i].astype(f32), p)
    return out


# --------------------------------------------------------------------------- #
# a4: voxel index / sort / intervals
# --------------------------------------------------------------------------- #

def voxel_rank(coor, grid):
    """Per-point voxel rank or -1 (lss_heightmap.py:329-354).  idx = trunc((p-lower)/interval)
    toward zero, kept iff 0 <= idx < size on every axis; rank = b*DzDyDx + z*DyDx + y*Dx + x."""
    lower, interval, size = grid_infos(**grid)
    b = coor.shape[0]
    pts = coor.reshape(b, -1, 3)
    q = ((pts - lower).astype(f32) / interval).astype(f32)
    idx = np.trunc(q).astype(np.int64)
    kept = np.ones(idx.shape[:2], bool)
    for a in range(3):
        kept &= (idx[..., a] >= 0) & (idx[..., a].astype(f32) < size[a])
    dx, dy, dz = int(size[0]), int(size[1]), int(size[2])
    rank = (np.arange(b, dtype=np.int64)[:, None] * (dz * dy * dx) + idx[..., 2] * (dy * dx)
            + idx[..., 1] * dx + idx[..., 0])
    return np.where(kept, rank, -1).astype(np.int32).reshape(-1)


def prepare_v2(coor, grid):
    """MGHS.voxel_pooling_prepare_v2 (lss_heightmap.py:303-371) in canonical form: the reference's
    argsort (:355) is unstable, so the within-voxel order is unspecified; here it is a stable sort,
    i.e. ascending point id inside each voxel.  Returns int32 arrays
    (ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths) or five Nones."""
    b, n, d, h, w, _ = coor.shape
    rank = voxel_rank(coor, grid)
    pid = np.arange(rank.size, dtype=np.int64)
    pix = (pid // (d * h * w)) * (h * w) + pid % (h * w)  # (b*N+n)*fH*fW + h*fW + w  (:322-327)
    kept = rank >= 0
    if not kept.any():
        return None, None, None, None, None
    order = np.argsort(rank[kept], kind='stable')
    rb = rank[kept][order]
    rd = pid[kept][order]
    rf = pix[kept][order]
    first = np.ones(rb.size, bool)
    first[1:] = rb[1:] != rb[:-1]
    starts = np.nonzero(first)[0]
    lengths = np.diff(np.append(starts, rb.size))
    i32 = np.int32
    return rb.astype(i32), rd.astype(i32), rf.astype(i32), starts.astype(i32), lengths.astype(i32)


# --------------------------------------------------------------------------- #
# a6-a8: the bev_pool_v2 operator
# --------------------------------------------------------------------------- #

def bev_pool_v2_forward(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                        interval_starts, interval_lengths):
    """bev_pool_v2_kernel (ops/bev_pool_v2/src/bev_pool_cuda.cu:21-50): for each interval,
    out[ranks_bev[start], c] = sum_i feat[ranks_feat[start+i], c] * depth[ranks_depth[start+i]],
    summed sequentially in float32.  Output (B,Dz,Dy,Dx,C), zero elsewhere (bev_pool.py:27)."""
    c = feat.shape[-1]
    out = np.zeros((int(np.prod(bev_feat_shape[:-1])), c), f32)
    if len(interval_starts) == 0:
        return out.reshape(bev_feat_shape)
    prod = (feat.reshape(-1, c)[ranks_feat] * depth.reshape(-1)[ranks_depth][:, None]).astype(f32)
    sums = _segment_sum(prod, interval_starts, interval_lengths)
    out[ranks_bev[interval_starts]] = sums
    return out.reshape(bev_feat_shape)


def _segment_sum(rows, starts, lengths):
    """Sequential float32 sum of rows[start:start+len] per segment."""
    out = np.zeros((len(starts), rows.shape[1]), f32)
    maxlen = int(lengths.max())
    for i in range(maxlen):
        m = lengths > i
        out[m] = (out[m] + rows[starts[m] + i]).astype(f32)
    return out


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
    """bev_pool_v2 (bev_pool.py:86-106): operator + permute to (B,C,Dz,Dy,Dx)."""
    x = bev_pool_v2_forward(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                            interval_starts, interval_lengths)
    return np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))


def bev_pool_v2_backward(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev):
    """QuickCumsumCuda.backward + bev_pool_grad_kernel (bev_pool.py:44-83; bev_pool_cuda.cu:69-123).
    out_grad is (B,Dz,Dy,Dx,C).  depth_grad[p] = <out_grad[voxel(p)], feat[pixel(p)]>;
    feat_grad[pixel] = sum_p out_grad[voxel(p)] * depth[p] over the pixel's points."""
    c = feat.shape[-1]
    g = out_grad.reshape(-1, c)[ranks_bev]
    f = feat.reshape(-1, c)[ranks_feat]
    depth_grad = np.zeros(depth.size, f32)
    dg = np.zeros(len(ranks_bev), f32)
    for ch in range(c):  # sequential over channels, as the kernel's inner loop (:99-103)
        dg = (dg + (g[:, ch] * f[:, ch]).astype(f32)).astype(f32)
    depth_grad[ranks_depth] = dg
    feat_grad = np.zeros((feat.size // c, c), f32)
    order = np.argsort(ranks_feat, kind='stable')
    rf = ranks_feat[order]
    first = np.ones(rf.size, bool)
    first[1:] = rf[1:] != rf[:-1]
    starts = np.nonzero(first)[0]
    lengths = np.diff(np.append(starts, rf.size))
    contrib = (g[order] * depth.reshape(-1)[ranks_depth[order]][:, None]).astype(f32)
    feat_grad[rf[starts]] = _segment_sum(contrib, starts, lengths)
    return depth_grad.reshape(depth.shape), feat_grad.reshape(feat.shape)


# --------------------------------------------------------------------------- #
# a9: height map -> band masks
# --------------------------------------------------------------------------- #

def band_index(height_idx, height_range, mask_range):
    """height_feature_to_height_map + create_mask_3 (lss_heightmap.py:528-564) as a per-pixel band
    id: 0 = [h_min,thr1), 1 = [thr1,thr2), 2 = [thr2,h_max), 255 = in no band.  Comparisons are
    made on float32 values exactly as torch does (python scalars against a float32 tensor)."""
    hr = np.array(height_range, f64).astype(f32)
    h = hr[height_idx.astype(np.int64)]
    h_min, t1, t2, h_max = [f32(v) for v in mask_range]
    band = np.full(h.shape, 255, np.uint8)
    band[(h >= h_min) & (h < t1)] = 0
    band[(h >= t1) & (h < t2)] = 1
    band[(h >= t2) & (h < h_max)] = 2
    return band


# --------------------------------------------------------------------------- #
# a5 / a10: voxel_pooling_v2 and view_transform, forward and backward
# --------------------------------------------------------------------------- #

def voxel_pooling_v2(coor, depth, feat_nchw, grid, collapse_z=True):
    """MGHS.voxel_pooling_v2 (lss_heightmap.py:261-300).  depth (B,N,D,fH,fW), feat (B,N,C,fH,fW)."""
    b, n, c = feat_nchw.shape[:3]
    _, _, size = grid_infos(**grid)
    dx, dy, dz = int(size[0]), int(size[1]), int(size[2])
    rb, rd, rf, st, ln = prepare_v2(coor, grid)
    if rb is None:
        out = np.zeros((b, c, dz, dy, dx), f32)
    else:
        feat = np.ascontiguousarray(feat_nchw.transpose(0, 1, 3, 4, 2))
        out = bev_pool_v2(depth, feat, rd, rf, rb, (b, dz, dy, dx, c), st, ln)
    if collapse_z:
        out = np.concatenate([out[:, :, z] for z in range(dz)], axis=1)  # channel = z*C + c (:299)
    return out


def view_transform(cfg, calib, depth, tran_feat, height_idx, inv_post_rot=None, combine=None,
                   collapse_z=True):
    """MGHS.view_transform (lss_heightmap.py:407-459).

    cfg: dict with grid_config['depth'], input_size, downsample, height_range, mask_range,
    mask_{1,2,3}_grid.  calib: (sensor2ego, ego2global, intrin, post_rot, post_tran, bda).
    depth (B*N,D,fH,fW), tran_feat (B*N,C,fH,fW), height_idx (B*N,fH,fW) = argmax of the
    height distribution.  Returns [bev, low, mid, high] each (B, C*Dz, Dy, Dx)."""
    s2e, _, intrin, post_rot, post_tran, bda = calib
    b, n = s2e.shape[:2]
    axes = frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    coor = ego_coor(axes, s2e, intrin, post_rot, post_tran, bda, inv_post_rot, combine)
    dd, fh, fw = coor.shape[2:5]
    c = tran_feat.shape[1]
    dep = depth.reshape(b, n, dd, fh, fw)
    band = band_index(height_idx, cfg['height_range'], cfg['mask_range'])
    outs = [voxel_pooling_v2(coor, dep, tran_feat.reshape(b, n, c, fh, fw), FULL_GRID, collapse_z)]
    for k, name in enumerate(('mask_1_grid', 'mask_2_grid', 'mask_3_grid')):
        masked = (tran_feat * (band == k)[:, None].astype(f32)).astype(f32)  # :436-442
        g = {a: cfg[name][a] for a in 'xyz'}
        outs.append(voxel_pooling_v2(coor, dep, masked.reshape(b, n, c, fh, fw), g, collapse_z))
    return outs


def view_transform_backward(cfg, calib, depth, tran_feat, height_idx, out_grads,
                            inv_post_rot=None, combine=None):
    """Gradients of sum_k <out_k, out_grads[k]> w.r.t. depth and tran_feat (masks carry no
    gradient: argmax/bool, lss_heightmap.py:434-442).  out_grads are (B, C*Dz, Dy, Dx)."""
    s2e, _, intrin, post_rot, post_tran, bda = calib
    b, n = s2e.shape[:2]
    axes = frustum_axes(cfg['grid_config']['depth'], cfg['input_size'], cfg['downsample'])
    coor = ego_coor(axes, s2e, intrin, post_rot, post_tran, bda, inv_post_rot, combine)
    dd, fh, fw = coor.shape[2:5]
    c = tran_feat.shape[1]
    dep = depth.reshape(b, n, dd, fh, fw)
    band = band_index(height_idx, cfg['height_range'], cfg['mask_range'])
    dgrad = np.zeros(dep.shape, f32)
    fgrad = np.zeros((b * n, c, fh, fw), f32)
    grids = [FULL_GRID] + [{a: cfg[k][a] for a in 'xyz'} for k in ('mask_1_grid', 'mask_2_grid', 'mask_3_grid')]
    for k, grid in enumerate(grids):
        _, _, size = grid_infos(**grid)
        dx, dy, dz = int(size[0]), int(size[1]), int(size[2])
        rb, rd, rf, st, ln = prepare_v2(coor, grid)
        if rb is None:
            continue
        mask = np.ones((b * n, 1, fh, fw), f32) if k == 0 else (band == k - 1)[:, None].astype(f32)
        feat = np.ascontiguousarray((tran_feat * mask).astype(f32).reshape(b, n, c, fh, fw).transpose(0, 1, 3, 4, 2))
        og = out_grads[k].reshape(b, dz, c, dy, dx).transpose(0, 1, 3, 4, 2)  # (B,Dz,Dy,Dx,C)
        dg, fg = bev_pool_v2_backward(np.ascontiguousarray(og), dep, feat, rd, rf, rb)
        dgrad = (dgrad + dg).astype(f32)
        fg = fg.reshape(b * n, fh, fw, c).transpose(0, 3, 1, 2)
        fgrad = (fgrad + fg * mask).astype(f32)
    return dgrad.reshape(depth.shape), fgrad


# --------------------------------------------------------------------------- #
# a13: get_mlp_input
# --------------------------------------------------------------------------- #

def get_mlp_input(sensor2ego, ego2global, intrin, post_rot, post_tran, bda):
    """MGHS.get_mlp_input (lss_heightmap.py:493-526) -> (B,N,27)."""
    b, n = sensor2ego.shape[:2]
    bd = np.broadcast_to(bda[:, None], (b, n, 3, 3))
    cols = [intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2], intrin[:, :, 1, 2],
            post_rot[:, :, 0, 0], post_rot[:, :, 0, 1], post_tran[:, :, 0],
            post_rot[:, :, 1, 0], post_rot[:, :, 1, 1], post_tran[:, :, 1],
            bd[:, :, 0, 0], bd[:, :, 0, 1], bd[:, :, 1, 0], bd[:, :, 1, 1], bd[:, :, 2, 2]]
    a = np.stack(cols, -1)
    return np.concatenate([a, sensor2ego[:, :, :3, :].reshape(b, n, 12)], -1).astype(f32)


# --------------------------------------------------------------------------- #
# a16: height loss labels
# --------------------------------------------------------------------------- #

def _min_pool_nonzero(gt, ds):
    """16x16 min-pool that ignores zeros (lss_heightmap.py:632-647,677-690)."""
    b, n, h, w = gt.shape
    t = np.where(gt == 0.0, f32(1e5), gt).astype(f32)
    t = t.reshape(b * n, h // ds, ds, w // ds, ds).transpose(0, 1, 3, 2, 4).reshape(b * n, h // ds, w // ds, ds * ds)
    return t.min(-1)


def downsampled_gt_depth(gt_depth, ds, depth_cfg, n_depth):
    """MGHS.get_downsampled_gt_depth (lss_heightmap.py:625-667), sid=False.  depth_cfg is the
    grid_config['depth'] *current at call time* (view_transform leaves it at mask_3_grid's
    [1, 45, 0.5] while D stays 44 -- reproduced)."""
    g = _min_pool_nonzero(gt_depth, ds)
    # python-double scalars applied to a float32 tensor: each op rounds to float32
    g = ((g - f32(depth_cfg[0] - depth_cfg[2])).astype(f32) / f32(depth_cfg[2])).astype(f32)
    g = np.where((g < n_depth + 1) & (g >= 0.0), g, f32(0)).astype(f32)
    lab = np.trunc(g).astype(np.int64).reshape(-1)
    oh = np.zeros((lab.size, n_depth + 1), f32)
    oh[np.arange(lab.size), lab] = 1
    return oh[:, 1:]


def downsampled_gt_height(gt_height, ds, height_range, height_interval):
    """MGHS.get_downsampled_gt_height (lss_heightmap.py:670-701)."""
    nh = len(height_range)
    g = _min_pool_nonzero(gt_height, ds)
    g = ((g - f32(height_range[0])).astype(f32) / f32(height_interval)).astype(f32)
    g = np.where((g < nh + 1) & (g >= 0.0), g, f32(0)).astype(f32)
    lab = np.trunc(g).astype(np.int64).reshape(-1)
    oh = np.zeros((lab.size, nh + 1), f32)
    oh[np.arange(lab.size), lab] = 1
    return oh[:, 1:]


def height_loss(gt_depth, gt_height, height, ds, depth_cfg, n_depth, height_range, height_interval, weight):
    """MGHS.get_height_loss (lss_heightmap.py:595-622): BCE(sum) over foreground pixels / max(1, n_fg).
    F.binary_cross_entropy clamps log at -100."""
    hl = downsampled_gt_height(gt_height, ds, height_range, height_interval)
    dl = downsampled_gt_depth(gt_depth, ds, depth_cfg, n_depth)
    fg = dl.max(1) > 0.0
    nh = len(height_range)
    pred = height.transpose(0, 2, 3, 1).reshape(-1, nh)[fg].astype(f64)
    lab = hl[fg].astype(f64)
    with np.errstate(divide='ignore'):
        lp = np.maximum(np.log(pred), -100.0)
        l1p = np.maximum(np.log1p(-pred), -100.0)
    loss = -(lab * lp + (1 - lab) * l1p).sum() / max(1.0, float(fg.sum()))
    return weight * loss


# --------------------------------------------------------------------------- #
# a14: SFA channel/spatial attention stage (models/necks/mix.py:37-59)
# --------------------------------------------------------------------------- #

def sfa_stage(x, fc1_w, fc1_b, fc2_w, fc2_b, conv1_w, conv1_b, bn1, conv2_w, conv2_b, bn2, eps=1e-5, training=False,
              out_grad=None):
    """channel_spatial_stage.forward (mix.py:37-59) in float64.
    x (B,2C,H,W); bn = (gamma, beta, running_mean, running_var).  training=True: BatchNorm normalises with the
    batch statistics (biased variance), as nn.BatchNorm2d does in train mode.
    With out_grad (B,C,H,W) also returns the gradients of <out, out_grad>: (out, dx, dict of the 12 parameter
    gradients keyed fc1_w, fc1_b, fc2_w, fc2_b, conv1_w, conv1_b, bn1_w, bn1_b, conv2_w, conv2_b, bn2_w, bn2_b)."""
    b, c2, h, w = x.shape
    c = c2 // 2
    X = x.astype(f64)
    xb, xv = X[:, :c], X[:, c:]
    s = X.mean(-1).mean(-1)                                                   # mix.py:41
    z1 = s @ fc1_w.T.astype(f64) + fc1_b
    r1 = np.maximum(z1, 0)
    z2 = r1 @ fc2_w.T.astype(f64) + fc2_b
    a1 = (1 / (1 + np.exp(-z2)))[:, :, None, None]                            # mix.py:43-44
    xb1 = a1 * xb
    xv1 = (1 - a1) * xv
    u = xb1 + xv1                                                             # mix.py:49
    W1 = conv1_w.reshape(c, c).astype(f64)
    W2 = conv2_w.reshape(c, c).astype(f64)

    def bn_fwd(t, p):
        g, be, m, v = [q.astype(f64)[None, :, None, None] for q in p]
        if training:
            m = t.mean((0, 2, 3), keepdims=True)
            v = t.var((0, 2, 3), keepdims=True)
        rstd = 1 / np.sqrt(v + eps)
        xh = (t - m) * rstd
        return xh * g + be, xh, rstd

    y1 = np.einsum('oc,bchw->bohw', W1, u) + conv1_b[None, :, None, None]
    n1, xh1, rstd1 = bn_fwd(y1, bn1)
    t1 = np.maximum(n1, 0)
    y2 = np.einsum('oc,bchw->bohw', W2, t1) + conv2_b[None, :, None, None]
    n2, xh2, rstd2 = bn_fwd(y2, bn2)
    a2 = 1 / (1 + np.exp(-n2))                                                # mix.py:51-53
    out = a2 * xb1 + (1 - a2) * xv1                                           # mix.py:55-58
    if out_grad is None:
        return out.astype(f32)

    def bn_bwd(gn, xh, rstd, gamma):
        gw, gb = (gn * xh).sum((0, 2, 3)), gn.sum((0, 2, 3))
        gxh = gn * gamma.astype(f64)[None, :, None, None]
        if training:
            m = b * h * w
            gy = rstd * (gxh - gxh.mean((0, 2, 3), keepdims=True) - xh * (gxh * xh).sum((0, 2, 3), keepdims=True) / m)
        else:
            gy = gxh * rstd
        return gy, gw, gb

    go = out_grad.astype(f64)
    ga2 = go * (xb1 - xv1)
    gxb1 = go * a2
    gxv1 = go * (1 - a2)
    gn2 = ga2 * a2 * (1 - a2)
    gy2, g_bn2w, g_bn2b = bn_bwd(gn2, xh2, rstd2, bn2[0])
    g_w2 = np.einsum('bohw,bchw->oc', gy2, t1)
    g_b2 = gy2.sum((0, 2, 3))
    gt1 = np.einsum('oc,bohw->bchw', W2, gy2)
    gn1 = gt1 * (n1 > 0)
    gy1, g_bn1w, g_bn1b = bn_bwd(gn1, xh1, rstd1, bn1[0])
    g_w1 = np.einsum('bohw,bchw->oc', gy1, u)
    g_b1 = gy1.sum((0, 2, 3))
    gu = np.einsum('oc,bohw->bchw', W1, gy1)
    gxb1 = gxb1 + gu
    gxv1 = gxv1 + gu
    ga1 = (gxb1 * xb - gxv1 * xv).sum((2, 3))                                  # (B, C)
    gxb = gxb1 * a1
    gxv = gxv1 * (1 - a1)
    a1s = a1[:, :, 0, 0]
    gz2 = ga1 * a1s * (1 - a1s)
    g_fc2w, g_fc2b = gz2.T @ r1, gz2.sum(0)
    gz1 = (gz2 @ fc2_w.astype(f64)) * (z1 > 0)
    g_fc1w, g_fc1b = gz1.T @ s, gz1.sum(0)
    gs = gz1 @ fc1_w.astype(f64)                                               # (B, 2C)
    dx = np.concatenate([gxb, gxv], 1) + gs[:, :, None, None] / (h * w)
    grads = dict(fc1_w=g_fc1w, fc1_b=g_fc1b, fc2_w=g_fc2w, fc2_b=g_fc2b, conv1_w=g_w1.reshape(conv1_w.shape), conv1_b=g_b1,
                 bn1_w=g_bn1w, bn1_b=g_bn1b, conv2_w=g_w2.reshape(conv2_w.shape), conv2_b=g_b2, bn2_w=g_bn2w, bn2_b=g_bn2b)
    return out.astype(f32), dx, grads


def mghs_depth_view_transform(cfg, calib, depth, tran_feat, height_idx, inv_post_rot=None, combine=None):
    """MGHS_Depth.view_transform (lss_heightmap.py:793-856) with collapse_z=False: the full-height grid as
    (B,C,1,Dy,Dx) and the three band grids concatenated along z in the order low, mid, high (:845) -> (B,C,16,Dy,Dx)."""
    bev, lo, mid, hi = view_transform(cfg, calib, depth, tran_feat, height_idx, inv_post_rot, combine, collapse_z=False)
    return bev, np.concatenate([lo, mid, hi], axis=2)


# ---------------------------------------------------------------------------------------------
# Occupancy-head losses (the caller row after the hot path)
# ---------------------------------------------------------------------------------------------

def _nll_clamped(x):
    """binary_cross_entropy_with_logits(inverse_sigmoid(x), 1): inverse_sigmoid's loops
    (semkitti_loss.py:8-16) move x into [1e-5, 1-1e-5) in steps of 1e-5; the result is -log(x')."""
    x = float(np.float32(x))
    while x >= 1 - 1e-5:
        x -= 1e-5
    while x < 1e-5:
        x += 1e-5
    return -np.log(x)


def occ_losses(logits, labels, mask_camera, class_weights, ignore_index=255, non_empty_idx=17):
    """predictor.loss (dense_heads/occ_head.py:102-139) in float64, written like the reference: the
    class-balanced, camera-masked cross entropy (losses/cross_entropy_loss.py:12-63 with weight = mask,
    avg_factor = sum_i #(valid voxels of class i) * w_i), sem_scal_loss_with_mask (semkitti_loss.py:171-226)
    and geo_scal_loss_with_mask (:136-169).  Returns (loss_occ, loss_sem_scal, loss_geo_scal) before the
    head's weight_ce / weight_sem / weight_geo factors (all 1 in DHD-S)."""
    z = logits.astype(np.float64)
    z = z - z.max(1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(1, keepdims=True)
    n_cls = p.shape[1]
    cam = mask_camera.astype(bool)
    # cross entropy
    valid_labels = labels[cam]
    avg = sum(float((valid_labels == i).sum()) * float(class_weights[i]) for i in range(n_cls))
    keep = labels != ignore_index
    nll = np.zeros(len(labels))
    nll[keep] = -np.log(p[keep, labels[keep]]) * class_weights.astype(np.float64)[labels[keep]]
    loss_occ = float((nll * cam).sum() / avg)
    # semantic scal
    m = keep & cam
    loss, count = 0.0, 0
    tgt = labels[m]
    for i in range(n_cls - 1):
        pi = p[m, i]
        ct = (tgt == i).astype(np.float64)
        if ct.sum() > 0:
            count += 1
            nom = (pi * ct).sum()
            if pi.sum() > 0:
                loss += _nll_clamped(nom / (pi.sum() + 1e-5))
            loss += _nll_clamped(nom / (ct.sum() + 1e-5))
            if (1 - ct).sum() > 0:
                loss += _nll_clamped(((1 - pi) * (1 - ct)).sum() / ((1 - ct).sum() + 1e-5))
    loss_sem = loss / count
    # geometric scal
    empty = p[m, non_empty_idx]
    nonempty_t = (tgt != non_empty_idx).astype(np.float64)
    inter = (nonempty_t * (1 - empty)).sum()
    loss_geo = (_nll_clamped(inter / ((1 - empty).sum() + 1e-5)) + _nll_clamped(inter / (nonempty_t.sum() + 1e-5)) +
                _nll_clamped(((1 - nonempty_t) * empty).sum() / ((1 - nonempty_t).sum() + 1e-5)))
    return loss_occ, loss_sem, loss_geo


def occ_confusion(logits, labels, mask_camera, n_cls=18):
    """predictor.get_occ (occ_head.py:141-153) + Metric_mIoU.hist_info (occ_metrics.py:79-104):
    pred = argmax softmax(logits); hist[gt, pred] over camera-visible voxels with 0 <= gt < n_cls."""
    z = logits.astype(np.float64)
    p = np.exp(z - z.max(1, keepdims=True))
    pred = (p / p.sum(1, keepdims=True)).argmax(1)
    cam = mask_camera.astype(bool)
    gt, pr = labels[cam], pred[cam]
    k = (gt >= 0) & (gt < n_cls)
    hist = np.bincount(n_cls * gt[k].astype(int) + pr[k].astype(int), minlength=n_cls ** 2).reshape(n_cls, n_cls)
    with np.errstate(divide='ignore', invalid='ignore'):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    return pred.astype(np.uint8), hist, iu


# ---------------------------------------------------------------------------------------------
# LiDAR -> per-camera sparse depth / height maps (the label side before the hot path)
# ---------------------------------------------------------------------------------------------

def points_to_maps(points, height, width, downsample=1, depth_range=(1.0, 45.0), return_ties=False):
    """PointToMultiViewDepthandHeight.points2depthmap / points2heightmap
    (datasets/pipelines/loading_new.py:35-99): points (N,4) = (u, v, d, h) in augmented image coordinates.
    Pixel = round-half-even(uv / downsample); among the points of a pixel the one with the smallest float32 key
    pixel_rank + d/100 wins (the reference sorts by that key and keeps the first of each pixel; equal keys --
    depths closer than the float32 spacing at the rank, up to ~1.5 m at the bottom of a 256x704 image -- keep
    their input order here, i.e. a stable sort; the reference's torch.argsort is NOT stable, so at such pixels it
    may keep any of the tied points: `return_ties` adds the mask of pixels whose winning key is shared).
    Returns (depth_map, height_map, height_mask[, tie_mask])."""
    h, w = height // downsample, width // downsample
    pts = points.astype(f32)
    cu = np.rint(pts[:, 0] / f32(downsample)).astype(f32)
    cv = np.rint(pts[:, 1] / f32(downsample)).astype(f32)
    d = pts[:, 2]
    kept = (cu >= 0) & (cu < w) & (cv >= 0) & (cv < h) & (d < f32(depth_range[1])) & (d >= f32(depth_range[0]))
    cu, cv, d, hv = cu[kept], cv[kept], d[kept], pts[kept, 3]
    ranks = (cu + cv * f32(w)).astype(f32)
    key = (ranks + d / f32(100.0)).astype(f32)
    order = np.argsort(key, kind='stable')
    cu, cv, d, hv, ranks = cu[order], cv[order], d[order], hv[order], ranks[order]
    first = np.ones(len(ranks), bool)
    first[1:] = ranks[1:] != ranks[:-1]
    x, y = cu[first].astype(np.int64), cv[first].astype(np.int64)
    depth_map = np.zeros((h, w), f32)
    height_map = np.zeros((h, w), f32)
    mask = np.zeros((h, w), bool)
    depth_map[y, x] = d[first]
    height_map[y, x] = hv[first]
    mask[y, x] = True
    if return_ties:
        key = key[order]
        tied = np.zeros(len(ranks), bool)
        tied[:-1] = first[:-1] & (key[1:] == key[:-1]) & (ranks[1:] == ranks[:-1])
        ties = np.zeros((h, w), bool)
        ties[cv[tied].astype(np.int64), cu[tied].astype(np.int64)] = True
        return depth_map, height_map, mask, ties
    return depth_map, height_map, mask


# ---------------------------------------------------------------------------------------------
# weight EMA of the training loop (core/hook/ema.py:31-59)
# ---------------------------------------------------------------------------------------------

def ema_decay(decay, updates):
    """ema.py:44: exponential ramp of the decay, a Python double."""
    import math
    return decay * (1 - math.exp(-updates / 2000))


def ema_update(ema, model, d):
    """ema.py:55-59 for one float32 tensor: `v *= d; v += (1.0 - d) * m` -- torch rounds the two
    Python doubles d and 1-d to float32 and each of the three operations to float32."""
    v = ema.astype(f32) * f32(d)
    return (v + (f32(1.0 - d) * model.astype(f32)).astype(f32)).astype(f32)
