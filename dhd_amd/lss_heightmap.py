"""MGHS / MGHS_Depth / MGHS_Stereo view transformers with the reference's registry names,
constructor kwargs, method names and return tuples
(projects/mmdet3d_plugin/models/necks/lss_heightmap.py: MGHS :13, MGHS_Depth :704, MGHS_Stereo :900),
running the lift-splat on the gfx950 kernels of libdhd_amd.so.

What changed underneath (and only underneath):
  * `view_transform` issues ONE geometry+grouping pass and ONE pooling launch for the full-height
    BEV grid and the three height-band grids, instead of 4x get_ego_coor + 4x
    voxel_pooling_prepare_v2 + 4x bev_pool_v2 + permute + cat (:425-457);
  * band membership is a per-pixel uint8 (argmax -> LUT); the three masked copies of tran_feat
    (:436-442) are never materialised;
  * the backward pass reuses the forward's grouping (the reference re-argsorts, bev_pool.py:47).
Reference quirks that are observable from outside are kept: `view_transform` overwrites
`self.grid_config` with the hard-coded full grid and (for MGHS) leaves it at `mask_3_grid`
(:425-457), which changes what get_downsampled_gt_depth computes afterwards (:652-654).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mghs_op
from .bev_pool_v2 import bev_pool_v2
from .depthnet import DepthNet, HeightNet
from .registry import NECKS

_FULL_GRID = {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4], 'depth': [1.0, 45.0, 0.5]}


@NECKS.register_module()
class MGHS(nn.Module):
    def __init__(self, grid_config, input_size, downsample=16, in_channels=512, out_channels=64,
                 heightnet_cfg=dict(), accelerate=False, sid=False, collapse_z=True,
                 height_range=[-1.5, -1, 0, 0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4], height_interval=0.5,
                 mask_range=[-5, 0, 0.4, 5], loss_height_weight=1.0,
                 mask_1_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 2.2, 0.4], 'depth': [1.0, 45.0, 0.5]},
                 mask_2_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [2.2, 3.8, 0.4], 'depth': [1.0, 45.0, 0.5]},
                 mask_3_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [3.8, 5.4, 0.4], 'depth': [1.0, 45.0, 0.5]}):
        super().__init__()
        self.grid_config = grid_config
        self.downsample = downsample
        self.create_grid_infos(**grid_config)
        self.sid = sid
        self.input_size = input_size
        self.frustum = self.create_frustum(grid_config['depth'], input_size, downsample)
        self.accelerate = accelerate
        self.initial_flag = True
        self.out_channels = out_channels
        self.in_channels = in_channels
        self.depth_net = nn.Conv2d(in_channels, self.D + self.out_channels, kernel_size=1, padding=0)
        self.H = len(height_range)
        self.height_net = HeightNet(in_channels=self.in_channels, mid_channels=self.in_channels,
                                    depth_channels=self.H, **heightnet_cfg)
        self.collapse_z = collapse_z
        self.height_range = height_range
        self.mask_range = mask_range
        self.height_interval = height_interval
        self.loss_height_weight = loss_height_weight
        self.mask_1_grid = mask_1_grid
        self.mask_2_grid = mask_2_grid
        self.mask_3_grid = mask_3_grid
        self._plans = {}
        self._axes_dev = {}
        self._static = {}           # plan -> (workspace, stamp, tensors kept alive) of an accelerate=True static rig
        self.deterministic = None   # None: mghs_op's default (DHD_MGHS_DETERMINISTIC); True / False: this module's plans
        self.amp_outputs = True     # under autocast: pooled tensors in the autocast dtype (= the float32 result cast, see _pool)

    # ------------------------------------------------------------------ grid / frustum ---
    def create_grid_infos(self, x, y, z, **kwargs):
        """lower bound, interval and size per axis as float32 tensors (reference :86-102)."""
        axes = (x, y, z)
        self.grid_lower_bound = torch.Tensor([a[0] for a in axes])
        self.grid_interval = torch.Tensor([a[2] for a in axes])
        self.grid_size = torch.Tensor([(a[1] - a[0]) / a[2] for a in axes])

    def create_frustum(self, depth_cfg, input_size, downsample):
        """(D, fH, fW, 3) template of (u, v, d) (reference :105-134); also sets self.D."""
        h_in, w_in = input_size
        fh, fw = h_in // downsample, w_in // downsample
        d = torch.arange(*depth_cfg, dtype=torch.float)
        self.D = d.shape[0]
        if self.sid:
            cfg_t = torch.tensor(depth_cfg).float()
            steps = torch.arange(self.D).float()
            d = torch.exp(torch.log(cfg_t[0]) + steps / (self.D - 1) * torch.log((cfg_t[1] - 1) / cfg_t[0]))
        u = torch.linspace(0, w_in - 1, fw, dtype=torch.float)
        v = torch.linspace(0, h_in - 1, fh, dtype=torch.float)
        self._axes = (u.clone(), v.clone(), d.clone())
        return torch.stack((u.view(1, 1, fw).expand(self.D, fh, fw), v.view(1, fh, 1).expand(self.D, fh, fw),
                            d.view(-1, 1, 1).expand(self.D, fh, fw)), -1)

    def _axes_on(self, device):
        key = str(device)
        if key not in self._axes_dev:
            self._axes_dev[key] = tuple(a.to(device) for a in self._axes)
        return self._axes_dev[key]

    def _current_grid(self):
        return mghs_op.grid_struct(self.grid_lower_bound.tolist(), self.grid_interval.tolist(),
                                   self.grid_size.tolist())

    def _plan(self, batch, n_cams, fh, fw, grids_key, grids):
        det = mghs_op.is_deterministic() if self.deterministic is None else bool(self.deterministic)
        key = (batch, n_cams, self.D, fh, fw, self.out_channels, grids_key, det)
        if key not in self._plans:
            # the pooling backward hands the context gradient back in tran_feat's own (B*N, C, fH, fW) layout
            self._plans[key] = mghs_op.Plan(batch, n_cams, self.D, fh, fw, self.out_channels, grids, deterministic=det,
                                            feat_grad_nchw=True)
        return self._plans[key]

    def _calib(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        return mghs_op.make_calib(sensor2ego, cam2imgs, post_rots, post_trans, bda, self._axes_on(sensor2ego.device))

    # ------------------------------------------------------------------ geometry API ------
    def get_ego_coor(self, sensor2ego, ego2global, cam2imgs, post_rots, post_trans, bda):
        """Frustum points in the ego frame, (B, N, D, fH, fW, 3) (reference :179-231)."""
        B, N = sensor2ego.shape[:2]
        fh, fw = self.frustum.shape[1:3]
        plan = self._plan(B, N, fh, fw, ('cur', tuple(self.grid_size.tolist())), [self._current_grid()])
        calib, keep = self._calib(sensor2ego, cam2imgs, post_rots, post_trans, bda)
        _, ego = mghs_op.voxel_index(plan, calib, 0, want_ego=True)
        return ego

    get_lidar_coor = get_ego_coor

    def voxel_pooling_prepare_v2(self, coor):
        """Index preparation for ONE grid from explicit coordinates (reference :303-371).  Kept for
        API parity (the dormant `accelerate` path and external callers); the hot path never
        materialises `coor`.  Same arithmetic as the reference; within-voxel order is ascending
        point id (the reference's argsort leaves it unspecified)."""
        B, N, D, H, W, _ = coor.shape
        n_pts = B * N * D * H * W
        dev = coor.device
        idx = ((coor - self.grid_lower_bound.to(coor)) / self.grid_interval.to(coor)).long().view(n_pts, 3)
        size = self.grid_size.to(dev)
        kept = ((idx >= 0) & (idx < size)).all(dim=1)
        pid = torch.arange(n_pts, dtype=torch.int, device=dev)
        pix = (torch.arange(n_pts // D, dtype=torch.int, device=dev).view(B, N, 1, H, W)
               .expand(B, N, D, H, W).reshape(-1))
        nx, ny, nz = (int(s) for s in self.grid_size.tolist())
        batch = torch.arange(B, device=dev).view(B, 1).expand(B, n_pts // B).reshape(-1)
        rank = batch * (nz * ny * nx) + idx[:, 2] * (ny * nx) + idx[:, 1] * nx + idx[:, 0]
        rank, pid, pix = rank[kept], pid[kept], pix[kept]
        if rank.numel() == 0:
            return None, None, None, None, None
        order = torch.argsort(rank, stable=True)
        rank, pid, pix = rank[order], pid[order], pix[order]
        _, lengths = torch.unique_consecutive(rank, return_counts=True)
        starts = torch.cumsum(lengths, 0) - lengths
        return (rank.int().contiguous(), pid.int().contiguous(), pix.int().contiguous(),
                starts.int().contiguous(), lengths.int().contiguous())

    def init_acceleration_v2(self, coor):
        (self.ranks_bev, self.ranks_depth, self.ranks_feat, self.interval_starts,
         self.interval_lengths) = self.voxel_pooling_prepare_v2(coor)

    def pre_compute(self, input):
        if self.initial_flag:
            self.init_acceleration_v2(self.get_ego_coor(*input[1:7]))
            self.initial_flag = False

    def voxel_pooling_v2(self, coor, depth, feat):
        """coor (B,N,D,fH,fW,3), depth (B,N,D,fH,fW), feat (B,N,C,fH,fW) -> (B, C*Dz, Dy, Dx)
        through the operator-level seam (reference :261-300)."""
        ranks_bev, ranks_depth, ranks_feat, starts, lengths = self.voxel_pooling_prepare_v2(coor)
        nx, ny, nz = (int(s) for s in self.grid_size.tolist())
        if ranks_feat is None:
            print('warning ---> no points within the predefined bev receptive field')
            return torch.zeros((feat.shape[0], feat.shape[2] * nz, ny, nx)).to(feat)
        feat = feat.permute(0, 1, 3, 4, 2)
        out = bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                          (depth.shape[0], nz, ny, nx, feat.shape[-1]), starts, lengths)
        if self.collapse_z:
            out = torch.cat(out.unbind(dim=2), 1)
        return out

    # ------------------------------------------------------------------ hot path ----------
    def _pool(self, input, depth, tran_feat, height, grid_cfgs, layout=None):
        """One dhd_mghs_lift (band ids from `height`, context re-layout, geometry + grouping) + one pooling call for
        all of `grid_cfgs` (grid 0 pools every pixel, grids 1.. the pixels of their height band)."""
        sensor2ego, _, cam2imgs, post_rots, post_trans, bda = input[1:7]
        B, N = sensor2ego.shape[:2]
        fh, fw = depth.shape[-2:]
        grids = [mghs_op.grid_from_cfg(g) for g in grid_cfgs]
        key = tuple(tuple(tuple(g[a]) for a in 'xyz') for g in grid_cfgs)
        plan = self._plan(B, N, fh, fw, key, grids)
        calib, keep = self._calib(sensor2ego, cam2imgs, post_rots, post_trans, bda)
        layout = layout or ('collapsed' if self.collapse_z else 'split')
        # Under autocast the reference's operator returns float32 (bev_pool.py:20-21) and the first convolution behind it casts
        # the 174 MB per sample to half at once.  `amp_outputs` (default on) lets the writer emit that half tensor directly --
        # bit-identical to the cast of the float32 result -- and the backward read half gradients.
        odt = (torch.get_autocast_dtype('cuda') if self.amp_outputs and torch.is_autocast_enabled() and plan.half_outputs_supported
               else torch.float32)
        if odt not in (torch.float16, torch.bfloat16):
            odt = torch.float32
        needs_grad = torch.is_grad_enabled() and (depth.requires_grad or tran_feat.requires_grad)
        if self.accelerate and not needs_grad:
            # Static rig at inference (the reference's dormant accelerate / pre_compute idea, :234-258,374-378): while the
            # SAME calibration tensors are passed unmodified, camera matrices, geometry and the grouping of the full-height
            # grid (which pools every pixel whatever its band) are reused; per frame only the band grids' entries are
            # counted / scanned / scattered again (dhd_mghs_lift_static), or nothing at all for a single-grid call.  The
            # cache entry owns its scratch (the grouping must survive other view transforms on the stream) and keeps the
            # caller's calibration tensors alive, so that their addresses cannot be recycled for other data.
            srcs = (sensor2ego, cam2imgs, post_rots, post_trans, bda)
            stamp = (depth.device,) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in srcs)
            hit = self._static.get(plan)
            if hit is None or hit[1] != stamp:
                ws = plan.new_workspace(depth.device, private_scratch=True)
                outs = mghs_op.mghs_lift_pool(plan, calib, height, self.height_range, self.mask_range, depth, tran_feat, ws, layout, out_dtype=odt)
                self._static = {p_: e for p_, e in self._static.items() if e[1] == stamp}   # one rig at a time
                self._static[plan] = (ws, stamp, srcs, keep)
                return list(outs)
            if len(grids) == 1:
                tf = tran_feat.float().contiguous()
                return list(mghs_op._MGHSPool.apply(depth.float(), tf, plan, hit[0], mghs_op._nchw_to_nhwc(tf), layout, odt))
            return list(mghs_op.mghs_lift_pool(plan, calib, height, self.height_range, self.mask_range, depth, tran_feat,
                                               hit[0], layout, static=True, out_dtype=odt))
        return list(mghs_op.mghs_lift_pool(plan, calib, height, self.height_range, self.mask_range, depth, tran_feat, layout=layout, out_dtype=odt))

    def view_transform_core(self, input, depth, tran_feat):
        """Single-grid lift-splat on the CURRENT grid_config (reference :380-405)."""
        cfg = {a: self.grid_config[a] for a in 'xyz'}
        return self._pool(input, depth, tran_feat, None, [cfg])[0], depth

    def _set_grid(self, cfg):
        self.grid_config = cfg
        self.create_grid_infos(**cfg)

    def _band(self, height):
        return mghs_op.height_band(height, self.height_range, self.mask_range)

    def _four_grid_pool(self, input, depth, tran_feat, height, layout=None):
        cfgs = [dict(_FULL_GRID), self.mask_1_grid, self.mask_2_grid, self.mask_3_grid]
        return self._pool(input, depth, tran_feat, height, cfgs, layout)

    def view_transform(self, input, depth, tran_feat, height):
        """-> (bev_feat, depth, height, low, mid, high) (reference :407-459)."""
        outs = self._four_grid_pool(input, depth, tran_feat, height)
        self._set_grid(dict(_FULL_GRID))
        self._set_grid(self.mask_3_grid)  # state the reference leaves behind (:455-456)
        bev, lo, mid, hi = outs  # (B, nz*C, ny, nx) each, or (B, C, nz, ny, nx) with collapse_z=False
        return bev, depth, height, lo, mid, hi

    def forward(self, input, stereo_metas=None):
        x, _, _, _, _, _, _, mlp_input = input[:8]
        B, N, C, H, W = x.shape
        x = x.view(B * N, C, H, W)
        depth, tran_feat, height = self._head(self.depth_net(x), self.height_net(x, mlp_input, stereo_metas))
        return self.view_transform(input, depth, tran_feat, height)

    fused_head = not __import__('os').environ.get('DHD_PLAIN_SOFTMAX')   # A/B switch: the torch formulation below

    def _head(self, x_d, h_logits):
        """depth = softmax(x_d[:, :D]), tran_feat = x_d[:, D:D+C], height = softmax(h_logits[:, :H]) (reference :484-489).  On
        the GPU one launch each way (mghs_op.depth_height_head: the softmax of torch's own kernel, bit for bit, float32 out as
        under autocast); the band ids of the lift are the argmax of exactly these height probabilities."""
        if self.fused_head and x_d.is_cuda and x_d.dim() == 4 and h_logits.dim() == 4:
            depth, tran_feat, height, _ = mghs_op.depth_height_head(x_d, h_logits, self.D, self.out_channels, self.height_range, self.mask_range)
            return depth, tran_feat, height
        return (x_d[:, :self.D].softmax(dim=1), x_d[:, self.D:self.D + self.out_channels], h_logits[:, :self.H].softmax(dim=1))

    # ------------------------------------------------------------------ small helpers -----
    def get_mlp_input(self, sensor2ego, ego2global, intrin, post_rot, post_tran, bda):
        """(B, N, 27) camera descriptor (reference :493-526)."""
        B, N = sensor2ego.shape[:2]
        bda = bda.view(B, 1, 3, 3).expand(B, N, 3, 3)
        head = torch.stack([intrin[..., 0, 0], intrin[..., 1, 1], intrin[..., 0, 2], intrin[..., 1, 2],
                            post_rot[..., 0, 0], post_rot[..., 0, 1], post_tran[..., 0],
                            post_rot[..., 1, 0], post_rot[..., 1, 1], post_tran[..., 1],
                            bda[..., 0, 0], bda[..., 0, 1], bda[..., 1, 0], bda[..., 1, 1], bda[..., 2, 2]], dim=-1)
        return torch.cat([head, sensor2ego[:, :, :3, :].reshape(B, N, -1)], dim=-1)

    def height_feature_to_height_map(self, height_feature, height_range):
        if height_feature.dim() != 4:
            raise ValueError('Input tensor must have 4 dimensions (BxN, H, fH, fW)')
        table = torch.tensor(height_range, device=height_feature.device)
        return table[torch.argmax(height_feature, dim=1)]

    def create_mask_3(self, input_tensor, h_min, thr1, thr2, h_max):
        return ((input_tensor >= h_min) & (input_tensor < thr1),
                (input_tensor >= thr1) & (input_tensor < thr2),
                (input_tensor >= thr2) & (input_tensor < h_max))

    @staticmethod
    def _min_pool_nonzero(maps, factor):
        """factor x factor min-pool that ignores zeros; empty cells stay 0 (reference :566-592)."""
        B, N, H, W = maps.shape
        assert H % factor == 0 and W % factor == 0
        big = torch.where(maps == 0.0, torch.full_like(maps, 1e5), maps)
        big = big.view(B * N, H // factor, factor, W // factor, factor).permute(0, 1, 3, 2, 4)
        return big.reshape(B * N, H // factor, W // factor, factor * factor).min(dim=-1).values

    def downsample_sparse_map(self, height_maps, downsample_factor=16):
        B, N, H, W = height_maps.shape
        m = self._min_pool_nonzero(height_maps, downsample_factor)
        m = torch.where(m == 1e5, torch.zeros_like(m), m)
        return m.view(B, N, H // downsample_factor, W // downsample_factor)

    def get_downsampled_gt_depth(self, gt_depths):
        """(B, N, H, W) sparse depth -> (B*N*fH*fW, D) one-hot (reference :625-667).  Reads the
        depth interval from the *current* self.grid_config."""
        g = self._min_pool_nonzero(gt_depths, self.downsample)
        dcfg = self.grid_config['depth']
        if not self.sid:
            g = (g - (dcfg[0] - dcfg[2])) / dcfg[2]
        else:
            g = torch.log(g) - torch.log(torch.tensor(dcfg[0]).float())
            g = g * (self.D - 1) / torch.log(torch.tensor(dcfg[1] - 1.).float() / dcfg[0])
            g = g + 1.
        g = torch.where((g < self.D + 1) & (g >= 0.0), g, torch.zeros_like(g))
        return F.one_hot(g.long(), num_classes=self.D + 1).view(-1, self.D + 1)[:, 1:].float()

    def get_downsampled_gt_height(self, gt_height):
        """(B, N, H, W) sparse height -> (B*N*fH*fW, H) one-hot (reference :670-701)."""
        g = self._min_pool_nonzero(gt_height, self.downsample)
        g = (g - self.height_range[0]) / self.height_interval
        g = torch.where((g < self.H + 1) & (g >= 0.0), g, torch.zeros_like(g))
        return F.one_hot(g.long(), num_classes=self.H + 1).view(-1, self.H + 1)[:, 1:].float()

    def _fg_bce(self, pred, labels, fg):
        with torch.autocast(device_type=pred.device.type, enabled=False):
            loss = F.binary_cross_entropy(pred.float()[fg], labels[fg], reduction='none').sum()
            return loss / max(1.0, fg.sum())

    def _hip_labels(self, gt_depth, gt_height):
        from . import label_loss
        return label_loss.bin_labels(gt_depth.float(), gt_height.float(), self.downsample, self.grid_config['depth'], self.D,
                                     self.height_range[0], self.height_interval, self.H, sid=self.sid)

    def get_height_loss(self, gt_depth, gt_height, height):
        """BCE over foreground pixels (those with a depth label), reference :595-622.  On the GPU labels are bin
        indices and the loss is one HIP operator (csrc/label_loss.hip)."""
        if height.is_cuda:
            from . import label_loss
            dbin, hbin = self._hip_labels(gt_depth, gt_height)
            return label_loss.fg_bce(height, hbin, dbin, self.loss_height_weight)
        height_labels = self.get_downsampled_gt_height(gt_height)
        depth_labels = self.get_downsampled_gt_depth(gt_depth)
        fg = depth_labels.max(dim=1).values > 0.0
        pred = height.permute(0, 2, 3, 1).contiguous().view(-1, self.H)
        return self.loss_height_weight * self._fg_bce(pred, height_labels, fg)


@NECKS.register_module()
class MGHS_Depth(MGHS):
    def __init__(self, loss_depth_weight=3.0, depthnet_cfg=dict(), **kwargs):
        super().__init__(**kwargs)
        self.loss_depth_weight = loss_depth_weight
        self.depth_net = DepthNet(in_channels=self.in_channels, mid_channels=self.in_channels,
                                  context_channels=self.out_channels, depth_channels=self.D, **depthnet_cfg)

    def forward(self, input, stereo_metas=None):
        x, _, _, _, _, _, _, mlp_input = input[:8]
        B, N, C, H, W = x.shape
        x = x.view(B * N, C, H, W)
        depth, tran_feat, height = self._head(self.depth_net(x, mlp_input, stereo_metas), self.height_net(x, mlp_input, stereo_metas=None))
        return self.view_transform(input, depth, tran_feat, height)

    def view_transform(self, input, depth, tran_feat, height):
        """-> (bev_feat, bev_feat_w_z (B,C,16,ny,nx), depth, height) (reference :793-856); unlike the
        base class the grid_config is reset to the full grid afterwards (:848-854)."""
        if self.collapse_z:
            # not used by any shipped config; the reference's cat(dim=2) of 4-D tensors (:845) would
            # stack the bands along y, which is reproduced literally
            outs = self._four_grid_pool(input, depth, tran_feat, height)
            bev, bev_w_z = outs[0], torch.cat(outs[1:], dim=2)
        else:
            # the three band grids are written straight into one (B, C, 16, ny, nx) tensor
            bev, bev_w_z = self._four_grid_pool(input, depth, tran_feat, height, layout='stacked')
        self._set_grid(dict(_FULL_GRID))
        return bev, bev_w_z, depth, height

    def get_depth_and_height_loss(self, gt_depth, gt_height, depth, height):
        """reference :859-897 -> (loss_depth, loss_height)."""
        if height.is_cuda:
            from . import label_loss
            dbin, hbin = self._hip_labels(gt_depth, gt_height)
            return (label_loss.fg_bce(depth, dbin, dbin, self.loss_depth_weight),
                    label_loss.fg_bce(height, hbin, dbin, self.loss_height_weight))
        height_labels = self.get_downsampled_gt_height(gt_height)
        depth_labels = self.get_downsampled_gt_depth(gt_depth)
        fg = depth_labels.max(dim=1).values > 0.0
        hp = height.permute(0, 2, 3, 1).contiguous().view(-1, self.H)
        dp = depth.permute(0, 2, 3, 1).contiguous().view(-1, self.D)
        return (self.loss_depth_weight * self._fg_bce(dp, depth_labels, fg),
                self.loss_height_weight * self._fg_bce(hp, height_labels, fg))


@NECKS.register_module()
class MGHS_Stereo(MGHS_Depth):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        axes = self._axes
        self.cv_frustum = self.create_frustum(kwargs['grid_config']['depth'], kwargs['input_size'], downsample=4)
        self._axes = axes  # the lift keeps the /16 frustum; cv_frustum is only the stereo template
