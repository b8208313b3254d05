"""Functional layer over the fused MGHS entry points of libdhd_amd.so (include/dhd_amd.h section 2).

`mghs_pool` is the autograd-visible replacement for the reference's
4x (get_ego_coor -> voxel_pooling_prepare_v2 -> bev_pool_v2 -> permute -> cat) chain,
models/necks/lss_heightmap.py:380-459.  Gradients flow to `depth` and `tran_feat` only; the
height branch enters through an argmax/bool mask and carries none (:434-442).
"""
import ctypes as C

import torch

from . import _lib
from .trace import traced


def grid_struct(lower, interval, size):
    """dhd_grid from the three float32 triples MGHS.create_grid_infos produces (:99-102)."""
    g = _lib.Grid()
    for a in range(3):
        g.lower[a] = float(lower[a])
        g.interval[a] = float(interval[a])
        g.size[a] = float(size[a])
        g.n[a] = int(size[a])
    return g


def grid_from_cfg(cfg):
    """Same arithmetic as create_grid_infos: python doubles, rounded to float32 by torch.Tensor."""
    axes = [cfg[k] for k in ('x', 'y', 'z')]
    lower = torch.Tensor([a[0] for a in axes])
    interval = torch.Tensor([a[2] for a in axes])
    size = torch.Tensor([(a[1] - a[0]) / a[2] for a in axes])
    return grid_struct(lower.tolist(), interval.tolist(), size.tolist())


_default_deterministic = None   # module default of Plan(deterministic=None); None = environment DHD_MGHS_DETERMINISTIC


def set_deterministic(on=True):
    """Default for plans built afterwards without an explicit `deterministic=` (Plan, MGHS modules): reproducible forward
    sums, i.e. prepare orders the entries of every voxel by point id (include/dhd_amd.h: DHD_MGHS_DETERMINISTIC; one
    ranking pass + a second scatter per prepare).  A Python-side default only: the C ABI takes the flag per call
    (dhd_mghs_desc.flags), so two models in one process can differ by passing `deterministic=` themselves."""
    global _default_deterministic
    _default_deterministic = bool(on)


def is_deterministic():
    if _default_deterministic is None:
        import os
        return os.environ.get('DHD_MGHS_DETERMINISTIC', '0') not in ('', '0')
    return _default_deterministic


class _ScratchPool:
    """Grow-only device scratch, one buffer per (device, stream, tag).  A larger request allocates a new buffer and
    keeps the old ones alive: a captured HIP graph (or a kernel still in flight) may hold their raw pointers, and
    memory handed back to the allocator could be given to someone else while they run.  Calls that share a buffer are
    ordered by their stream, which is what scratch needs."""

    def __init__(self):
        self._cur = {}
        self._retired = []

    def get(self, device, nbytes, tag=''):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
        buf = self._cur.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                self._retired.append(buf)
            buf = self._cur[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return buf

    def bytes_held(self):
        return sum(b.numel() for b in self._cur.values()) + sum(b.numel() for b in self._retired)


scratch_pool = _ScratchPool()


class Workspace:
    """Device memory of one view transform (include/dhd_amd.h: dhd_mghs_workspace): `state` is what the backward pass
    needs from prepare (held by the autograd graph), `scratch` everything else (shared per stream by default)."""

    def __init__(self, state, scratch):
        self.state, self.scratch = state, scratch
        self.device = state.device
        c = self.c = _lib.MghsWorkspace()
        c.state, c.state_bytes = state.data_ptr(), state.numel()
        c.scratch, c.scratch_bytes = scratch.data_ptr(), scratch.numel()


class Plan:
    """Static description of one view transform: sizes + grids + flags (no device state)."""

    def __init__(self, batch, n_cams, n_depth, fh, fw, channels, grids, deterministic=None, feat_grad_nchw=False):
        if not 1 <= len(grids) <= _lib.DHD_MAX_GRIDS:
            raise ValueError('1..4 grids')
        d = _lib.MghsDesc()
        d.batch, d.n_cams, d.n_depth, d.fh, d.fw, d.channels = batch, n_cams, n_depth, fh, fw, channels
        d.n_grids = len(grids)
        for i in range(_lib.DHD_MAX_GRIDS):
            d.grid[i] = grids[min(i, len(grids) - 1)]
        self.deterministic = is_deterministic() if deterministic is None else bool(deterministic)
        self.feat_grad_nchw = bool(feat_grad_nchw)
        d.flags = (_lib.MGHS_DETERMINISTIC if self.deterministic else 0) | (_lib.MGHS_FEAT_GRAD_NCHW if feat_grad_nchw else 0)
        self.desc = d
        self.grids = list(grids)
        st, sc = C.c_size_t(0), C.c_size_t(0)
        _lib.check(_lib.load().dhd_mghs_workspace_bytes(C.byref(d), C.byref(st), C.byref(sc)), 'dhd_mghs_workspace_bytes')
        self.state_bytes, self.scratch_bytes = int(st.value), int(sc.value)

    @property
    def half_outputs_supported(self):
        """dhd_tensor_view.dtype DHD_F16 / DHD_BF16 need the segment writer (include/dhd_amd.h): C = 64, ny % 4 == 0, nx % 4 == 0."""
        return self.desc.channels == 64 and all(g.n[1] % 4 == 0 and g.n[0] % 4 == 0 and 4 * g.n[0] <= 1024 for g in self.grids)

    def out_shapes(self):
        d = self.desc
        return [(d.batch, g.n[2] * d.channels, g.n[1], g.n[0]) for g in self.grids]

    def new_workspace(self, device, private_scratch=False):
        """A fresh state + the stream's shared scratch (or, `private_scratch`, one of its own: needed when the prepared
        grouping must survive other view transforms on the stream, e.g. a static-rig cache)."""
        device = torch.device(device)
        state = torch.empty(self.state_bytes, dtype=torch.uint8, device=device)
        scratch = (torch.empty(self.scratch_bytes, dtype=torch.uint8, device=device) if private_scratch
                   else scratch_pool.get(device, self.scratch_bytes, 'mghs'))
        return Workspace(state, scratch)


def _f32(t, name):
    return _lib.require_gpu_tensor(t.detach(), torch.float32, name)


def make_calib(sensor2ego, intrin, post_rot, post_tran, bda, frustum_axes, inv_post_rot=None, combine=None):
    """Pack device tensors into a dhd_calib.  Returns (struct, keepalive list)."""
    u, v, d = frustum_axes
    dev = sensor2ego.device
    keep = []

    def prep(t, name):
        if t is None:
            return None
        t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(t)
        return _f32(t, name)

    c = _lib.Calib()
    for name, t in (('sensor2ego', sensor2ego), ('intrin', intrin), ('post_rot', post_rot),
                    ('post_tran', post_tran), ('bda', bda), ('inv_post_rot', inv_post_rot),
                    ('combine', combine), ('frustum_u', u), ('frustum_v', v), ('frustum_d', d)):
        tt = prep(t, name)
        setattr(c, name, tt.data_ptr() if tt is not None else None)
    return c, keep


def height_band(height, height_range, mask_range):
    """(B*N, H, fH, fW) height distribution -> uint8 band id per pixel (0/1/2, 255 = none).
    height_feature_to_height_map + create_mask_3, lss_heightmap.py:528-564."""
    lib = _lib.load()
    h = _f32(height.float().contiguous(), 'height')
    bn, nh, fh, fw = h.shape
    if nh != len(height_range):
        raise ValueError('height has %d bins, height_range %d' % (nh, len(height_range)))
    hr, mr = _range_arrays(height_range, mask_range)
    band = torch.empty((bn, fh, fw), dtype=torch.uint8, device=h.device)
    with torch.cuda.device(h.device):
        rc = lib.dhd_height_band(_lib.ptr(h), bn, nh, fh, fw, hr, mr, _lib.ptr(band), _lib.stream_ptr(h.device))
    _lib.check(rc, 'dhd_height_band')
    return band


def _nchw_to_nhwc(x):
    lib = _lib.load()
    bn, c, fh, fw = x.shape
    out = torch.empty((bn, fh, fw, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dhd_feat_nchw_to_nhwc(_lib.ptr(x), _lib.ptr(out), bn, c, fh * fw, _lib.stream_ptr(x.device))
    _lib.check(rc, 'dhd_feat_nchw_to_nhwc')
    return out


def _nhwc_to_nchw(x):
    lib = _lib.load()
    bn, fh, fw, c = x.shape
    out = torch.empty((bn, c, fh, fw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dhd_feat_nhwc_to_nchw(_lib.ptr(x), _lib.ptr(out), bn, c, fh * fw, _lib.stream_ptr(x.device))
    _lib.check(rc, 'dhd_feat_nhwc_to_nchw')
    return out


def _dense_layout(t):
    """(tensor, nhwc flag) for a 4-D tensor the head kernels can read where it lies: dense NCHW or dense channels_last (scalar
    accesses: any alignment); anything else is made NCHW-contiguous first."""
    if t.is_contiguous():
        return t, 0
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t, 1
    return t.contiguous(), 0


class _DepthHeightHead(torch.autograd.Function):
    """softmax(depth bins), the context slice, softmax(height bins) and the band id per pixel in one launch; the two softmax
    Jacobians + the context gradient in one launch back (csrc/mghs_softmax.hip, dhd_mghs_softmax_forward / _backward).
    Outputs are float32 dense NCHW whatever the inputs' dtype and layout (under autocast torch's softmax returns float32 too)."""

    @staticmethod
    @traced('dhd.mghs.head.forward')
    def forward(ctx, x_d, h_logits, d, c, h_bins, hr, mr):
        lib = _lib.load()
        x_d, x_nhwc = _dense_layout(x_d)
        bn, ct, fh, fw = x_d.shape
        dev = x_d.device
        hl_code, h_nhwc, ht = 0, 0, 0
        if h_logits is not None:
            h_logits, h_nhwc = _dense_layout(h_logits)
            hl_code, ht = _lib.dtype_code(h_logits.dtype), h_logits.shape[1]
        with torch.cuda.device(dev):
            depth = torch.empty((bn, d, fh, fw), dtype=torch.float32, device=dev)
            feat = torch.empty((bn, c, fh, fw), dtype=torch.float32, device=dev)
            height = band = None
            if h_logits is not None:
                height = torch.empty((bn, h_bins, fh, fw), dtype=torch.float32, device=dev)
                band = torch.empty((bn, fh, fw), dtype=torch.uint8, device=dev)
            _lib.check(lib.dhd_mghs_softmax_forward(_lib.ptr(x_d), _lib.dtype_code(x_d.dtype), x_nhwc, ct, _lib.ptr(h_logits), hl_code,
                                                    h_nhwc, ht, bn, fh * fw, d, c, h_bins, hr, mr, _lib.ptr(depth), _lib.ptr(feat),
                                                    _lib.ptr(height), _lib.ptr(band), _lib.stream_ptr(dev)), 'dhd_mghs_softmax_forward')
        ctx.save_for_backward(depth, height)
        ctx.meta = (d, c, h_bins, x_d.dtype, x_nhwc, ct, None if h_logits is None else h_logits.dtype, h_nhwc, ht, bn, fh, fw)
        if h_logits is None:
            return depth, feat
        ctx.mark_non_differentiable(band)
        return depth, feat, height, band

    @staticmethod
    @traced('dhd.mghs.head.backward')
    def backward(ctx, g_depth, g_feat, g_height=None, _g_band=None):
        lib = _lib.load()
        depth, height = ctx.saved_tensors
        d, c, h_bins, x_dt, x_nhwc, ct, h_dt, h_nhwc, ht, bn, fh, fw = ctx.meta
        dev = depth.device
        f32 = lambda g: None if g is None else g.float().contiguous()
        g_depth, g_feat, g_height = f32(g_depth), f32(g_feat), f32(g_height)
        want_x, want_h = ctx.needs_input_grad[0], h_dt is not None and ctx.needs_input_grad[1]
        with torch.cuda.device(dev):
            fmt = lambda nhwc: torch.channels_last if nhwc else torch.contiguous_format
            g_xd = torch.empty((bn, ct, fh, fw), dtype=x_dt, device=dev, memory_format=fmt(x_nhwc)) if want_x else None
            g_hl = torch.empty((bn, ht, fh, fw), dtype=h_dt, device=dev, memory_format=fmt(h_nhwc)) if want_h else None
            if want_x or want_h:
                _lib.check(lib.dhd_mghs_softmax_backward(_lib.ptr(g_depth), _lib.ptr(g_feat), _lib.ptr(g_height), _lib.ptr(depth),
                                                         _lib.ptr(height), bn, fh * fw, d, c, h_bins,
                                                         _lib.ptr(g_xd), _lib.dtype_code(x_dt), x_nhwc, ct, _lib.ptr(g_hl),
                                                         0 if h_dt is None else _lib.dtype_code(h_dt), h_nhwc, ht, _lib.stream_ptr(dev)),
                           'dhd_mghs_softmax_backward')
        return g_xd, g_hl, None, None, None, None, None


def depth_height_head(x_d, h_logits, n_depth, n_context, height_range=None, mask_range=None):
    """lss_heightmap.py:484-489 on the GPU: (depth_net output, height_net output or None) -> (depth, tran_feat, height, band),
    float32 NCHW (band uint8; height / band None without a height branch)."""
    hr = mr = None
    h_bins = 0
    if h_logits is not None:
        h_bins = len(height_range)
        hr, mr = _range_arrays(height_range, mask_range)
    for t in (x_d, h_logits):
        if t is not None and not (t.is_cuda and t.dim() == 4 and t.dtype in (torch.float32, torch.float16, torch.bfloat16)):
            raise _lib.DhdError('depth_height_head: 4-D float32 / float16 / bfloat16 GPU tensors only')
    out = _DepthHeightHead.apply(x_d, h_logits, n_depth, n_context, h_bins, hr, mr)
    return out if h_logits is not None else (out[0], out[1], None, None)


@traced('dhd.mghs.prepare')
def prepare(plan, calib, band, workspace):
    lib = _lib.load()
    dev = workspace.device
    if band is not None:
        _lib.require_gpu_tensor(band, torch.uint8, 'band')
    with torch.cuda.device(dev):
        rc = lib.dhd_mghs_prepare(C.byref(plan.desc), C.byref(calib), _lib.ptr(band), C.byref(workspace.c), _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_mghs_prepare')


def _range_arrays(height_range, mask_range):
    # float32 values exactly as torch.tensor(height_range) and the python-scalar comparisons see them
    hr = (C.c_float * len(height_range))(*torch.tensor(list(height_range), dtype=torch.float32).tolist())
    mr = (C.c_float * 4)(*torch.tensor(list(mask_range), dtype=torch.float32).tolist())
    return hr, mr


@traced('dhd.mghs.lift')
def lift(plan, calib, height, height_range, mask_range, tran_feat, workspace, static=False):
    """The lift side of MGHS.view_transform in one C call (dhd_mghs_lift: band ids from the height distribution, the
    context re-laid out to (B*N,fH,fW,C), geometry + grouping).  height may be None for a single-grid plan.  `static`:
    dhd_mghs_lift_static -- `workspace` holds an earlier lift of the same plan and calibration.
    Returns (band or None, feat_nhwc)."""
    lib = _lib.load()
    tf = _lib.require_gpu_tensor(tran_feat, torch.float32, 'tran_feat')
    bn, c, fh, fw = tf.shape
    dev = tf.device
    band, hptr, nh, hr, mr = None, None, 0, None, None
    if plan.desc.n_grids > 1:
        h = _f32(height.float().contiguous(), 'height')
        nh = h.shape[1]
        if nh != len(height_range):
            raise ValueError('height has %d bins, height_range %d' % (nh, len(height_range)))
        hr, mr = _range_arrays(height_range, mask_range)
        band = torch.empty((bn, fh, fw), dtype=torch.uint8, device=dev)
        hptr = _lib.ptr(h)
    feat_nhwc = torch.empty((bn, fh, fw, c), dtype=torch.float32, device=dev)
    fn = lib.dhd_mghs_lift_static if static else lib.dhd_mghs_lift
    with torch.cuda.device(dev):
        rc = fn(C.byref(plan.desc), C.byref(calib), hptr, nh, hr, mr, _lib.ptr(tf), _lib.ptr(band), _lib.ptr(feat_nhwc),
                C.byref(workspace.c), _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_mghs_lift_static' if static else 'dhd_mghs_lift')
    return band, feat_nhwc


def _ptr_array(tensors):
    arr = (C.c_void_p * _lib.DHD_MAX_GRIDS)()
    for i in range(_lib.DHD_MAX_GRIDS):
        arr[i] = tensors[i].data_ptr() if i < len(tensors) else None
    return arr


def pool_forward(plan, depth, feat_nhwc, workspace):
    lib = _lib.load()
    dev = depth.device
    outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in plan.out_shapes()]
    arr = _ptr_array(outs)
    with torch.cuda.device(dev):
        rc = lib.dhd_mghs_forward(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                  C.byref(workspace.c), _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_mghs_forward')
    return outs


def pool_forward_phases(plan, depth, feat_nhwc, workspace, between=None):
    """pool_forward as its two C-ABI phases; `between()` runs after the gather launch (bench.py
    records a HIP event there so that the streaming kernel is timed on its own)."""
    lib = _lib.load()
    dev = depth.device
    outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in plan.out_shapes()]
    arr = _ptr_array(outs)
    with torch.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        _lib.check(lib.dhd_mghs_forward_gather(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc),
                                               C.byref(workspace.c), st), 'dhd_mghs_forward_gather')
        if between is not None:
            between()
        _lib.check(lib.dhd_mghs_forward_stream(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                               C.byref(workspace.c), st), 'dhd_mghs_forward_stream')
    return outs


def pool_backward(plan, depth, feat_nhwc, out_grads, workspace):
    lib = _lib.load()
    dev = depth.device
    depth_grad = torch.empty_like(depth)
    bn, fh, fw, c = feat_nhwc.shape
    feat_grad = torch.empty((bn, c, fh, fw) if plan.feat_grad_nchw else (bn, fh, fw, c), dtype=torch.float32, device=dev)
    arr = _ptr_array(out_grads)
    with torch.cuda.device(dev):
        rc = lib.dhd_mghs_backward(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                   _lib.ptr(depth_grad), _lib.ptr(feat_grad), C.byref(workspace.c),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_mghs_backward')
    return depth_grad, feat_grad


# Output layouts of the pooled tensors
#   'collapsed' : per grid (B, nz*C, ny, nx), channel = z*C + c   (collapse_z=True, lss_heightmap.py:298-299)
#   'split'     : per grid (B, C, nz, ny, nx)                     (collapse_z=False, bev_pool.py:105)
#   'stacked'   : grid 0 as (B, C, nz0, ny, nx) and ALL band grids in one (B, C, sum nz, ny, nx) tensor
#                 (MGHS_Depth's bev_feat_w_z, lss_heightmap.py:845) -- written in place, no cat
LAYOUTS = ('collapsed', 'split', 'stacked')


def _alloc_outputs(plan, layout, device, dtype=torch.float32):
    d = plan.desc
    if layout == 'collapsed':
        return [torch.empty(s, dtype=dtype, device=device) for s in plan.out_shapes()]
    if layout == 'split':
        return [torch.empty((d.batch, d.channels, g.n[2], g.n[1], g.n[0]), dtype=dtype, device=device)
                for g in plan.grids]
    bands = plan.grids[1:]
    if not bands or any((g.n[0], g.n[1]) != (bands[0].n[0], bands[0].n[1]) for g in bands):
        raise ValueError("layout 'stacked' needs band grids of equal (nx, ny)")
    g0 = plan.grids[0]
    return [torch.empty((d.batch, d.channels, g0.n[2], g0.n[1], g0.n[0]), dtype=dtype, device=device),
            torch.empty((d.batch, d.channels, sum(g.n[2] for g in bands), bands[0].n[1], bands[0].n[0]),
                        dtype=dtype, device=device)]


def _views(plan, layout, tensors):
    """dhd_tensor_view per grid for tensors laid out as `layout` (see LAYOUTS)."""
    d = plan.desc
    arr = (_lib.TensorView * _lib.DHD_MAX_GRIDS)()
    zoff = 0
    for i, g in enumerate(plan.grids):
        plane = g.n[1] * g.n[0]
        if layout == 'collapsed':
            t, v = tensors[i], (g.n[2] * d.channels * plane, d.channels * plane, plane, 0)
        elif layout == 'split' or i == 0:
            t, v = tensors[i], (d.channels * g.n[2] * plane, plane, g.n[2] * plane, 0)
        else:
            t = tensors[1]
            zt = t.shape[2]
            v = (d.channels * zt * plane, plane, zt * plane, zoff * plane)
            zoff += g.n[2]
        if not t.is_contiguous() or not t.is_cuda or t.dtype != tensors[0].dtype:
            raise _lib.DhdError('pooled tensors must be contiguous GPU tensors of one dtype')
        arr[i].ptr = t.data_ptr() + t.element_size() * v[3]
        arr[i].batch_stride, arr[i].z_stride, arr[i].channel_stride = v[0], v[1], v[2]
        arr[i].dtype = _lib.dtype_code(t.dtype)
    return arr


class _MGHSPool(torch.autograd.Function):
    """depth (B*N,D,fH,fW), tran_feat (B*N,C,fH,fW) -> pooled tensors in the requested layout.  `feat_nhwc` is the
    re-laid-out context the lift produced for `workspace` (mghs_op.lift); tran_feat itself only carries the gradient."""

    @staticmethod
    @traced('dhd.mghs.pool.forward')
    def forward(ctx, depth, tran_feat, plan, workspace, feat_nhwc, layout='collapsed', out_dtype=torch.float32):
        # float32 inputs: callers cast outside the node (see mghs_pool) so that autograd casts the gradients back to whatever
        # dtype an autocast region produced.  out_dtype float16 / bfloat16: the writer rounds its float32 sums on the way out
        # (= the float32 result followed by .half(), at half the bytes) and the backward reads half gradients as they come.
        depth = _lib.require_gpu_tensor(depth.contiguous(), torch.float32, 'depth')
        lib = _lib.load()
        dev = depth.device
        outs = _alloc_outputs(plan, layout, dev, out_dtype)
        arr = _views(plan, layout, outs)
        with torch.cuda.device(dev):
            rc = lib.dhd_mghs_forward_views(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                            C.byref(workspace.c), _lib.stream_ptr(dev))
        _lib.check(rc, 'dhd_mghs_forward_views')
        ctx.plan, ctx.layout, ctx.out_dtype = plan, layout, out_dtype
        ctx.save_for_backward(depth, feat_nhwc, workspace.state)
        return tuple(outs)

    @staticmethod
    @traced('dhd.mghs.pool.backward')
    def backward(ctx, *grads):
        depth, feat_nhwc, state = ctx.saved_tensors
        plan, layout = ctx.plan, ctx.layout
        lib = _lib.load()
        dev = depth.device
        shapes = [tuple(t.shape) for t in _alloc_shapes(plan, layout)]
        gdt = ctx.out_dtype if all(g is None or g.dtype == ctx.out_dtype for g in grads) else torch.float32
        gs = [torch.zeros(s, dtype=gdt, device=dev) if g is None else g.to(gdt).contiguous() for g, s in zip(grads, shapes)]
        arr = _views(plan, layout, gs)
        bn, fh, fw, c = feat_nhwc.shape
        with torch.cuda.device(dev):
            # the scratch of the stream the backward runs on (autograd replays the forward's stream)
            ws = Workspace(state, scratch_pool.get(dev, plan.scratch_bytes, 'mghs'))
            depth_grad = torch.empty_like(depth)
            feat_grad = torch.empty((bn, c, fh, fw) if plan.feat_grad_nchw else (bn, fh, fw, c), dtype=torch.float32, device=dev)
            rc = lib.dhd_mghs_backward_views(C.byref(plan.desc), _lib.ptr(depth), _lib.ptr(feat_nhwc), C.byref(arr),
                                             _lib.ptr(depth_grad), _lib.ptr(feat_grad), C.byref(ws.c), _lib.stream_ptr(dev))
        _lib.check(rc, 'dhd_mghs_backward_views')
        return depth_grad, (feat_grad if plan.feat_grad_nchw else _nhwc_to_nchw(feat_grad)), None, None, None, None, None


class _Shape:
    def __init__(self, shape):
        self.shape = shape


def _alloc_shapes(plan, layout):
    d = plan.desc
    if layout == 'collapsed':
        return [_Shape(s) for s in plan.out_shapes()]
    if layout == 'split':
        return [_Shape((d.batch, d.channels, g.n[2], g.n[1], g.n[0])) for g in plan.grids]
    g0, bands = plan.grids[0], plan.grids[1:]
    return [_Shape((d.batch, d.channels, g0.n[2], g0.n[1], g0.n[0])),
            _Shape((d.batch, d.channels, sum(g.n[2] for g in bands), bands[0].n[1], bands[0].n[0]))]


def mghs_pool(plan, calib, band, depth, tran_feat, workspace=None, layout='collapsed', out_dtype=torch.float32):
    """Prepare (geometry + grouping) from a given band map and pool.  `workspace` may be passed to reuse memory in
    inference; under autograd a fresh state is held by the graph until backward has run."""
    if workspace is None:
        workspace = plan.new_workspace(depth.device)
    prepare(plan, calib, band, workspace)
    tf = _lib.require_gpu_tensor(tran_feat.float().contiguous(), torch.float32, 'tran_feat')
    return _MGHSPool.apply(depth.float(), tf, plan, workspace, _nchw_to_nhwc(tf.detach()), layout, out_dtype)


def mghs_lift_pool(plan, calib, height, height_range, mask_range, depth, tran_feat, workspace=None, layout='collapsed',
                   static=False, out_dtype=torch.float32):
    """MGHS.view_transform's whole device side: dhd_mghs_lift (band ids, context re-layout, geometry + grouping: four
    launches) and the pooling node.  height: (B*N, H, fH, fW) distribution or logits (only its argmax matters)."""
    if workspace is None:
        workspace = plan.new_workspace(depth.device)
    tf = tran_feat.float().contiguous()
    _, feat_nhwc = lift(plan, calib, height, height_range, mask_range, tf.detach(), workspace, static=static)
    return _MGHSPool.apply(depth.float(), tf, plan, workspace, feat_nhwc, layout, out_dtype)


def voxel_index(plan, calib, grid_index, want_ego=False):
    """Per-point voxel rank of one grid (-1 = dropped) and optionally the ego coordinates
    (B,N,D,fH,fW,3): the device twin of lss_heightmap.py:179-231 + :329-354, for parity checks and
    for the voxel_pooling_prepare_v2 mirror."""
    lib = _lib.load()
    d = plan.desc
    dev = torch.device('cuda', torch.cuda.current_device())
    npts = d.batch * d.n_cams * d.n_depth * d.fh * d.fw
    rank = torch.empty(npts, dtype=torch.int32, device=dev)
    ego = torch.empty((d.batch, d.n_cams, d.n_depth, d.fh, d.fw, 3), dtype=torch.float32, device=dev) if want_ego else None
    rc = lib.dhd_mghs_voxel_index(C.byref(d), C.byref(calib), grid_index, _lib.ptr(rank), _lib.ptr(ego),
                                  _lib.stream_ptr(dev))
    _lib.check(rc, 'dhd_mghs_voxel_index')
    return rank, ego


def debug_keys(plan, workspace):
    """dhd_mghs_debug_keys: the voxel keys of the last prepare on `workspace` as computed by the product's counting kernel,
    (2, P) int32: row 0 = global voxel id in grid 0, row 1 = in the pixel's band grid, -1 = dropped."""
    lib = _lib.load()
    d = plan.desc
    npts = d.batch * d.n_cams * d.n_depth * d.fh * d.fw
    keys = torch.empty((2, npts), dtype=torch.int32, device=workspace.device)
    with torch.cuda.device(workspace.device):
        rc = lib.dhd_mghs_debug_keys(C.byref(d), C.byref(workspace.c), _lib.ptr(keys), _lib.stream_ptr(workspace.device))
    _lib.check(rc, 'dhd_mghs_debug_keys')
    return keys


def stats(plan, workspace):
    lib = _lib.load()
    kept = (C.c_int32 * _lib.DHD_MAX_GRIDS)()
    ivs = (C.c_int32 * _lib.DHD_MAX_GRIDS)()
    with torch.cuda.device(workspace.device):
        rc = lib.dhd_mghs_stats(C.byref(plan.desc), C.byref(workspace.c), C.byref(kept), C.byref(ivs),
                                _lib.stream_ptr(workspace.device))
    _lib.check(rc, 'dhd_mghs_stats')
    n = plan.desc.n_grids
    return list(kept)[:n], list(ivs)[:n]
