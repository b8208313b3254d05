"""CPU oracle for deformable convolution v1 -- TEST INFRASTRUCTURE ONLY (see oracle/mghs_oracle.py's header:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

What it restates.  The reference's HeightNet / DepthNet append `build_conv_layer(cfg=dict(type='DCN', kernel_size=3,
padding=1, groups=4, im2col_step=128))` to their conv stacks (projects/mmdet3d_plugin/models/model_utils/depthnet.py:225-236
and :466-477).  `DCN` is mmcv-full 1.5.3's `DeformConv2dPack` (doc/install.md:7 pins the version); its source is NOT
under /root/reference and mmcv is not installed here, so this file follows the *published algorithm* of that
version's `deform_conv2d` (ops/csrc/common/cuda/deform_conv_cuda_kernel.cuh: deformable_im2col_gpu_kernel,
deformable_im2col_bilinear, deformable_col2im_gpu_kernel / get_gradient_weight, deformable_col2im_coord_gpu_kernel /
get_coordinate_weight):

  * offsets (B, deform_groups * 2 * kh * kw, Ho, Wo): channel 2t is the ROW (dy) and 2t+1 the COLUMN (dx) offset of
    tap t = i * kw + j;
  * sampling position of output pixel (ho, wo), tap (i, j):  h = ho*stride - pad + i*dil + dy,  w likewise;
  * the sample is 0 unless  -1 < h < H and -1 < w < W;  otherwise bilinear over the four integer neighbours
    (h_low = floor(h), h_high = h_low + 1, ...), each neighbour counted only when it lies inside the image;
  * columns col[(c*kh + i)*kw + j, ho, wo]; output = weight.view(g, O/g, (C/g)*kh*kw) @ col per group;
  * backward: d/dx spreads every column gradient over the same four neighbours with the same weights; d/doffset is
    the analytic derivative of the bilinear form with the floor held fixed (zero where the sample was cut to 0).

Parity status: this pins the ALGORITHM, not mmcv's bits ("parity unpinned" with respect to an mmcv binary: no
reference test, fixture or golden vector exists for it).  It is written with explicit neighbour
gathers in float64 and shares no code with the product's HIP kernel (csrc/deform.hip) or with the grid_sample
formulation in dhd_amd/depthnet.py.
"""
import numpy as np

f64 = np.float64


def _taps(offset, h, w, k, pad, dil, stride=1):
    """Sampling rows / columns (B, k*k, Ho, Wo) in float64."""
    b, _, ho, wo = offset.shape
    off = offset.astype(f64).reshape(b, k * k, 2, ho, wo)
    ii, jj = np.divmod(np.arange(k * k), k)
    base_h = (np.arange(ho) * stride - pad)[None, None, :, None] + (ii * dil)[None, :, None, None]
    base_w = (np.arange(wo) * stride - pad)[None, None, None, :] + (jj * dil)[None, :, None, None]
    return base_h + off[:, :, 0], base_w + off[:, :, 1]


def _neighbours(ph, pw, h, w):
    """Four (row, col, weight, d weight / d row, d weight / d col, valid) tuples per sampling position."""
    inside = (ph > -1) & (pw > -1) & (ph < h) & (pw < w)
    hl, wl = np.floor(ph), np.floor(pw)
    lh, lw = ph - hl, pw - wl
    hh, hw = 1 - lh, 1 - lw
    hl, wl = hl.astype(np.int64), wl.astype(np.int64)
    out = []
    for r, c, wt, dwh, dww in ((hl, wl, hh * hw, -hw, -hh), (hl, wl + 1, hh * lw, -lw, hh),
                               (hl + 1, wl, lh * hw, hw, -lh), (hl + 1, wl + 1, lh * lw, lw, lh)):
        ok = inside & (r >= 0) & (r <= h - 1) & (c >= 0) & (c <= w - 1)
        out.append((np.clip(r, 0, h - 1), np.clip(c, 0, w - 1), wt, dwh, dww, ok))
    return out


def deform_im2col(x, offset, k=3, pad=1, dil=1, stride=1):
    """x (B,C,H,W), offset (B,2*k*k,Ho,Wo) -> col (B, C*k*k, Ho*Wo) float64."""
    b, c, h, w = x.shape
    ph, pw = _taps(offset, h, w, k, pad, dil, stride)
    ho, wo = ph.shape[2:]
    xs = x.astype(f64)
    col = np.zeros((b, c, k * k, ho, wo), f64)
    bi = np.arange(b)[:, None, None, None]
    for r, cc, wt, _, _, ok in _neighbours(ph, pw, h, w):
        v = xs[bi, :, r, cc]                       # (B, kk, Ho, Wo, C)
        col += np.moveaxis(v * (wt * ok)[..., None], -1, 1)
    return col.reshape(b, c * k * k, ho * wo)


def deform_conv2d(x, offset, weight, pad=1, dil=1, groups=1, stride=1):
    """mmcv deform_conv2d forward (deform_groups = 1, no bias) -> (B, O, Ho, Wo) float64."""
    b, c, h, w = x.shape
    o, cg, k, _ = weight.shape
    ho, wo = offset.shape[2:]
    col = deform_im2col(x, offset, k, pad, dil, stride).reshape(b, groups, cg * k * k, ho * wo)
    wg = weight.astype(f64).reshape(groups, o // groups, cg * k * k)
    return np.einsum('gok,bgkp->bgop', wg, col).reshape(b, o, ho, wo)


def deform_conv2d_backward(gout, x, offset, weight, pad=1, dil=1, groups=1, stride=1):
    """Gradients of <gout, deform_conv2d(x, offset, weight)> -> (dx, doffset, dweight), float64."""
    b, c, h, w = x.shape
    o, cg, k, _ = weight.shape
    ho, wo = offset.shape[2:]
    kk = k * k
    col = deform_im2col(x, offset, k, pad, dil, stride).reshape(b, groups, cg * kk, ho * wo)
    wg = weight.astype(f64).reshape(groups, o // groups, cg * kk)
    go = gout.astype(f64).reshape(b, groups, o // groups, ho * wo)
    dweight = np.einsum('bgop,bgkp->gok', go, col).reshape(weight.shape)
    dcol = np.einsum('gok,bgop->bgkp', wg, go).reshape(b, c, kk, ho, wo)
    ph, pw = _taps(offset, h, w, k, pad, dil, stride)
    xs = x.astype(f64)
    dx = np.zeros((b, c, h, w), f64)
    doff = np.zeros((b, kk, 2, ho, wo), f64)
    bi = np.broadcast_to(np.arange(b)[:, None, None, None], ph.shape)
    for r, cc, wt, dwh, dww, ok in _neighbours(ph, pw, h, w):
        contrib = dcol * (wt * ok)[:, None]                                  # (B, C, kk, Ho, Wo)
        for ch in range(c):
            np.add.at(dx[:, ch], (bi, r, cc), contrib[:, ch])
        v = np.moveaxis(xs[bi, :, r, cc], -1, 1)                             # (B, C, kk, Ho, Wo)
        doff[:, :, 0] += (dcol * v).sum(1) * (dwh * ok)
        doff[:, :, 1] += (dcol * v).sum(1) * (dww * ok)
    return dx, doff.reshape(offset.shape), dweight
