"""Deterministic synthetic inputs for the MGHS/SFA hot path.

Everything here is built from integer hashing (splitmix64) followed by exact
float32 arithmetic, so the same arrays are regenerated bit-for-bit on any host
(no libm, no RNG-stream dependence).  Used by bench.py, the tests and the golden
generator; shapes follow SURVEY.md section 8(d) "Config 1" and "Config 2"
(reference configs: projects/configs/DHD/DHD-S.py:13-155).

Calibrations involve sin/cos and are therefore *stored* in golden fixtures
rather than regenerated when bit-exact reproduction matters.
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u32(seed, n):
    """n pseudo-random uint32 values, a pure function of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1))
        return (_splitmix64(idx ^ base) >> np.uint64(32)).astype(np.uint32)


def hash_uniform(seed, shape):
    """float32 in [0, 1) with 24 random mantissa bits (exact)."""
    n = int(np.prod(shape))
    u = (hash_u32(seed, n) >> np.uint32(8)).astype(np.float32)
    return (u * np.float32(1.0 / (1 << 24))).reshape(shape)


def hash_signed(seed, shape):
    """Roughly bell-shaped float32 in (-2, 2): sum of four 12-bit uniforms, exact."""
    n = int(np.prod(shape))
    a = hash_u32(seed, n)
    b = hash_u32(seed + 7919, n)
    s = ((a & np.uint32(0xFFF)).astype(np.int64) + ((a >> np.uint32(12)) & np.uint32(0xFFF))
         + (b & np.uint32(0xFFF)) + ((b >> np.uint32(12)) & np.uint32(0xFFF)) - 2 * 4095)
    return (s.astype(np.float32) * np.float32(1.0 / 4096.0)).reshape(shape)


def depth_like(seed, shape):
    """Positive, peaky float32 weights k/4096 (exact); stands in for softmax(depth)."""
    n = int(np.prod(shape))
    w = (hash_u32(seed, n) & np.uint32(0xFFF)).astype(np.int64)
    w = (w * w) >> 12
    w = (w * w) >> 12
    return ((w + 1).astype(np.float32) * np.float32(1.0 / 4096.0)).reshape(shape)


def height_index(seed, shape, n_bins):
    """Per-pixel argmax index of the height distribution, uint8 in [0, n_bins)."""
    n = int(np.prod(shape))
    return (hash_u32(seed, n) % np.uint32(n_bins)).astype(np.uint8).reshape(shape)


def height_probs_from_index(idx, n_bins):
    """A (.., n_bins, fH, fW)-style softmax-like tensor whose argmax over the bin
    axis (axis 1) is `idx` (BN, fH, fW).  Values are exact multiples of 2^-10."""
    bn, fh, fw = idx.shape
    p = np.full((bn, n_bins, fh, fw), 1.0 / 1024.0, dtype=np.float32)
    b, h, w = np.meshgrid(np.arange(bn), np.arange(fh), np.arange(fw), indexing="ij")
    p[b, idx.astype(np.int64), h, w] = np.float32(0.5)
    return p


def _rot_xyz(rx, ry, rz):
    cx, sx = math.cos(rx), math.sin(rx)
    cy, sy = math.cos(ry), math.sin(ry)
    cz, sz = math.cos(rz), math.sin(rz)
    mx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    my = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    mz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return mz @ my @ mx


def make_calibration(seed, batch, n_cams, input_size=(256, 704), src_size=(900, 1600)):
    """Randomised (non axis-aligned) multi-camera ring calibration.

    Mirrors the ranges the reference's image augmentation produces
    (datasets/pipelines/loading.py:55-146 with DHD-S.py:15-33): resize
    U(-0.06, 0.11) around input_w/src_w, rotation U(-5.4, 5.4) deg, random flip,
    crop; bda = diag(+-1, +-1, 1) (DHD-S.py:161-166).

    Returns float32 arrays: sensor2ego (B,N,4,4), ego2global (B,N,4,4),
    intrin (B,N,3,3), post_rot (B,N,3,3), post_tran (B,N,3), bda (B,3,3).
    """
    u = hash_uniform(seed, (batch, n_cams + 1, 16)).astype(np.float64)
    fin_h, fin_w = input_size
    src_h, src_w = src_size
    s2e = np.zeros((batch, n_cams, 4, 4), np.float64)
    intrin = np.zeros((batch, n_cams, 3, 3), np.float64)
    post_rot = np.zeros((batch, n_cams, 3, 3), np.float64)
    post_tran = np.zeros((batch, n_cams, 3), np.float64)
    bda = np.zeros((batch, 3, 3), np.float64)
    cam_axes = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])  # camera (x right, y down, z fwd) -> ego
    step = 2.0 * math.pi / n_cams
    for b in range(batch):
        for n in range(n_cams):
            r = u[b, n]
            jit = (r[0:3] * 2 - 1) * math.radians(3.0)
            rot = _rot_xyz(jit[0], jit[1], step * n + jit[2]) @ cam_axes
            s2e[b, n, :3, :3] = rot
            rad = 0.5 + 0.5 * r[3]
            s2e[b, n, :3, 3] = [rad * math.cos(step * n), rad * math.sin(step * n), 1.5 + 0.1 * (r[4] * 2 - 1)]
            s2e[b, n, 3, 3] = 1.0
            f = 1266.0 * (1 + 0.02 * (r[5] * 2 - 1))
            intrin[b, n] = [[f, 0, 816.0 * (1 + 0.02 * (r[6] * 2 - 1))],
                            [0, f * (1 + 0.002 * (r[7] - 0.5)), 491.0 * (1 + 0.02 * (r[8] * 2 - 1))],
                            [0, 0, 1]]
            resize = fin_w / src_w + (-0.06 + 0.17 * r[9])
            ang = math.radians(5.4) * (r[10] * 2 - 1)
            flip = r[11] < 0.5
            new_w, new_h = src_w * resize, src_h * resize
            crop_h = int(new_h) - fin_h
            crop_w = int(r[12] * max(0.0, new_w - fin_w))
            a = np.eye(2) * resize
            t = -np.array([crop_w, crop_h], np.float64)
            if flip:
                fm = np.array([[-1.0, 0], [0, 1]])
                a = fm @ a
                t = fm @ t + np.array([fin_w, 0.0])
            rm = np.array([[math.cos(ang), math.sin(ang)], [-math.sin(ang), math.cos(ang)]])
            ctr = np.array([fin_w, fin_h]) / 2.0
            t = rm @ (t - ctr) + ctr
            a = rm @ a
            post_rot[b, n, :2, :2] = a
            post_rot[b, n, 2, 2] = 1.0
            post_tran[b, n, :2] = t
        rb = u[b, n_cams]
        bda[b] = np.diag([1.0 if rb[0] < 0.5 else -1.0, 1.0 if rb[1] < 0.5 else -1.0, 1.0])
    e2g = np.broadcast_to(np.eye(4), (batch, n_cams, 4, 4)).copy()
    f32 = np.float32
    return (s2e.astype(f32), e2g.astype(f32), intrin.astype(f32), post_rot.astype(f32),
            post_tran.astype(f32), bda.astype(f32))


# ---- the two standard shapes -------------------------------------------------

def dhd_s_config():
    """DHD-S view-transformer hyper-parameters (projects/configs/DHD/DHD-S.py:33-98)."""
    hr = [round(-1.0 + 0.1 * i, 1) for i in range(65)]
    return dict(
        grid_config={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 6.4], 'depth': [1.0, 45.0, 1.0]},
        input_size=(256, 704), downsample=16, in_channels=256, out_channels=64,
        height_range=hr, height_interval=0.1, mask_range=[-1.0, 0.6, 2.2, 5.4], loss_height_weight=0.1,
        mask_1_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 0.6, 0.4], 'depth': [1.0, 45.0, 0.5]},
        mask_2_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [0.6, 2.2, 0.4], 'depth': [1.0, 45.0, 0.5]},
        mask_3_grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [2.2, 5.4, 0.4], 'depth': [1.0, 45.0, 0.5]},
        sid=False, collapse_z=True)


def smoke_config():
    """1 camera 64x176, 50x50 band grids, 17 height bins, C=16."""
    hr = [round(-1.0 + 0.1 * i, 1) for i in range(17)]
    return dict(
        grid_config={'x': [-10, 10, 0.4], 'y': [-10, 10, 0.4], 'z': [-1, 0.6, 1.6], 'depth': [1.0, 45.0, 1.0]},
        input_size=(64, 176), downsample=16, in_channels=32, out_channels=16,
        height_range=hr, height_interval=0.1, mask_range=[-1.0, -0.6, -0.2, 0.6], loss_height_weight=0.1,
        mask_1_grid={'x': [-10, 10, 0.4], 'y': [-10, 10, 0.4], 'z': [-1, -0.6, 0.4], 'depth': [1.0, 45.0, 0.5]},
        mask_2_grid={'x': [-10, 10, 0.4], 'y': [-10, 10, 0.4], 'z': [-0.6, -0.2, 0.4], 'depth': [1.0, 45.0, 0.5]},
        mask_3_grid={'x': [-10, 10, 0.4], 'y': [-10, 10, 0.4], 'z': [-0.2, 0.6, 0.4], 'depth': [1.0, 45.0, 0.5]},
        sid=False, collapse_z=True)


def lift_inputs(seed, batch, n_cams, n_depth, fh, fw, channels, n_height):
    """depth (B*N,D,fH,fW), tran_feat (B*N,C,fH,fW), height index (B*N,fH,fW) uint8."""
    bn = batch * n_cams
    depth = depth_like(seed + 1, (bn, n_depth, fh, fw))
    feat = hash_signed(seed + 2, (bn, channels, fh, fw))
    hidx = height_index(seed + 3, (bn, fh, fw), n_height)
    return depth, feat, hidx


def hashed_state(shapes, seed):
    """name -> float32 array for the floating-point entries of a state dict, a pure function of
    (seed, name, shape) -- independent of the order of the entries: matrices / kernels ~ signed / sqrt(fan_in), 1-D `weight` and `running_var`
    in [0.5, 1.5), biases and `running_mean` ~ 0.1 * signed.  Lets a golden fixture hold only outputs."""
    import zlib
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        j = zlib.crc32(name.encode()) % 1000003
        leaf = name.rsplit('.', 1)[-1]
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = hash_signed(seed + j, shape) * np.float32(1.0 / math.sqrt(fan_in))
        elif leaf in ('weight', 'running_var'):
            a = hash_uniform(seed + j, shape) + np.float32(0.5)
        else:
            a = hash_signed(seed + j, shape) * np.float32(0.1)
        out[name] = a.astype(np.float32)
    return out
