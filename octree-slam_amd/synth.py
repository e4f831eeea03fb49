"""Synthetic RGB-D stream of BASELINE.md config 3/4: an analytic room (axis-aligned
box 6 x 3 x 6 m) with one sphere (r = 0.5 m), seen by a pinhole camera that moves
on a circle of radius 0.5 m at height 1.5 m and yaws +0.1 deg per frame.

Stands in for sensor::OpenNIDevice (src/sensor/openni_device.cpp:96-150): it
produces what readFrame() would hand to the pipeline -- depth uint16 in mm
(0 = no measurement) and RGB888, row-major, origin top-left -- plus the focal
length (f = 570.3 * W/640).  Depth = round(1000 z) with additive Gaussian noise
(sigma = 2 mm, seed 5678) and 1 % dropouts (seed 1234); colour = 3-D checker of
the world position (cell 0.25 m).  Written with torch ops so the same code
generates on the CPU (tests) and directly in HBM (bench); the two devices use
different random streams (torch.Generator per device), both fixed by the seeds.
"""
import math

import torch

ROOM_MIN = (-3.0, 0.0, -3.0)
ROOM_MAX = (3.0, 3.0, 3.0)
SPHERE_C = (0.9, 0.8, 1.6)
SPHERE_R = 0.5
CHECKER = 0.25
SEED_DROPOUT = 1234
SEED_NOISE = 5678


def focal_length(width):
    return 570.3 * width / 640.0


def camera_pose(frame, radius=0.5, height=1.5, deg_per_frame=0.1):
    """Ground-truth camera-to-world pose of frame k: position (3,), yaw (rad)."""
    th = math.radians(deg_per_frame * frame)
    return (radius * math.cos(th), height, radius * math.sin(th)), th


def render_frame(frame, width, height, device="cpu", noise_sigma_mm=2.0, dropout=0.01, dtype=torch.float64):
    """Returns (depth int16-viewed-uint16 [H,W], rgb uint8 [H,W,3]) on `device`.

    The depth tensor has dtype torch.int16 holding the uint16 bit pattern (torch has
    no first-class uint16); view it with .cpu().numpy().view(numpy.uint16)."""
    f = focal_length(width)
    (px, py, pz), yaw = camera_pose(frame)
    u = torch.arange(width, device=device, dtype=dtype)
    v = torch.arange(height, device=device, dtype=dtype)
    # camera frame of image_kernels.cu:49-51: x right, y up, z forward
    dx = ((u - width / 2.0) / f)[None, :].expand(height, width)
    dy = ((height / 2.0 - v) / f)[:, None].expand(height, width)
    dz = torch.ones((height, width), device=device, dtype=dtype)
    c, s = math.cos(yaw), math.sin(yaw)
    # yaw about +y; camera forward at yaw 0 is +z
    wx = c * dx + s * dz
    wy = dy
    wz = -s * dx + c * dz
    big = torch.full((height, width), 1e30, device=device, dtype=dtype)
    t = big.clone()
    # box interior: the exit distance along each axis
    for w_, o, lo, hi in ((wx, px, ROOM_MIN[0], ROOM_MAX[0]), (wy, py, ROOM_MIN[1], ROOM_MAX[1]), (wz, pz, ROOM_MIN[2], ROOM_MAX[2])):
        tt = torch.where(w_ > 0, (hi - o) / w_, torch.where(w_ < 0, (lo - o) / w_, big))
        t = torch.minimum(t, tt)
    # sphere
    ox, oy, oz = px - SPHERE_C[0], py - SPHERE_C[1], pz - SPHERE_C[2]
    a = wx * wx + wy * wy + wz * wz
    b = 2.0 * (ox * wx + oy * wy + oz * wz)
    cc = ox * ox + oy * oy + oz * oz - SPHERE_R * SPHERE_R
    disc = b * b - 4.0 * a * cc
    ts = torch.where(disc > 0, (-b - torch.sqrt(torch.clamp(disc, min=0.0))) / (2.0 * a), big)
    hit_sphere = (ts > 0) & (ts < t)
    t = torch.where(hit_sphere, ts, t)
    # world hit point -> checker colour
    hx, hy, hz = px + t * wx, py + t * wy, pz + t * wz
    eps = 1e-6
    ix = torch.floor(hx / CHECKER + eps).to(torch.int64)
    iy = torch.floor(hy / CHECKER + eps).to(torch.int64)
    iz = torch.floor(hz / CHECKER + eps).to(torch.int64)
    par = (ix + iy + iz) & 1
    r = torch.where(par == 1, 210, 60) + (ix & 3) * 8
    g = torch.where(par == 1, 180, 70) + (iy & 3) * 8
    bl = torch.where(par == 1, 90, 200) + (iz & 3) * 8
    r = torch.where(hit_sphere, 240 - (par * 100), r)
    rgb = torch.stack([r, g, bl], dim=-1).clamp(0, 255).to(torch.uint8)
    # depth in mm (z of the camera frame == t because dz == 1)
    gen_n = torch.Generator(device=device); gen_n.manual_seed(SEED_NOISE + 7919 * frame)
    gen_d = torch.Generator(device=device); gen_d.manual_seed(SEED_DROPOUT + 104729 * frame)
    noise = torch.randn((height, width), generator=gen_n, device=device, dtype=torch.float32).to(dtype) * noise_sigma_mm
    depth_mm = torch.round(1000.0 * t + noise).clamp(0, 65535)
    drop = torch.rand((height, width), generator=gen_d, device=device, dtype=torch.float32) < dropout
    depth_mm = torch.where(drop, torch.zeros_like(depth_mm), depth_mm)
    d16 = depth_mm.to(torch.int32)
    d16 = torch.where(d16 >= 32768, d16 - 65536, d16).to(torch.int16)  # uint16 bit pattern
    return d16.contiguous(), rgb.contiguous()


def render_stream(num_frames, width, height, device="cpu", start=0):
    """Stacked frames: depth [K,H,W] int16(uint16 bits), rgb [K,H,W,3] uint8."""
    ds, cs = [], []
    for k in range(start, start + num_frames):
        d, c = render_frame(k, width, height, device=device)
        ds.append(d); cs.append(c)
    return torch.stack(ds), torch.stack(cs)
