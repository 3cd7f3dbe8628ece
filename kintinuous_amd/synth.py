"""Deterministic synthetic RGB-D sequences for tests and bench.py.

Scenes are analytic (axis-aligned room / wall + spheres + a cube) and rendered by exact ray casting in
numpy, so every frame is reproducible bit for bit from (scene, trajectory, frame index) alone -- there
is no dataset and no network in this environment.

Conventions: camera frame x right, y down, z forward (metres).  Scene coordinates == the frame of camera 0.
Depth = z-depth in millimetres (uint16, 0 = no return); rgb24 = bytes as they sit in a raw .klg log.
The tracker's volume frame is scene + volume_basis (the initial pose is R = I, t = volume_basis,
KintinuousTracker.cpp:101-110).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

FX = FY = 528.01442863461716  # MainController.cpp:222-226
CX, CY = 320.0, 267.0


@dataclass
class Camera:
    cols: int = 640
    rows: int = 480
    fx: float = FX
    fy: float = FY
    cx: float = CX
    cy: float = CY

    @staticmethod
    def scaled(scale: int) -> "Camera":
        return Camera(640 * scale, 480 * scale, FX * scale, FY * scale, CX * scale, CY * scale)

    @staticmethod
    def small(cols: int, rows: int) -> "Camera":
        """Reduced-resolution camera with the same field of view (for fast CPU-checked tests)."""
        s = cols / 640.0
        return Camera(cols, rows, FX * s, FY * s, CX * s, CY * s)


def _rot_y(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


# ---- trajectories: list of (R_wc, c_w) = camera-to-scene rotation and camera centre ------------------
def orbit_trajectory(n: int = 300, period: int = 300) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Config 2: gentle orbit about the point 2 m ahead plus a small translation loop (<= ~12 mm/frame)."""
    out = []
    pivot = np.array([0.0, 0.0, 2.0])
    for k in range(n):
        ph = 2 * np.pi * k / period
        th = 0.25 * np.sin(ph)
        R = _rot_y(th) @ _rot_x(0.05 * np.sin(2 * ph))
        t = np.array([0.3 * np.sin(ph), 0.05 * np.sin(2 * ph), 0.3 * (1 - np.cos(ph))]) * 0.5
        c = pivot - R @ pivot + t
        out.append((R, c))
    return out


def crabwalk_trajectory(n: int = 420) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Config 3: face the wall and crab-walk +x for 200 frames, back for 200, rest for 20 (forces volume shifts)."""
    out = []
    x = 0.0
    for k in range(n):
        if 0 < k <= 200:
            x += 0.015
        elif 200 < k <= 400:
            x -= 0.015
        z = 0.3 * np.sin(2 * np.pi * k / 420)
        y = 0.04 * np.sin(4 * np.pi * k / 420)
        yaw = 0.05 * np.sin(2 * np.pi * k / 210)
        out.append((_rot_y(yaw), np.array([x, y, z])))
    return out


def static_trajectory(n: int = 50) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Config 5: <= 2 mm / frame drift in front of a far wall."""
    return [(_rot_y(0.0005 * k), np.array([0.002 * k * 0.5, 0.0, 0.0])) for k in range(n)]


# ---- scenes ------------------------------------------------------------------------------------------
@dataclass
class Scene:
    kind: str = "room"  # "room" (S0), "wall" (S1), "farwall" (config 5)
    seed: int = 1234

    def _offsets(self):
        rng = np.random.default_rng(self.seed)
        d = rng.uniform(-0.15, 0.15, size=6) if self.seed != 1234 else np.zeros(6)
        return d

    def spheres(self, cam_x: float = 0.0):
        d = self._offsets()
        if self.kind == "room":
            return [((0.3 + d[0], 0.2 + d[1], 1.8 + d[2]), 0.4)]
        if self.kind == "wall":
            j0 = int(np.floor(cam_x / 0.8))
            return [((0.8 * j, 0.3 * (-1) ** j, 2.1), 0.25) for j in range(j0 - 4, j0 + 6)]
        return [((0.3, 0.2, 1.8), 0.4)]

    def cubes(self):
        d = self._offsets()
        if self.kind in ("room", "farwall"):
            c = np.array([-0.7 + d[3], 0.4 + d[4], 2.2 + d[5]])
            return [(c - 0.25, c + 0.25)]
        return []

    def planes_box(self):
        """Inward-facing bounding box (lo, hi) the camera sits inside; None entries are open sides."""
        if self.kind == "room":
            return np.array([-1.6, -1.1, -2.2]), np.array([1.6, 1.0, 2.8])
        if self.kind == "wall":
            return np.array([-1e9, -1.1, -1e9]), np.array([1e9, 1.0, 2.6])
        return np.array([-1e9, -1e9, -1e9]), np.array([1e9, 1e9, 6.2])


def render(scene: Scene, cam: Camera, R: np.ndarray, c: np.ndarray, noise_mm: float = 0.0, rng=None) -> Tuple[np.ndarray, np.ndarray]:
    """Exact z-depth (uint16 mm) and textured rgb24 for one pose."""
    u, v = np.meshgrid(np.arange(cam.cols, dtype=np.float64), np.arange(cam.rows, dtype=np.float64))
    d_c = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], axis=-1)  # z component 1 => t == z-depth
    d = d_c @ R.T
    o = c.astype(np.float64)
    t_hit = np.full(u.shape, np.inf)

    lo, hi = scene.planes_box()
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax in range(3):
            for bound, sign in ((hi[ax], 1.0), (lo[ax], -1.0)):
                if abs(bound) > 1e8:
                    continue
                t = (bound - o[ax]) / d[..., ax]
                ok = (t > 1e-6) & (d[..., ax] * sign > 0)
                p = o + d * t[..., None]
                for a2 in range(3):
                    if a2 != ax:
                        ok &= (p[..., a2] >= lo[a2] - 1e-9) & (p[..., a2] <= hi[a2] + 1e-9)
                t_hit = np.where(ok & (t < t_hit), t, t_hit)
        for (sc, r) in scene.spheres(float(o[0])):
            oc = o - np.array(sc)
            a = np.sum(d * d, axis=-1)
            b = 2 * np.sum(d * oc, axis=-1)
            cc = float(oc @ oc) - r * r
            disc = b * b - 4 * a * cc
            t = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
            ok = (disc > 0) & (t > 1e-6)
            t_hit = np.where(ok & (t < t_hit), t, t_hit)
        for (blo, bhi) in scene.cubes():
            t1 = (blo - o) / d
            t2 = (bhi - o) / d
            tn = np.max(np.minimum(t1, t2), axis=-1)
            tf = np.min(np.maximum(t1, t2), axis=-1)
            ok = (tn < tf) & (tn > 1e-6)
            t_hit = np.where(ok & (tn < t_hit), tn, t_hit)

    valid = np.isfinite(t_hit) & (t_hit < 8.0)
    z = np.where(valid, t_hit, 0.0)
    if noise_mm > 0:
        rng = rng or np.random.default_rng(0)
        z = z + np.where(valid, rng.normal(0.0, noise_mm * 1e-3, size=z.shape), 0.0)
    depth = np.where(valid, np.rint(1000.0 * z), 0).astype(np.uint16)

    p = o + d * np.where(valid, t_hit, 0.0)[..., None]
    rgb = np.zeros((cam.rows, cam.cols, 3), dtype=np.uint8)
    for ch, ph in enumerate((0.0, 2.1, 4.2)):
        val = 128 + 60 * np.sin(9 * p[..., 0] + ph) * np.sin(7 * p[..., 1] + ph) + 40 * np.sin(11 * p[..., 2] + ph)
        rgb[..., ch] = np.clip(np.rint(val), 1, 255).astype(np.uint8)
    return depth, rgb


def sequence(config: str, n: int, cam: Camera = None, seed: int = 1234):
    """Returns (camera, frames[(depth, rgb)], gt_poses[(R, c)], tracker kwargs) for a BASELINE config name."""
    cam = cam or Camera()
    if config == "orbit":  # configs 1, 2, 4
        scene, traj, kw = Scene("room", seed), orbit_trajectory(n), dict(volume_size=6.0)
    elif config == "crabwalk":  # config 3
        scene, traj, kw = Scene("wall", seed), crabwalk_trajectory(n), dict(volume_size=7.0, use_rgbd_icp=1)
    elif config == "farwall":  # config 5
        scene, traj, kw = Scene("farwall", seed), static_trajectory(n), dict(volume_size=6.0, static_mode=1)
    else:
        raise ValueError(config)
    frames = [render(scene, cam, R, c) for (R, c) in traj[:n]]
    return cam, frames, traj[:n], kw


def ground_truth_rows(poses) -> np.ndarray:
    """Camera-to-world poses (R, c) in the tracker's camera convention (x right, y down, z forward) -> the rows
    x y z qx qy qz qw of a -p trajectory file (x forward, y left, z up): T = M C M^-1, so that the tracker's
    M^-1 (Ta^-1 Tb) M is Ca^-1 Cb (GroundTruthOdometry.cpp:52-68).  float32 [n, 7]."""
    from scipy.spatial.transform import Rotation
    M = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float64)
    rows = []
    for R, c in poses:
        C4 = np.eye(4)
        C4[:3, :3], C4[:3, 3] = R, c
        T = M @ C4 @ M.T
        rows.append(np.concatenate([T[:3, 3], Rotation.from_matrix(T[:3, :3]).as_quat()]))
    return np.array(rows, np.float32)


def write_trajectory_file(path: str, stamps, rows) -> None:
    """-p file format of KintinuousTracker::loadTrajectory: utime,x,y,z,qx,qy,qz,qw per line."""
    with open(path, "w") as f:
        for ts, r in zip(stamps, rows):
            f.write(str(int(ts)) + "," + ",".join(repr(float(np.float32(v))) for v in r) + "\n")
