"""Synthetic input generators shared by tests and bench.py (data only -- no reference or oracle code).
PnP scene of SURVEY.md 8d: N correspondences, points in the frustum of camera b at 0.5-20 m rounded to float32
(mirrors the CV_32FC3 depth image, PointFeatureMatching.cpp:124-141), pose yaw U(-30,30) deg, pitch/roll U(-5,5) deg,
|t| <= 1 m, pixel noise at f = 458 (EuRoC-like), a fraction of uniformly random outliers."""
from __future__ import annotations

import numpy as np


def make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242, focal=458.0):
    rng = np.random.default_rng(seed)
    yaw = np.deg2rad(rng.uniform(-30, 30)); pitch = np.deg2rad(rng.uniform(-5, 5)); roll = np.deg2rad(rng.uniform(-5, 5))
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    R = Ry @ Rx @ Rz
    t = rng.standard_normal(3); t *= rng.uniform(0, 1) / np.linalg.norm(t)
    # points in the frustum of camera b, depth 0.5..20 m, expressed in frame a, rounded to float32 (CV_32FC3)
    depth = rng.uniform(0.5, 20.0, N)
    uvb = np.stack([rng.uniform(-0.8, 0.8, N), rng.uniform(-0.5, 0.5, N)], axis=1)
    Xb = np.concatenate([uvb * depth[:, None], depth[:, None]], axis=1)
    Xa = ((Xb - t) @ R).astype(np.float32).astype(np.float64)           # X_a = R^T (X_b - t)
    proj = Xa @ R.T + t
    uv = proj[:, :2] / proj[:, 2:3] + rng.standard_normal((N, 2)) * (noise_px / focal)
    out = rng.random(N) < outlier_frac
    uv[out] = np.stack([rng.uniform(-0.8, 0.8, out.sum()), rng.uniform(-0.5, 0.5, out.sum())], axis=1)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return Xa, uv, T, ~out


def make_icp_scene(N=300, outlier_frac=0.2, noise=0.02, seed=1):
    """3-D / 3-D correspondences for the Umeyama-ICP-RANSAC row (P3P_ICP, DlsPnpWithRansac.cpp:15-16: uv_X, uvd_Y): the PnP scene's
    points in frame a, the same points in frame b with Gaussian noise, a fraction displaced by up to 3 m."""
    X, uv, T, inl = make_scene(N=N, outlier_frac=0.0, noise_px=0.0, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    A = X
    B = X @ T[:3, :3].T + T[:3, 3] + rng.standard_normal((N, 3)) * noise
    out = rng.random(N) < outlier_frac
    B[out] += rng.uniform(-3, 3, (out.sum(), 3))
    return A, B, T, ~out
