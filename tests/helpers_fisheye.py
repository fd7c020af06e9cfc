"""Test helper: constructed two-view fisheye keypoint tables (used by the CPU known-answer tests and the GPU parity tests)."""
import numpy as np


def fisheye_tables(oracle, cam1, cam2, R, t, n=80, seed=3, noise=0.2):
    """n 3-D points seen by both cameras as matched keypoints (+ noise), descriptors = random rows (right = left with 3 bits
    flipped), in front of 10 'mono' keypoints per eye."""
    rng = np.random.default_rng(seed)
    P1 = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(1.0, 4.0, n)], 1).astype(np.float32)
    R = np.asarray(R, np.float32).reshape(3, 3)
    t = np.asarray(t, np.float32)
    P2 = (R.T @ (P1 - t).T).T.astype(np.float32)                    # x2 = R21 x1 + t21, R21 = R12^T, t21 = -R21 t12
    mono = 10
    kpL = np.zeros(mono + n, oracle.KEYPOINT_DT)
    kpR = np.zeros(mono + n, oracle.KEYPOINT_DT)
    for k in (kpL, kpR):
        k["x"][:mono] = rng.uniform(0, 40, mono); k["y"][:mono] = rng.uniform(0, 400, mono)
    for i in range(n):
        kpL["x"][mono + i], kpL["y"][mono + i] = oracle.kb8_project(cam1, P1[i]) + rng.normal(0, noise, 2).astype(np.float32)
        kpR["x"][mono + i], kpR["y"][mono + i] = oracle.kb8_project(cam2, P2[i]) + rng.normal(0, noise, 2).astype(np.float32)
    kpL["octave"] = rng.integers(0, 3, mono + n); kpR["octave"] = rng.integers(0, 3, mono + n)
    dL = rng.integers(0, 256, (mono + n, 32), dtype=np.uint8)
    dR = dL.copy()
    dR[:mono] = rng.integers(0, 256, (mono, 32), dtype=np.uint8)
    dR[np.arange(mono, mono + n), rng.integers(0, 32, n)] ^= 0x15
    perm = mono + rng.permutation(n)                                  # the right table in another order
    kpR[mono:], dR[mono:] = kpR[perm].copy(), dR[perm].copy()
    return P1, kpL, dL, kpR, dR, mono, perm
