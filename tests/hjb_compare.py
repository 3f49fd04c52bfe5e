"""Shared comparison helpers of the full-size HJB tests (checker side only).

Why two phi metrics.  The reference integrates a diffusion equation with an EXPLICIT Runge-Kutta pair whose step the
controller pushes to the stability limit (h = 3.3 / (8 (sigma^2/2)/dx^2)); there the grid's highest
mode -- the checkerboard (-1)^(i+j) -- is amplified a little in every step until it shows up in the error estimate.
Rounding noise is its seed: two float64 implementations that differ only in summation order already disagree by
1.7e-6 in that mode after a T = 20 solve (amplification 1e10, measured: tests/diagnostics/dbg_forced.py), and float32 seeds
(6e-8) saturate at whatever the controller tolerates, which is an RMS over ALL cells (scipy common.py:63-66): on a
2048^2 grid a ripple of a few per cent confined to the thousand cells next to a door still has an error norm < 1.
The ripple is invisible to what the path consumes: central differences (optimals.py:174-175) take both neighbours from
the same checkerboard colour, so a ripple of locally uniform amplitude cancels in the direction field.  Hence: a loose
pointwise bound that documents the ripple, a tighter one on the 2x2-block average (which removes the mode where its
amplitude varies slowly), and 1e-2 rad on the direction field for all but a stated fraction of the cells (those where
the ripple's amplitude changes from one cell to the next, next to doors and wall corners).  On a 200 x 200 grid the
norm dilutes less and SURVEY 8d's figures (5e-3, 1e-2 rad everywhere) hold as they stand.
"""
import numpy as np

from oracle import oc_oracle as oc


def block_average(a, Ny, Nx):
    a = a.reshape(Ny, Nx)
    return 0.25 * (a[:-1, :-1] + a[:-1, 1:] + a[1:, :-1] + a[1:, 1:])


def phi_errors(got, want, V, Ny, Nx):
    """(largest pointwise relative error where phi > lim on free cells, the same for the 2x2-block averages)"""
    free = (V.reshape(Ny, Nx) >= 0)
    sel = free.ravel() & (want > oc.LIM)
    point = float((np.abs(got - want)[sel] / want[sel]).max())
    fb = free[:-1, :-1] & free[:-1, 1:] & free[1:, :-1] & free[1:, 1:]
    w = block_average(want, Ny, Nx)
    fb &= w > oc.LIM
    avg = float((np.abs(block_average(got, Ny, Nx) - w)[fb] / w[fb]).max())
    return point, avg


def clear_cells(phi, Ny, Nx, gs, mu):
    """cells whose |v| (oracle phi) is at least 10 % away from the unit-or-zero threshold `lim`"""
    ph = phi.reshape(Ny, Nx)
    p = ph * (ph > oc.LIM) + oc.LIM * (ph < oc.LIM)
    nrm = np.hypot((p[1:-1, 2:] - p[1:-1, :-2]) / (2 * gs), (p[2:, 1:-1] - p[:-2, 1:-1]) / (2 * gs)) / (mu * p[1:-1, 1:-1])
    return np.abs(nrm - oc.LIM) > 0.1 * oc.LIM


def direction_errors(gx, gy, hx, hy, clear):
    """(angles between the two unit fields on cells that move in both and are clear of the threshold, flip fraction)"""
    both = (np.hypot(gx, gy) > 0.5) & (np.hypot(hx, hy) > 0.5)
    ang = np.abs(np.arctan2(gx * hy - gy * hx, gx * hx + gy * hy))
    flip = (np.hypot(gx, gy) > 0.5) != (np.hypot(hx, hy) > 0.5)
    return ang[both & clear], float(flip.mean())


def describe(a):
    """summary of an error sample for assertion messages"""
    if a.size == 0:
        return "empty"
    return "n=%d max=%.3e p99=%.3e p99.99=%.3e frac>1e-2=%.2e" % (
        a.size, a.max(), np.percentile(a, 99), np.percentile(a, 99.99), (a > 1e-2).mean())


def assert_steps_acceptable(err, limit=1.1):
    """err: the float64 oracle's error norm of every step the float32 GPU solve accepted.  The GPU's own norm of the same
    step is a float32 evaluation of a sum that hovers just below 1 (the controller aims there), so a step or two may read
    1.0x in float64; none may be a step the oracle would clearly have rejected."""
    assert (err < limit).all() and (err >= 1.0).sum() <= 3, (float(err.max()), int((err >= 1.0).sum()))


def record_margin(config, slice_index, phi_point, phi_avg, angles, flips, err=None, mode="float32", against="forced"):
    """One line of measured parity margins per (config, slice) into gpurun_out/parity_margins.jsonl (merged back from the GPU
    box; tools/collect_profiles.py summarises it into profiles/rNN_parity_margins.json).  The float32 bounds of the full-size
    tests are stated as <= 1.5 x the largest value on record: the ripple's amplitude is chaotic in the step sequence, so a
    regression shows up here -- and in the test -- instead of hiding under a bound with 2 x headroom."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    a = np.asarray(angles, dtype=np.float64)
    rec = {"config": config, "mode": mode, "against": against, "slice": int(slice_index), "phi_point": float(phi_point),
           "phi_avg": float(phi_avg), "angle_max": float(a.max()) if a.size else None,
           "angle_p99": float(np.percentile(a, 99)) if a.size else None, "frac_gt_1e-2": float((a > 1e-2).mean()) if a.size else None,
           "clear_cells": int(a.size), "flips": float(flips),
           "max_err_norm": None if err is None else float(np.max(err)),
           "steps_at_or_above_1": None if err is None else int((np.asarray(err) >= 1.0).sum())}
    with open(os.path.join(d, "parity_margins.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    return rec
