"""Mirror of ``gp_edge_tracing/gpet_utils.py`` for the parts on (or feeding) the hot path.

``comp_grad_img`` / ``normalise`` run on the GPU through libgpet_hip.so (a1); ``kernel_builder``
is host-side setup (a 11x5 table).  ``construct_test_img`` is this package's own generator of the
reference's synthetic test image recipe (gpet_utils.py:163-253).  Called like the reference (no ``seed``) it returns the
reference's own image bit for bit: scikit-image 0.18's ``random_noise(..., seed=1)`` (gpet_utils.py:251) is
``np.random.seed(1); image + np.random.normal(0, sqrt(var), shape)`` clipped to [0, 1] -- numpy's frozen legacy stream, which
needs no scikit-image (pinned by tests/golden/readme_image.npz, made by the unmodified reference under skimage 0.18.3).  With an
explicit ``seed`` the noise comes from numpy's Generator: independent images for batches.  Metrics restate gpet_utils.py:256-313.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib

_default_ctx = None


def _ctx():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(0)
    return _default_ctx


def kernel_builder(size, b2d=False, normalize=False, vertical_edges=False, unit=False):
    """Sobel-like (rows x cols) edge kernel (gpet_utils.py:10-61)."""
    rows, cols = size
    mid_r, mid_c = rows // 2, cols // 2
    kernel = np.zeros(size)
    for i in range(mid_r):
        for j in range(cols):
            kernel[i, j] = 1 if unit else 1 + max(0, mid_r + 1 - abs(i - mid_r) - abs(j - mid_c))
    kernel[mid_r + 1:, :] = -kernel[0:mid_r, :][::-1]
    if b2d:
        kernel = np.flipud(kernel)
    if vertical_edges:
        kernel = kernel.T
    if normalize:
        kernel = kernel / kernel.max()
    return kernel


def normalise(img, minmax_val=(0, 1), astyp=np.float32, ctx=None):
    """float32 min-max normalisation on the GPU (gpet_utils.py:65-91)."""
    lo, hi = minmax_val
    out = (ctx or _ctx()).normalise_f32(np.asarray(img).astype(np.float32))
    if (lo, hi) != (0, 1):
        out = out * np.float32(hi - lo) + np.float32(lo)
    return out.astype(astyp)


def comp_grad_img(img, kernel, norm=True, astyp=np.float32, ctx=None):
    """Image gradient: convolution (clamp-to-edge) + ReLU + float32 min-max, one fused GPU pass
    (gpet_utils.py:95-119).  Like the reference it always normalises (``norm`` is ignored there:
    ``if normalise:`` tests the function object, gpet_utils.py:114)."""
    out = (ctx or _ctx()).grad_image(np.asarray(img, dtype=np.float64), np.asarray(kernel, dtype=np.float64))
    return out.astype(astyp)


def construct_test_img(size, amplitude, curvature, noise_level, ltype, intensity, gaps=False, seed=None):
    """Synthetic step-edge image + ground-truth edge (yx), recipe of gpet_utils.py:163-253.  ``seed=None`` (the reference's
    signature has no seed): the reference's own noise, ``random_noise(..., seed=1)`` of scikit-image 0.18 = numpy's legacy
    ``RandomState(1).normal``; an integer: numpy ``Generator(seed)`` noise (this package's extension)."""
    M, N = size
    img = np.zeros((M, N))
    x = np.linspace(-np.pi, np.pi, N)
    A = M // 2 if amplitude > M else amplitude // 2
    cols = np.arange(N)
    wave2 = None  # second edge of the two multi-sinusoidal types (gpet_utils.py:203-220)
    if ltype in ("sinusoidal", "multi-sinusoidal", "close multi-sinusoidal"):
        wave = (np.rint(A * np.sin(N * curvature * x)) + M // 2).astype(int)
        if ltype != "sinusoidal":
            wave2 = wave + (A // 2 if ltype == "multi-sinusoidal" else A // 6)
    elif ltype == "co-sinusoidal":
        wave = (np.rint(A * np.cos(N * curvature * x)) + M // 2).astype(int)
    elif ltype == "diag":
        wave = cols.copy()
    elif ltype == "straight":
        wave = np.full(N, M // 2, dtype=int)
    else:
        raise ValueError("ltype must be one of sinusoidal, multi-sinusoidal, close multi-sinusoidal, co-sinusoidal, "
                         "diag, straight")
    rows = np.arange(M)[:, None]
    img[rows >= wave[None, :]] = intensity
    if wave2 is not None:  # the band below the second edge is 1 - intensity
        img[rows >= wave2[None, :]] = 1 - intensity
    if gaps:
        img[:, 20:30] = 0
        img[:, N // 2:(N // 2 + 10)] = 0
        img[:, N - 100:N - 90] = 0
        img[:, N // 4:(N // 4 + 20)] = 0
    if seed is None:  # gpet_utils.py:251 under scikit-image 0.18: np.random.seed(1); np.random.normal(mean, var ** 0.5, shape)
        noise = np.random.RandomState(1).normal(0.0, noise_level ** 0.5, img.shape)
    else:
        noise = np.random.default_rng(seed).normal(0.0, math.sqrt(noise_level), img.shape)
    img = np.clip(img + noise, 0.0, 1.0)
    edge = np.stack([wave, cols], axis=1)
    if wave2 is not None:  # both edges, one after the other (2N rows), as the reference returns them
        edge = np.concatenate([edge, np.stack([wave2, cols], axis=1)], axis=0)
    return img, edge


def trace_MSE(edge_pred, edge_true):
    n = edge_pred.shape[0]
    return np.round((1 / n) * np.sum((edge_pred.reshape(n, -1)[:, 0] - edge_true[:, 0]) ** 2), 4)


def trace_relarea(edge_pred, edge_true):
    n = edge_pred.shape[0]
    ta = np.sum(n - edge_true[:, 0]) / n ** 2
    pa = np.sum(n - edge_pred.reshape(n, -1)[:, 0]) / n ** 2
    return np.round(np.abs((ta - pa) / ta), 5)


def trace_dicecoef(edge_pred, edge_true, jaccard=False):
    n = edge_pred.shape[0]
    rows = np.arange(n)[:, None]
    pb = (rows >= edge_pred.reshape(n, -1)[:, 0].astype(int)[None, :]).astype(float)
    tb = (rows >= edge_true[:, 0].astype(int)[None, :]).astype(float)
    jacc = np.sum(pb * tb) / np.sum(np.clip(pb + tb, 0, 1))
    return np.round(jacc, 4) if jaccard else np.round(2 * jacc / (jacc + 1), 4)
