"""Harness that imports the UNMODIFIED reference from /root/reference (this container only).

Used only by ``make_fixtures.py`` (to generate golden vectors) and by tests marked
``needs_reference`` (which skip when /root/reference is absent, e.g. on the GPU box).
The reference's files are never copied; four harness-side shims make the imports work
in this image:

  1. ``scipy.integrate.simps``  -> ``scipy.integrate.simpson`` (removed in scipy >= 1.14;
     bit-identical for the odd sample counts used here),
  2. stub ``skimage`` modules (only imported by gpet_utils, not used by the traced path),
  3. ``GaussianProcessRegressor._validate_data`` -> ``sklearn.utils.validation.validate_data``
     (removed from BaseEstimator in scikit-learn >= 1.6),
  4. a stand-in ``KDEpy.FFTKDE`` built on the oracle's restatement (KDEpy is not
     installed: everything downstream of the KDE is labelled ``kde_standin``).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gp_edge_tracing"))


def _install_shims():
    import scipy.integrate

    if not hasattr(scipy.integrate, "simps"):
        scipy.integrate.simps = lambda y, x=None, **k: scipy.integrate.simpson(y, x=x, **k)

    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        util = types.ModuleType("skimage.util")
        metrics = types.ModuleType("skimage.metrics")
        measure = types.ModuleType("skimage.measure")
        restoration = types.ModuleType("skimage.restoration")

        def _absent(*a, **k):
            raise RuntimeError("skimage is not installed (harness stub)")

        util.random_noise = _absent
        metrics.peak_signal_noise_ratio = _absent
        metrics.structural_similarity = _absent
        metrics.normalized_root_mse = _absent
        measure.shannon_entropy = _absent
        sk.util, sk.metrics, sk.measure, sk.restoration = util, metrics, measure, restoration
        for name, mod in [("skimage", sk), ("skimage.util", util), ("skimage.metrics", metrics),
                          ("skimage.measure", measure), ("skimage.restoration", restoration)]:
            sys.modules[name] = mod

    if "KDEpy" not in sys.modules:
        if _REPO not in sys.path:
            sys.path.insert(0, _REPO)
        from oracle import gpet_oracle as orc

        class FFTKDE:  # stand-in, PARITY UNPINNED
            def __init__(self, kernel="gaussian", bw=1, norm=2):
                assert kernel == "gaussian"
                self.bw = bw

            def fit(self, data, weights=None):
                self.data = np.asarray(data, dtype=np.float64)
                self.weights = np.ones(len(self.data)) if weights is None else np.asarray(weights, float)
                return self

            def evaluate(self, grid_points):
                gp = np.asarray(grid_points)
                nx = int(gp[:, 0].max() - gp[:, 0].min()) + 1
                ny = int(gp[:, 1].max() - gp[:, 1].min()) + 1
                dens = orc.fftkde_grid(self.data, self.weights, ny - 2, nx - 2, bw=self.bw)
                return dens.reshape(-1)

        kd = types.ModuleType("KDEpy")
        kd.FFTKDE = FFTKDE
        sys.modules["KDEpy"] = kd


def load_reference():
    """Returns the reference's ``gp_edge_tracing`` package (imported from /root/reference)."""
    if not reference_available():
        raise RuntimeError("reference not mounted")
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import gp_edge_tracing  # noqa
    from gp_edge_tracing import sklearn_gpr
    from sklearn.utils.validation import validate_data

    if not hasattr(sklearn_gpr.GaussianProcessRegressor, "_validate_data"):
        sklearn_gpr.GaussianProcessRegressor._validate_data = (
            lambda self, *a, **k: validate_data(self, *a, **k))
    return gp_edge_tracing
