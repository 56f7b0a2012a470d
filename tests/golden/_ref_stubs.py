"""Stub modules that let the *reference* sBayes (read-only at /root/reference) be imported
in the build container, where numba / pyproj / cartopy / libpysal / tables / ruamel.yaml /
unidecode are not installed.  TEST INFRASTRUCTURE ONLY: used by make_golden.py to generate
the fixtures committed under tests/golden/.  Nothing here (and nothing of the reference)
is imported by the product, the GPU tests, smoke() or bench.py.

The numba stub turns @njit/@jit into identity decorators, so the reference's two jitted
functions run as the plain NumPy code they are written as (that NumPy path is the CPU
baseline BASELINE.json names).  `numba.vectorize` is mapped to scipy.special.gammaln
because math.lgamma(0) raises while numba's lgamma(0) returns +inf.
"""
import sys
import types

import numpy as np
import scipy.special
import yaml

REFERENCE_ROOT = "/root/reference"


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


class _Dummy:
    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


# (module level, so that objects holding them pickle: the MC3 parent sends `data` -- with its CRS -- to its workers,
#  sbayes/mcmc_setup.py:299, :554)
class CRS:
    def __init__(self, *a, **k):
        pass

    @classmethod
    def from_user_input(cls, *a, **k):
        return cls()

    @classmethod
    def from_epsg(cls, *a, **k):
        return cls()


class Transformer:
    @classmethod
    def from_crs(cls, *a, **k):
        return cls()

    def transform(self, x, y, *a, **k):
        return x, y


class Geodesic:
    def inverse(self, loc, pts):
        loc = np.asarray(loc, dtype=float).reshape(1, -1)
        pts = np.asarray(pts, dtype=float)
        d = np.sqrt(((pts - loc) ** 2).sum(axis=-1))
        out = np.zeros((len(pts), 3))
        out[:, 0] = d
        return out


def install():
    if "numba" in sys.modules and getattr(sys.modules["numba"], "_sbayes_amd_stub", False):
        return
    nb = types.ModuleType("numba")
    nb._sbayes_amd_stub = True
    nb.jit = _identity_decorator
    nb.njit = _identity_decorator
    nb.vectorize = lambda *a, **k: (lambda fn: scipy.special.gammaln)
    for name in ("float32", "float64", "int64", "boolean"):
        setattr(nb, name, _Dummy())
    sys.modules["numba"] = nb

    ud = types.ModuleType("unidecode")
    ud.unidecode = lambda s: s
    sys.modules["unidecode"] = ud

    pyproj = types.ModuleType("pyproj")

    pyproj.CRS = CRS
    transformer = types.ModuleType("pyproj.transformer")

    transformer.Transformer = Transformer
    pyproj.transformer = transformer
    pyproj.Transformer = Transformer
    sys.modules["pyproj"] = pyproj
    sys.modules["pyproj.transformer"] = transformer

    cartopy = types.ModuleType("cartopy")
    cartopy.__version__ = "0.22.0"
    geodesic = types.ModuleType("cartopy.geodesic")

    geodesic.Geodesic = Geodesic
    cartopy.geodesic = geodesic
    crs_mod = types.ModuleType("cartopy.crs")
    cartopy.crs = crs_mod
    sys.modules["cartopy"] = cartopy
    sys.modules["cartopy.geodesic"] = geodesic
    sys.modules["cartopy.crs"] = crs_mod

    sys.modules["libpysal"] = types.ModuleType("libpysal")
    sys.modules["tables"] = types.ModuleType("tables")

    ruamel = types.ModuleType("ruamel")
    ruamel_yaml = types.ModuleType("ruamel.yaml")

    class YAML:
        def __init__(self, typ=None, **k):
            pass

        def load(self, stream):
            return yaml.safe_load(stream)

        def dump(self, data, stream=None):
            return yaml.safe_dump(data, stream)

    ruamel_yaml.YAML = YAML
    ruamel_yaml.CommentedMap = dict
    ruamel.yaml = ruamel_yaml
    comments = types.ModuleType("ruamel.yaml.comments")
    comments.CommentedMap = dict
    sys.modules["ruamel"] = ruamel
    sys.modules["ruamel.yaml"] = ruamel_yaml
    sys.modules["ruamel.yaml.comments"] = comments

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # import order matters (circular import state -> model_shapes -> sbayes.model -> likelihood -> counts -> state)
    import sbayes.model  # noqa: F401
    import sbayes.sampling.state  # noqa: F401
