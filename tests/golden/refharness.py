"""Import harness for the *reference* HypAD sources (this container only).

TEST INFRASTRUCTURE.  Used solely by ``tests/golden/gen_fixtures.py`` to run the
reference's own Python on CPU and record golden input/output vectors.  Nothing
here travels to the GPU box in executable form: ``/root/reference`` does not
exist there, and every test that runs on the GPU box reads only the committed
``.npz`` fixtures.

The reference needs three things this image lacks:

* ``geoopt`` (0.5.0 in ``environment.yml:91``) -- its stereographic math module
  is vendored at ``/root/reference/math_.py`` but imports five helpers from
  ``geoopt.utils`` which are *not* vendored.  They are restated below from the
  published geoopt definitions.  ``geoopt.optim.RiemannianAdam`` is not
  vendored at all: the stand-in registered here is ``oracle.radam`` (parity
  UNPINNED for that one class; see oracle/radam.py).
* ``torchvision.transforms`` -- import-time only stub.  ``pyts.metrics.dtw`` -- an independently written
  classic DTW (memoised path search; NOT oracle/scoring.py) so that the reference's own ``_dtw_error``
  framing can run; ``scipy.integrate.trapz`` (removed in SciPy 1.14) is aliased to ``numpy.trapezoid``.
* a GPU -- ``.cuda()`` is patched to the identity.

``PYTORCH_JIT=0`` must be set before torch is imported (torch 2.10 cannot
script ``math_.py:1315``) and bytecode writing is disabled so that nothing is
ever written under /root/reference.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _geoopt_utils_module():
    import torch

    m = types.ModuleType("geoopt.utils")

    def sign(x):
        return torch.sign(x.sign() + 0.5)

    def sabs(x, eps: float = 1e-15):
        return x.abs().add_(eps)

    def clamp_abs(x, eps: float = 1e-15):
        s = sign(x)
        return s * sabs(x, eps=eps)

    def list_range(end: int):
        return list(range(end))

    def drop_dims(tensor, dims):
        seen = 0
        for d in dims:
            tensor = tensor.squeeze(d - seen)
            seen += 1
        return tensor

    m.sign, m.sabs, m.clamp_abs = sign, sabs, clamp_abs
    m.list_range, m.drop_dims = list_range, drop_dims
    return m


def install(repo_root=None):
    """Put the reference on sys.path behind the stubs.  Idempotent."""
    if "geoopt" in sys.modules and getattr(sys.modules["geoopt"], "_hypad_stub", False):
        return sys.modules["geoopt"]
    if os.environ.get("PYTORCH_JIT", "1") != "0":
        raise RuntimeError("set PYTORCH_JIT=0 before importing torch (math_.py cannot be scripted)")
    sys.dont_write_bytecode = True
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present; fixtures can only be generated in the build container")

    import torch

    repo_root = repo_root or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)

    # ---- geoopt package skeleton -------------------------------------------------
    geoopt = types.ModuleType("geoopt")
    geoopt._hypad_stub = True
    geoopt.__path__ = []
    manifolds = types.ModuleType("geoopt.manifolds")
    manifolds.__path__ = []
    stereo = types.ModuleType("geoopt.manifolds.stereographic")
    stereo.__path__ = []
    utils = _geoopt_utils_module()
    sys.modules.update({
        "geoopt": geoopt,
        "geoopt.utils": utils,
        "geoopt.manifolds": manifolds,
        "geoopt.manifolds.stereographic": stereo,
    })
    geoopt.utils, geoopt.manifolds, manifolds.stereographic = utils, manifolds, stereo

    spec = importlib.util.spec_from_file_location(
        "geoopt.manifolds.stereographic.math", os.path.join(REFERENCE_ROOT, "math_.py"))
    gmath = importlib.util.module_from_spec(spec)
    sys.modules["geoopt.manifolds.stereographic.math"] = gmath
    spec.loader.exec_module(gmath)
    stereo.math = gmath

    class PoincareBall:
        def __init__(self, c=1.0):
            self.c = torch.as_tensor(c, dtype=torch.float32)
            self.k = -self.c

    class Sphere:
        pass

    class ManifoldParameter(torch.nn.Parameter):
        def __new__(cls, data=None, manifold=None, requires_grad=True):
            inst = torch.nn.Parameter.__new__(cls, data.data if isinstance(data, torch.nn.Parameter) else data,
                                              requires_grad)
            inst.manifold = manifold
            return inst

    geoopt.PoincareBall = PoincareBall
    geoopt.ManifoldParameter = ManifoldParameter
    geoopt.ManifoldTensor = ManifoldParameter
    manifolds.Sphere = Sphere
    manifolds.PoincareBall = PoincareBall

    optim = types.ModuleType("geoopt.optim")
    from oracle.radam import RiemannianAdam  # UNPINNED stand-in (not in the reference tree)
    optim.RiemannianAdam = RiemannianAdam
    geoopt.optim = optim
    sys.modules["geoopt.optim"] = optim

    # ---- import-time stubs --------------------------------------------------------
    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    pyts = types.ModuleType("pyts")
    pyts.__path__ = []
    pm = types.ModuleType("pyts.metrics")

    def _dtw_independent(x, y, **kwargs):
        """Stand-in for ``pyts.metrics.dtw(x, y)`` with its defaults (pyts==0.12.0 is not installed):
        classic DTW, squared point cost, square root of the cheapest warping path's cost.  Written as a
        memoised search over warping paths from the END cell backwards -- deliberately NOT the forward
        table fill of ``oracle/scoring.py:dtw_classic`` -- so that the fixtures it produces pin the
        reference's own padding / loop bound / framing (``utils/anomaly_detection_utils.py:834-861``)
        and cross-check the oracle's recurrence with independently written code."""
        import functools
        import math
        if kwargs:
            raise RuntimeError("the reference calls dtw(x, y) with defaults only")
        xs, ys = [float(v) for v in x], [float(v) for v in y]

        @functools.lru_cache(maxsize=None)
        def cheapest(i, j):          # cheapest path from (0, 0) to (i, j), both inclusive
            here = (xs[i] - ys[j]) * (xs[i] - ys[j])
            if i == 0 and j == 0:
                return here
            prev = []
            if i > 0:
                prev.append(cheapest(i - 1, j))
            if j > 0:
                prev.append(cheapest(i, j - 1))
            if i > 0 and j > 0:
                prev.append(cheapest(i - 1, j - 1))
            return here + min(prev)

        return math.sqrt(cheapest(len(xs) - 1, len(ys) - 1))

    pm.dtw = _dtw_independent
    pyts.metrics = pm
    sys.modules.setdefault("pyts", pyts)
    sys.modules.setdefault("pyts.metrics", pm)

    # ---- SciPy >= 1.14 dropped integrate.trapz (utils/anomaly_detection_utils.py:802,807 call it) ----------
    import numpy as _np
    import scipy.integrate as _integrate
    if not hasattr(_integrate, "trapz"):
        _integrate.trapz = _np.trapezoid     # the same function under its current name (SciPy's trapz WAS numpy's)

    # ---- no GPU here ----------------------------------------------------------------
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    return geoopt
