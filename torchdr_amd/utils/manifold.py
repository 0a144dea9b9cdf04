"""Manifolds of the hyperbolic estimator (reference ``utils/manifold.py``: ``Manifold`` :13-78, ``ManifoldParameter``
:81-96, ``EuclideanManifold`` :99-174, ``PoincareBallManifold`` :243-383).

COSNE itself runs on HIP kernels (``csrc/tdr_cosne.hip``: pair terms, closed-form gradient, the Riemannian-Adam step with
its exponential map and parallel transport fused per row); these classes are the host-side counterpart for code that
works WITH an embedding on the ball -- distances, log / exp maps, Moebius arithmetic -- written with torch ops and the
reference's conventions: points satisfy ``c * |x|^2 < 1``, curvature ``-c``, and the same guards (norms floored at 1e-15,
``tanh`` arguments clipped to +-15, ``artanh`` arguments to +-(1 - 1e-15), the ball shrunk by 4e-3 / 1e-5 for float32 /
float64 projections).  Values agree with the reference's to float64 rounding (``tests/golden/manifold.npz``)."""

import torch
from torch.nn import Parameter

_FLOOR = 1e-15          # smallest norm / denominator
_TANH_CLIP = 15.0
_BALL_MARGIN = {torch.float32: 4e-3, torch.float64: 1e-5}


def _dot(a, b, dim=-1):
    return (a * b).sum(dim=dim, keepdim=True)


def _len(a):
    return a.norm(dim=-1, p=2, keepdim=True).clamp_min(_FLOOR)


def tanh(x, clamp=_TANH_CLIP):
    """tanh of the argument clipped to +-clamp (reference :206-225)."""
    return x.clamp(-clamp, clamp).tanh()


_tanh = tanh


class Artanh(torch.autograd.Function):
    """0.5 (log(1 + x) - log(1 - x)) evaluated in float64 on the clipped argument; derivative 1 / (1 - x^2)."""

    @staticmethod
    def forward(ctx, x):
        xc = x.clamp(-1 + 1e-15, 1 - 1e-15)
        ctx.save_for_backward(xc)
        z = xc.double()
        return (0.5 * (torch.log1p(z) - torch.log1p(-z))).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return g / (1 - xc * xc)


def artanh(x):
    return Artanh.apply(x)


class Manifold:
    """Interface (reference :13-78); every operation takes the curvature parameter ``c`` last."""

    name = "Manifold"
    eps = 1e-7

    def __init__(self):
        pass

    def _todo(self, *a, **k):
        raise NotImplementedError

    sqdist = egrad2rgrad = proj = proj_tan = proj_tan0 = expmap = logmap = expmap0 = logmap0 = _todo
    mobius_add = mobius_matvec = init_weights = inner = ptransp = ptransp0 = _todo


class ManifoldParameter(Parameter):
    """A parameter that knows the manifold it lives on and its curvature (reference :81-96)."""

    def __new__(cls, data, requires_grad, manifold, c):
        return Parameter.__new__(cls, data, requires_grad)

    def __init__(self, data, requires_grad, manifold, c):
        self.manifold, self.c = manifold, c

    def __repr__(self):
        return f"{self.manifold.name} Parameter containing:\n" + super(Parameter, self).__repr__()


class EuclideanManifold(Manifold):
    """Flat space: every map is the obvious one (reference :99-174, including its ``ptransp0(x, v) = x + v``)."""

    name = "Euclidean"

    def normalize(self, p):
        p.view(-1, p.size(-1)).renorm_(2, 0, 1.0)
        return p

    def sqdist(self, p1, p2, c):
        return ((p1 - p2) ** 2).sum(dim=-1)

    def egrad2rgrad(self, p, dp, c):
        return dp

    def proj(self, p, c):
        return p

    def proj_tan(self, u, p, c):
        return u

    def proj_tan0(self, u, c):
        return u

    def expmap(self, u, p, c):
        return p + u

    def logmap(self, p1, p2, c):
        return p2 - p1

    def expmap0(self, u, c):
        return u

    def logmap0(self, p, c):
        return p

    def mobius_add(self, x, y, c, dim=-1):
        return x + y

    def mobius_matvec(self, m, x, c):
        return x @ m.transpose(-1, -2)

    def init_weights(self, w, c, irange=1e-5):
        w.data.uniform_(-irange, irange)
        return w

    def inner(self, p, c, u, v=None, keepdim=False):
        return (u * (u if v is None else v)).sum(dim=-1, keepdim=keepdim)

    def ptransp(self, x, y, v, c):
        return v

    def ptransp0(self, x, v, c):
        return x + v


class PoincareBallManifold(Manifold):
    """The ball ``{x : c |x|^2 < 1}`` with the metric ``lambda_x^2 <.,.>``, ``lambda_x = 2 / (1 - c |x|^2)``
    (reference :243-383)."""

    name = "PoincareBall"

    def __init__(self):
        super().__init__()
        self.min_norm = _FLOOR
        self.eps = dict(_BALL_MARGIN)

    # -- building blocks -------------------------------------------------------------------------------------------
    def _lambda_x(self, x, c):
        return 2 / (1.0 - c * _dot(x.data, x.data)).clamp_min(_FLOOR)

    def mobius_add(self, x, y, c, dim=-1):
        """x (+) y = ((1 + 2c<x,y> + c|y|^2) x + (1 - c|x|^2) y) / (1 + 2c<x,y> + c^2 |x|^2 |y|^2)."""
        xx, yy, xy = _dot(x, x, dim), _dot(y, y, dim), _dot(x, y, dim)
        top = (1 + 2 * c * xy + c * yy) * x + (1 - c * xx) * y
        return top / (1 + 2 * c * xy + c ** 2 * xx * yy).clamp_min(_FLOOR)

    def _gyration(self, u, v, w, c, dim: int = -1):
        """gyr[u, v] w, the rotation by which Moebius addition fails to be associative."""
        uu, vv, uv = _dot(u, u, dim), _dot(v, v, dim), _dot(u, v, dim)
        uw, vw = _dot(u, w, dim), _dot(v, w, dim)
        cc = c ** 2
        on_u = -cc * uw * vv + c * vw + 2 * cc * uv * vw
        on_v = -cc * vw * uu - c * uw
        return w + 2 * (on_u * u + on_v * v) / (1 + 2 * c * uv + cc * uu * vv).clamp_min(_FLOOR)

    # -- geometry ----------------------------------------------------------------------------------------------------
    def sqdist(self, p1, p2, c):
        """(2 / sqrt(c) * artanh(sqrt(c) |(-p1) (+) p2|))^2."""
        rc = c ** 0.5
        gap = self.mobius_add(-p1, p2, c, dim=-1).norm(dim=-1, p=2, keepdim=False)
        return (artanh(rc * gap) * 2 / rc) ** 2

    def egrad2rgrad(self, p, dp, c):
        """Riemannian gradient = Euclidean gradient / lambda_p^2 (scaled in place, as the reference does)."""
        dp /= self._lambda_x(p, c).pow(2)
        return dp

    def proj(self, x, c):
        """Points outside the ball of radius (1 - eps) / sqrt(c) are pulled back onto it."""
        r = _len(x)
        limit = (1 - self.eps[x.dtype]) / (c ** 0.5)
        return torch.where(r > limit, x / r * limit, x)

    def proj_tan(self, u, p, c):
        return u

    def proj_tan0(self, u, c):
        return u

    def expmap(self, u, p, c):
        rc = c ** 0.5
        ul = _len(u)
        return self.mobius_add(p, _tanh(rc / 2 * self._lambda_x(p, c) * ul) * u / (rc * ul), c)

    def logmap(self, p1, p2, c):
        rc = c ** 0.5
        d = self.mobius_add(-p1, p2, c)
        dl = _len(d)
        return 2 / rc / self._lambda_x(p1, c) * artanh(rc * dl) * d / dl

    def expmap0(self, u, c):
        rc = c ** 0.5
        ul = _len(u)
        return _tanh(rc * ul) * u / (rc * ul)

    def logmap0(self, p, c):
        rc = c ** 0.5
        pl = _len(p)
        return (1.0 / rc * artanh(rc * pl) / pl) * p

    def mobius_matvec(self, m, x, c):
        """M (x) x = tanh(|Mx| / |x| artanh(sqrt(c) |x|)) Mx / (|Mx| sqrt(c)); 0 where Mx = 0."""
        rc = c ** 0.5
        xl = _len(x)
        mx = x @ m.transpose(-1, -2)
        ml = _len(mx)
        out = _tanh(ml / xl * artanh(rc * xl)) * mx / (ml * rc)
        vanishes = (mx == 0).prod(-1, keepdim=True, dtype=torch.uint8)
        return torch.where(vanishes.bool(), torch.zeros(1, dtype=out.dtype, device=out.device), out)

    def init_weights(self, w, c, irange=1e-5):
        w.data.uniform_(-irange, irange)
        return w

    def inner(self, x, c, u, v=None, keepdim=False):
        return self._lambda_x(x, c) ** 2 * (u * (u if v is None else v)).sum(dim=-1, keepdim=keepdim)

    def ptransp(self, x, y, u, c):
        """Parallel transport of u from x to y: gyr[y, -x] u * lambda_x / lambda_y."""
        return self._gyration(y, -x, u, c) * self._lambda_x(x, c) / self._lambda_x(y, c)

    ptransp_ = ptransp

    def ptransp0(self, x, u, c):
        return 2 * u / self._lambda_x(x, c).clamp_min(_FLOOR)

    def to_hyperboloid(self, x, c):
        K = 1.0 / c
        rK = K ** 0.5
        s = torch.norm(x, p=2, dim=1, keepdim=True) ** 2
        return rK * torch.cat([K + s, 2 * rK * x], dim=1) / (K - s)
