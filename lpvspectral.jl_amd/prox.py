"""Host-side mirrors of the ProximalOperators.jl objects the reference passes around
(``using ProximalOperators`` at src/lasso.jl:1; call sites :51,:53-55,:88,:98,:108,:119-121).

They carry parameters only; all arithmetic runs in the HIP library.  ``proxg`` objects map
to the four device prox kernels, ``proxf`` objects to the device Gram problem.
"""
from __future__ import annotations

from . import _lib


class NormL1:
    """``NormL1(λ)``: g(x) = λ‖x‖₁ (default proxg, src/lasso.jl:88,108)."""
    kind = _lib.PROX_L1

    def __init__(self, λ=1.0):
        self.λ = float(λ)

    def device_params(self, n):
        return self.kind, self.λ, 0


class NormL0:
    """``NormL0(λ)``: g(x) = λ·nnz(x) (README.md:73-77)."""
    kind = _lib.PROX_L0

    def __init__(self, λ=1.0):
        self.λ = float(λ)

    def device_params(self, n):
        return self.kind, self.λ, 0


class IndBallL0:
    """``IndBallL0(r)``: indicator of {x : nnz(x) ≤ r} (README.md:79-83)."""
    kind = _lib.PROX_BALL_L0

    def __init__(self, r):
        if int(r) != r or r < 1:
            raise ValueError("r must be a positive integer")
        self.r = int(r)

    def device_params(self, n):
        return self.kind, float(self.r), 0


class NormL2:
    """``NormL2(λ)``: g(x) = λ‖x‖₂ -- used through :class:`SlicedSeparableSum`."""

    def __init__(self, λ=1.0):
        self.λ = float(λ)


class SlicedSeparableSum:
    """``SlicedSeparableSum(gs, idxs)`` as built at src/lasso.jl:53-55: equal ``NormL2(λ)`` terms on
    contiguous, equally long slices ``((f-1)*L+1 : f*L,)``.  Other slicings are not supported on device."""
    kind = _lib.PROX_GROUP_L2

    def __init__(self, gs, idxs):
        gs, idxs = list(gs), list(idxs)
        if len(gs) != len(idxs) or not gs:
            raise ValueError("gs and idxs must be non-empty and of equal length")
        lam = {g.λ for g in gs if isinstance(g, NormL2)}
        if len(lam) != 1 or not all(isinstance(g, NormL2) for g in gs):
            raise NotImplementedError("device group prox needs identical NormL2(λ) terms")
        self.λ = lam.pop()
        rngs = [tuple(i[0]) if isinstance(i, tuple) and len(i) == 1 and not isinstance(i[0], int) else tuple(i) for i in idxs]
        L = len(rngs[0])
        for f, r in enumerate(rngs):  # 1-based ranges like Julia's (f-1)*L+1 : f*L
            if len(r) != L or r[0] != f * L + 1 or r[-1] != (f + 1) * L:
                raise NotImplementedError("device group prox needs contiguous equal slices (f-1)*L+1:f*L")
        self.group_len, self.ngroups = L, len(rngs)

    @classmethod
    def frequency_groups(cls, λ, Nf, group_len):
        obj = cls.__new__(cls)
        obj.λ, obj.group_len, obj.ngroups = float(λ), int(group_len), int(Nf)
        return obj

    def device_params(self, n):
        return self.kind, self.λ, self.group_len


class LeastSquares:
    """``LeastSquares(A, b; iterative=true)``: f(x) = ½‖Ax−b‖² (src/lasso.jl:51,98).  On device the
    x-update (A'A + I/μ)x = A'b + v/μ is solved from the resident Gram instead of by CG."""
    linear_sign = _lib.LINEAR_LEAST_SQUARES

    def __init__(self, A, b, iterative=True):
        self.A, self.b, self.W = A, b, None


class Quadratic:
    """``Quadratic(Q, q; iterative=true)``: f(x) = ½x'Qx + q'x (src/lasso.jl:121):
    (Q + I/μ)x = v/μ − q."""
    linear_sign = _lib.LINEAR_QUADRATIC_AS_WRITTEN

    def __init__(self, Q, q, iterative=True):
        self.Q, self.q = Q, q
