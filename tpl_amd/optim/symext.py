"""Symbolic building blocks for OCP definitions (mirror of the reference's
``tpl.optim.symext`` interface, library/tpl/optim/symext.py:13-157).

* :class:`ArraySymbol` -- an opaque array-valued parameter; only usable as an
  argument of the interpolation functions below.
* ``lerp, lerp_angle, box_interp, blerp, lerp_wrap, get_array_value`` -- opaque
  sympy functions.  sympy cannot differentiate them, so their derivatives stay
  ``Derivative`` atoms that the generator turns into finite differences
  (genopt.py, rule FD).  Their run-time semantics live in
  ``csrc/ilqr/builtins.h``.
* :func:`fixed` / :func:`unfixed` -- freeze sub-expressions during
  differentiation by renaming their symbols.
"""

import sympy as sp
from sympy.core import cache as _sp_cache

FIXED_PREFIX = "fixed_"


class ArraySymbol(sp.Symbol):
    """Array-valued parameter of undetermined shape (float64, C order)."""

    def __new__(cls, name, **assumptions):
        return super().__new__(cls, name, **assumptions)


def _opaque(name, n_args, doc):
    return type(name, (sp.Function,), {"nargs": (n_args,), "__doc__": doc})


get_array_value = type("get_array_value", (sp.Function,), {
    "__doc__": "get_array_value(arr, i): element i of a flat array (no bounds check)."})
box_interp = _opaque("box_interp", 3, "box_interp(dx, x, arr): arr[floor(x/dx)], index clamped.")
lerp = _opaque("lerp", 4, "lerp(x0, dx, x, arr): linear interpolation on a uniform grid.")
lerp_angle = _opaque("lerp_angle", 4, "lerp_angle(x0, dx, x, arr): lerp along the shorter arc.")
blerp = _opaque("blerp", 7, "blerp(x0, y0, dx, dy, x, y, arr): bilinear interpolation, arr[rows=y][cols=x].")
lerp_wrap = _opaque("lerp_wrap", 5, "lerp_wrap(length, dx, x, xs, arr): periodic linear interpolation.")

OPAQUE_FUNCTIONS = (get_array_value, box_interp, lerp, lerp_angle, blerp, lerp_wrap)


def clear_cache():
    _sp_cache.clear_cache()


def fixed(expr):
    """Rename every free symbol ``s`` of ``expr`` to ``fixed_s``: the result is
    a constant with respect to the original symbols during differentiation."""
    expr = sp.sympify(expr)
    ren = {s: sp.Symbol(FIXED_PREFIX + s.name) for s in expr.free_symbols
           if not s.name.startswith(FIXED_PREFIX) and not isinstance(s, ArraySymbol)}
    return expr.xreplace(ren)


def unfixed(expr):
    """Inverse of :func:`fixed`."""
    ren = {s: sp.Symbol(s.name[len(FIXED_PREFIX):]) for s in expr.free_symbols
           if s.name.startswith(FIXED_PREFIX)}
    return expr.xreplace(ren) if ren else expr
