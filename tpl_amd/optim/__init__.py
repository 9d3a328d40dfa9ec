"""tpl_amd.optim -- HP-2: generated, batched iLQR solvers on MI355X.

Mirrors the layout of the reference's ``tpl.optim`` package
(library/tpl/optim/{genopt,symext,optimizers}.py): ``symext`` holds the
symbolic building blocks, ``genopt`` turns a ``Config`` into a solver library,
``optimizers`` defines the problems tpl ships, ``solver`` is the Python face of
a compiled library.
"""
from . import symext  # noqa: F401
from . import genopt  # noqa: F401
from . import optimizers  # noqa: F401
from . import solver  # noqa: F401
