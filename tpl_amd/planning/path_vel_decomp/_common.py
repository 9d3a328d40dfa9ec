import time


def runtime(func):
    """tpl.util.runtime (util.py:54-67): wall time of the last call as ``<method>.runtime``."""
    def inner(*args, **kwargs):
        start = time.perf_counter()
        res = func(*args, **kwargs)
        inner.runtime = time.perf_counter() - start
        return res
    inner.runtime = 0.0
    return inner


def make_solver(factory_name, solver=None):
    """The solver object a glue class drives: the HIP solver of tpl_amd.optim by default; tests (and the CPU leg of
    a comparison) hand in any object with the reference's ``Optim`` surface."""
    if solver is not None:
        return solver
    from ...optim import optimizers as opts
    if not hasattr(opts, factory_name):
        opts.build_optimizers()                    # the reference's applications do this once at start-up
    return getattr(opts, factory_name)()
