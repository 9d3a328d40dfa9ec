"""Glue between tpl's path/velocity-decomposition planner and the iLQR solvers: what
``tpl.planning.path_vel_decomp`` and ``tpl.planning.utils`` do around ``Optim.update()``, written against the
reference's call patterns so that ``PathVelDecompPlanner`` can run on ``tpl_amd.optim``'s solvers."""
