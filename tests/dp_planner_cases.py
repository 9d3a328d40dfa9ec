"""The synthetic drives of the planner-plugin tests live in the package (benchkit/drive.py: bench.py and the examples use
them too); this name is what the tests and the fixture generator import."""
from benchkit.drive import *  # noqa: F401,F403
from benchkit.drive import DT, Drive, trajectory_row  # noqa: F401
