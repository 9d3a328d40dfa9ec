"""Workload generators and timing loops for bench.py, the tests, the examples and tools/ (synthetic scenarios, the
synthetic drive, the mini simulator, latency loops, the batched-iLQR measurement).  Not part of the product package:
nothing under tpl_amd/ imports from here."""
