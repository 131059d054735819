"""Deterministic synthetic inputs for tests, benchmarks and examples: the linearised Navier-Stokes pair (A, M) on a
structured Taylor-Hood triangulation of the cylinder channel, and the membrane pair of the reference's benchmark.

Input generation only -- no solver arithmetic lives here.  ``oracle/`` (the CPU checker) and the product's examples and
``bench.py`` all draw their matrices from this one place, so that the GPU path and the checker see identical inputs.
"""
