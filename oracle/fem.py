"""Alias kept for the oracle and its tests: the synthetic (A, M) generator lives in ``synthetic/fem.py`` (input
generation is not part of the checker; the product's examples and ``bench.py`` import it from there)."""

from synthetic.fem import *  # noqa: F401,F403
from synthetic.fem import CASES, GRADING, SIGMA_RE50  # noqa: F401  (explicit for linters and readers)
