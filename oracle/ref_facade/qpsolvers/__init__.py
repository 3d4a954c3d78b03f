"""TEST-TIME TOOL: a ``qpsolvers``-shaped facade (Problem / solve_problem) over the
restated Goldfarb-Idnani solver in oracle/ik_oracle.py, so that the unmodified
reference ``mink.solve_ik`` runs end to end when capturing golden vectors."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle.ik_oracle import solve_qp as _solve_qp  # noqa: E402


class Problem:
    def __init__(self, P, q, G=None, h=None, A=None, b=None, lb=None, ub=None):
        self.P, self.q, self.G, self.h = P, q, G, h

    def unpack(self):
        return self.P, self.q, self.G, self.h, None, None, None, None


class Solution:
    def __init__(self, problem, x):
        self.problem = problem
        self.x = x


def solve_problem(problem, solver, **kwargs):
    x, status, _, _ = _solve_qp(problem.P, problem.q, problem.G, problem.h)
    return Solution(problem, x if status == 0 else None)
