"""Stand-in for the un-vendored third-party `munkres==1.0.12` (reference requirements.txt:12),
used ONLY by oracle/make_golden.py to import the reference's src/utils/hungarian.py in the
build container.  Backed by scipy's linear_sum_assignment (same optimal cost; tie-breaking
between equal-cost assignments is not pinned by the reference)."""
import numpy as np
from scipy.optimize import linear_sum_assignment


class Munkres(object):
    def compute(self, cost_matrix):
        r, c = linear_sum_assignment(np.asarray(cost_matrix, dtype=np.float64))
        return list(zip(r.tolist(), c.tolist()))
