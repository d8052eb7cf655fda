"""Multipole conversions of the reference's admp/multipole.py that callers use during set-up."""
import numpy as np

from .systems import convert_cart2harm  # noqa: F401  (admp/multipole.py:36-77)

rt3 = 1.73205080757          # admp/multipole.py:14
inv_rt3 = 1.0 / rt3
C1_h2c = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=np.float64)      # admp/multipole.py:17-19
C1_c2h = C1_h2c.T                                                           # admp/multipole.py:20
