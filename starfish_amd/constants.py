"""Physical constants used on the log-likelihood path (values as in Starfish/constants.py:4-7)."""
from math import pi  # noqa: F401

c_ang = 2.99792458e18  # A s^-1
c_kms = 2.99792458e5  # km s^-1
