"""Numeric constants shared across the package (cavour/utils/global_vars.py:3-4)."""

gDaysInYear = 365.0
g_small = 1e-12
ONE_MILLION = 1_000_000
