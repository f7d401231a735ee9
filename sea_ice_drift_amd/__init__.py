"""MI355X-native pattern-matching (PM) hot path of sea_ice_drift.

Only the per-grid-point rotated-template MCC sweep is implemented here (as a HIP kernel
for gfx950 behind the C ABI of include/sid_pm.h) together with the host-side callers that
keep the reference's call shape: ``pattern_matching`` and ``SeaIceDrift.get_drift_PM``.
"""
__version__ = '0.1.0'
