"""`InterpTypes` under the import path the reference's users expect
(cavour/market/curves/interpolator.py:18-26 defines a second copy of the enum
with the same values as cavour/utils/global_types.py:76-84; one shared enum is
enough because the engine only compares ``.value``)."""
from ...utils.global_types import InterpTypes  # noqa: F401
