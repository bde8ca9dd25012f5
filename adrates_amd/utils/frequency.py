"""Payment frequencies (cavour/utils/frequency.py:39-71)."""
from enum import Enum

from .error import LibError


class FrequencyTypes(Enum):
    ZERO = -1
    SIMPLE = 0
    ANNUAL = 1
    SEMI_ANNUAL = 2
    TRI_ANNUAL = 3
    QUARTERLY = 4
    MONTHLY = 12
    CONTINUOUS = 99


_PER_YEAR = {
    FrequencyTypes.CONTINUOUS: -1,
    FrequencyTypes.ZERO: 1.0,  # no coupon; 1 avoids a division by zero downstream
    FrequencyTypes.ANNUAL: 1.0,
    FrequencyTypes.SEMI_ANNUAL: 2.0,
    FrequencyTypes.TRI_ANNUAL: 3.0,
    FrequencyTypes.QUARTERLY: 4.0,
    FrequencyTypes.MONTHLY: 12.0,
}


def annual_frequency(freq_type: FrequencyTypes):
    """Number of payments per year; ``None`` for SIMPLE, as in the reference
    (cavour/utils/frequency.py:50-71 falls off the end for SIMPLE)."""
    if not isinstance(freq_type, FrequencyTypes):
        raise LibError("Unknown frequency type")
    return _PER_YEAR.get(freq_type)
