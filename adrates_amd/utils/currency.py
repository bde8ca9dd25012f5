"""Currency codes (cavour/utils/currency.py:47-62); values kept identical."""
from enum import Enum


class CurrencyTypes(Enum):
    USD = 1
    EUR = 2
    GBP = 3
    CHF = 4
    CAD = 5
    AUD = 6
    NZD = 7
    DKK = 8
    SEK = 9
    HKD = 10
    JPY = 11
    NOK = 12
    PLN = 13
    RON = 14
    NONE = 15
