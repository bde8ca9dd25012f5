"""adrates_amd - MI355X-native PV + AD-Greeks path for OIS portfolios."""
__version__ = "0.1.0"
