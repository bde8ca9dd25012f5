"""Library exception type (mirrors cavour/utils/error.py:21-30)."""


class LibError(Exception):
    """Raised for every error that originates in this library.

    The reference keeps the text in ``_message`` (cavour/utils/error.py:27) and
    callers/tests read that attribute, so it is preserved here; ``str(e)`` also
    works because the message is forwarded to ``Exception``.
    """

    def __init__(self, message: str):
        super().__init__(message)
        self._message = message

    def _print(self):
        print("LibError:", self._message)
