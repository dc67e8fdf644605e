"""Minimal stand-in for `loguru` so the reference's modules import in this container.

Test infrastructure only (used by oracle/make_golden.py). Not shipped, not imported by llmc_amd.
"""
import logging as _logging


class _Logger:
    def __init__(self):
        self._log = _logging.getLogger('llmc-ref')

    def _emit(self, level, msg, *a, **k):
        self._log.log(level, str(msg))

    def info(self, msg, *a, **k):
        self._emit(_logging.INFO, msg)

    def warning(self, msg, *a, **k):
        self._emit(_logging.WARNING, msg)

    def error(self, msg, *a, **k):
        self._emit(_logging.ERROR, msg)

    def debug(self, msg, *a, **k):
        self._emit(_logging.DEBUG, msg)

    def remove(self, *a, **k):
        pass

    def add(self, *a, **k):
        return 0


logger = _Logger()
