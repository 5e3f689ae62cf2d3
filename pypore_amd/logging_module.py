"""Logger "PyPore" (lib/logging_module.py:6-23): console handler at INFO, file handler ``pypore.log`` at DEBUG.

The reference opens ``pypore.log`` in the current directory as a side effect of importing the module (:16); here the
file handler is attached by ``attach_file_handler()``, which the CLI entry points call, so that importing the package
from a library or a test writes nothing."""
from __future__ import annotations

import logging
import sys

log = logging.getLogger("PyPore")

_COLORS = {"WARNING": 33, "INFO": 32, "DEBUG": 34, "CRITICAL": 33, "ERROR": 31}


class ColoredFormatter(logging.Formatter):
    """ANSI-coloured level names on a terminal (lib/colored_log.py:9-32); plain text when not a tty."""

    def format(self, record):
        msg = super().format(record)
        if sys.stderr.isatty() and record.levelname in _COLORS:
            msg = msg.replace(record.levelname, "\033[1;%dm%s\033[0m" % (_COLORS[record.levelname], record.levelname), 1)
        return msg


ch = logging.StreamHandler()
ch.setLevel(logging.INFO)
ch.setFormatter(ColoredFormatter("[%(name)s][%(levelname)s] - %(message)s"))
if not log.handlers:
    log.addHandler(ch)
log.setLevel(logging.INFO)

_fh = None


def attach_file_handler(path="pypore.log"):
    global _fh
    if _fh is None:
        _fh = logging.FileHandler(path)
        _fh.setLevel(logging.DEBUG)
        _fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        log.addHandler(_fh)
    return _fh


def verbosity(ver):
    """-v 0..3 (lib/seq_routines.py:785-795).  In the reference ``ver`` arrives as a string and no branch fires; the
    intended behaviour is implemented here."""
    ver = int(ver)
    if ver == 0:
        log.setLevel(logging.WARN)
    elif ver == 1:
        log.setLevel(logging.INFO)
    elif ver >= 2:
        log.setLevel(logging.DEBUG)
        if ver == 3:
            ch.setLevel(logging.DEBUG)
