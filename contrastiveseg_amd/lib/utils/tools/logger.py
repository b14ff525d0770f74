"""Logging surface of the reference (lib/utils/tools/logger.py:31-204): a process-wide `Logger` with
info/warn/error/debug/info_once class methods and the `[file, line]` prefix the reference prints."""
import logging
import os
import sys

_FMT = "%(asctime)s %(levelname)-7s %(message)s"


class Logger(object):
    _log = None
    _once = set()

    @classmethod
    def init(cls, logfile_level="info", stdout_level="info", log_file=None, log_format=_FMT, rewrite=False):
        log = logging.getLogger("contrastiveseg_amd")
        log.handlers = []
        log.propagate = False
        log.setLevel(logging.DEBUG)
        fmt = logging.Formatter(log_format or _FMT)
        if log_file:
            os.makedirs(os.path.dirname(os.path.abspath(log_file)), exist_ok=True)
            fh = logging.FileHandler(log_file, mode="w" if rewrite else "a")
            fh.setLevel(getattr(logging, str(logfile_level).upper(), logging.INFO))
            fh.setFormatter(fmt)
            log.addHandler(fh)
        if stdout_level is not None:
            sh = logging.StreamHandler(sys.stdout)
            sh.setLevel(getattr(logging, str(stdout_level).upper(), logging.INFO))
            sh.setFormatter(fmt)
            log.addHandler(sh)
        cls._log = log

    @classmethod
    def _emit(cls, level, message):
        if cls._log is None:
            cls.init(stdout_level=os.environ.get("CSEG_LOG_LEVEL", "warning"))
        frame = sys._getframe(2)
        prefix = "[%s, %d]" % (os.path.basename(frame.f_code.co_filename), frame.f_lineno)
        cls._log.log(level, "%s %s", prefix, message)

    @classmethod
    def debug(cls, message):
        cls._emit(logging.DEBUG, message)

    @classmethod
    def info(cls, message):
        cls._emit(logging.INFO, message)

    @classmethod
    def info_once(cls, message):
        if message not in cls._once:
            cls._once.add(message)
            cls._emit(logging.INFO, message)

    @classmethod
    def warn(cls, message):
        cls._emit(logging.WARNING, message)

    @classmethod
    def error(cls, message):
        cls._emit(logging.ERROR, message)

    @classmethod
    def critical(cls, message):
        cls._emit(logging.CRITICAL, message)
