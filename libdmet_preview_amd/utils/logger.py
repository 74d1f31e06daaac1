"""
Minimal logger with the reference's level names and call shapes (utils/logger.py:27-55).
Warnings are log lines, never exceptions; `eassert` raises a bare Exception like the reference.
Nothing is logged from inner loops.
"""
import sys

Level = {"FATAL": 0, "ERROR": 1, "WARNING": 2, "SECTION": 3, "RESULT": 4, "INFO": 5,
         "DEBUG0": 6, "DEBUG1": 7, "DEBUG2": 8}
verbose = "WARNING"
stream = sys.stderr
warnings_seen = []      # (message) of every warn() call; tests read this


def _emit(level, msg, args):
    if Level[verbose] >= Level[level]:
        try:
            text = msg % args if args else msg
        except TypeError:
            text = "%s %s" % (msg, args)
        stream.write("%8s: %s\n" % (level, text))


def warn(msg, *args):
    try:
        warnings_seen.append(msg % args if args else msg)
    except TypeError:
        warnings_seen.append(str(msg))
    _emit("WARNING", msg, args)


warning = warn


def info(msg, *args):
    _emit("INFO", msg, args)


def result(msg, *args):
    _emit("RESULT", msg, args)


def error(msg, *args):
    _emit("ERROR", msg, args)


def debug(level, msg, *args):
    _emit("DEBUG%d" % min(max(int(level), 0), 2), msg, args)


def eassert(cond, msg, *args):
    if not cond:
        error(msg, *args)
        raise Exception(msg % args if args else msg)


def check(cond, msg, *args):
    """utils/logger.py:57-60: a failed check is a warning, not an error."""
    if not cond:
        warn(msg, *args)
