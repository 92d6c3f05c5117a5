"""`create_logger(output_dir, dist_rank, name)` of the pre-training entry points.

Contract (reference logger.py:7-32): one `logging.Logger` per (directory, rank, name); every rank appends to
`<output_dir>/log_rank<rank>_<name>.txt`; rank 0 additionally prints to stdout with the timestamp/name in green and the
call site in yellow; line layout `[time name] (file line): LEVEL message`, seconds resolution; no propagation to the
root logger; calling it twice with the same arguments returns the same object without stacking handlers.
"""
import logging
import os
import sys

_LINE = "[%(asctime)s %(name)s] (%(filename)s %(lineno)d): %(levelname)s %(message)s"
_STAMP = "%Y-%m-%d %H:%M:%S"
_GREEN, _YELLOW, _RESET = "\033[32m", "\033[33m", "\033[0m"
_made = {}


def _console_layout() -> str:
    head, site, tail = "[%(asctime)s %(name)s]", "(%(filename)s %(lineno)d)", ": %(levelname)s %(message)s"
    if not sys.stdout.isatty() and os.environ.get("FORCE_COLOR") is None:
        return head + site + tail
    return _GREEN + head + _RESET + _YELLOW + site + _RESET + tail


def create_logger(output_dir, dist_rank=0, name=''):
    key = (str(output_dir), int(dist_rank), str(name))
    if key in _made:
        return _made[key]
    log = logging.getLogger(name)
    log.setLevel(logging.DEBUG)
    log.propagate = False
    sinks = []
    if dist_rank == 0:
        sinks.append((logging.StreamHandler(sys.stdout), _console_layout()))
    os.makedirs(output_dir, exist_ok=True)
    sinks.append((logging.FileHandler(os.path.join(output_dir, f"log_rank{dist_rank}_{name}.txt"), mode="a"), _LINE))
    for handler, layout in sinks:
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(logging.Formatter(fmt=layout, datefmt=_STAMP))
        log.addHandler(handler)
    _made[key] = log
    return log
