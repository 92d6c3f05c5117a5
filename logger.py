"""Rank-aware logger (logger.py:7-32 of the reference): console on rank 0 + per-rank file, same line format.
termcolor is optional (absent in this image)."""
import functools
import logging
import os
import sys

try:
    from termcolor import colored
except Exception:  # pragma: no cover
    def colored(s, *_a, **_k):
        return s


@functools.lru_cache()
def create_logger(output_dir, dist_rank=0, name=''):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    fmt = '[%(asctime)s %(name)s] (%(filename)s %(lineno)d): %(levelname)s %(message)s'
    color_fmt = colored('[%(asctime)s %(name)s]', 'green') + colored('(%(filename)s %(lineno)d)', 'yellow') + ': %(levelname)s %(message)s'
    if dist_rank == 0:
        console_handler = logging.StreamHandler(sys.stdout)
        console_handler.setLevel(logging.DEBUG)
        console_handler.setFormatter(logging.Formatter(fmt=color_fmt, datefmt='%Y-%m-%d %H:%M:%S'))
        logger.addHandler(console_handler)
    os.makedirs(output_dir, exist_ok=True)
    file_handler = logging.FileHandler(os.path.join(output_dir, f'log_rank{dist_rank}_{name}.txt'), mode='a')
    file_handler.setLevel(logging.DEBUG)
    file_handler.setFormatter(logging.Formatter(fmt=fmt, datefmt='%Y-%m-%d %H:%M:%S'))
    logger.addHandler(file_handler)
    return logger
