"""TEST INFRASTRUCTURE ONLY -- stand-in for `pebble` (imported by the reference's symbolic inner
products module).  `ProcessPool.map(fn, items, timeout=)` returns a future whose `result()` is an
iterator over the results in order; per-item timeouts are not implemented (the reference only
uses them to fall back from symbolic integration to quadrature, and the golden generator asks for
the quadrature directly)."""
import multiprocessing
import os


class _MapFuture(object):
    def __init__(self, it):
        self._it = it

    def result(self):
        return self._it


class ProcessPool(object):
    def __init__(self, max_workers=None, **kwargs):
        self._pool = multiprocessing.get_context('fork').Pool(max_workers or os.cpu_count())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._pool.terminate()
        self._pool.join()
        return False

    def map(self, function, iterable, timeout=None, chunksize=1):
        return _MapFuture(self._pool.imap(function, list(iterable), chunksize))
