"""TEST INFRASTRUCTURE ONLY -- stand-in for `pebble` (imported by the reference's
symbolic inner products module, never used on the analytic path)."""


class ProcessPool(object):
    def __init__(self, *args, **kwargs):
        raise RuntimeError("pebble stand-in: symbolic inner products are out of scope")
