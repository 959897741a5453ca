"""TEST INFRASTRUCTURE ONLY -- stand-in for `numba` so that the *Python reference*
(/root/reference, qgs) can be imported in the build container, where numba is not
installed.  `njit` is the identity decorator: CPython then executes the reference's
own loops statement by statement, i.e. the same IEEE-754 operation order that
numba would compile (the reference never asks for fastmath).

Used only by tests/golden/make_golden.py.  Never imported by qgs_amd.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(func):
        return func
    return wrap


jit = njit
