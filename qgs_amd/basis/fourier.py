"""Fourier basis bookkeeping (reference: qgs/basis/fourier.py).

Only the wavenumber tables needed by the analytic inner products are kept; the SymPy function objects of
the reference's `ChannelFourierBasis` / `BasinFourierBasis` are out of scope, so the two classes here are
light records of the spectral blocks.

Basis functions (nondimensional domain coordinates x, y; n = aspect ratio):
    'A'  sqrt(2) cos(P y)                 P = ny
    'K'  2 cos(M n x) sin(P y)            M = nx, P = ny
    'L'  2 sin(H n x) sin(P y)            H = nx, P = ny      (ocean basin: nx = H / 2)
"""
import numpy as np


class WaveNumber(object):
    """Type and wavenumbers of one basis function (fourier.py:172-222)."""

    __slots__ = ('type', 'P', 'M', 'H', 'nx', 'ny')

    def __init__(self, function_type, P, M, H, nx, ny):
        self.type = function_type
        self.P = P
        self.M = M
        self.H = H
        self.nx = nx
        self.ny = ny

    def __repr__(self):
        return "type = {}, P = {}, M= {},H={}, nx= {}, ny={}".format(self.type, self.P, self.M, self.H, self.nx, self.ny)


def channel_wavenumbers(spectral_blocks):
    """Channel (atmosphere / ground) modes: a block with nx == 1 gives A, K, L, the others K, L
    (fourier.py:225-255)."""
    out = []
    for nx, ny in np.asarray(spectral_blocks):
        nx, ny = int(nx), int(ny)
        if nx == 1:
            out.append(WaveNumber('A', ny, 0, 0, 0, ny))
        out.append(WaveNumber('K', ny, nx, 0, nx, ny))
        out.append(WaveNumber('L', ny, 0, nx, nx, ny))
    return np.array(out, dtype=object)


def basin_wavenumbers(spectral_blocks):
    """Closed-basin (ocean) modes: one L function per block with x-wavenumber H/2 (fourier.py:258-282)."""
    return np.array([WaveNumber('L', int(ny), 0, int(nx), int(nx) / 2., int(ny)) for nx, ny in np.asarray(spectral_blocks)],
                    dtype=object)


class _FourierBasisRecord(object):
    def __init__(self, spectral_blocks, aspect_ratio):
        self.spectral_blocks = np.asarray(spectral_blocks)
        self.aspect_ratio = float(aspect_ratio)

    def __len__(self):
        return len(self.wavenumbers)


class ChannelFourierBasis(_FourierBasisRecord):
    @property
    def wavenumbers(self):
        return channel_wavenumbers(self.spectral_blocks)


class BasinFourierBasis(_FourierBasisRecord):
    @property
    def wavenumbers(self):
        return basin_wavenumbers(self.spectral_blocks)
