"""numpy model of the wavefront-level data layout the HIP correlator kernel uses.

This is the executable specification of `gypsum_amd/csrc/corr_core.hpp`: the
polyphase split of an N = K*1023 circular correlation into K independent
1023-point correlations, each embedded in a 2048-point FFT that ONE 64-lane
wavefront computes as two 1024-point (32 x 32) half-wave transforms.  The model
is checked against `numpy.fft` in `tests/test_lane_model.py`, and the host-built
PRN spectrum table of the C library is checked against `prn_spectrum_lane_layout`.

Layout conventions (lane = 32*h + l, h = half-wave, l = 0..31; reg = 0..31):

  time   domain : value index m = 32*reg + l              (both halves hold the same m)
  freq   domain : bin f = 2*(l + 32*reg) + h
  output domain : lag q = l + 32*reg
"""
from __future__ import annotations

import numpy as np

M = 2048
NCHIP = 1023


def polyphase_inputs(xw: np.ndarray, k: int) -> np.ndarray:
    """y[r, m] = sum_{j<k} xw[(k*m + r + j) mod N]  for r < k, m < 1023."""
    n = len(xw)
    assert n == k * NCHIP
    idx = (k * np.arange(NCHIP)[None, :, None] + np.arange(k)[:, None, None] + np.arange(k)[None, None, :]) % n
    return xw[idx].sum(axis=2)


def fwd_lane_layout(y: np.ndarray) -> np.ndarray:
    """Forward 2048-point DFT of y zero-padded, returned as X[reg, lane] (32 x 64)."""
    a = np.zeros(1024, dtype=complex)
    a[:len(y)] = y
    out = np.zeros((32, 64), dtype=complex)
    n = np.arange(1024)
    for h in (0, 1):
        b = a * np.exp(-2j * np.pi * n * h / M)               # stage-0 twiddle of the odd half
        t = b.reshape(32, 32)                                  # t[n1, n2] = b[32*n1 + n2]; lane n2, reg n1
        t = np.fft.fft(t, axis=0)                              # in-lane FFT32 over n1 -> [g1, n2]
        g1 = np.arange(32)[:, None]
        n2 = np.arange(32)[None, :]
        t = t * np.exp(-2j * np.pi * g1 * n2 / 1024)           # inter-pass twiddle W1024^(g1*n2)
        t = t.T                                                # LDS transpose -> [n2, g1]: lane g1, reg n2
        t = np.fft.fft(t, axis=0)                              # in-lane FFT32 over n2 -> [g2, g1]
        out[:, 32 * h:32 * h + 32] = t                         # reg g2, lane g1 holds X[2*(g1+32*g2)+h]
    return out


def freq_index_of(reg: np.ndarray, lane: np.ndarray) -> np.ndarray:
    return 2 * ((lane % 32) + 32 * reg) + (lane // 32)


def inv_lane_layout(yf: np.ndarray) -> np.ndarray:
    """Un-normalised inverse of `fwd_lane_layout` restricted to lags 0..1023: returns c[q], q = 0..1023."""
    z = np.zeros((2, 1024), dtype=complex)
    for h in (0, 1):
        t = yf[:, 32 * h:32 * h + 32]                          # [g2, g1]: lane g1, reg g2
        t = np.fft.ifft(t, axis=0) * 32                        # in-lane inverse FFT32 over g2 -> [qa, g1]
        qa = np.arange(32)[:, None]
        g1 = np.arange(32)[None, :]
        t = t * np.exp(+2j * np.pi * qa * g1 / 1024)           # W1024^-(g1*qa)
        t = t.T                                                # transpose -> [g1, qa]: lane qa, reg g1
        t = np.fft.ifft(t, axis=0) * 32                        # inverse FFT32 over g1 -> [qb, qa]
        z[h] = t.reshape(-1)                                   # q = qa + 32*qb  == reg*32 + lane
    q = np.arange(1024)
    return z[0] + np.exp(+2j * np.pi * q / M) * z[1]


def prn_spectrum_lane_layout(chips01: np.ndarray) -> np.ndarray:
    """conj(FFT2048(periodic +-1 code)) / 2048 as P[reg, lane]; multiply-and-inverse then yields
    out[q] = sum_m y[m] * code[(m - q) mod 1023]."""
    code = chips01.astype(np.float64) * 2 - 1
    pp = np.zeros(M)
    pp[:NCHIP] = code
    j = np.arange(1, NCHIP)
    pp[M - j] = code[NCHIP - j]
    spec = np.conj(np.fft.fft(pp)) / M
    reg = np.arange(32)[:, None]
    lane = np.arange(64)[None, :]
    return spec[freq_index_of(reg, lane)]


def correlate_lane_model(xw: np.ndarray, chips01: np.ndarray, k: int) -> np.ndarray:
    """Full N-point circular correlation of wiped samples `xw` with the k-times upsampled code."""
    y = polyphase_inputs(xw, k)
    p = prn_spectrum_lane_layout(chips01)
    out = np.zeros(k * NCHIP, dtype=complex)
    for r in range(k):
        c = inv_lane_layout(fwd_lane_layout(y[r]) * p)
        out[r::k] = c[:NCHIP]                                   # out[k*q + r] = c_r[q]
    return out
