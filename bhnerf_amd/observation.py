"""Visibility-domain measurement operators for the EHT losses (reference: bhnerf/observation.py wraps
ehtim, an external package).  Only what the hot path consumes is provided: the direct-DFT matrix
``A[k, p] = exp(-2 pi i (u_k x_p + v_k y_p))`` that maps an image vector to complex visibilities, in the
shape ``loss_fn_eht`` expects (network.py:541-544).  Not a re-implementation of ehtim's pulse functions or
its sign/ordering conventions (parity unpinned: SURVEY 8c iii)."""
import numpy as np


def dft_matrix(uv, fov, npix):
    """uv: (nvis, 2) baselines in wavelengths; fov: field of view in radians; npix: image is npix x npix
    (row-major, H then W as flattened by loss_fn_eht).  Returns complex64 (nvis, npix*npix)."""
    uv = np.asarray(uv, dtype=np.float64)
    x = (np.arange(npix) - (npix - 1) / 2.0) * (fov / npix)
    yy, xx = np.meshgrid(x, x, indexing='ij')
    phase = -2.0 * np.pi * (uv[:, 0:1] * xx.reshape(1, -1) + uv[:, 1:2] * yy.reshape(1, -1))
    return np.exp(1j * phase).astype(np.complex64)


def closure_triangles(nsites):
    """All site triples (i<j<k) of an array with nsites stations."""
    return [(i, j, k) for i in range(nsites) for j in range(i + 1, nsites) for k in range(j + 1, nsites)]
