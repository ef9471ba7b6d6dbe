"""Kerr null geodesics for the image plane: an own ray tracer in place of the external ``kgeo`` package.

The reference's ``bhnerf.kgeo.image_plane_geos`` (``kgeo.py:6-63``) wraps ``kgeo.raytrace_ana`` of an external,
un-vendored package and returns an ``xarray.Dataset``.  Neither is available here, so this module integrates the
geodesics itself and returns a ``Geodesics`` record (attribute access to NumPy arrays with the same names and the
same ``(alpha, beta, geo)`` layout).  **Parity unpinned** against the external tracer; what is checked instead
(``tests/test_geodesics_cpu.py``): the constants of motion along every ray (``(dr/dλ)² = R(r)``, ``(dθ/dλ)² = Θ(θ)``),
straight lines and Euclidean light-travel time for ``M → 0``, the Schwarzschild shadow radius ``√27 M`` and the
equatorial reflection symmetry.

Method (Boyer-Lindquist coordinates, ``G = c = 1``, photon energy at infinity ``E``): in Mino time ``λ``
(``d(affine) = Σ dλ``) the radial and polar motions separate,

    (dr/dλ)² = R(r) = (r² + a² − a ℓ)² − Δ (η + (ℓ − a)²),      (dθ/dλ)² = Θ(θ) = η + a² cos²θ − ℓ² cot²θ,
    dφ/dλ = a (r² + a² − a ℓ)/Δ − a + ℓ / sin²θ,               dt/dλ = (r² + a²)(r² + a² − a ℓ)/Δ + a (ℓ − a sin²θ),

with ``ℓ = −α sin i`` and ``η = (α² − a²) cos² i + β²`` for the image-plane coordinates ``(α, β)`` of an observer at
inclination ``i`` (Bardeen 1973).  The second-order form ``r'' = R'(r)/2``, ``θ'' = Θ'(θ)/2`` is integrated backwards
from the observer with classical RK4 — no sign bookkeeping at the turning points; away from them the radial
velocity is re-derived from ``R(r)`` after every step — with a per-ray step
``dλ = h (1 + r/r_c) / r²`` (radial steps of ≈ h(1 + r/r_c): a few hundredths of M near the hole, geometric far away), until the ray falls
through the horizon or is back at the observer's radius (finished rays are compacted away).  A second pass with the
now known total Mino time of every ray writes its ``ngeo`` samples, uniform in Mino time, as the steps cross them, which is where ``dtau`` (the Mino step) of the radiative-transfer integrand ``g² · dtau · Σ`` comes from.
"""
import numpy as np


class Geodesics(dict):
    """Mapping with attribute access: the fields of the reference's geodesic dataset as NumPy arrays."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    @property
    def dims(self):
        return {'alpha': self['r'].shape[0], 'beta': self['r'].shape[1], 'geo': self['r'].shape[2]}

    def fillna(self, value):
        """Copy with the NaNs of every floating array replaced (``xarray.Dataset.fillna``, used by alma.py:43)."""
        out = Geodesics()
        for k, v in self.items():
            arr = np.asarray(v)
            out[k] = np.where(np.isnan(arr), value, arr) if arr.dtype.kind == 'f' and arr.ndim else v
        return out


def kerr_functions(r, theta, spin, M=1.0):
    """Δ, Σ, Ξ and the frame-dragging frequency ω of the Kerr metric (spin in units of M)."""
    a = spin * M
    Delta = r ** 2 - 2.0 * M * r + a ** 2
    Sigma = r ** 2 + a ** 2 * np.cos(theta) ** 2
    Xi = (r ** 2 + a ** 2) ** 2 - Delta * a ** 2 * np.sin(theta) ** 2
    omega = 2.0 * a * M * r / Xi
    return Delta, Sigma, Xi, omega


def radial_potential(r, a, lam, eta, M=1.0):
    Delta = r ** 2 - 2.0 * M * r + a ** 2
    return (r ** 2 + a ** 2 - a * lam) ** 2 - Delta * (eta + (lam - a) ** 2)


def angular_potential(theta, a, lam, eta):
    return eta + a ** 2 * np.cos(theta) ** 2 - lam ** 2 / np.tan(theta) ** 2


def _rhs(y, a, lam, eta, M):
    r, th, vr, vth = y[0], y[1], y[4], y[5]
    s, c = np.sin(th), np.cos(th)
    Delta = r ** 2 - 2.0 * M * r + a ** 2
    P = r ** 2 + a ** 2 - a * lam
    # R'(r)/2 and Θ'(θ)/2
    ar = 2.0 * r * P - (r - M) * (eta + (lam - a) ** 2)
    ath = -a ** 2 * s * c + lam ** 2 * c / s ** 3
    dph = a * P / Delta - a + lam / s ** 2
    dt = (r ** 2 + a ** 2) * P / Delta + a * (lam - a * s ** 2)
    return np.stack([vr, vth, dph, dt, ar, ath])


def _initial_state(alpha, beta, a, inc, distance, M):
    lam = -alpha * np.sin(inc)
    eta = (alpha ** 2 - a ** 2) * np.cos(inc) ** 2 + beta ** 2
    r0 = np.full_like(alpha, float(distance))
    th0 = np.full_like(alpha, inc)
    vr0 = -np.sqrt(np.clip(radial_potential(r0, a, lam, eta, M), 0.0, None))            # inwards, back in time
    vth0 = -np.sign(beta) * np.sqrt(np.clip(angular_potential(th0, a, lam, eta), 0.0, None))
    return np.stack([r0, th0, np.zeros_like(r0), np.zeros_like(r0), vr0, vth0]), lam, eta


def _integrate(alpha, beta, spin, inclination, distance, M, h, r_c, max_steps, targets=None):
    """RK4 over all rays with the finished ones compacted away.  Without ``targets``: returns each ray's final Mino
    time and state.  With ``targets`` (n, ngeo) increasing Mino times: returns the states interpolated at them,
    shape (7, n, ngeo) with rows (mino, r, theta, phi, t, vr, vth)."""
    alpha = np.asarray(alpha, dtype=np.float64).ravel()
    beta = np.asarray(beta, dtype=np.float64).ravel()
    a = float(spin) * M
    inc = float(inclination)
    if not (0.0 < inc <= 0.5 * np.pi + 1e-12):
        raise ValueError('inclination must be in (0, pi/2]')
    n = alpha.size
    y, lam, eta = _initial_state(alpha, beta, a, inc, distance, M)
    r_hor = M + np.sqrt(max(M * M - a * a, 0.0))
    idx = np.arange(n)                                   # rays still being integrated (compacted)
    mino = np.zeros(n)
    end_mino, end_y = np.zeros(n), y.copy()
    out = nxt = None
    if targets is not None:
        ngeo = targets.shape[1]
        out = np.zeros((7, n, ngeo))
        nxt = np.zeros(n, dtype=np.int64)                # next sample of every ray (global indexing)
    for step in range(1, max_steps + 1):
        lm, et = lam[idx], eta[idx]
        # (the centrifugal term of Theta is stiff next to the poles: up to 20x smaller steps there)
        dl = h * (1.0 + y[0] / r_c) / y[0] ** 2 * np.clip((np.sin(y[1]) / 0.25) ** 2, 0.05, 1.0)
        k1 = _rhs(y, a, lm, et, M)
        k2 = _rhs(y + 0.5 * dl * k1, a, lm, et, M)
        k3 = _rhs(y + 0.5 * dl * k2, a, lm, et, M)
        k4 = _rhs(y + dl * k3, a, lm, et, M)
        y_new = y + dl / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        # keep the first integrals (dr/dlambda)^2 = R(r), (dtheta/dlambda)^2 = Theta(theta): far from the hole dr/dlambda ~ r^2 ~
        # 1e6 and a relative RK4 error of 1e-10 there is an absolute error of 1e2 in (dr/dlambda)^2 at the turning point; away
        # from turning points the velocities are therefore re-derived from the potentials (sign kept), near them the
        # second-order form runs free
        Rn = radial_potential(y_new[0], a, lm, et, M)
        y_new[4] = np.where(Rn > 1e-2 * y_new[0] ** 4, np.sign(y_new[4]) * np.sqrt(np.abs(Rn)), y_new[4])
        Tn = angular_potential(y_new[1], a, lm, et)
        y_new[5] = np.where(Tn > 1e-2 * (et + a * a + lm ** 2), np.sign(y_new[5]) * np.sqrt(np.abs(Tn)), y_new[5])
        captured = ~(y_new[0] > r_hor * 1.02)              # (also catches a NaN): the step is not taken, the ray ends here
        mino_new = mino + dl
        if targets is not None:                            # samples crossed by this step: cubic Hermite in Mino time
            ok = ~captured                                 # (value and derivative at both ends of the step: O(h^4) like
            f_new = None                                   #  the RK4 step itself; linear interpolation left O(h^2))
            while True:
                g_idx = idx
                k = np.minimum(nxt[g_idx], ngeo - 1)
                tgt = targets[g_idx, k]
                hit = ok & (nxt[g_idx] < ngeo) & (tgt <= mino_new)
                if not hit.any():
                    break
                w = ((tgt - mino) / np.where(dl > 0, dl, 1.0))[hit]
                rows = g_idx[hit]
                if f_new is None:
                    f_new = _rhs(y_new, a, lm, et, M)
                w2, w3 = w * w, w * w * w
                out[0, rows, k[hit]] = tgt[hit]
                out[1:, rows, k[hit]] = ((2 * w3 - 3 * w2 + 1) * y[:, hit] + (w3 - 2 * w2 + w) * dl[hit] * k1[:, hit]
                                         + (3 * w2 - 2 * w3) * y_new[:, hit] + (w3 - w2) * dl[hit] * f_new[:, hit])
                nxt[rows] += 1
        y = np.where(captured, y, y_new)
        mino = np.where(captured, mino, mino_new)
        done = captured | ((y[0] > distance) & (y[4] > 0.0))
        if done.any():
            rows = idx[done]
            end_mino[rows], end_y[:, rows] = mino[done], y[:, done]
            keep = ~done
            idx, y, mino = idx[keep], y[:, keep], mino[keep]
            if idx.size == 0:
                break
    else:
        raise RuntimeError('geodesic integration did not terminate in %d steps' % max_steps)
    if targets is None:
        return end_mino, end_y, lam, eta
    # rounding can leave the very last sample (target = end of the ray) unwritten: it is the end state
    miss = nxt < ngeo
    for j in np.nonzero(miss)[0]:
        out[0, j, nxt[j]:] = end_mino[j]
        out[1:, j, nxt[j]:] = end_y[:, j, None]
    return out, lam, eta


def trace(alpha, beta, spin, inclination, distance=1000.0, M=1.0, h=0.02, r_c=5.0, max_steps=400000):
    """Integrate one geodesic per (alpha, beta) backwards from the observer until it is captured or has escaped.
    Returns ``(mino_end, state_end, lam, eta)`` with ``state_end`` rows ``(r, theta, phi, t, vr, vth)`` (phi and t of
    the forward equations: negate them for the backward ray)."""
    return _integrate(alpha, beta, spin, inclination, distance, M, h, r_c, max_steps)


def image_plane_geos(spin, inclination, alpha_range, beta_range, ngeo=100, num_alpha=64, num_beta=64, distance=1000.0,
                     E=1.0, M=1.0, randomize_subpixel_rays=False, verbose=False, h=0.02, chunk=65536):
    """Kerr geodesics for the whole image plane (signature of ``bhnerf.kgeo.image_plane_geos``, kgeo.py:6-63).

    Returns a ``Geodesics`` record with arrays of shape ``(num_alpha, num_beta, ngeo)`` (per-ray constants
    ``(num_alpha, num_beta)``): ``r, theta, phi, t, x, y, z, mino, affine, dtau, Sigma, Delta, Xi, omega, R, Theta``,
    ``alpha, beta, lam, eta`` and the scalars ``spin, inc, M, E, r_o``.  ``t`` is the coordinate time relative to the
    arrival at the observer (negative along the ray), ``dtau`` the Mino step between consecutive samples."""
    alpha_1d = np.linspace(*alpha_range, num_alpha)
    beta_1d = np.linspace(*beta_range, num_beta)
    if randomize_subpixel_rays:
        alpha_1d = alpha_1d + (np.random.random(num_alpha) - 0.5) * (alpha_range[1] - alpha_range[0]) / max(num_alpha - 1, 1)
        beta_1d = beta_1d + (np.random.random(num_beta) - 0.5) * (beta_range[1] - beta_range[0]) / max(num_beta - 1, 1)
    alpha, beta = np.meshgrid(alpha_1d, beta_1d, indexing='ij')
    n = alpha.size
    out = {k: np.empty((n, ngeo)) for k in ('r', 'theta', 'phi', 't', 'mino', 'vr', 'vth')}
    lam_all, eta_all = np.empty(n), np.empty(n)
    af, bf = alpha.ravel(), beta.ravel()
    # beta = 0 exactly sits on the theta turning point of the observer: nudge it (measure-zero set of rays)
    bf = np.where(bf == 0.0, 1e-9, bf)
    for c0 in range(0, n, chunk):
        sl = slice(c0, min(c0 + chunk, n))
        mino_end, _, lam, eta = _integrate(af[sl], bf[sl], spin, inclination, distance, M, h, 5.0, 400000)
        lam_all[sl], eta_all[sl] = lam, eta
        target = (np.arange(1, ngeo + 1)[None, :] / float(ngeo)) * mino_end[:, None]            # uniform in Mino time
        samp, _, _ = _integrate(af[sl], bf[sl], spin, inclination, distance, M, h, 5.0, 400000, targets=target)
        for row, name in enumerate(('mino', 'r', 'theta', 'phi', 't', 'vr', 'vth')):
            out[name][sl] = samp[row]
        if verbose:
            print('traced rays %d-%d of %d' % (c0, sl.stop, n))
    shape3 = (num_alpha, num_beta, ngeo)
    g = Geodesics()
    a = float(spin) * M
    r, th = out['r'].reshape(shape3), out['theta'].reshape(shape3)
    # the rays were followed backwards: coordinate time and azimuth run the other way
    g.update(r=r, theta=th, phi=-out['phi'].reshape(shape3), t=-out['t'].reshape(shape3), mino=-out['mino'].reshape(shape3))
    g['Delta'], g['Sigma'], g['Xi'], g['omega'] = kerr_functions(r, th, spin, M)
    g['x'] = r * np.sin(th) * np.cos(g['phi'])
    g['y'] = r * np.sin(th) * np.sin(g['phi'])
    g['z'] = r * np.cos(th)
    lam3, eta3 = lam_all.reshape(num_alpha, num_beta)[..., None], eta_all.reshape(num_alpha, num_beta)[..., None]
    g['R'] = radial_potential(r, a, lam3, eta3, M)
    g['Theta'] = angular_potential(th, a, lam3, eta3)
    g['vr'], g['vth'] = out['vr'].reshape(shape3), out['vth'].reshape(shape3)
    dtau = -g['mino'][..., :1] * np.ones(shape3)                                 # uniform Mino step of each ray (positive)
    g['dtau'] = dtau
    g['affine'] = -np.cumsum(g['Sigma'] * dtau, axis=-1)          # 0 at the observer, decreasing into the past: d(affine) = Σ dλ
    g['alpha'], g['beta'] = alpha, beta
    g['lam'], g['eta'] = lam_all.reshape(num_alpha, num_beta), eta_all.reshape(num_alpha, num_beta)
    g['spin'], g['inc'], g['M'], g['E'], g['r_o'] = float(spin), float(inclination), float(M), float(E), float(distance)
    return g
