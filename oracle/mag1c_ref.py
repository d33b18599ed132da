"""CPU restatement of the mag1c matched filters and their drivers.

TEST INFRASTRUCTURE -- not product code (see oracle/__init__.py).  PINNED: checked against outputs of the
reference itself (tests/golden/g1_filters.npz, g2_groups_*.npz, g3_templates.npz, written by
tests/golden/make_golden.py importing /root/reference/starcop/models/mag1c.py) in tests/test_oracle.py.

Follows /root/reference/starcop/models/mag1c.py:
  rmf :284-348, acrwl1mf :177-280, func_by_groups :116-174, get_mask_bad_bands :98-113,
  generate_template_from_bands :60-95 (constants NODATA/SCALING/EPSILON :55-57)
and the two drivers' group semantics: starcop/models/mag1c_emit.py:40-90, starcop/process_aviris.py:189-219.

Formulation: one group (pixels x bands matrix) at a time in numpy, in the dtype of the input (float32 data is
filtered in float32, float64 in float64 -- as torch does in the reference).  The batched [b, p, s] entry points
loop over b.
"""
import numpy as np

NODATA = -9999
SCALING = 1e5
EPSILON = 1e-9


class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def _chol_solve(Cm, rhs):
    """C^{-1} rhs through a lower Cholesky factor (torch.linalg.cholesky + cholesky_solve in the reference)."""
    try:
        L = np.linalg.cholesky(Cm)
    except np.linalg.LinAlgError as e:
        raise NotPositiveDefinite(str(e)) from e
    import scipy.linalg as sla
    y = sla.solve_triangular(L, rhs, lower=True, check_finite=False)
    return sla.solve_triangular(L.T, y, lower=False, check_finite=False).astype(Cm.dtype)


def _covariance(m, n_div, alpha, dt):
    """(m^T m)/N blended towards its own diagonal: C + alpha*(diag(C) - C)   (mag1c.py:320-321, 248-249)."""
    Cm = (m.T @ m) / dt(n_div)
    return (Cm + dt(alpha) * (np.diag(np.diag(Cm)) - Cm)).astype(dt)


def _energy(xmm, Cm):
    """sum of ALL entries of (x-mu) C^{-1} (x-mu)^T, formed as the reference forms it (mag1c.py:271-275, 338): a P x P matrix."""
    return float(np.sum(xmm @ _chol_solve(Cm, np.ascontiguousarray(xmm.T))))


def rmf_group(x, template, alpha=0.0, zero_override=False, albedo_override=False, apply_scaling=True, mask=None, energy=None):
    """x [P,S], template [S] -> (mf [P], R [P] or scalar 1).  Statistics over ``mask`` pixels, divided by ALL P.
    ``energy``: a list; gets (norm residual, N/2 * log(1 / prod diag chol C)) appended (compute_energy, mag1c.py:337-343)."""
    dt = x.dtype.type
    P = x.shape[0]
    t = template.astype(x.dtype)
    stat = x if mask is None else x[mask]
    mu = stat.mean(axis=0, dtype=x.dtype)
    target = t * mu
    Cm0 = _covariance(stat - mu, P, alpha, dt)
    Cit = _chol_solve(Cm0, target)
    if energy is not None:
        energy.append((_energy(x - mu, Cm0), np.diag(np.linalg.cholesky(Cm0)).astype(x.dtype), P))
    normalizer = target @ Cit
    R = np.ones(P, dtype=x.dtype) if albedo_override else (x @ mu) / (mu @ mu)
    mf = ((x - mu) @ Cit) / (R * normalizer)
    if not zero_override:
        mf = np.maximum(mf, 0)
    if apply_scaling and energy is None:      # (the reference returns (mf, R, energy) BEFORE its scaling line: mag1c.py:337-346)
        mf = mf * dt(SCALING)
    return mf.astype(x.dtype), R.astype(x.dtype)


def acrwl1mf_group(x, template, num_iter=30, albedo_override=False, zero_override=False, sparse_override=False,
                   covariance_update_scaling=1.0, alpha=0.0, mask=None, energy=None):
    """Albedo-corrected reweighted-L1 matched filter for one group (mag1c.py:177-280).
    ``energy``: a list; gets the rmf pair first, then one norm residual per iteration (mag1c.py:270-275)."""
    dt = x.dtype.type
    P = x.shape[0]
    t = template.astype(x.dtype)
    mf, R = rmf_group(x, t, alpha=alpha, zero_override=zero_override, albedo_override=albedo_override,
                      apply_scaling=False, mask=mask, energy=energy)
    sel = slice(None) if mask is None else mask
    target = t * x[sel].mean(axis=0, dtype=x.dtype)
    k = dt(covariance_update_scaling)
    for _ in range(num_iter):
        modx = x[sel] - (k * R[sel] * mf[sel])[:, None] * target[None, :]
        mu = modx.mean(axis=0, dtype=x.dtype)
        target = t * mu
        Cmk = _covariance(modx - mu, P, alpha, dt)
        Cit = _chol_solve(Cmk, target)
        if energy is not None:
            energy.append(_energy(x - mu, Cmk))
        reg = dt(0) if sparse_override else dt(1) / (R * (mf + dt(EPSILON)))
        normalizer = max(target @ Cit, dt(1))            # clamp_(min=1) when < 1  (:264-266)
        mf = np.maximum((((x - mu) @ Cit) - reg) / (R * normalizer), 0).astype(x.dtype)   # relu unconditional (:268)
    return (mf * dt(SCALING)).astype(x.dtype), R


def _rmf_energy(per_group):
    """the reference's value, a scalar: the norm residuals summed over the WHOLE batch + N/2 log(1 / prod of ALL diagonal entries of all
    the batch's Cholesky factors), the product taken in the data's dtype (mag1c.py:338-341: torch.prod without a dim; in float32 it
    underflows to 0 for a few dozen bands and the reference returns inf)"""
    total = sum(e[0][0] for e in per_group)
    diags = np.concatenate([e[0][1] for e in per_group])
    n = per_group[0][0][2]
    with np.errstate(divide="ignore", over="ignore"):
        det = diags.dtype.type(1) / np.prod(diags, dtype=diags.dtype)
        return float(total + n / 2 * np.log(det))


def rmf(x, template, compute_energy=False, **kw):
    """batched [b,p,s] -> (mf [b,p,1], R [b,p,1])  (+ the scalar energy with compute_energy)"""
    en = [[] for _ in range(x.shape[0])] if compute_energy else [None] * x.shape[0]
    outs = [rmf_group(x[b], template, energy=en[b], **kw) for b in range(x.shape[0])]
    res = np.stack([o[0] for o in outs])[..., None], np.stack([o[1] for o in outs])[..., None]
    return res + (_rmf_energy(en),) if compute_energy else res


def acrwl1mf(x, template, compute_energy=False, **kw):
    """(+ with compute_energy: the list the reference returns -- [rmf energy, then per iteration the residual summed over the batch])"""
    en = [[] for _ in range(x.shape[0])] if compute_energy else [None] * x.shape[0]
    outs = [acrwl1mf_group(x[b], template, energy=en[b], **kw) for b in range(x.shape[0])]
    res = np.stack([o[0] for o in outs])[..., None], np.stack([o[1] for o in outs])[..., None]
    if not compute_energy:
        return res
    n_it = len(en[0]) - 1
    return res + ([_rmf_energy(en)] + [sum(e[1 + k] for e in en) for k in range(n_it)],)


def acrwl1mf_batched(x, template, num_iter=30, alpha=0.0, covariance_update_scaling=1.0):
    """The same filter for b groups at once in the reference's own tensor formulation -- torch CPU ops on [b, P, S] (bmm for the
    covariance, torch.linalg.cholesky + cholesky_solve; mag1c.py:203-210 allows the batch dimension, :236-276 is the loop) --
    used as the BATCHED CPU baseline leg of bench.py and pinned against acrwl1mf_group / golden G1 in
    tests/test_oracle.py.  Default flags only (no overrides, no pixel mask).  x: torch tensor [b, P, S]; returns (mf, R) [b, P, 1]."""
    import torch
    with torch.no_grad():
        t = torch.as_tensor(template, dtype=x.dtype).reshape(1, 1, -1)
        b, P, S = x.shape

        def stats(m):
            mu = m.mean(dim=1, keepdim=True)
            target = t * mu
            d = m - mu
            Cm = torch.bmm(d.transpose(1, 2), d) / P
            Cm = torch.lerp(Cm, torch.diag_embed(torch.diagonal(Cm, dim1=-2, dim2=-1)), alpha)
            L = torch.linalg.cholesky(Cm)
            Cit = torch.cholesky_solve(target.transpose(1, 2), L)                 # [b, S, 1]
            return mu, target, Cit, torch.bmm(target, Cit)                        # normaliser [b, 1, 1]
        mu, target, Cit, nrm = stats(x)
        R = torch.bmm(x, mu.transpose(1, 2)) / torch.bmm(mu, mu.transpose(1, 2))  # [b, P, 1]
        mf = torch.relu(torch.bmm(x - mu, Cit) / (R * nrm))
        for _ in range(num_iter):
            modx = x - covariance_update_scaling * R * mf * target
            mu, target, Cit, nrm = stats(modx)
            reg = 1.0 / (R * (mf + EPSILON))
            mf = torch.relu((torch.bmm(x - mu, Cit) - reg) / (R * nrm.clamp(min=1)))
        return mf * SCALING, R


def func_by_groups(func, x, groups, mask=None, min_pixels=10):
    """AVIRIS driver semantics (mag1c.py:116-174): every group id present under ``mask`` is filtered on its own
    valid pixels; groups with <= 10 valid pixels and invalid pixels keep NODATA.  ``func(xg [P,S]) -> (mf, R)``."""
    groups = np.asarray(groups)
    if mask is None:
        mask = np.all(x > NODATA, axis=-1)
    mf_out = np.full(x.shape[:2], NODATA, dtype=x.dtype)
    alb_out = np.full(x.shape[:2], NODATA, dtype=x.dtype)
    for g in np.unique(groups[mask]):
        sel = (groups == g) & mask
        if sel.sum() <= min_pixels:
            continue
        mf, R = func(np.ascontiguousarray(x[sel, :]))
        mf_out[sel], alb_out[sel] = mf, R
    return mf_out, alb_out


def mag1c_columns(raw, template, fill_value, column_step=None, num_iter=30, alpha=1e-4):
    """EMIT driver semantics (mag1c_emit.py:50-90): raw (rows, cols, S) float32, blocks of ``column_step`` columns,
    invalid = any band == fill, computed in float64, outputs float32 filled with ``fill_value``."""
    rows, cols, _ = raw.shape
    invalid = np.any(raw == fill_value, axis=-1)
    mf_out = np.full((rows, cols), fill_value, dtype=np.float64)
    alb_out = np.full((rows, cols), fill_value, dtype=np.float64)
    step = column_step or cols
    for c0 in range(0, cols, step):
        c1 = min(c0 + step, cols)
        valid = ~invalid[:, c0:c1]
        if not valid.any():
            continue
        xg = raw[:, c0:c1][valid, :].astype(np.float64)
        mf, R = acrwl1mf_group(xg, template.astype(np.float64), num_iter=num_iter, alpha=alpha)
        mf_out[:, c0:c1][valid] = mf
        alb_out[:, c0:c1][valid] = R
    return mf_out.astype(np.float32), alb_out.astype(np.float32)


def get_mask_bad_bands(wave):
    """keep 400..2485 nm minus the water bands (1350,1420) and (1800,1945), open intervals (mag1c.py:98-113)."""
    wave = np.asarray(wave)
    drop = (wave < 400) | (wave > 2485) | ((wave > 1350) & (wave < 1420)) | ((wave > 1800) & (wave < 1945))
    return ~drop


def read_ch4_lut(hdr_path, lut_path):
    """ENVI BSQ float64 reader for the CH4 radiance look-up table -> (rads [7, n_wave], wave [n_wave])."""
    import re
    txt = open(hdr_path).read()
    geti = lambda k: int(re.search(rf"{k}\s*=\s*(\d+)", txt).group(1))   # noqa: E731
    ns, nl, nb = geti("samples"), geti("lines"), geti("bands")
    wl = re.search(r"wavelength\s*=\s*\{([^}]*)\}", txt, re.S).group(1)
    wave = np.array([float(v) for v in wl.replace("\n", " ").split(",") if v.strip()])
    raw = np.fromfile(lut_path, dtype="<f8").reshape(nb, nl, ns)
    return raw.transpose(1, 2, 0).squeeze(), wave


def generate_template_from_bands(centers, fwhm, rads, wave):
    """Unit absorption spectrum (mag1c.py:60-95): Gaussian band responses (sigma = fwhm / 2.3548) normalised to unit
    sum, LUT resampled, log, least-squares slope vs concentration, x1e5.  Bands whose response does not overlap the
    LUT span are returned as NaN (the reference leaves them uninitialised)."""
    centers, fwhm = np.asarray(centers, dtype=np.float64), np.asarray(fwhm, dtype=np.float64)
    if np.any(~np.isfinite(centers)) or np.any(~np.isfinite(fwhm)):
        raise RuntimeError("Band Wavelengths Centers/FWHM data contains non-finite data (NaN or Inf).")
    if centers.shape[0] != fwhm.shape[0]:
        raise RuntimeError("Length of band center wavelengths and band fwhm arrays must be equal.")
    conc = np.array([0, 500, 1000, 2000, 4000, 8000, 16000], dtype=np.float64)
    var = (fwhm / (2.0 * np.sqrt(2.0 * np.log(2.0)))) ** 2
    resp = np.exp(-(wave[:, None] - centers[None, :]) ** 2 / (2 * var)) / np.sqrt(2 * np.pi * var)
    tot = resp.sum(axis=0)
    ok = tot > 0
    spectrum = np.full(centers.shape[0], np.nan)
    resampled = rads @ (resp[:, ok] / tot[ok])
    A = np.stack([np.ones_like(conc), conc], axis=1)
    good = np.all(resampled > 0, axis=0)
    slope = np.linalg.lstsq(A, np.log(resampled[:, good]), rcond=None)[0]
    idx = np.flatnonzero(ok)[good]
    spectrum[idx] = slope[1] * SCALING
    return np.stack([centers, spectrum], axis=1)
