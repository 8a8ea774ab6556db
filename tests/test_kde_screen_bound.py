"""The error budget of kde_mode_kernel's fp32 screening pass (hypad_amd/csrc/scoring.hip, written out at its threshold), checked by
emulating the pass's arithmetic step by step in NumPy float32: centring in fp64, the fp32 scale, samples rounded once, the direct
form exp2(-(x - v)^2) or -- when every centred sample lies within 8 units -- the factored form exp2(-x^2) * sum exp2(2 x v - v^2)
with a fused multiply-add, four partial sums per sample.  Claims under test:
  * every screened density is within eps = 8.8e-6 (+ 0.5e-6 for the fp32 scale) of the exact fp64 density;
  * hence the sample scipy would select (first maximum of the fp64 densities) always survives the 4e-5 margin.
No GPU involved: this pins the analysis the GPU kernel's margin rests on (its selections are compared with scipy in the GPU suite)."""
import numpy as np

F = np.float32
EPS = 8.8e-6 + 0.5e-6
MARGIN = 4e-5


def _fma32(a, b, c):           # a * b exact in fp64 (24 x 24 bits), + c, one rounding to fp32 (double rounding: negligible, < 2^-29 relative)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def screen(values):
    v = values.astype(F).astype(np.float64)                       # critic values are fp32
    cnt = len(v)
    mean = v.sum() / cnt
    var = ((v - mean) ** 2).sum() * (1.0 / (cnt - 1))
    cov = var * cnt ** -0.4
    c32 = F(1.0) / np.sqrt(F(cov) * F(1.3862943611198906), dtype=F)      # v_rsq_f32, 1 ulp
    y = ((v - mean) * np.float64(c32)).astype(F)
    factored = bool(np.max(np.abs(y)) <= F(8.0))
    if factored:
        arg = _fma32((F(2.0) * y)[:, None], y[None, :], (-(y * y))[None, :])
    else:
        d = (y[:, None] - y[None, :]).astype(F)
        arg = -(d * d).astype(F)
    terms = np.exp2(arg, dtype=F)
    pad = (-cnt) % 4
    if pad:
        terms = np.concatenate([terms, np.zeros((cnt, pad), F)], axis=1)
    part = [np.add.accumulate(terms[:, c::4], axis=1, dtype=F)[:, -1] for c in range(4)]      # sequential fp32 sums, as the lanes do
    dens = ((part[0] + part[1]).astype(F) + (part[2] + part[3]).astype(F)).astype(F)
    if factored:
        dens = (dens * np.exp2(-(y * y).astype(F), dtype=F)).astype(F)
    exact = np.exp(-((v[:, None] - v[None, :]) ** 2) * (0.5 / cov)).sum(axis=1)               # what the fp64 pass evaluates
    return dens.astype(np.float64), exact, factored


def _cases(rng, cnt):
    yield "normal", rng.standard_normal(cnt)
    yield "offset 3e4", 3.0e4 + 0.3 * rng.standard_normal(cnt)
    yield "offset -1e5", -1.0e5 + rng.standard_normal(cnt)
    yield "heavy tails", rng.standard_t(2, cnt)
    yield "two clusters", np.concatenate([-1 + 0.05 * rng.standard_normal(cnt // 2), 1 + 0.05 * rng.standard_normal(cnt - cnt // 2)])
    yield "one outlier", np.concatenate([0.01 * rng.standard_normal(cnt - 1), [40.0]])
    yield "two outliers together", np.concatenate([0.01 * rng.standard_normal(cnt - 2), [30.0, 30.001]])
    yield "uniform", rng.uniform(-1, 1, cnt)
    yield "tiny spread", 0.5 + 1e-6 * rng.standard_normal(cnt)
    yield "lattice", np.round(rng.standard_normal(cnt) * 4) / 4


def test_screen_error_stays_inside_its_budget_and_keeps_the_argmax():
    rng = np.random.default_rng(2024)
    worst, seen_factored, seen_direct = 0.0, 0, 0
    for cnt in (7, 64, 100, 101, 256):
        for rep in range(12):
            for name, vals in _cases(rng, cnt):
                dens, exact, factored = screen(np.asarray(vals))
                seen_factored += factored; seen_direct += not factored
                rel = np.max(np.abs(dens - exact) / exact)
                worst = max(worst, rel)
                assert rel <= EPS, (name, cnt, rel, factored)
                k_star = int(np.argmax(exact))                                    # scipy's choice: first maximum of the fp64 densities
                assert dens[k_star] >= dens.max() * (1.0 - MARGIN), (name, cnt, dens[k_star] / dens.max())
    assert seen_factored > 100 and seen_direct > 100                               # both forms were exercised
    assert worst > 1e-8                                                            # (and the emulation is not trivially exact)
