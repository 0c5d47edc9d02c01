"""CPU suite: a SECOND, structurally independent statement of the reward's Sinkhorn divergence.

geomloss 0.2.4 / pykeops 1.5 (requirements.txt:1,11 of the reference; call site search_dg.py:116,158-160) are not importable in this
image, so the C oracle (oracle/aadg_oracle.c: orc_sinkhorn_divergence[_f64]) is "parity unpinned".  What CAN be removed is the failure
mode "one transcription error in one C file": this file restates the published algorithm of geomloss 0.2.4's
`SamplesLoss("sinkhorn", p=2, blur=.05, scaling=.5, debias=True)` a second time, in dense numpy / scipy float64, organised like the
package itself (max_diameter -> epsilon_schedule -> softmin -> sinkhorn_loop -> sinkhorn_cost) and sharing no code, loop structure or
author session with the C file.  Both must agree: <= 1e-9 with the float64 master (the bounding-box norm is a float32 sum in both, in
different summation orders: one ulp of float32 on the diameter moves the whole eps schedule, ~2e-11 on the result; everything behind it
is float64), <= 1e-6 with the float32 oracle."""
import numpy as np
import pytest
from scipy.special import logsumexp


# ---- geomloss/sinkhorn_divergence.py (0.2.4), restated densely ------------------------------------------------------------------
def max_diameter(x, y):
    """norm of the bounding box of x u y; the package computes it on the float32 tensors and takes .item()"""
    mins = np.minimum(x.min(axis=0), y.min(axis=0))
    maxs = np.maximum(x.max(axis=0), y.max(axis=0))
    return float(np.sqrt(np.sum((maxs - mins).astype(np.float32) ** 2, dtype=np.float32)))


def epsilon_schedule(p, diameter, blur, scaling):
    return [diameter ** p] + [np.exp(e) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))] + [blur ** p]


def cosine_cost(x, y):
    """the reference's formula cost "IntCst(1) - (X|Y)/(Norm2(X)*Norm2(Y))" as a dense matrix C[i, j]"""
    x, y = x.astype(np.float64), y.astype(np.float64)
    return 1.0 - (x @ y.T) / (np.linalg.norm(x, axis=1)[:, None] * np.linalg.norm(y, axis=1)[None, :])


def softmin(eps, C, f):
    """-eps * log sum_j exp(f_j - C_ij / eps)"""
    return -eps * logsumexp(f[None, :] - C / eps, axis=1)


def sinkhorn_loop(a_log, b_log, C_xx, C_yy, C_xy, C_yx, eps_s):
    eps = eps_s[0]
    # a decent initialisation of the dual vectors
    a_x, b_y = softmin(eps, C_xx, a_log), softmin(eps, C_yy, b_log)
    a_y, b_x = softmin(eps, C_yx, a_log), softmin(eps, C_xy, b_log)
    for eps in eps_s:                                             # eps-scaling descent, symmetrised updates
        at_x = softmin(eps, C_xx, a_log + a_x / eps)
        bt_y = softmin(eps, C_yy, b_log + b_y / eps)
        at_y = softmin(eps, C_yx, a_log + b_x / eps)
        bt_x = softmin(eps, C_xy, b_log + a_y / eps)
        a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
        a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
    # last extrapolation: "the cross-updates should be done in parallel"
    a_x, b_y = softmin(eps, C_xx, a_log + a_x / eps), softmin(eps, C_yy, b_log + b_y / eps)
    a_y, b_x = softmin(eps, C_yx, a_log + b_x / eps), softmin(eps, C_xy, b_log + a_y / eps)
    return a_x, b_y, a_y, b_x


def sinkhorn_cost(alpha, beta, a_x, b_y, a_y, b_x):
    """debias = True, no reach: <alpha, b_x - a_x> + <beta, a_y - b_y>"""
    return float(np.dot(alpha, b_x - a_x) + np.dot(beta, a_y - b_y))


def samples_loss_sinkhorn(x, y, p=2, blur=0.05, scaling=0.5):
    n, m = len(x), len(y)
    alpha, beta = np.full(n, 1.0 / n), np.full(m, 1.0 / m)
    eps_s = epsilon_schedule(p, max_diameter(x, y), blur, scaling)
    C_xy = cosine_cost(x, y)
    pots = sinkhorn_loop(np.log(alpha), np.log(beta), cosine_cost(x, x), cosine_cost(y, y), C_xy, C_xy.T, eps_s)
    return sinkhorn_cost(alpha, beta, *pots)


# ---- the comparison -------------------------------------------------------------------------------------------------------------
def _feat(rs, n, E=128, shift=None):
    v = rs.randn(n, E).astype(np.float32) * 0.5 + (rs.randn(E).astype(np.float32) if shift is None else shift)
    return np.where(v > 0, v, 0.2 * v).astype(np.float32)          # LeakyReLU(0.2) features, as the discriminator's embedding


@pytest.mark.parametrize("n,m", [(8, 8), (64, 64), (5, 13), (1, 7), (24, 3)])
def test_c_oracle_equals_the_numpy_statement(oracle, n, m):
    rs = np.random.RandomState(100 * n + m)
    x, y = _feat(rs, n), _feat(rs, m)
    want = samples_loss_sinkhorn(x, y)
    got64 = oracle.sinkhorn_divergence(x, y, f64=True)
    got32 = oracle.sinkhorn_divergence(x, y)
    assert abs(got64 - want) <= 1e-9 * max(1.0, abs(want)), (got64, want)
    assert abs(got32 - want) <= 1e-6, (got32, want)


def test_other_blur_and_scaling(oracle):
    rs = np.random.RandomState(7)
    x, y = _feat(rs, 16), _feat(rs, 12)
    for blur, scaling in ((0.05, 0.5), (0.1, 0.7), (0.01, 0.9)):
        want = samples_loss_sinkhorn(x, y, blur=blur, scaling=scaling)
        assert abs(oracle.sinkhorn_divergence(x, y, blur=blur, scaling=scaling, f64=True) - want) <= 1e-9 * max(1.0, abs(want))
        assert abs(oracle.sinkhorn_divergence(x, y, blur=blur, scaling=scaling) - want) <= 2e-6


def test_rewards_of_a_search_batch(oracle):
    """the reward loop of search_dg.py:150-162 over the numpy statement == oracle.sinkhorn_rewards (D = 3, B = 8, M = 6)"""
    rs = np.random.RandomState(11)
    D, B, M = 3, 8, 6
    means = rs.randn(D, 128).astype(np.float32)
    fe = np.concatenate([_feat(rs, 1, shift=means[(s // 1) % D]) for s in range(D * B) for _ in range(M)])
    got = oracle.sinkhorn_rewards(fe, D, B, M)
    for j in range(M):
        cl = [fe[[(b * D + d) * M + j for b in range(B)]] for d in range(D)]
        want = sum(samples_loss_sinkhorn(cl[p], cl[q]) for p in range(D) for q in range(p + 1, D))
        assert abs(got[j] - want) <= 3e-6, (j, got[j], want)
