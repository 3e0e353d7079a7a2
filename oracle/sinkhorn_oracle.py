"""CPU restatement of SuperGlue's log-domain optimal transport and of the hand-derived
reverse sweep used by the HIP backward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference restated: gluefactory_nonfree/superglue.py:186-191 (log_sinkhorn_iterations) and
:194-214 (log_optimal_transport).  ``backward_recurrence`` is the analytic gradient that
csrc/sinkhorn.hip implements (storing only the u/v iterates); ``tests/test_oracle_golden.py``
checks it against autograd of the restated forward in fp64 and the forward against the
reference's own output (tests/golden/superglue_ot.npz).
"""
import math

import torch


def couplings(scores, alpha):
    b, m, n = scores.shape
    a = alpha.to(scores).reshape(1, 1, 1)
    top = torch.cat([scores, a.expand(b, m, 1)], -1)
    bot = torch.cat([a.expand(b, 1, n), a.expand(b, 1, 1)], -1)
    return torch.cat([top, bot], 1)


def marginals(m, n, like):
    norm = -math.log(m + n)
    log_mu = torch.full((m + 1,), norm, dtype=like.dtype, device=like.device)
    log_nu = torch.full((n + 1,), norm, dtype=like.dtype, device=like.device)
    log_mu[-1] = math.log(n) + norm
    log_nu[-1] = math.log(m) + norm
    return log_mu, log_nu, norm


def sinkhorn(Z, log_mu, log_nu, iters):
    """Returns (Z + u + v, u_hist [T,B,R], v_hist [T,B,C])."""
    b = Z.shape[0]
    u = torch.zeros(b, Z.shape[1], dtype=Z.dtype, device=Z.device)
    v = torch.zeros(b, Z.shape[2], dtype=Z.dtype, device=Z.device)
    uh, vh = [], []
    for _ in range(iters):
        u = log_mu[None] - torch.logsumexp(Z + v[:, None, :], dim=2)
        v = log_nu[None] - torch.logsumexp(Z + u[:, :, None], dim=1)
        uh.append(u)
        vh.append(v)
    out = Z + u[:, :, None] + v[:, None, :]
    if iters == 0:
        return out, Z.new_zeros(0, *u.shape), Z.new_zeros(0, *v.shape)
    return out, torch.stack(uh), torch.stack(vh)


def log_optimal_transport(scores, alpha, iters):
    b, m, n = scores.shape
    Z = couplings(scores, alpha)
    log_mu, log_nu, norm = marginals(m, n, scores)
    out, _, _ = sinkhorn(Z, log_mu, log_nu, iters)
    return out - norm


def backward_recurrence(Z, gout, u_hist, v_hist, log_mu, log_nu):
    """dL/dZ of out = Z + u^T + v^T given gout, from the stored iterates only.

    ubar^k_i = [k==T] rowsum(G)_i - sum_j Q^k_ij vbar^k_j,   Q^k = exp(Z + u^k + v^k - log_nu)
    vbar^{k-1}_j = - sum_i R^k_ij ubar^k_i,                  R^k = exp(Z + u^k + v^{k-1} - log_mu)
    dZ = G - sum_k [ Q^k * vbar^k (cols) + R^k * ubar^k (rows) ],  vbar^T = colsum(G), v^0 = 0.
    """
    T = u_hist.shape[0]
    gZ = gout.clone()
    vbar = gout.sum(1)
    ubar0 = gout.sum(2)
    for k in range(T, 0, -1):
        u, v = u_hist[k - 1], v_hist[k - 1]
        vprev = v_hist[k - 2] if k >= 2 else torch.zeros_like(v)
        Q = torch.exp(Z + u[:, :, None] + (v - log_nu[None])[:, None, :])
        ubar = (ubar0 if k == T else 0.0) - (Q * vbar[:, None, :]).sum(2)
        gZ = gZ - Q * vbar[:, None, :]
        R = torch.exp(Z + (u - log_mu[None])[:, :, None] + vprev[:, None, :])
        gZ = gZ - R * ubar[:, :, None]
        vbar = -(R * ubar[:, :, None]).sum(1)
    return gZ
