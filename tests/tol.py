"""The floating-point bar of the GPU parity tests, in one place.

north_star: embeddings, scores, tables and losses within 1e-5 relative (fp32).  Checked twice: the max-norm
|a - b|_max <= RTOL |b|_max, and ELEMENTWISE |a - b| <= RTOL |b| + atol.  The absolute term is the float32 rounding of
sums the GPU forms in float32 and the host restatement in float64 (an element near zero after cancellation has no
relative accuracy to speak of); every caller states its own and why:
    ATOL_ROW    2e-6   one or two SGD updates of table rows touched by up to ~10^3 lookups (ulp(10^2) * lr ~ 1e-6)
    ATOL_TABLE  2e-5   small tables followed for tens of steps: a few hundred rows, each shared by thousands of lookups per
                       step, so row values reach ~10 and tens of updates of |sum| ~ 10^2..10^3 accumulate
    ATOL_SCORE  1e-6   scores (O(1): norm * batch / field sum, the sum of <= 10^5 norms in float32)"""
import numpy as np

RTOL = 1e-5
ATOL_ROW = 2e-6
ATOL_TABLE = 2e-5
ATOL_SCORE = 1e-6


def close(a, b, what, rtol=RTOL, atol=ATOL_ROW):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    if a.size == 0:
        return
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max-norm rel err {err:.3e}"
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements beyond {rtol} |b| + {atol}; worst "
                           f"|a-b| = {np.abs(a - b)[bad].max():.3e}")
