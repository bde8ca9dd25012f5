"""`InterpTypes` under the import path the reference's users expect
(cavour/market/curves/interpolator.py:18-26 defines a second copy of the enum
with the same values as cavour/utils/global_types.py:76-84; one shared enum is
enough because the engine only compares ``.value``)."""
from ...utils.global_types import InterpTypes  # noqa: F401

import numpy as np

from ...utils.error import LibError


def _point(t, times, dfs, method):
    """One discount factor from the node arrays (cavour/market/curves/interpolator.py:69-170).

    ``i`` ends up as the first node with ``times[i] >= t`` (1-based segments), or ``n`` when ``t`` lies
    beyond the last node; the first segment and the extrapolation region have their own formulas:
    LINEAR_ZERO_RATES holds the first / last node's zero rate flat, FLAT_FWD_RATES interpolates
    -ln(df) linearly (extrapolating with the last segment's slope), LINEAR_FWD_RATES interpolates the
    segments' forward rates."""
    n = times.size
    if t == times[0]:
        return dfs[0]
    i = 0
    while times[i] < t and i < n - 1:
        i += 1
    if t > times[i]:
        i = n

    if method == InterpTypes.LINEAR_ZERO_RATES.value:
        if i == 1:
            r1 = r2 = -np.log(dfs[1]) / times[1]
            lo, hi = times[0], times[1]
        elif i < n:
            r1 = -np.log(dfs[i - 1]) / times[i - 1]
            r2 = -np.log(dfs[i]) / times[i]
            lo, hi = times[i - 1], times[i]
        else:
            r1 = r2 = -np.log(dfs[n - 1]) / times[n - 1]
            lo, hi = times[n - 2], times[n - 1]
        rate = ((hi - t) * r1 + (t - lo) * r2) / (hi - lo)
        return np.exp(-rate * t)

    if method == InterpTypes.FLAT_FWD_RATES.value:
        a, b = (i - 1, i) if i < n else (n - 2, n - 1)
        rt1, rt2 = -np.log(dfs[a]), -np.log(dfs[b])
        rt = ((times[b] - t) * rt1 + (t - times[a]) * rt2) / (times[b] - times[a])
        return np.exp(-rt)

    if method == InterpTypes.LINEAR_FWD_RATES.value:
        small = 1e-10
        if i == 1:
            return np.exp(-(t * -np.log(dfs[1] + small) / (times[1] + small)))
        fwd1 = -np.log(dfs[i - 1] / dfs[i - 2]) / (times[i - 1] - times[i - 2])
        if i < n:
            fwd2 = -np.log(dfs[i] / dfs[i - 1]) / (times[i] - times[i - 1])
            fwd = ((times[i] - t) * fwd1 + (t - times[i - 1]) * fwd2) / (times[i] - times[i - 1])
        else:
            fwd = fwd1
        return dfs[i - 1] * np.exp(-fwd * (t - times[i - 1]))

    raise LibError("Invalid interpolation scheme.")


def interpolate(t, times, dfs, method: int):
    """Discount factor(s) at time(s) ``t`` (interpolator.py:35-62): float in, float out; array in, array out."""
    times = np.asarray(times, dtype=np.float64)
    dfs = np.asarray(dfs, dtype=np.float64)
    if isinstance(t, (float, np.float64)):
        if t < 0.0:
            raise LibError("Interpolate times must all be >= 0")
        return _point(t, times, dfs, method)
    if isinstance(t, np.ndarray):
        if np.any(t < 0.0):
            raise LibError("Interpolate times must all be >= 0")
        return np.array([_point(x, times, dfs, method) for x in t.ravel()]).reshape(t.shape)
    raise LibError("Unknown input type" + str(type(t)))
