// The fast kernels with the LINEAR_FWD_RATES node arithmetic compiled in (InterpolatorAd.simple_interpolate's third
// scheme, cavour/market/curves/interpolator_ad.py:236-238: the discount factor itself is interpolated linearly).
// Same source as kernels_fast.hip; see the note at its top.
#define ADR_FAST_LINDF 1
#include "kernels_fast.hip"
