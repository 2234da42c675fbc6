"""tests/golden/ssim_hand_derived.json: SSIM values DERIVED BY HAND for images whose window statistics are known in closed form,
evaluated in exact rational arithmetic — a pin of matchnerf_amd/metrics.py:ssim that does not go through any implementation of
the filter.  Definition: scikit-image 0.19 `structural_similarity(pred, gt, channel_axis=-1)` with its defaults (7 x 7 uniform
window, sample covariance, K1 = 0.01, K2 = 0.03; float images without data_range -> data_range 2), the call at
/root/reference/misc/metrics.py:43-45.  usage: python tools/gen_ssim_golden.py"""
import json
import os
from fractions import Fraction as F

K1, K2, L = F(1, 100), F(3, 100), F(2)
C1, C2 = (K1 * L) ** 2, (K2 * L) ** 2
cases = []

# A: two constant images a, b.  Variances and the covariance vanish in every window.
a, b = F(1, 5), F(3, 5)
ssim_a = (2 * a * b + C1) / (a * a + b * b + C1)
assert ssim_a == F(601, 1001)
cases.append(dict(name="constant_images", kind="constant", a=float(a), b=float(b), data_range=2.0, shape=[16, 20, 3],
                  ssim=float(ssim_a),
                  derivation="variances and covariance vanish in every window: SSIM = (2ab + C1)/(a^2 + b^2 + C1) * (C2/C2), "
                             "C1 = (0.01*2)^2; a = 1/5, b = 3/5: (0.24 + 0.0004)/(0.40 + 0.0004) = 601/1001"))

# B: opposite vertical stripes of period 2: x = m + a s(j), y = m - a s(j), s = +-1 alternating per column.
m, amp = F(1, 2), F(1, 4)
mux, muy = m + amp / 7, m - amp / 7  # 4 columns of one sign, 3 of the other in every 7 x 7 window (either way round: symmetric)
var, cov = amp * amp, -amp * amp     # population variance a^2 (1 - 1/49), times 49/48 (sample form) = a^2
ssim_b = (2 * mux * muy + C1) / (mux * mux + muy * muy + C1) * (2 * cov + C2) / (2 * var + C2)
cases.append(dict(name="opposite_stripes", kind="stripes", m=float(m), amp=float(amp), data_range=2.0, shape=[21, 30, 3],
                  ssim=float(ssim_b),
                  derivation="every 7x7 window has 4 columns of one sign and 3 of the other: window means m +- a/7 (x) and m -+ a/7 "
                             "(y), population variance a^2 (1 - 1/49), times 49/48 for the sample form = a^2, covariance -a^2. SSIM = "
                             "(2 (m^2 - a^2/49) + C1)/(2 m^2 + 2 a^2/49 + C1) * (-2 a^2 + C2)/(2 a^2 + C2), the same in every window; "
                             "C1 = 0.0004, C2 = 0.0036, m = 1/2, a = 1/4"))

# C: the stripes against themselves at half the contrast: y = m + (a/2) s(j).
cs = (2 * (amp * amp / 2) + C2) / (amp * amp + amp * amp / 4 + C2)
vals = []
for sign in (1, -1):  # windows that start on a + column / on a - column: 12 window columns of each kind at width 30
    ux, uy = m + sign * amp / 7, m + sign * amp / 14
    vals.append((2 * ux * uy + C1) / (ux * ux + uy * uy + C1) * cs)
ssim_c = sum(vals) / 2
cases.append(dict(name="half_contrast_stripes", kind="stripes_scaled", m=float(m), amp=float(amp), data_range=2.0, shape=[21, 30, 3],
                  ssim=float(ssim_c),
                  derivation="sample variances a^2 and a^2/4, covariance a^2/2; window means (m + a/7, m + a/14) for windows that "
                             "start on a + column and (m - a/7, m - a/14) for the others; width 30 gives 24 window columns, 12 of each "
                             "kind: SSIM = mean of the two values"))

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ssim_hand_derived.json")
json.dump(dict(source="hand-derived closed forms of scikit-image 0.19 structural_similarity(channel_axis=-1) defaults (7x7 uniform "
                      "window, sample covariance, K1 = 0.01, K2 = 0.03, float images without data_range -> 2): the call at "
                      "/root/reference/misc/metrics.py:43-45; exact rationals from tools/gen_ssim_golden.py",
               cases=cases), open(out, "w"), indent=1)
print(out, [c["ssim"] for c in cases])
