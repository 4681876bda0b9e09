"""Reproduces the erf polynomial of csrc/fvit_common.h::gelu_fast and its error bound (numpy + scipy, CPU)."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf

Z, DEG = 3.0, 8
z = np.linspace(1e-6, Z, 400001)
t = 2 * z * z / (Z * Z) - 1
V = Ch.chebvander(t, DEG) * z[:, None]
w = np.ones_like(z)
for _ in range(60):  # Lawson iteration towards the minimax fit of erf(z) ~ z * Q(z^2)
    c, *_ = np.linalg.lstsq(V * w[:, None], erf(z) * w, rcond=None)
    e = np.abs(V @ c - erf(z))
    w = w * (e / e.max()) ** 0.5 + 1e-12
    w /= w.max()
pu = np.zeros(1)
for k, ck in enumerate(Ch.cheb2poly(c)):
    pu = P.polyadd(pu, ck * P.polypow([-1.0, 2 / (Z * Z)], k))
pu32 = pu.astype(np.float32)
print("Q coefficients (u^0 .. u^8):", ", ".join(f"{v:.9e}f" for v in pu32))
x = np.linspace(-8, 8, 2000001).astype(np.float32)
zz = np.clip(x * np.float32(0.70710678118654752), -Z, Z).astype(np.float32)
uu = (zz * zz).astype(np.float32)
q = np.full_like(uu, pu32[-1])
for ck in pu32[-2::-1]:
    q = (q * uu + ck).astype(np.float32)
hx = (np.float32(0.5) * x).astype(np.float32)
g = (hx * (zz * q).astype(np.float32) + hx).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
print("max |erf error| :", np.abs((zz * q).astype(np.float64) - erf(x.astype(np.float64) / np.sqrt(2))).max())
print("max |GELU error|:", np.abs(g - ref).max())
