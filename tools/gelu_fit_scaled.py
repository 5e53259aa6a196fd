# tools/gelu_fit_scaled.py -- the transcendental-free GELU of convffn32_kernel in the SCALED variable y = x / 4 (round 3).
# W1 and b1 are divided by 4 at pack time (exact: a power of two), so the first product yields y; then
#     u   = clamp(y * y, 0, 1)                    one v_pk_mul_f32 with the clamp bit -- no v_med3 on y
#     phi = clamp(0.5 + y * Q(u), 0, 1)           Q of degree n-1, constraint 0.5 + Q(1) >= 1: beyond |y| = 1 the tails are exact
#     g   = y * phi  = GELU(x) / 4                the 4 goes into W2 at pack time (exact)
# LP minimax of |4 g - GELU(4 y)| over y in [0, 1] (odd/even symmetry covers y < 0), evaluated in fp32 like the kernel.
import numpy as np
from scipy.optimize import linprog
from scipy.special import erf
MARGIN = 4e-6
gelu = lambda x: 0.5 * x * (1 + erf(x / np.sqrt(2)))
def solve(n):
    y = np.linspace(0, 1, 4001)
    A = np.stack([4 * y ** (2 * k + 2) for k in range(n)], 1)      # 4 y * y * sum q_k y^{2k}
    b = gelu(4 * y) - 2 * y
    Aub = np.block([[A, -np.ones((len(y), 1))], [-A, -np.ones((len(y), 1))]])
    bub = np.concatenate([b, -b])
    Aub = np.vstack([Aub, np.concatenate([-np.ones(n), [0]])])       # Q(1) >= 0.5 + margin
    bub = np.append(bub, -0.5 - MARGIN)
    c = np.zeros(n + 1); c[-1] = 1
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=[(None, None)] * n + [(0, None)], method="highs")
    return r.x[:n], r.x[-1]
def approx32(q, x):
    y = (x / 4).astype(np.float32)
    u = np.clip(y * y, 0, 1).astype(np.float32)
    p = np.full_like(y, np.float32(q[-1]))
    for k in range(len(q) - 2, -1, -1):
        p = (p * u + np.float32(q[k])).astype(np.float32)
    phi = np.clip(np.float32(0.5) + y * p, 0, 1).astype(np.float32)
    return 4.0 * (y * phi).astype(np.float64)
xs = np.concatenate([np.linspace(-100, 100, 400001), np.linspace(-6, 6, 1200001)])
for n in (6, 7, 8):
    q, t = solve(n)
    e = np.max(np.abs(approx32(q, xs) - gelu(xs)))
    print(n, "coefficients: max |err| fp32 %.3g (lp %.3g)" % (e, t))
    print("   ", ", ".join("%.9gf" % v for v in q))
