"""Is ONE Markstein correction after a Newton-refined v_rcp_f32 enough for a correctly rounded fp32 quotient?  (K1 / K3 sample
positions, csrc/warp_lds.hip positions2.)  numpy emulation: fma through float64 (exact for fp32 operands), v_rcp_f32 modelled as
the correctly rounded reciprocal moved by -1 / 0 / +1 ulp at random.  Prints the number of quotients that differ from RN(a / b)."""
import numpy as np
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
rng = np.random.default_rng(1)
for label, zlo, zhi, alo, ahi, N in (("DTU range", 400.0, 950.0, 1e-1, 4e5, 4_000_000), ("wide", 1e-2, 1e5, 1e-3, 1e7, 8_000_000)):
    z = (np.exp(rng.uniform(np.log(zlo), np.log(zhi), N)) * rng.choice([-1, 1], N)).astype(f32)
    a = (np.exp(rng.uniform(np.log(alo), np.log(ahi), N)) * rng.choice([-1, 1], N)).astype(f32)
    exact = (a.astype(np.float64) / z.astype(np.float64)).astype(f32)
    y0 = (1.0 / z.astype(np.float64)).astype(f32)
    k = rng.integers(-1, 2, N)
    y0 = np.where(k == 0, y0, np.nextafter(y0, np.where(k > 0, np.inf, -np.inf).astype(f32)))
    e = fma(-z, y0, np.ones(N, f32)); y = fma(e, y0, y0)                  # one Newton step
    q = (a * y).astype(f32); r = fma(-z, q, a); q1 = fma(r, y, q)         # one Markstein correction
    r2 = fma(-z, q1, a); q2 = fma(r2, y, q1)                              # a second one
    print(f"{label}: {N} operand pairs; wrong quotients: no correction {(q != exact).sum()}, one {(q1 != exact).sum()}, two {(q2 != exact).sum()}")
