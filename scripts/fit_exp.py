# minimax-ish (Chebyshev interpolation) fit of g(r) = (exp(r) - 1 - r) / r^2 on [-b, b], b = ln2/2 * (1 + 2^-10), degree 9
from decimal import Decimal as D, getcontext
import math, struct
getcontext().prec = 80
def dexp(x):
    return x.exp()
def g(r):
    if abs(r) < D(10) ** -30:
        return D(1) / 2
    return (dexp(r) - 1 - r) / (r * r)
pi = D("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")
def dcos(x):   # Taylor, x in [0, pi]
    s, t, k = D(1), D(1), 0
    while abs(t) > D(10) ** -75:
        k += 2
        t = -t * x * x / (k * (k - 1))
        s += t
    return s
deg = 9
b = D(2).ln() / 2 * (1 + D(2) ** -10)
n = deg + 1
nodes = [b * dcos(pi * (2 * i + 1) / (2 * n)) for i in range(n)]
vals = [g(x) for x in nodes]
# Newton divided differences -> monomial coefficients
coef = list(vals)
for j in range(1, n):
    for i in range(n - 1, j - 1, -1):
        coef[i] = (coef[i] - coef[i - 1]) / (nodes[i] - nodes[i - j])
# expand Newton form to monomials
poly = [D(0)] * n
poly[0] = coef[n - 1]
for i in range(n - 2, -1, -1):
    # poly = poly * (x - nodes[i]) + coef[i]
    new = [D(0)] * n
    for k in range(n - 1):
        new[k + 1] += poly[k]
        new[k] -= poly[k] * nodes[i]
    new[0] += coef[i]
    poly = new
print("coefficients of g (c0 + c1 r + ...):")
for k, c in enumerate(poly):
    f = float(c)
    print(k, f.hex(), repr(f), " taylor", float(D(1) / math.factorial(k + 2)))
# error of the double-rounded polynomial in exact arithmetic
cf = [D(float(c)) for c in poly]
worst = D(0)
N = 4001
for i in range(N):
    r = -b + 2 * b * i / (N - 1)
    p = D(0)
    for c in reversed(cf):
        p = p * r + c
    e = 1 + r + r * r * p
    rel = abs(e / dexp(r) - 1)
    worst = max(worst, rel)
print("max rel err of exact evaluation:", float(worst), " = %.3f ulp" % float(worst / D(2) ** -53))
