# -*- coding: utf-8 -*-
"""
Coefficients of the per-node-sample 2^f polynomial of qm_kernels.hpp (development aid).

Remez exchange in 60-digit arithmetic (mpmath) for the polynomial P of degree D that minimises
max |P(f) / 2^f - 1| over f in [-1/2, 1/2]; the coefficients are then rounded to float64 and
the error of the ROUNDED polynomial (evaluated exactly) is reported -- that is the truncation
bound quoted in the kernel; the D Horner roundings add at most ~D * 1.2e-16 to it.

usage: python tools/exp2_minimax.py [degrees ...]
"""
import sys

from mpmath import mp, mpf, matrix, lu_solve, cos, pi, power, findroot, diff

mp.dps = 60


def remez(deg, iters=12):
    n = deg + 2
    xs = [mpf("0.5") * cos(pi * k / (n - 1)) for k in range(n)][::-1]
    coeff = None
    for _ in range(iters):
        # solve sum_i c_i x^i / 2^x - 1 = (-1)^k E  at the n reference points
        A = matrix(n, n)
        b = matrix(n, 1)
        for k, x in enumerate(xs):
            w = power(2, -x)
            for i in range(deg + 1):
                A[k, i] = x ** i * w
            A[k, deg + 1] = (-1) ** k
            b[k] = 1
        sol = lu_solve(A, b)
        coeff = [sol[i] for i in range(deg + 1)]

        def err(x):
            return sum(c * x ** i for i, c in enumerate(coeff)) * power(2, -x) - 1

        # new reference: extrema of err between consecutive zeros (dense scan + polish)
        grid = [mpf(-0.5) + mpf(k) / 4000 for k in range(4001)]
        vals = [err(x) for x in grid]
        ext = [grid[0]]
        for k in range(1, 4000):
            if (vals[k] - vals[k - 1]) * (vals[k + 1] - vals[k]) <= 0:
                try:
                    r = findroot(lambda t: diff(err, t), grid[k])
                    if grid[k - 1] <= r <= grid[k + 1]:
                        ext.append(r)
                        continue
                except Exception:
                    pass
                ext.append(grid[k])
        ext.append(grid[-1])
        # keep n alternating extrema of largest magnitude
        pts = []
        for x in ext:
            if pts and (err(x) > 0) == (err(pts[-1]) > 0):
                if abs(err(x)) > abs(err(pts[-1])):
                    pts[-1] = x
            else:
                pts.append(x)
        if len(pts) < n:
            break
        while len(pts) > n:
            if abs(err(pts[0])) < abs(err(pts[-1])):
                pts.pop(0)
            else:
                pts.pop()
        xs = pts
    return coeff


def main():
    degrees = [int(a) for a in sys.argv[1:]] or [6, 7, 8]
    for deg in degrees:
        c = remez(deg)
        cd = [float(x) for x in c]
        worst = mpf(0)
        for k in range(20001):
            x = mpf(-0.5) + mpf(k) / 20000
            p = sum(mpf(v) * x ** i for i, v in enumerate(cd))
            worst = max(worst, abs(p * power(2, -x) - 1))
        print(f"// degree {deg}: max |P(f)/2^f - 1| on [-1/2, 1/2] = {float(worst):.3e} "
              f"(float64-rounded coefficients, exact evaluation)")
        print("{" + ", ".join(repr(v) for v in cd) + "}")


if __name__ == "__main__":
    main()
