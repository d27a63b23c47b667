"""Prototype: W_m(z) as monomial (Horner) polynomials vs the recurrence; accuracy for L up to 20."""
import numpy as np
from numpy.polynomial import polynomial as Pn
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import shapes

def alpha(n, m): return np.sqrt((4.0*n*n-1)/(n*n-m*m))
def beta(n, m): return np.sqrt(((2.0*n+1)*(n+m-1)*(n-m-1))/((n-m)*(n+m)*(2.0*n-3)))
def pmm(m):
    p = np.sqrt(1/(4*np.pi))
    for k in range(1, m+1): p = -p*np.sqrt((2.0*k+1)/(2.0*k))
    return p

for L in (6, 12, 16, 20):
    anm = shapes.random_shape(L, 3, amp=0.1).reshape(-1, 2)
    rng = np.random.default_rng(0)
    zs = rng.uniform(-1, 1, 2000)
    worst = 0.0
    for m in range(L+1):
        # Pi_n^m(z) polynomials in z (np.longdouble), monomial coefficients ascending
        polys = {}
        polys[m] = np.array([pmm(m)], dtype=np.longdouble)
        if m+1 <= L: polys[m+1] = np.array([0, alpha(m+1, m)*pmm(m)], dtype=np.longdouble)
        for n in range(m+2, L+1):
            a, b = np.longdouble(alpha(n, m)), np.longdouble(beta(n, m))
            polys[n] = Pn.polysub(Pn.polymul([0, a], polys[n-1]), b*polys[n-2])
        fac = 1.0 if m == 0 else 2.0
        wr = np.zeros(L-m+1, dtype=np.longdouble)
        for n in range(m, L+1):
            c = fac*anm[n*(n+1)//2+m, 0]
            wr[:len(polys[n])] += c*polys[n]
        wr64 = wr.astype(np.float64)
        # Horner in float64
        h = np.zeros_like(zs)
        for k in range(L-m, -1, -1): h = h*zs + wr64[k]
        # reference: recurrence in float64
        p2 = np.zeros_like(zs); p1 = np.full_like(zs, pmm(m)); ref = fac*anm[m*(m+1)//2+m, 0]*p1
        for n in range(m+1, L+1):
            b = 0.0 if n-m < 2 else beta(n, m)
            p = alpha(n, m)*zs*p1 - b*p2
            ref = ref + fac*anm[n*(n+1)//2+m, 0]*p
            p2, p1 = p1, p
        # exact in longdouble
        ex = np.zeros_like(zs, dtype=np.longdouble)
        for k in range(L-m, -1, -1): ex = ex*zs.astype(np.longdouble) + wr[k]
        worst = max(worst, np.abs(h-ex.astype(np.float64)).max())
        rec = np.abs(ref-ex.astype(np.float64)).max()
    print(f"L={L}: max |Horner64 - exact| = {worst:.2e}  (recurrence err of last m: {rec:.1e}), max |coef| = {np.abs(wr64).max():.2e}")
