#!/usr/bin/env python3
"""Fit the polynomial coefficients used by include/ptrace_fmath.h.

Least-squares fits on Chebyshev nodes (close to minimax), done in float64 and
rounded to float32.  Run once; the printed constants are pasted into the header.
"""
import numpy as np

def cheb_nodes(a, b, n=4000):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(fun_of_w, w_lo, w_hi, deg, weight=None):
    w = cheb_nodes(w_lo, w_hi)
    y = fun_of_w(w)
    A = np.vander(w, deg + 1, increasing=True)
    if weight is not None:
        wt = weight(w)
        A = A * wt[:, None]
        y = y * wt
    c, *_ = np.linalg.lstsq(A, y, rcond=None)
    return c

def show(name, c):
    c32 = np.asarray(c, dtype=np.float32)
    print(f"// {name}")
    for i, v in enumerate(c32):
        print(f"#define {name}_{i} {float(v).hex()}f  /* {v:.9e} */")
    return c32

# sin(r) = r + r^3 * S(w), w = r^2, r in [-pi/4, pi/4]
def S(w):
    r = np.sqrt(np.maximum(w, 1e-300))
    out = (np.sin(r) - r) / (r ** 3)
    return np.where(w < 1e-8, -1.0 / 6.0, out)
cs = show("PT_SIN_C", fit(S, 1e-6, (np.pi / 4 * 1.01) ** 2, 3))

# cos(r) = 1 - w/2 + w^2 * C(w)
def C(w):
    r = np.sqrt(w)
    return (np.cos(r) - 1.0 + 0.5 * w) / (w * w)
cc = show("PT_COS_C", fit(C, 1e-3, (np.pi / 4 * 1.01) ** 2, 3))

# log2(m) = s * L(w), s=(m-1)/(m+1), w = s^2, m in [sqrt(.5), sqrt(2)]
smax = (np.sqrt(2) - 1) / (np.sqrt(2) + 1)
def L(w):
    s = np.sqrt(w)
    return 2.0 * np.arctanh(s) / s / np.log(2.0)
cl = show("PT_LOG2_C", fit(L, 1e-8, (smax * 1.01) ** 2, 4))

# exp2(f) on [-0.5, 0.5] : direct polynomial in f
def E(f):
    return np.exp2(f)
ce = show("PT_EXP2_C", fit(E, -0.5, 0.5, 6))

# asin(z) = z + z^3 * R(w), w = z^2 <= 0.25
def R(w):
    z = np.sqrt(w)
    return (np.arcsin(z) - z) / z ** 3
cr = show("PT_ASIN_C", fit(R, 1e-6, 0.2501, 5))

# atan(z) = z + z^3 * Q(w), |z| <= tan(pi/8)
tmax = np.tan(np.pi / 8)
def Q(w):
    z = np.sqrt(w)
    return (np.arctan(z) - z) / z ** 3
cq = show("PT_ATAN_C", fit(Q, 1e-6, (tmax * 1.01) ** 2, 4))

# --- error report in float32 arithmetic (no fma; indicative only) ---
f32 = np.float32
def horner(c, w):
    acc = np.full_like(w, c[-1], dtype=np.float32)
    for k in c[-2::-1]:
        acc = acc * w + f32(k)
    return acc
r = np.linspace(-np.pi / 4, np.pi / 4, 200001).astype(np.float32)
w = r * r
print("sin err", np.max(np.abs((r + r * w * horner(cs, w)).astype(np.float64) - np.sin(r.astype(np.float64)))))
print("cos err", np.max(np.abs((f32(1) - f32(.5) * w + w * w * horner(cc, w)).astype(np.float64) - np.cos(r.astype(np.float64)))))
m = np.linspace(np.sqrt(.5), np.sqrt(2), 200001).astype(np.float32)
s = (m - f32(1)) / (m + f32(1)); w = s * s
print("log2 err", np.max(np.abs((s * horner(cl, w)).astype(np.float64) - np.log2(m.astype(np.float64)))))
f = np.linspace(-.5, .5, 200001).astype(np.float32)
print("exp2 relerr", np.max(np.abs(horner(ce, f).astype(np.float64) / np.exp2(f.astype(np.float64)) - 1)))
z = np.linspace(-.5, .5, 200001).astype(np.float32); w = z * z
print("asin err", np.max(np.abs((z + z * w * horner(cr, w)).astype(np.float64) - np.arcsin(z.astype(np.float64)))))
z = np.linspace(-tmax, tmax, 200001).astype(np.float32); w = z * z
print("atan err", np.max(np.abs((z + z * w * horner(cq, w)).astype(np.float64) - np.arctan(z.astype(np.float64)))))
