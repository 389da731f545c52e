"""Numpy prototype of the block active-set split-weight solver (development aid; the product is csrc/fnn_splits.hip).
usage: block_nnls.py n [kfrac] [kmin]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import nnet_oracle as O


def prefix2(X):
    return np.cumsum(np.cumsum(X, axis=0), axis=1)


def Pm(P, i, j):  # P[i][j] with -1 -> 0 (vectorised)
    i = np.asarray(i); j = np.asarray(j)
    out = P[np.maximum(i, 0), np.maximum(j, 0)]
    return np.where((i < 0) | (j < 0), 0.0, out)


def A_mul(X):  # X upper grid of split weights -> pair grid (a<b)
    n = X.shape[0]
    P = prefix2(X)
    a, b = np.triu_indices(n, 1)
    first = Pm(P, a - 1, b - 1) - Pm(P, a - 1, a - 1)
    second = (Pm(P, b - 1, n - 1) - Pm(P, a - 1, n - 1)) - (Pm(P, b - 1, b - 1) - Pm(P, a - 1, b - 1))
    Y = np.zeros_like(X)
    Y[a, b] = first + second
    return Y


def At_mul(Y):
    n = Y.shape[0]
    P = prefix2(Y)
    S = Y + Y.T
    RS = np.cumsum(S.sum(axis=1))
    i, j = np.triu_indices(n, 1)
    out = np.zeros_like(Y)
    out[i, j] = (RS[j] - RS[i]) - 2.0 * (P[j, j] - P[i, j])
    return out


def h_entry(n, i, j, k, l):
    lo = np.maximum(i, k); hi = np.minimum(j, l)
    st = np.maximum(hi - lo, 0).astype(np.float64)
    s = (j - i).astype(np.float64); t = (l - k).astype(np.float64)
    return st * (n - s - t + st) + (s - st) * (t - st)


def Hblock(n, Fa, Fb):
    return h_entry(n, Fa[:, 0][:, None], Fa[:, 1][:, None], Fb[:, 0][None, :], Fb[:, 1][None, :])


class Inv:
    """explicit inverse of H_FF with block border / block eliminate"""
    def __init__(self, n):
        self.n = n; self.F = np.zeros((0, 2), dtype=np.int64); self.G = np.zeros((0, 0)); self.flops = 0.0
    def add(self, K):
        f = len(self.F); k = len(K)
        C = Hblock(self.n, K, K)
        if f == 0:
            self.G = np.linalg.inv(C); self.F = K.copy(); return
        B = Hblock(self.n, self.F, K)
        U = self.G @ B
        S = C - B.T @ U
        Si = np.linalg.inv(S)
        V = U @ Si
        G = np.empty((f + k, f + k))
        G[:f, :f] = self.G + V @ U.T
        G[:f, f:] = -V; G[f:, :f] = -V.T; G[f:, f:] = Si
        self.G = G; self.F = np.vstack([self.F, K]); self.flops += 4.0 * f * f * k
    def remove(self, idx):
        f = len(self.F); idx = np.asarray(idx); r = len(idx)
        keep = np.setdiff1d(np.arange(f), idx)
        W = self.G[:, idx]
        T = np.linalg.inv(self.G[np.ix_(idx, idx)])
        G = self.G - W @ T @ W.T
        self.G = G[np.ix_(keep, keep)]; self.F = self.F[keep]; self.flops += 2.0 * f * f * r


def solve(n, seed=1, kfrac=0.25, kmin=8, kmax=10**9, verbose=True, mode="all"):
    D = O.synth(n, seed)
    order, _, _ = O.run(D, threads=2, want_events=False)
    p = order[1:] - 1
    d = np.triu(D[np.ix_(p, p)], 1)
    c = At_mul(d)
    iu = np.triu_indices(n, 1)
    cmax = c[iu].max(); tol = 1e-12 * cmax
    inv = Inv(n)
    xF = np.zeros(0); phi = 0.0
    outer = inner = adds = removes = rejects = 0
    rem_old = 0; exact = os.environ.get('EXACT') == '1'
    k_scale = 1.0
    S0 = int(os.environ.get('S0', 0))
    if S0:
        ii, jj = np.triu_indices(n, 1); sz = jj - ii; sel = (np.minimum(sz, n - sz) <= S0)
        F0 = np.stack([ii[sel], jj[sel]], 1)
        inv.add(F0); adds += len(F0)
        while True:
            inner += 1
            cF = c[inv.F[:, 0], inv.F[:, 1]]; s = np.linalg.solve(Hblock(n, inv.F, inv.F), cF) if exact else inv.G @ cF
            neg = np.nonzero(s <= 0)[0]
            if len(neg) == 0: break
            inv.remove(neg); removes += len(neg)
            if exact: inv.G = np.linalg.inv(Hblock(n, inv.F, inv.F))
        xF = s; phi = -0.5 * cF @ s
        print(f" start: S0={S0} |F0|={len(F0)} -> {len(inv.F)} after {inner} solves, phi={phi:.6e}")
    while True:
        outer += 1
        X = np.zeros((n, n)); X[inv.F[:, 0], inv.F[:, 1]] = xF
        w = c - At_mul(A_mul(X))
        wm = w.copy(); wm[inv.F[:, 0], inv.F[:, 1]] = -np.inf; wm[np.tril_indices(n)] = -np.inf
        NMS = int(os.environ.get('NMS', 0))
        ncand_all = int((wm > tol).sum())
        if NMS and ncand_all > 0:
            from scipy.ndimage import maximum_filter
            wz = np.where(np.isfinite(wm), wm, -1e300)
            if os.environ.get('SCORE') == 'norm':
                ii, jj = np.indices((n, n)); szz = np.clip(jj - ii, 1, n - 1).astype(np.float64); wz = np.where(wz > 0, wz / np.sqrt(szz * (n - szz)), wz)
            mx = maximum_filter(wz, size=2 * NMS + 1, mode='constant', cval=-1e300)
            loc = (wz >= mx) & (wm > tol)
            cand = np.argwhere(loc)
        else:
            cand = np.argwhere(wm > tol)
        if len(cand) == 0:
            break
        vals = wm[cand[:, 0], cand[:, 1]]
        SC = os.environ.get('SCORE', 'w')
        if SC == 'norm':
            sz = (cand[:, 1] - cand[:, 0]).astype(np.float64); vals = vals / np.sqrt(sz * (n - sz))
        k = int(min(max(kmin, kfrac * k_scale * max(len(inv.F), 1)), len(cand), kmax))
        k = max(k, 1)
        top = cand[np.argsort(-vals)[:k]]
        # save state
        sF, sG, sx = inv.F.copy(), inv.G.copy(), xF.copy()
        fold = len(inv.F); fold0 = fold
        inv.add(top); adds += len(top)
        if exact: inv.G = np.linalg.inv(Hblock(n, inv.F, inv.F))
        xcur = np.concatenate([xF, np.zeros(len(top))])
        while True:
            inner += 1
            cF = c[inv.F[:, 0], inv.F[:, 1]]
            s = inv.G @ cF
            neg = np.nonzero(s <= 0)[0]
            if len(neg) == 0:
                break
            if mode == "all":
                rem = neg
            else:  # Lawson-Hanson ratio step, remove everything that reaches zero
                al = np.min(xcur[neg] / (xcur[neg] - s[neg]))
                xcur = xcur + al * (s - xcur)
                rem = np.nonzero((xcur <= 1e-300) & (s <= 0))[0]
                if len(rem) == 0:
                    rem = neg[np.argmin(xcur[neg] / (xcur[neg] - s[neg]))][None]
            keep = np.setdiff1d(np.arange(len(inv.F)), rem)
            rem_old += int((rem < fold).sum()); fold -= int((rem < fold).sum())
            inv.remove(rem); removes += len(rem)
            if exact: inv.G = np.linalg.inv(Hblock(n, inv.F, inv.F))
            xcur = xcur[keep]
        cF = c[inv.F[:, 0], inv.F[:, 1]]
        phi_new = -0.5 * cF @ s
        if not (phi_new < phi):
            rejects += 1
            inv.F, inv.G, xF = sF, sG, sx
            if k == 1 and mode == "lh":
                raise RuntimeError("no progress with k = 1 in LH mode")
            if k == 1:
                mode = "lh"
            k_scale *= 0.25
            if kfrac * k_scale * max(len(inv.F), 1) < 1: kmin = 1
            if verbose: print(f"  reject at |F|={len(inv.F)} k={k} phi {phi_new} vs {phi}")
            continue
        if os.environ.get('TRACE'): print(f'  it {outer}: f {fold0}+{k} -> {len(inv.F)} (old kept {fold}) cand {len(cand)} phi {phi_new:.8e}')
        phi = phi_new; xF = s; k_scale = min(1.0, k_scale * 2)
        if verbose and outer % int(os.environ.get('EVERY', 10)) == 0:
            print(f" outer {outer} |F|={len(inv.F)} cand={len(cand)} k={k} inner={inner} phi={phi:.6e}", flush=True)
    return dict(n=n, outer=outer, inner=inner, adds=adds, removes=removes, rejects=rejects, rem_old=rem_old, F=len(inv.F), flops=inv.flops,
                flops_over_F3=inv.flops / max(len(inv.F), 1) ** 3), inv, xF, c


if __name__ == "__main__":
    n = int(sys.argv[1]); kfrac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    kmin = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    mode = sys.argv[4] if len(sys.argv) > 4 else "all"
    t = time.time()
    st, inv, xF, c = solve(n, kfrac=kfrac, kmin=kmin, mode=mode)
    print(st, f"{time.time()-t:.1f}s")
    # KKT check
    X = np.zeros((n, n)); X[inv.F[:, 0], inv.F[:, 1]] = xF
    w = c - At_mul(A_mul(X))
    iu = np.triu_indices(n, 1)
    wz = w.copy(); wz[inv.F[:, 0], inv.F[:, 1]] = 0
    print("kkt: max w on Z", wz[iu].max() / c[iu].max(), "max |w| on F", np.abs(w[inv.F[:, 0], inv.F[:, 1]]).max() / c[iu].max(), "min x", xF.min())
