"""Numpy prototype, second form: the free set's normal equations through W = L^-1 (H = L L^T), block appends that never
touch existing entries, deletions as a projection (deleted set R, Gram matrix of W's columns), periodic refactoring.
Development aid for csrc/fnn_splits.hip.   usage: block_nnls2.py n [kfrac] [nms] [rfrac]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.linalg as sl
from scipy.ndimage import maximum_filter
from block_nnls import At_mul, A_mul, Hblock, O


DELPOS = []; DELAGE = []; SMAX = [int(os.environ.get('SMAX', 0))]
class Factor:
    def __init__(self, n, c):
        self.n = n; self.c = c
        self.F = np.zeros((0, 2), dtype=np.int64); self.W = np.zeros((0, 0)); self.z = np.zeros(0)
        self.dead = np.zeros(0, dtype=bool)
        self.flops = 0.0; self.refactors = 0
    def cF(self, F): return self.c[F[:, 0], F[:, 1]]
    def append(self, K):
        f = len(self.F); k = len(K)
        C = Hblock(self.n, K, K)
        if f:
            B = Hblock(self.n, self.F, K)
            T = self.W @ B; S = C - T.T @ T
        else:
            T = np.zeros((0, k)); S = C
        L22 = np.linalg.cholesky(S)      # raises LinAlgError if not positive definite
        L22i = sl.solve_triangular(L22, np.eye(k), lower=True)
        W = np.zeros((f + k, f + k))
        W[:f, :f] = self.W
        if f: W[f:, :f] = -L22i @ (T.T @ self.W)
        W[f:, f:] = L22i
        zk = L22i @ (self.cF(K) - (T.T @ self.z if f else 0.0))
        self.W = W; self.F = np.vstack([self.F, K]); self.z = np.concatenate([self.z, zk]); self.dead = np.concatenate([self.dead, np.zeros(k, bool)])
        self.flops += 2.0 * f * f * k
    def truncate(self, f, dead):
        self.W = self.W[:f, :f].copy(); self.F = self.F[:f]; self.z = self.z[:f]; self.dead = dead.copy()
    def solve(self):
        R = np.nonzero(self.dead)[0]
        v = self.z
        if len(R):
            Y = self.W[:, R]
            lam = np.linalg.solve(Y.T @ Y, Y.T @ self.z)
            v = self.z - Y @ lam
        x = self.W.T @ v
        x[R] = 0.0
        return x
    def refactor(self):
        keep = ~self.dead
        self.F = self.F[keep]; f = len(self.F)
        L = np.linalg.cholesky(Hblock(self.n, self.F, self.F))
        self.W = sl.solve_triangular(L, np.eye(f), lower=True)
        self.z = self.W @ self.cF(self.F); self.dead = np.zeros(f, bool)
        self.flops += (2.0 / 3.0) * f ** 3; self.refactors += 1


def solve(n, seed=1, kfrac=0.1, kmin=8, nms=3, rfrac=0.2, verbose=True):
    D = O.synth(n, seed)
    order, _, _ = O.run(D, threads=2, want_events=False)
    p = order[1:] - 1
    d = np.triu(D[np.ix_(p, p)], 1)
    c = At_mul(d)
    iu = np.triu_indices(n, 1)
    cmax = c[iu].max(); tol = 1e-12 * cmax
    fac = Factor(n, c)
    x = np.zeros(0); phi = 0.0
    outer = inner = adds = dels = rejects = 0
    born = np.zeros(0, dtype=np.int64); k_scale = 1.0; lh_mode = False; lh_steps = 0
    lower = np.tril_indices(n)
    while True:
        outer += 1
        live = ~fac.dead
        X = np.zeros((n, n)); X[fac.F[live, 0], fac.F[live, 1]] = x[live] if len(x) else 0
        w = c - At_mul(A_mul(X))
        wm = w.copy(); wm[lower] = -np.inf
        wdead = wm[fac.F[fac.dead, 0], fac.F[fac.dead, 1]] if fac.dead.any() else np.zeros(0)
        wm[fac.F[:, 0], fac.F[:, 1]] = -np.inf
        if SMAX[0]:
            ii, jj = np.indices((n, n)); szm = np.minimum(jj - ii, n - (jj - ii)); wm = np.where(szm <= SMAX[0], wm, -np.inf)
        pos = wm > tol
        if not pos.any() and SMAX[0]:
            print(f'  phase with splits of size <= {SMAX[0]} done at outer {outer}: live {int((~fac.dead).sum())} adds {adds} dels {dels}'); SMAX[0] = SMAX[0] * 4 if SMAX[0] * 4 < n // 2 else 0; continue
        if not pos.any():
            if (wdead > tol).any():
                born = born[~fac.dead]; fac.refactor(); x = fac.solve(); continue
            break
        wz = np.where(np.isfinite(wm), wm, -1e300)
        mx = maximum_filter(wz, size=2 * nms + 1, mode='constant', cval=-1e300) if nms else wz
        cand = np.argwhere((wz >= mx) & pos)
        vals = wm[cand[:, 0], cand[:, 1]]
        nlive = int(live.sum())
        k = int(min(max(kmin, kfrac * k_scale * max(nlive, 1)), len(cand)))
        k = max(k, 1)
        if lh_mode: k = 1; lh_steps += 1
        top = cand[np.argsort(-vals)[:k]]
        f0, dead0 = len(fac.F), fac.dead.copy()
        while True:
            try:
                fac.append(top); break
            except np.linalg.LinAlgError:
                top = top[: max(1, len(top) // 2)]
                if verbose: print("  block not positive definite: halved to", len(top))
        adds += len(top); born = np.concatenate([born[:f0], np.full(len(fac.F) - f0, outer)])
        xcur = np.concatenate([x, np.zeros(len(top))])
        while True:
            inner += 1
            s = fac.solve()
            neg = np.nonzero((s <= 0) & ~fac.dead)[0]
            if len(neg) == 0:
                break
            if lh_mode:   # Lawson-Hanson: as far towards s as feasibility allows, what reaches zero leaves
                ratio = xcur[neg] / (xcur[neg] - s[neg])
                al = ratio.min()
                xcur = xcur + al * (s - xcur)
                out = neg[ratio <= al]
                xcur[out] = 0.0
                neg = out
            fac.dead[neg] = True; dels += len(neg); DELPOS.extend((neg / len(fac.F)).tolist()); DELAGE.extend((outer - born[neg]).tolist())
        livem = ~fac.dead
        phi_new = -0.5 * fac.cF(fac.F[livem]) @ s[livem]
        if not (phi_new < phi):
            rejects += 1
            fac.truncate(f0, dead0)
            k_scale *= 0.25
            if verbose: print(f"  reject at f={f0} k={k}: {phi_new} vs {phi}")
            if lh_mode: raise RuntimeError("no progress with a single split in Lawson-Hanson mode")
            if k <= 4: lh_mode = True
            if kfrac * k_scale * max(nlive, 1) < 1: kmin = 1
            continue
        phi = phi_new; x = s; k_scale = min(1.0, 2 * k_scale); lh_mode = False
        if fac.dead.sum() > rfrac * len(fac.F):
            born = born[~fac.dead]; fac.refactor(); x = fac.solve()
        if verbose and outer % 10 == 0:
            print(f" outer {outer} f={len(fac.F)} dead={int(fac.dead.sum())} cand={len(cand)} k={k} inner={inner} phi={phi:.8e}", flush=True)
    live = ~fac.dead
    st = dict(n=n, outer=outer, inner=inner, adds=adds, dels=dels, rejects=rejects, refactors=fac.refactors, lh_steps=lh_steps, F=int(live.sum()),
              flops_over_F3=fac.flops / max(int(live.sum()), 1) ** 3)
    return st, fac.F[live], x[live], c


if __name__ == "__main__":
    n = int(sys.argv[1]); kfrac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    nms = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rfrac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
    t = time.time()
    st, F, xF, c = solve(n, kfrac=kfrac, nms=nms, rfrac=rfrac, kmin=int(os.environ.get("KMIN", 8)), verbose=bool(os.environ.get("V")))
    print(st, f"{time.time()-t:.1f}s")
    X = np.zeros((n, n)); X[F[:, 0], F[:, 1]] = xF
    w = c - At_mul(A_mul(X))
    iu = np.triu_indices(n, 1)
    wz = w.copy(); wz[F[:, 0], F[:, 1]] = 0
    print("deletion position quantiles (p/f)", np.quantile(DELPOS, [0.1, 0.25, 0.5, 0.75, 0.9]).round(3), "age in outer iterations", np.quantile(DELAGE, [0.25, 0.5, 0.75, 0.9]))
    print("kkt: max w on Z", wz[iu].max() / c[iu].max(), "max |w| on F", np.abs(w[F[:, 0], F[:, 1]]).max() / c[iu].max(), "min x", xF.min())
