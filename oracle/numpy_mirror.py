"""Independent NumPy/SciPy restatement, used ONLY to cross-check gof_oracle.c.

TEST INFRASTRUCTURE ONLY (see gof_oracle.c).  PARITY UNPINNED.
Written separately from the C so that a coding slip in either shows up as a
disagreement in tests/test_oracle.py; both follow SURVEY.md Appendix A.
"""
import numpy as np
from scipy import ndimage

F = np.float32


def gaussian_blur(x, kx, ky):
    """Fixed-point separable blur, mirror (= REFLECT_101) border, int64 exact."""
    a = x.astype(np.int64)
    a = ndimage.correlate1d(a, np.asarray(kx, np.int64), axis=1, mode="mirror")
    a = ndimage.correlate1d(a, np.asarray(ky, np.int64), axis=0, mode="mirror")
    return ((a + 32768) >> 16).astype(np.uint8)


def erode(mask, kernel):
    k = np.asarray(kernel) != 0
    return ndimage.grey_erosion(mask, footprint=k, mode="constant", cval=255,
                                origin=_origin(k))


def dilate(mask, kernel):
    k = np.asarray(kernel) != 0
    # scipy's grey_dilation mirrors the footprint; mirror it back so that the
    # window is {in(y+dy-ay, x+dx-ax): K[dy,dx] != 0} as in cv.dilate.
    return ndimage.grey_dilation(mask, footprint=k[::-1, ::-1], mode="constant", cval=0,
                                 origin=_origin_dil(k))


def _origin(k):
    # scipy centres a footprint at shape//2 already == OpenCV's default anchor
    return 0


def _origin_dil(k):
    # after mirroring an even-sized footprint the centre moves by one
    return tuple(-1 if s % 2 == 0 else 0 for s in k.shape)


def fd_apply(x, prev, thr):
    """[REF segmentation.py:55-68] with numpy doing exactly what the reference does."""
    if prev is None:
        f = np.zeros_like(x)
    else:
        d = np.abs(x.astype(np.int16) - prev.astype(np.int16))
        f = np.where(d > int(np.floor(thr)), 255, 0).astype(np.uint8)
    if f.ndim > 2:
        f = np.mean(f, axis=-1, dtype=np.uint8, keepdims=False)
    return f


class MOG2:
    """Vectorised over pixels; every float op is a separate float32 numpy op."""

    def __init__(self, history=500, K=5, Tb=16.0, TB=0.9, Tg=9.0, var_init=15.0,
                 var_min=4.0, var_max=75.0, ct=0.05, tau=0.5, shadows=True, shadow_value=127):
        self.history, self.K = history, K
        self.Tb, self.TB, self.Tg = F(Tb), F(TB), F(Tg)
        self.var_init, self.var_min, self.var_max = F(var_init), F(var_min), F(var_max)
        self.ct, self.tau = F(ct), F(tau)
        self.shadows, self.shadow_value = shadows, shadow_value
        self.nframes = 0
        self.w = None

    def _init(self, n, c):
        K = self.K
        self.w = np.zeros((K, n), F)
        self.var = np.zeros((K, n), F)
        self.mu = np.zeros((K, c, n), F)
        self.nm = np.zeros(n, np.int32)

    def _swap(self, i, sel):
        for a in (self.w, self.var):
            t = a[i, sel].copy()
            a[i, sel] = a[i - 1, sel]
            a[i - 1, sel] = t
        t = self.mu[i][:, sel].copy()
        self.mu[i][:, sel] = self.mu[i - 1][:, sel]
        self.mu[i - 1][:, sel] = t

    def apply(self, x):
        h, w_ = x.shape[:2]
        c = 1 if x.ndim == 2 else x.shape[2]
        n = h * w_
        if self.w is None:
            self._init(n, c)
        data = x.reshape(n, c).T.astype(F)  # (c, n)
        self.nframes += 1
        lr = 1.0 / min(2 * self.nframes, self.history)
        alpha = F(lr)
        alpha1 = F(1.0) - alpha
        prune = F(-lr * float(self.ct))
        K = self.K
        w, var, mu = self.w, self.var, self.mu
        nm = self.nm.copy()
        bg = np.zeros(n, bool)
        fits = np.zeros(n, bool)
        tot = np.zeros(n, F)
        for m in range(K):
            act = m < nm  # live bound
            if not act.any():
                break
            wt = alpha1 * w[m] + prune
            test = act & ~fits
            d = mu[m] - data
            if c == 3:
                dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
            else:
                dist2 = np.zeros(n, F)
                for ch in range(c):
                    dist2 = dist2 + d[ch] * d[ch]
            bg |= test & (tot < self.TB) & (dist2 < self.Tb * var[m])
            fit = test & (dist2 < self.Tg * var[m])
            fits |= fit
            wfit = wt + alpha
            with np.errstate(divide="ignore", invalid="ignore"):
                k = alpha / wfit
            wt = np.where(fit, wfit, wt)
            for ch in range(c):
                mu[m, ch] = np.where(fit, mu[m, ch] - k * d[ch], mu[m, ch])
            vn = var[m] + k * (dist2 - var[m])
            vn = np.where(vn > self.var_min, vn, self.var_min)
            vn = np.where(vn < self.var_max, vn, self.var_max)
            var[m] = np.where(fit, vn, var[m])
            # prune (a fitted mode can never be pruned: wt >= 0.95 alpha > 0.05 alpha)
            pr = act & (wt < -prune)
            wt = np.where(pr, F(0), wt)
            nm = np.where(pr, nm - 1, nm)
            # store, then bubble the fitted mode up (the stored weight travels
            # with it: same end state as OpenCV's "swap, then write at m-swaps")
            w[m] = np.where(act, wt, w[m])
            tot = np.where(act, tot + wt, tot)
            go = fit.copy()
            for i in range(m, 0, -1):
                go &= ~(w[i] < w[i - 1])
                sel = np.nonzero(go)[0]
                if sel.size == 0:
                    break
                self._swap(i, sel)
        with np.errstate(divide="ignore"):
            inv = np.where(np.abs(tot) > np.finfo(F).eps, F(1) / tot, F(0)).astype(F)
        for m in range(K):
            w[m] = np.where(m < nm, w[m] * inv, w[m])
        add = ~fits & (alpha > 0)
        slot = np.where(nm == K, K - 1, nm)
        nm = np.where(add & (nm < K), nm + 1, nm)
        first = add & (nm == 1)
        for m in range(K):
            here = add & (slot == m)
            w[m] = np.where(here, np.where(first, F(1), alpha), w[m])
            w[m] = np.where(add & ~first & (m < nm - 1), w[m] * alpha1, w[m])
            var[m] = np.where(here, self.var_init, var[m])
            for ch in range(c):
                mu[m, ch] = np.where(here, data[ch], mu[m, ch])
        go = add.copy()
        for i in range(K - 1, 0, -1):
            can = go & (i <= nm - 1)
            stop = can & (alpha < w[i - 1])
            go &= ~stop
            sel = np.nonzero(can & ~stop)[0]
            if sel.size:
                self._swap(i, sel)
        self.nm = nm
        mask = np.where(bg, 0, 255).astype(np.uint8)
        if self.shadows:
            sh = self._shadow(data, nm, c)
            mask = np.where(~bg & sh, self.shadow_value, mask).astype(np.uint8)
        return mask.reshape(h, w_)

    def _shadow(self, data, nm, c):
        n = data.shape[1]
        res = np.zeros(n, bool)
        done = np.zeros(n, bool)
        tw = np.zeros(n, F)
        for m in range(self.K):
            live = ~done & (m < nm)
            num = np.zeros(n, F)
            den = np.zeros(n, F)
            for ch in range(c):
                num = num + data[ch] * self.mu[m, ch]
                den = den + self.mu[m, ch] * self.mu[m, ch]
            z = live & (den == 0)
            done |= z
            live &= ~z
            cand = live & (num <= den) & (num >= self.tau * den)
            with np.errstate(divide="ignore", invalid="ignore"):
                a = num / den
            d2 = np.zeros(n, F)
            for ch in range(c):
                dd = a * self.mu[m, ch] - data[ch]
                d2 = d2 + dd * dd
            hit = cand & (d2 < self.Tb * self.var[m] * a * a)
            res |= hit
            done |= hit
            live &= ~hit
            tw = np.where(live, tw + self.w[m], tw)
            done |= live & (tw > self.TB)
        return res
