"""Pure-Python model of the crack-cycle contour extraction that csrc/gof_blobs.hip runs on
the GPU (test infrastructure): the same successor rule, sums, sign / nesting tests and
point order, written as plain loops.  tests/test_blob_model.py checks it against the host
border follower on CPU; the GPU tests check the kernels against the same follower."""
import numpy as np

from pygofpid_amd import hostcv

# sides in ccw walking order: W (walk S), S (walk E), E (walk N), N (walk W)
# side codes chosen so that N is the smallest: N=0, W=1, S=2, E=3
N_, W_, S_, E_ = 0, 1, 2, 3
WALK = {W_: (0, 1), S_: (1, 0), E_: (0, -1), N_: (-1, 0)}      # (dx, dy)
RIGHT = {W_: (-1, 0), S_: (0, 1), E_: (1, 0), N_: (0, -1)}     # towards bg
NEXT = {W_: S_, S_: E_, E_: N_, N_: W_}                        # turn left (same pixel)
PREV = {v: k for k, v in NEXT.items()}                         # turn right (diagonal pixel)

def fg(m, x, y):
    h, w = m.shape
    return 0 <= x < w and 0 <= y < h and m[y, x] != 0

def cracks_of(m, x, y):
    return [s for s in (N_, W_, S_, E_) if not fg(m, x + RIGHT[s][0], y + RIGHT[s][1])]

def succ(m, x, y, s):
    dx, dy = WALK[s]; rx, ry = RIGHT[s]
    bx, by = x + dx + rx, y + dy + ry
    cx, cy = x + dx, y + dy
    if fg(m, bx, by):
        return bx, by, PREV[s]
    if fg(m, cx, cy):
        return cx, cy, s
    return x, y, NEXT[s]

def extract(m, min_a2=200):
    h, w = m.shape
    key = lambda x, y, s: (y * w + x) * 4 + s
    allc = {}
    for y in range(h):
        for x in range(w):
            if m[y, x]:
                for s in cracks_of(m, x, y):
                    allc[key(x, y, s)] = (x, y, s)
    seen = set(); cycles = []
    for k in sorted(allc):
        if k in seen: continue
        cyc = []; cur = allc[k]
        while key(*cur) not in seen:
            seen.add(key(*cur)); cyc.append(cur); cur = succ(m, *cur)
        cycles.append(cyc)     # starts at its min key
    recs = []
    root_of = {}
    for ci, cyc in enumerate(cycles):
        a00 = a10 = a01 = 0; n = len(cyc); npts = 0
        for i in range(n):
            x0, y0, _ = cyc[i]; x1, y1, _ = cyc[(i + 1) % n]
            root_of[key(*cyc[i])] = ci
            if (x0, y0) != (x1, y1):
                cr = x0 * y1 - x1 * y0
                a00 += cr; a10 += cr * (x0 + x1); a01 += cr * (y0 + y1); npts += 1
        xs = [c[0] for c in cyc]; ys = [c[1] for c in cyc]
        recs.append(dict(a00=a00, a10=a10, a01=a01, npts=max(1, npts), bbox=(min(xs), min(ys), max(xs), max(ys)), cyc=cyc))
    out = []
    for ci, r in enumerate(recs):
        if r['a00'] > 0 or -r['a00'] < min_a2:    # hole, or too small
            continue
        x0, y0, s0 = r['cyc'][0]
        assert s0 == N_
        depth = 0
        for x in range(x0):
            if m[y0, x]:
                for s, d in ((W_, 1), (E_, -1)):
                    k = key(x, y0, s)
                    if k in allc and recs[root_of[k]]['a00'] <= 0:
                        depth += d
        if depth != 0:
            continue
        pts = []
        cyc = r['cyc']
        for i, (x, y, s) in enumerate(cyc):
            if i == 0 or (x, y) != (cyc[i - 1][0], cyc[i - 1][1]):
                pts.append((x, y))
        if len(pts) > 1 and pts[-1] == pts[0]:
            pts.pop()
        assert len(pts) == r['npts'], (len(pts), r['npts'])
        out.append(dict(pts=np.array(pts, np.int32).reshape(-1, 1, 2), **{k: r[k] for k in ('a00', 'a10', 'a01', 'bbox')}))
    return out

def check(m, min_a2=200):
    ref = [c for c in hostcv.findContours(m) if hostcv.contourArea(c) * 2 >= min_a2]
    got = hostcv.in_contour_order(extract(m, min_a2))   # (the model lists borders in raster order)
    assert len(ref) == len(got), (len(ref), len(got))
    for c, g in zip(ref, got):
        assert c.shape == g['pts'].shape and np.array_equal(c, g['pts']), (c.reshape(-1, 2)[:8], g['pts'].reshape(-1, 2)[:8])
        mo = hostcv.moments(c)
        s = -1.0 if g['a00'] < 0 else 1.0
        assert mo['m00'] == s * g['a00'] / 2.0 and mo['m10'] == s * g['a10'] / 6.0 and mo['m01'] == s * g['a01'] / 6.0
        x, y, ww, hh = hostcv.boundingRect(c)
        assert (x, y, x + ww - 1, y + hh - 1) == g['bbox']
    return len(ref)

