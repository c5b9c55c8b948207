"""Independent brute-force numpy re-implementations of the integer primitives (written from the definitions in
SURVEY.md Appendix C, not from the oracle's code) -- used to cross-check the oracle and, later, the HIP kernels."""
import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_score_map(img, min_th):
    """S(p) = max over the 16 contiguous 9-arcs (both polarities) of min |v - x| , minus 1; 0 unless that max > min_th.
    Defined on the interior [3, h-3) x [3, w-3); zero elsewhere."""
    h, w = img.shape
    I = img.astype(np.int32)
    S = np.zeros((h, w), np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING], 0)  # v - x
    best = np.zeros_like(c)
    for s in range(16):
        idx = [(s + k) % 16 for k in range(9)]
        arc = d[idx]
        best = np.maximum(best, arc.min(0))      # all darker than centre by at least that
        best = np.maximum(best, (-arc).min(0))   # all brighter
    S[3:h - 3, 3:w - 3] = np.where(best > min_th, best - 1, 0)
    return S


def fast_nms(img, th):
    """cv::FAST(img, th, nms=True) by definition: corners at th, strict 8-neighbour maximum of the score among
    corners at th (non-corners count as 0); row-major order."""
    S = fast_score_map(img, th)
    h, w = S.shape
    P = np.pad(S, 1)
    keep = S > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            keep &= S > P[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, S[ys, xs]], 1).astype(np.int32)


def cell_grid(w, h):
    """ORBExtractor.cpp:745-773 geometry for one level; returns list of (i, j, x0, x1, y0, y1, wCell, hCell)."""
    minB = 16; maxBX = w - 16; maxBY = h - 16
    width = np.float32(maxBX - minB); height = np.float32(maxBY - minB)
    nCols = int(width / np.float32(30)); nRows = int(height / np.float32(30))
    wCell = int(np.ceil(width / nCols)); hCell = int(np.ceil(height / nRows))
    cells = []
    for i in range(nRows):
        iniY = minB + i * hCell; maxY = iniY + hCell + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nCols):
            iniX = minB + j * wCell; maxX = iniX + wCell + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            cells.append((i, j, iniX, maxX, iniY, maxY, wCell, hCell))
    return cells


def level_candidates_scoremap(img, ini_th, min_th):
    """The single-pass formulation the HIP kernel uses: one global score map at min_th, per-cell strict NMS over the
    cell's evaluated rectangle, ini/min threshold fallback per cell. Output order = (cellRow, cellCol, y, x)."""
    h, w = img.shape
    S = fast_score_map(img, min_th)
    out = []
    for (i, j, x0, x1, y0, y1, wCell, hCell) in cell_grid(w, h):
        ex0, ex1, ey0, ey1 = x0 + 3, x1 - 3, y0 + 3, y1 - 3
        if ex1 <= ex0 or ey1 <= ey0:
            continue
        sub = S[ey0:ey1, ex0:ex1]
        P = np.pad(sub, 1)
        keep = sub > 0
        hh, ww = sub.shape
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= sub > P[1 + dy:1 + dy + hh, 1 + dx:1 + dx + ww]
        sel = keep & (sub >= ini_th)
        if not sel.any():
            sel = keep
        ys, xs = np.nonzero(sel)
        for y, x in zip(ys, xs):
            out.append((x + ex0 - 16, y + ey0 - 16, sub[y, x]))
    return np.array(out, np.int32).reshape(-1, 3)


def resize_linear(src, dw, dh):
    sh, sw = src.shape
    sx_scale = 1.0 / (dw / sw); sy_scale = 1.0 / (dh / sh)
    def coeffs(n, sn, scale, clamp_frac):
        f = ((np.arange(n) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_frac:
            lo = s < 0; f[lo] = 0; s[lo] = 0
            hi = s >= sn - 1; f[hi] = 0; s[hi] = sn - 1
        a0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
        a1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
        return s, a0, a1
    xs, ax0, ax1 = coeffs(dw, sw, sx_scale, True)
    ys, by0, by1 = coeffs(dh, sh, sy_scale, False)
    S = src.astype(np.int64)
    xs1 = np.minimum(xs + 1, sw - 1)
    rows = S[:, xs] * ax0[None, :] + S[:, xs1] * ax1[None, :]
    y0 = np.clip(ys, 0, sh - 1); y1 = np.clip(ys + 1, 0, sh - 1)
    out = (((by0[:, None] * (rows[y0] >> 4)) >> 16) + ((by1[:, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def blur7(src):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    h, w = src.shape
    P = np.pad(src.astype(np.int64), 3, mode="reflect")
    rows = sum(k[t] * P[:, t:t + w] for t in range(7))
    out = sum(k[t] * rows[t:t + h, :] for t in range(7))
    return np.clip((out + 32768) >> 16, 0, 255).astype(np.uint8)


def remap_bilinear(src, m1, m2):
    sh, sw = src.shape
    sx = np.rint((m1.astype(np.float32) * np.float32(32)).astype(np.float64)).astype(np.int64)
    sy = np.rint((m2.astype(np.float32) * np.float32(32)).astype(np.float64)).astype(np.int64)
    X = sx >> 5; Y = sy >> 5; ax = sx & 31; ay = sy & 31
    P = np.zeros((sh + 2, sw + 2), np.int64)
    P[1:-1, 1:-1] = src
    def at(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok, P[np.clip(yy, -1, sh) + 1, np.clip(xx, -1, sw) + 1], 0)
    v = (at(Y, X) * (32 - ax) * (32 - ay) + at(Y, X + 1) * ax * (32 - ay) + at(Y + 1, X) * (32 - ax) * ay + at(Y + 1, X + 1) * ax * ay) * 32
    return np.clip((v + 16384) >> 15, 0, 255).astype(np.uint8)


def fast_atan2(y, x):
    f = np.float32
    p1 = f(0.9997878412794807) * f(180 / np.pi); p3 = f(-0.3258083974640975) * f(180 / np.pi)
    p5 = f(0.1555786518463281) * f(180 / np.pi); p7 = f(-0.04432655554792128) * f(180 / np.pi)
    y = f(y); x = f(x)
    ax, ay = abs(x), abs(y)
    eps = f(2.220446049250313e-16)
    if ax >= ay:
        c = ay / (ax + eps); c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps); c2 = c * c
        a = f(90) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180) - a
    if y < 0:
        a = f(360) - a
    return f(a)


def popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def octree_arrayform(xys, W, H, N):
    """Array-form restatement of DistributeOctTree (the shape the HIP kernel uses): no linked list, no per-node key
    vectors -- nodes are rectangles + counts, every candidate carries its node id, the list order is rebuilt per pass.
    xys rows are (x, y, response) in candidate order (order = tie-break priority of the final arg-max)."""
    n = len(xys)
    if n == 0:
        return np.zeros((0, 3), np.int32)
    X = xys[:, 0].astype(np.int64); Y = xys[:, 1].astype(np.int64); R = xys[:, 2].astype(np.int64)
    rect = [(0, W, 0, H)]          # x0, x1, y0, y1 per node id
    cnt = [n]
    node_of = np.zeros(n, np.int64)
    order = [0]

    def split(p):
        x0, x1, y0, y1 = rect[p]
        hx = int(np.ceil(np.float32(x1 - x0) / 2)); hy = int(np.ceil(np.float32(y1 - y0) / 2))
        mx, my = x0 + hx, y0 + hy
        sel = np.nonzero(node_of == p)[0]
        q = (X[sel] >= mx).astype(np.int64) + 2 * (Y[sel] >= my).astype(np.int64)   # 0:n1 1:n2 2:n3 3:n4
        rects = [(x0, mx, y0, my), (mx, x1, y0, my), (x0, mx, my, y1), (mx, x1, my, y1)]
        kids = []
        for c in range(4):
            m = int((q == c).sum())
            if m == 0:
                continue
            rect.append(rects[c]); cnt.append(m)
            node_of[sel[q == c]] = len(rect) - 1
            kids.append(len(rect) - 1)
        return kids

    finish = False
    while not finish:
        prev = len(order)
        C = []
        keep = []
        for p in order:
            if cnt[p] == 1:
                keep.append(p)
            else:
                C += split(p)
        any_split = len(C) > 0
        order = C[::-1] + keep
        Ex = [c for c in C if cnt[c] > 1]
        size = len(order)
        if size >= N or (size == prev and size >= N // 100):
            finish = True
        elif size + 3 * len(Ex) > N:
            while not finish:
                prev2 = size
                srt = sorted(Ex, key=lambda c: (cnt[c], c))   # node id == creation sequence
                Ex = []
                C2 = []
                removed = set()
                for p in srt[::-1]:
                    kids = split(p)
                    C2 += kids
                    Ex += [c for c in kids if cnt[c] > 1]
                    removed.add(p)
                    size += len(kids) - 1
                    if size >= N:
                        break
                order = C2[::-1] + [o for o in order if o not in removed]
                assert len(order) == size
                if size >= N or size == prev2:
                    finish = True
        elif not any_split:
            finish = True
    out = []
    for p in order:
        sel = np.nonzero(node_of == p)[0]
        b = sel[np.argmax(R[sel])]     # first maximum in candidate order
        out.append((X[b], Y[b], R[b]))
    return np.array(out, np.int32).reshape(-1, 3)
