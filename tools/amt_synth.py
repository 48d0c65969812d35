"""Deterministic synthetic clips and logos.

Test / bench utility, not product code.  Everything is integer-hash based (no library RNG) so the
same frames come out of numpy on the CPU and of torch on the GPU.

  * ``make_logo``      -- text-like alpha shape -> LogoData planes A = 1/(1-alpha), B = -alpha*c/(1-alpha)
                          (observed = (bg - B*maxv)/A  <=>  bg = A*obs + B*maxv, LogoScan.hpp:320-333,247)
  * ``make_clip_np``   -- numpy YUV420 planar interlaced (TFF) frames with the logo blended in
  * ``make_clip_torch``-- the same arithmetic with torch ops on any device
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def _mix_np(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def logo_alpha(w: int, h: int, seed: int = 0x10600001) -> np.ndarray:
    """Anti-aliased text-like alpha in [0, 0.6]; zero on the outer 8-px ring."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.zeros((h, w), np.float64)
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    nglyph = max(3, w // 40)
    gx0 = 12
    gw = (w - 24) / nglyph
    for g in range(nglyph):
        cx = gx0 + gw * (g + 0.5)
        cy = h * 0.5 + rng.uniform(-0.06, 0.06) * h
        kind = g % 4
        rx, ry = gw * 0.36, h * 0.30
        if kind == 0:      # ring
            d = np.abs(np.hypot((xx - cx) / rx, (yy - cy) / ry) - 1.0) * min(rx, ry)
            a = np.maximum(a, np.clip(2.2 - d, 0, 1))
        elif kind == 1:    # vertical bar + horizontal bar (T)
            d1 = np.maximum(np.abs(xx - cx) - 2.0, np.abs(yy - cy) - ry)
            d2 = np.maximum(np.abs(xx - cx) - rx, np.abs(yy - (cy - ry)) - 2.0)
            a = np.maximum(a, np.clip(1.0 - np.minimum(d1, d2), 0, 1))
        elif kind == 2:    # diagonal strokes (X)
            d1 = np.abs((xx - cx) / rx - (yy - cy) / ry) * min(rx, ry) * 0.7
            d2 = np.abs((xx - cx) / rx + (yy - cy) / ry) * min(rx, ry) * 0.7
            inside = (np.abs(xx - cx) <= rx) & (np.abs(yy - cy) <= ry)
            a = np.maximum(a, np.where(inside, np.clip(2.0 - np.minimum(d1, d2), 0, 1), 0))
        else:              # filled blob with soft edge
            d = np.hypot((xx - cx) / (rx * 0.8), (yy - cy) / (ry * 0.8))
            a = np.maximum(a, np.clip((1.0 - d) * 4.0, 0, 1))
    ring = 8
    a[:ring, :] = 0
    a[-ring:, :] = 0
    a[:, :ring] = 0
    a[:, -ring:] = 0
    return (a * 0.6).astype(np.float64)


def make_logo(w: int = 256, h: int = 128, seed: int = 0x10600001, strength: float = 1.0):
    """Returns (data, alphaY, alphaUV): data = aY,bY,aU,bU,aV,bV fp32 (AMTLogo.hpp:204-212)."""
    alpha = logo_alpha(w, h, seed) * strength
    cY, cC = 235.0 / 255.0, 128.0 / 255.0
    aY = (1.0 / (1.0 - alpha)).astype(np.float32)
    bY = (-alpha * cY / (1.0 - alpha)).astype(np.float32)
    alphaUV = alpha.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    aC = (1.0 / (1.0 - alphaUV)).astype(np.float32)
    bC = (-alphaUV * cC / (1.0 - alphaUV)).astype(np.float32)
    data = np.concatenate([aY.ravel(), bY.ravel(), aC.ravel(), bC.ravel(), aC.ravel(), bC.ravel()]).astype(np.float32)
    return data, alpha, alphaUV


def logo_presence(n: np.ndarray, period: int = 900, fade: int = 12) -> np.ndarray:
    """Logo opacity per frame: on for (n//period)%2==0, linear `fade`-frame ramps at transitions."""
    n = np.asarray(n, np.int64)
    seg = n // period
    pos = n % period
    on = (seg % 2 == 0).astype(np.float64)
    # ramp in at the start of an "on" segment (except the very first), ramp out at its end
    ramp_in = np.clip((pos + 1) / float(fade), 0, 1)
    ramp_out = np.clip((period - pos) / float(fade), 0, 1)
    vis = np.where(seg % 2 == 0, np.minimum(np.where(seg > 0, ramp_in, 1.0), ramp_out), 0.0)
    return vis * on


def frame_planes_np(n: int, W: int, H: int, seed: int, bits: int = 8, cadence: str = "30i"):
    """One synthetic interlaced frame (Y,U,V uint16 arrays in `bits` range) without logo."""
    maxv = (1 << bits) - 1
    scale = maxv / 255.0
    scene = n // 97
    sseed = (seed * 0x9E3779B97F4A7C15 + scene * 0xD1B54A32D192ED03) & MASK64

    def field_time(parity):
        if cadence == "24p":        # 3:2 pulldown AABBBCCDDD over 5 frames (10 fields)
            fidx = 2 * n + parity
            grp, r = divmod(fidx, 10)
            return grp * 4 + (0 if r < 2 else 1 if r < 5 else 2 if r < 7 else 3)
        if cadence == "30p":
            return 2 * n
        return 2 * n + parity       # 30i: every field its own time

    def picture(t):
        # smooth panning texture (2 px per field time) + moving box + a little temporal noise; a new scene
        # (every 97 frames) re-draws the texture and shifts the level
        yy, xx = np.mgrid[0:H, 0:W]
        xs = (xx + 2 * t).astype(np.float64)
        f1 = 37.0 + float((sseed >> 8) & 0x1F)
        f2 = 23.0 + float((sseed >> 16) & 0xF)
        p1 = float((sseed >> 24) & 0xFF) / 40.0
        p2 = float((sseed >> 32) & 0xFF) / 40.0
        level = 95.0 + float((sseed >> 40) & 0x3F)
        base = level + 40.0 * np.sin((xs + 0.6 * yy) / f1 * 2.0 + p1) + 22.0 * np.sin((xs * 0.7 - yy) / f2 * 2.0 + p2)
        bx = (int(sseed & 0x3FF) + 7 * t) % max(1, W - 160)
        by = (int((sseed >> 10) & 0x1FF) + 3 * t) % max(1, H - 120)
        box = ((xx >= bx) & (xx < bx + 160) & (yy >= by) & (yy < by + 120))
        base = np.where(box, 200.0 - 0.25 * base, base)
        key = np.uint64((sseed ^ (t * 0x9E3779B97F4A7C15)) & MASK64)
        idx = (yy.astype(np.uint64) * np.uint64(W) + xx.astype(np.uint64)) + key
        hsh = _mix_np(idx)
        noise = ((hsh & np.uint64(0x7)).astype(np.float64) + ((hsh >> np.uint64(4)) & np.uint64(0x7)).astype(np.float64) - 7.0) * 0.5
        return base + noise, hsh

    pt, ht = picture(field_time(0))
    pb, hb = picture(field_time(1))
    Yf = np.empty((H, W), np.float64)
    Yf[0::2] = pt[0::2]
    Yf[1::2] = pb[1::2]
    hs = np.where((np.arange(H) % 2 == 0)[:, None], ht, hb)
    cyy, cxx = np.mgrid[0:H // 2, 0:W // 2]
    Uf = 128.0 + 20.0 * np.sin((cxx + scene * 13) / 37.0) + (((hs[0::2, 0::2] >> np.uint64(12)) & np.uint64(7)).astype(np.float64) - 3.5)
    Vf = 128.0 + 20.0 * np.cos((cyy + scene * 7) / 29.0) + (((hs[0::2, 0::2] >> np.uint64(16)) & np.uint64(7)).astype(np.float64) - 3.5)
    Y = np.clip(np.rint(Yf * scale), 0, maxv).astype(np.uint16)
    U = np.clip(np.rint(Uf * scale), 0, maxv).astype(np.uint16)
    V = np.clip(np.rint(Vf * scale), 0, maxv).astype(np.uint16)
    return Y, U, V


def blend_logo_np(Y, U, V, alpha, alphaUV, imgx, imgy, vis: float, bits: int = 8):
    """obs = (1 - vis*alpha) * bg + vis*alpha * c * maxv on the logo rectangle (in place)."""
    if vis <= 0:
        return
    maxv = (1 << bits) - 1
    h, w = alpha.shape
    a = alpha * vis
    r = Y[imgy:imgy + h, imgx:imgx + w].astype(np.float64)
    Y[imgy:imgy + h, imgx:imgx + w] = np.clip(np.rint((1 - a) * r + a * (235.0 / 255.0) * maxv), 0, maxv).astype(Y.dtype)
    au = alphaUV * vis
    for P in (U, V):
        r = P[imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2].astype(np.float64)
        P[imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2] = np.clip(
            np.rint((1 - au) * r + au * (128.0 / 255.0) * maxv), 0, maxv).astype(P.dtype)


def make_clip_np(N: int, W: int, H: int, seed: int, alpha=None, alphaUV=None, imgx: int = 0, imgy: int = 0,
                 bits: int = 8, cadence: str = "30i", period: int = 900, fade: int = 12,
                 flat_every: int = 8, pitchY: int | None = None, pitchUV: int | None = None, start: int = 0):
    """Returns dict(Y,U,V) arrays of shape (N, H, pitch) / (N, H/2, pitchUV), dtype u8 or u16.

    Frames whose hash says so (1 in `flat_every`) get a flat 2-px ring just inside the logo rectangle
    border so LogoScan::AddFrame accepts them (LogoScan.hpp:616-649)."""
    dt = np.uint8 if bits <= 8 else np.uint16
    pitchY = pitchY or W
    pitchUV = pitchUV or W // 2
    Ys = np.zeros((N, H, pitchY), dt)
    Us = np.zeros((N, H // 2, pitchUV), dt)
    Vs = np.zeros((N, H // 2, pitchUV), dt)
    maxv = (1 << bits) - 1
    for i in range(N):
        n = start + i
        Y, U, V = frame_planes_np(n, W, H, seed, bits, cadence)
        if alpha is not None:
            h, w = alpha.shape
            hv = int(_mix_np(np.array([seed ^ (n * 0x632BE59BD9B4E019 & MASK64)], np.uint64))[0])
            if flat_every and (hv % flat_every) == 0:
                lvl = 16 + (hv >> 8) % 180
                lv = int(round(lvl * maxv / 255.0))
                cu = int(round((112 + (hv >> 20) % 32) * maxv / 255.0))
                cv = int(round((112 + (hv >> 28) % 32) * maxv / 255.0))
                # flat background over the whole rectangle +- small noise inside (keeps the border flat)
                rect = Y[imgy:imgy + h, imgx:imgx + w]
                inner = (rect.astype(np.int64) % 5) - 2
                rect[:] = np.clip(lv + inner, 0, maxv).astype(rect.dtype)
                rect[:2, :] = lv; rect[-2:, :] = lv; rect[:, :2] = lv; rect[:, -2:] = lv
                U[imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2] = cu
                V[imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2] = cv
            vis = float(logo_presence(np.array([n]), period, fade)[0])
            blend_logo_np(Y, U, V, alpha, alphaUV, imgx, imgy, vis, bits)
        Ys[i, :, :W] = Y.astype(dt)
        Us[i, :, :W // 2] = U.astype(dt)
        Vs[i, :, :W // 2] = V.astype(dt)
    return {"Y": Ys, "U": Us, "V": Vs}


# ------------------------------------------------------------------------------------------------
# torch generator for device-resident bench inputs (cheaper picture model, same layout; values are
# NOT identical to make_clip_np -- parity tests never mix the two)
# ------------------------------------------------------------------------------------------------
def make_clip_torch(N: int, W: int, H: int, seed: int, alpha, alphaUV, imgx: int, imgy: int, device,
                    bits: int = 8, period: int = 900, fade: int = 12, pitchY: int | None = None,
                    pitchUV: int | None = None, start: int = 0, chunk: int = 64, cadence: str = "30i", chroma: bool = True,
                    rows: tuple | None = None, flat_every: int = 0, noise: str = "white"):
    """chroma=False: Y plane only (the all-frames LogoFrame scan reads nothing else); U/V come back as None.
    rows=(y0, y1) (both even): only those luma rows (and chroma rows y0/2 .. y1/2) are generated and stored -- the same samples the
    full frames hold there; for passes that read nothing but the logo rectangle's rows.
    flat_every=k: one frame in k (by a hash of its index) gets a flat background over the logo rectangle with a flat 2-px ring just
    inside its border, so that LogoScan::AddFrame accepts it (LogoScan.hpp:616-649) -- as make_clip_np does.
    noise="white": every sample of every field time gets its own +-22 noise value (the logo passes' default: nothing in them looks at
    neighbouring rows of the whole frame); noise="soft": band-limited grain -- 4x4 blocks, +-7, plus +-1 per sample -- as SURVEY.md 8d
    specifies for the clips the field-difference / combing detectors are run on (white noise of that size buries the combing signal)."""
    import torch
    dt = torch.uint8 if bits <= 8 else torch.int16
    maxv = (1 << bits) - 1
    pitchY = pitchY or W
    pitchUV = pitchUV or W // 2
    ry0, ry1 = rows if rows is not None else (0, H)
    assert ry0 % 2 == 0 and ry1 % 2 == 0 and 0 <= ry0 < ry1 <= H
    Ys = torch.zeros((N, ry1 - ry0, pitchY), dtype=dt, device=device)
    Us = torch.zeros((N, (ry1 - ry0) // 2, pitchUV), dtype=dt, device=device) if chroma else None
    Vs = torch.zeros((N, (ry1 - ry0) // 2, pitchUV), dtype=dt, device=device) if chroma else None
    yy = torch.arange(ry0, ry1, device=device, dtype=torch.int64)[:, None]
    xx = torch.arange(W, device=device, dtype=torch.int64)[None, :]
    imgy = imgy - ry0                       # the rectangle's first row within the generated rows
    al = torch.as_tensor(alpha, dtype=torch.float32, device=device) if alpha is not None else None
    alc = torch.as_tensor(alphaUV, dtype=torch.float32, device=device) if alphaUV is not None else None
    par = (yy % 2)
    for c0 in range(0, N, chunk):
        c1 = min(N, c0 + chunk)
        n = torch.arange(start + c0, start + c1, device=device, dtype=torch.int64)[:, None, None]
        scene = n // 97
        if cadence == "24p":
            fidx = 2 * n + par
            grp, r = fidx // 10, fidx % 10
            t = grp * 4 + (r >= 2).long() + (r >= 5).long() + (r >= 7).long()
        elif cadence == "30p":
            t = 2 * n + 0 * par
        else:
            t = 2 * n + par
        sv = (scene * 2654435761 + seed) & 0x7FFFFFFF
        if noise == "soft":     # the detectors' clip: a scene change replaces the picture (slope and phase of the ramp), not just shifts it
            base = 40 + (120 * ((xx * (1 + sv % 5) + yy * (1 + (sv >> 3) % 4) + (sv & 0xFFF)) % (W + H))) // (W + H)
        else:
            base = 40 + (120 * ((xx * 3 + yy * 2 + (sv & 0xFF)) % (W + H))) // (W + H)
        bx = ((sv & 0x3FF) + 7 * t) % max(1, W - 160)
        by = (((sv >> 10) & 0x1FF) + 3 * t) % max(1, H - 120)
        box = (xx >= bx) & (xx < bx + 160) & (yy >= by) & (yy < by + 120)
        base = torch.where(box, 200 - base // 4, base)
        hsh = (yy * W + xx + t * 40503 + sv * 69069)
        hsh = (hsh ^ (hsh >> 13)) * 1274126177
        hsh = (hsh ^ (hsh >> 16)) & 0xFFFFFFFF
        if noise == "soft":
            hb = ((yy >> 2) * ((W >> 2) + 1) + (xx >> 2) + t * 40503 + sv * 69069)
            hb = (hb ^ (hb >> 13)) * 1274126177
            hb = (hb ^ (hb >> 16)) & 0xFFFFFFFF
            nz = (hb & 7) + ((hb >> 4) & 7) - 7 + (hsh & 3) - 1
        else:
            nz = (hsh & 0xF) + ((hsh >> 4) & 0xF) + ((hsh >> 8) & 0xF) - 22
        Y = (base + nz).clamp(0, 255).to(torch.float32) * (maxv / 255.0)
        if chroma:
            hc = hsh[:, 0::2, 0::2]
            cxx = xx[:, 0::2] // 2
            cyy = yy[0::2] // 2
            U = (128 + ((cxx + scene * 13) % 41) - 20 + ((hc >> 12) & 7) - 3).to(torch.float32) * (maxv / 255.0)
            V = (128 + ((cyy + scene * 7) % 37) - 18 + ((hc >> 16) & 7) - 3).to(torch.float32) * (maxv / 255.0)
        if al is not None and flat_every:
            h, w = al.shape
            hv = ((n * 2654435761 + seed) ^ ((n >> 3) * 40503)) & 0x7FFFFFFF
            flat = (hv % flat_every) == 0
            lvl = (16 + (hv >> 8) % 180).to(torch.float32) * (maxv / 255.0)
            r = Y[:, imgy:imgy + h, imgx:imgx + w]
            fr = (lvl.round() + ((r.long() % 5) - 2).to(torch.float32)).clamp(0, maxv)
            ring = torch.zeros((1, h, w), dtype=torch.bool, device=device)
            ring[:, :2, :] = True; ring[:, -2:, :] = True; ring[:, :, :2] = True; ring[:, :, -2:] = True
            fr = torch.where(ring, lvl.round().expand_as(fr), fr)
            Y[:, imgy:imgy + h, imgx:imgx + w] = torch.where(flat, fr, r)
            if chroma:
                cu = ((112 + (hv >> 20) % 32).to(torch.float32) * (maxv / 255.0)).round()
                cv = ((112 + (hv >> 28) % 8 * 4).to(torch.float32) * (maxv / 255.0)).round()
                for P, cval in ((U, cu), (V, cv)):
                    r = P[:, imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2]
                    P[:, imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2] = torch.where(flat, cval.expand_as(r), r)
        if al is not None:
            h, w = al.shape
            vis = torch.as_tensor(logo_presence(np.arange(start + c0, start + c1), period, fade),
                                  dtype=torch.float32, device=device)[:, None, None]
            a = al[None] * vis
            r = Y[:, imgy:imgy + h, imgx:imgx + w]
            Y[:, imgy:imgy + h, imgx:imgx + w] = (1 - a) * r + a * (235.0 / 255.0) * maxv
            if chroma:
                ac = alc[None] * vis
                for P in (U, V):
                    r = P[:, imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2]
                    P[:, imgy // 2:imgy // 2 + h // 2, imgx // 2:imgx // 2 + w // 2] = (1 - ac) * r + ac * (128.0 / 255.0) * maxv
        Ys[c0:c1, :, :W] = Y.round().clamp(0, maxv).to(dt)
        if chroma:
            Us[c0:c1, :, :W // 2] = U.round().clamp(0, maxv).to(dt)
            Vs[c0:c1, :, :W // 2] = V.round().clamp(0, maxv).to(dt)
    return {"Y": Ys, "U": Us, "V": Vs}
