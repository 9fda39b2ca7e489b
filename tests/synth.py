"""Synthetic u8 stacks for parity tests and golden vectors (own generator).

Black background + additive Gaussian-profile tubes A*exp(-d^2/2s^2) around
poly-lines (axis-aligned, oblique, helix, Y-junction) + uniform noise 0..10 from
a counter-based 32-bit hash, saturating at 255.
Deterministic in (w, h, l, seed); anisotropic stacks squeeze the tube
cross-section along z by `zdist`.
"""
import numpy as np


def _hash32(i, seed):
    """lowbias32-style integer mix on uint64 lanes masked to 32 bits."""
    M = np.uint64(0xFFFFFFFF)
    x = (i.astype(np.uint64) + np.uint64((seed * 0x9E3779B9) & 0xFFFFFFFF)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return x


def tube_polylines(w, h, l, seed, zdist=1.0):
    """List of (points[K,3] in voxel coords (x,y,z), s, A)."""
    rng = np.random.RandomState(1000 + seed)
    W, H, L = float(w - 1), float(h - 1), float(l - 1)
    tubes = []
    radii = [1.5, 2.0, 3.0, 4.0]
    # (i) axis-aligned line along x
    tubes.append((np.array([[0.08 * W, 0.30 * H, 0.45 * L], [0.92 * W, 0.30 * H, 0.45 * L]]), 2.0, 200.0))
    # (ii) oblique line
    tubes.append((np.array([[0.15 * W, 0.85 * H, 0.20 * L], [0.85 * W, 0.55 * H, 0.80 * L]]), 1.5, 180.0))
    # (iii) helix around the stack centre
    tt = np.linspace(0, 1, 48)
    hel = np.stack([0.5 * W + 0.28 * W * np.cos(4 * np.pi * tt),
                    0.5 * H + 0.28 * H * np.sin(4 * np.pi * tt),
                    0.15 * L + 0.7 * L * tt], 1)
    tubes.append((hel, 2.0, 160.0))
    # (iv) Y-junction
    j = np.array([0.5 * W, 0.62 * H, 0.5 * L])
    tubes.append((np.array([[0.5 * W, 0.95 * H, 0.5 * L], j]), 3.0, 220.0))
    tubes.append((np.array([j, [0.22 * W, 0.40 * H, 0.35 * L]]), 2.0, 190.0))
    tubes.append((np.array([j, [0.78 * W, 0.42 * H, 0.68 * L]]), 2.0, 170.0))
    # extra random oblique segments, count ~ volume^(1/3)
    nextra = max(0, int(round((w * h * l) ** (1.0 / 3.0) / 16.0)) - 2)
    for _ in range(nextra):
        a = rng.uniform(0.05, 0.95, 3) * [W, H, L]
        b = rng.uniform(0.05, 0.95, 3) * [W, H, L]
        tubes.append((np.array([a, b]), float(rng.choice(radii)), float(rng.uniform(120, 220))))
    return tubes


def synth(w, h, l, seed=1, zdist=1.0, noise=10):
    """Return uint8 array of shape (l, h, w) (x fastest)."""
    vol = np.zeros((l, h, w), np.float32)
    for pts, s, A in tube_polylines(w, h, l, seed, zdist):
        for a, b in zip(pts[:-1], pts[1:]):
            lo = np.floor(np.minimum(a, b) - 4 * s - 1).astype(int)
            hi = np.ceil(np.maximum(a, b) + 4 * s + 2).astype(int)
            x0, y0, z0 = max(lo[0], 0), max(lo[1], 0), max(lo[2], 0)
            x1, y1, z1 = min(hi[0], w), min(hi[1], h), min(hi[2], l)
            if x0 >= x1 or y0 >= y1 or z0 >= z1:
                continue
            zz, yy, xx = np.meshgrid(np.arange(z0, z1), np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
            P = np.stack([xx, yy, zz], -1).astype(np.float64)
            ab = b - a
            t = np.clip(((P - a) @ ab) / max(ab @ ab, 1e-12), 0.0, 1.0)
            D = P - (a + t[..., None] * ab)
            D[..., 2] *= zdist  # squeeze cross-section along z for anisotropic stacks
            d2 = (D * D).sum(-1)
            vol[z0:z1, y0:y1, x0:x1] = np.maximum(vol[z0:z1, y0:y1, x0:x1], (A * np.exp(-d2 / (2 * s * s))).astype(np.float32))
    n = w * h * l
    nz = (_hash32(np.arange(n, dtype=np.uint64), seed) % np.uint64(noise + 1)).astype(np.float32).reshape(l, h, w)
    out = np.clip(np.floor(vol + nz), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(out)


def synth_torch(w, h, l, seed=1, zdist=1.0, noise=10, device="cuda"):
    """Same construction on the GPU with torch (data generation only -- plumbing): returns a uint8
    torch tensor [l][h][w] resident in HBM.  Used by bench.py for the 512^3 / 1024^3 stacks, where
    the numpy generator would take minutes.  Same tubes and the same hash noise as synth()."""
    import torch
    vol = torch.zeros((l, h, w), dtype=torch.float32, device=device)
    for pts, s, A in tube_polylines(w, h, l, seed, zdist):
        for a, b in zip(pts[:-1], pts[1:]):
            lo = np.floor(np.minimum(a, b) - 4 * s - 1).astype(int)
            hi = np.ceil(np.maximum(a, b) + 4 * s + 2).astype(int)
            x0, y0, z0 = max(lo[0], 0), max(lo[1], 0), max(lo[2], 0)
            x1, y1, z1 = min(hi[0], w), min(hi[1], h), min(hi[2], l)
            if x0 >= x1 or y0 >= y1 or z0 >= z1:
                continue
            ab = b - a
            den = max(float(ab @ ab), 1e-12)
            # march the bounding box in z-chunks to bound temporaries for long oblique tubes
            zc = max(1, int(2 ** 24 // max(1, (y1 - y0) * (x1 - x0))))
            for za in range(z0, z1, zc):
                zb = min(z1, za + zc)
                zz = torch.arange(za, zb, device=device, dtype=torch.float32)[:, None, None]
                yy = torch.arange(y0, y1, device=device, dtype=torch.float32)[None, :, None]
                xx = torch.arange(x0, x1, device=device, dtype=torch.float32)[None, None, :]
                t = (((xx - a[0]) * ab[0] + (yy - a[1]) * ab[1] + (zz - a[2]) * ab[2]) / den).clamp(0.0, 1.0)
                dx = xx - (a[0] + t * ab[0])
                dy = yy - (a[1] + t * ab[1])
                dz = (zz - (a[2] + t * ab[2])) * zdist
                val = A * torch.exp(-(dx * dx + dy * dy + dz * dz) / (2 * s * s))
                sub = vol[za:zb, y0:y1, x0:x1]
                torch.maximum(sub, val, out=sub)
    out = torch.empty((l, h, w), dtype=torch.uint8, device=device)
    M = 0xFFFFFFFF
    add = (seed * 0x9E3779B9) & M
    wh = w * h
    zc = max(1, (1 << 26) // wh)
    for za in range(0, l, zc):
        zb = min(l, za + zc)
        i = torch.arange(za * wh, zb * wh, device=device, dtype=torch.int64)
        x = (i + add) & M
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & M
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & M
        x = x ^ (x >> 16)
        nz = (x % (noise + 1)).to(torch.float32).reshape(zb - za, h, w)
        out[za:zb] = torch.clamp(torch.floor(vol[za:zb] + nz), 0, 255).to(torch.uint8)
    return out


def add_somas(img, centres_radii, amplitude=230):
    """Bright solid balls (x, y, z, r) added to a synth() stack: the cell bodies the soma path looks for
    (somaradius > 0).  Returns a new uint8 array."""
    l, h, w = img.shape
    out = img.astype(np.float32)
    zz, yy, xx = np.meshgrid(np.arange(l), np.arange(h), np.arange(w), indexing="ij")
    for (cx, cy, cz, r) in centres_radii:
        d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2)
        out = np.maximum(out, amplitude * np.clip(r + 1.0 - d, 0.0, 1.0))
    return np.ascontiguousarray(np.clip(np.floor(out), 0, 255).astype(np.uint8))
