"""LDS-array cycles of the corner gathers of ph_sample for real particle clouds, by the bank rules of MI355X_MICROARCH.md (LDS):
ds_read_u8 / ds_read_b32: two groups of 32 lanes, bank = (a / 4) mod 32; ds_read_b64: two groups of 32 lanes, bank = (a / 4) mod 64,
two banks per lane; identical dwords broadcast; every further distinct address on a busy bank costs one more cycle.

Layouts compared, per template sample of a wavefront (64 lanes):
  P-u8   lane = particle (today): 8 ds_read_u8, cube pitch 56 B / plane 54 x 56 B
  P-b64  lane = particle, rows stored as overlapping 8-byte windows of 7 new voxels each (window k = bytes 7k .. 7k+7), so that one
         aligned ds_read_b64 always holds (x, x+1): 4 ds_read_b64
  S-u8   lane = 64 CONSECUTIVE template samples of ONE particle, 8 ds_read_u8
  S-b64  the same with the windows
The particle clouds come from the oracle's tracker (CPU) on a synthetic stack; no GPU is needed.

  python scripts/sim_lds_banks.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import synth  # noqa: E402


def frames(q):
    """znccBBB's local frame (tracker.cpp:1893-1917) for poses q[n, 6]"""
    x, y, z, vx, vy, vz = [q[:, i].astype(np.float64) for i in range(6)]
    nrm = np.sqrt(vx * vx + vy * vy)
    sg = np.where(vy < 0, -1.0, 1.0)
    ok = nrm > 1e-4
    nn = np.where(ok, nrm, 1.0)
    ux = np.where(ok, sg * vy / nn, 1.0)
    uy = np.where(ok, -sg * vx / nn, 0.0)
    uz = np.zeros_like(ux)
    wx = uy * vz - uz * vy
    wy = -ux * vz + uz * vx
    wz = ux * vy - uy * vx
    return np.stack([x, y, z], 1), np.stack([-vx, -vy, -vz], 1), np.stack([ux, uy, uz], 1), np.stack([wx, wy, wz], 1)


def group_cycles(dw, banks, width):
    """cycles of one 32-lane group: max over banks of the number of distinct dwords; `width` consecutive banks per lane"""
    best = 1
    occ = {}
    for d in np.unique(dw):
        for k in range(width):
            b = (d + k) % banks
            occ[b] = occ.get(b, 0) + 1
    for v in occ.values():
        best = max(best, v)
    return best


def wave_cycles_u8(addr):
    """addr[64] byte addresses of one ds_read_u8 wave-instruction"""
    d = addr // 4
    return group_cycles(d[:32], 32, 1) + group_cycles(d[32:], 32, 1)


def wave_cycles_b64(addr8):
    """addr8[64] 8-byte aligned byte addresses of one ds_read_b64 wave-instruction"""
    d = addr8 // 4
    return group_cycles(d[:32], 64, 2) + group_cycles(d[32:], 64, 2)


def main():
    L = orc.load_oracle()
    sigs = [2.0, 4.0, 6.0]
    img = synth.synth(96, 96, 64, seed=3)
    T = orc.Tracker(L, sigs, 2, 200, 12, 3.0, 0.3, zdist=2.0)
    # a seed on a tube: take the brightest voxel, direction unknown -> NaN (the tracker then draws directions from its table)
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, [2.0], 2.0)
    J8 = orc.j8(L, J, jmin, jmax)
    seeds = orc.extract_seeds(L, 5, J8, Vx, Vy, Vz)
    corr, _ = T.zncc(img, seeds[:, :6])
    order = np.argsort(-corr)[:4]
    rng = np.random.default_rng(0)
    CS, PITCH = 54, 56
    PLANE = CS * PITCH
    PLANE64 = PLANE + int(os.environ.get("SIM_PLANE_PAD", "0"))  # plane pitch of the window layout (multiple of 8)
    tot = {k: [0, 0] for k in ("P-u8", "P-b64", "S-u8", "S-b64")}
    for si in order:
        Tn, stop, xc, xf, idx, neff = T.trace(img, seeds[si, :6], max_dbg=12)
        for it in range(2, min(Tn, 12), 3):
            P = xf[it][:, :6]
            pos, nv, u, w = frames(P)
            org = np.floor(pos.mean(0)).astype(int) - CS // 2
            for s in (1, 2):
                vuw, wgt, avg = T.model(s)
                M = len(vuw)
                # ---- lane = particle: 64 particles, one sample per wave-instruction (sample a subset of the template)
                for k in rng.choice(M, 40, replace=False):
                    for g0 in (0, 64, 128):
                        sel = slice(g0, g0 + 64)
                        xyz = pos[sel] + vuw[k, 0] * nv[sel] + vuw[k, 1] * u[sel] + vuw[k, 2] * w[sel]
                        c = np.floor(xyz).astype(int) - org
                        c = np.clip(c, 0, CS - 2)
                        for dz in (0, 1):
                            for dy in (0, 1):
                                row = (c[:, 2] + dz) * PLANE + (c[:, 1] + dy) * PITCH
                                for dx in (0, 1):
                                    tot["P-u8"][0] += wave_cycles_u8(row + c[:, 0] + dx)
                                    tot["P-u8"][1] += 1
                                xw = np.minimum(c[:, 0], 48) // 7
                                row64 = (c[:, 2] + dz) * PLANE64 + (c[:, 1] + dy) * PITCH
                                tot["P-b64"][0] += wave_cycles_b64(row64 + 8 * xw)
                                tot["P-b64"][1] += 1
                # ---- lane = sample: 64 consecutive samples of one particle
                for p in rng.choice(len(P), 12, replace=False):
                    for k0 in rng.choice(M // 64, 10, replace=False) * 64:
                        t = vuw[k0:k0 + 64]
                        xyz = pos[p] + t[:, 0:1] * nv[p] + t[:, 1:2] * u[p] + t[:, 2:3] * w[p]
                        c = np.floor(xyz).astype(int) - org
                        c = np.clip(c, 0, CS - 2)
                        for dz in (0, 1):
                            for dy in (0, 1):
                                row = (c[:, 2] + dz) * PLANE + (c[:, 1] + dy) * PITCH
                                for dx in (0, 1):
                                    tot["S-u8"][0] += wave_cycles_u8(row + c[:, 0] + dx)
                                    tot["S-u8"][1] += 1
                                xw = np.minimum(c[:, 0], 48) // 7
                                row64 = (c[:, 2] + dz) * PLANE64 + (c[:, 1] + dy) * PITCH
                                tot["S-b64"][0] += wave_cycles_b64(row64 + 8 * xw)
                                tot["S-b64"][1] += 1
    print("layout   LDS cycles per wave-instruction   instructions per sample   LDS cycles per sample-wave")
    for k, (cyc, n) in tot.items():
        per = cyc / max(n, 1)
        ins = 8 if k.endswith("u8") else 4
        print(f"{k:7s}  {per:6.2f}                              {ins}                         {per * ins:6.1f}")


if __name__ == "__main__":
    main()
