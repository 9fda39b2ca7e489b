"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref/libpnr_ref.so =
/root/reference/pnr-vaa3d/{frangi,seed,node}.cpp compiled by oracle/Makefile).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The fixtures are data: inputs (synthetic stacks from tests/synth.py) and the reference's
outputs for Frangi::imgaussian, Frangi::hessian3d, Frangi::eigen_decomposition,
Frangi::frangi3d (J, Jmin, Jmax, Vx, Vy, Vz) and SeedExtractor::extractSeeds.
J8 is the orchestrator's min-max rule (Advantra_plugin.cpp:2499-2512), which lives in the
unbuildable plugin TU; the fixture's J8 is produced by the restatement orc_j8 from the
reference's J and is recorded as such (key "J8_restated").
"""
import os
import sys
import ctypes as C
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc    # noqa: E402
import synth  # noqa: E402

CASES = {
    # name: (w, h, l, synth seed, sigmas, zdist, tolerance)
    "g1_48x40x24_s2-3": (48, 40, 24, 1, [2.0, 3.0], 2.0, 5.0),
    "g2_64x64x32_s2-4-6": (64, 64, 32, 2, [2.0, 4.0, 6.0], 2.0, 5.0),
    "g3_40x36x20_z4_s2-4": (40, 36, 20, 5, [2.0, 4.0], 4.0, 10.0),
}


def main():
    R = orc.load_ref()
    L = orc.load_oracle()
    assert R is not None, "oracle/_ref/libpnr_ref.so missing (needs /root/reference)"
    for name, (w, h, l, seed, sigs, zdist, tol) in CASES.items():
        img = synth.synth(w, h, l, seed=seed, zdist=zdist if zdist > 2 else 1.0)
        J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(R, img, sigs, zdist, prefix="ref")
        J8 = orc.j8(L, J, jmin, jmax)
        seeds = orc.extract_seeds(R, tol, J8, Vx, Vy, Vz, prefix="ref")
        s0 = np.float32(sigs[0])
        F = np.zeros(img.shape, np.float32)
        R.ref_imgaussian3d(img, w, h, l, s0, zdist, F)
        H = [np.zeros(img.shape, np.float32) for _ in range(6)]
        R.ref_hessian3d(img, w, h, l, s0, zdist, *H)
        out = dict(img=img, sigs=np.float32(sigs), zdist=np.float32(zdist), tol=np.float32(tol), J=J, Jmin=np.float32(jmin),
                   Jmax=np.float32(jmax), Vx=Vx, Vy=Vy, Vz=Vz, J8_restated=J8, seeds=seeds, F_sig0=F,
                   Dzz=H[0], Dyy=H[1], Dyz=H[2], Dxx=H[3], Dxy=H[4], Dxz=H[5])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "Jmax", jmax, "seeds", len(seeds), os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")
    # 2-D mode (single-slice stack): Frangi::frangi2d / hessian2d / extractSeeds of the reference
    for name, (w, h, seed, sigs, tol) in {"p2d_96x80_s2-3": (96, 80, 4, [2.0, 3.0], 5.0), "p2d_33x21_s2": (33, 21, 6, [2.0], 5.0)}.items():
        img = np.ascontiguousarray(synth.synth(w, h, 9, seed=seed).max(0, keepdims=True))
        J, jmin, jmax, Vx, Vy, Vz = orc.frangi2d(R, img.copy(), sigs, prefix="ref")
        J8 = orc.j8(L, J, jmin, jmax)
        seeds = orc.extract_seeds(R, tol, J8, Vx, Vy, Vz, prefix="ref")
        D = [np.zeros(img.shape, np.float32) for _ in range(3)]
        R.ref_hessian2d(img.copy(), w, h, np.float32(sigs[0]), *D)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), img=img, sigs=np.float32(sigs), tol=np.float32(tol), J=J, Jmin=np.float32(jmin),
                            Jmax=np.float32(jmax), Vx=Vx, Vy=Vy, Vz=Vz, J8_restated=J8, seeds=seeds, Dyy=D[0], Dxy=D[1], Dxx=D[2])
        print(name, "Jmax", jmax, "seeds", len(seeds))
    # soma path: Frangi::imerode (xy) and the u8 Frangi::imgaussian of the reference
    for name, (w, h, l, seed, rad, somas) in {"soma_64x56x32_r3": (64, 56, 32, 2, 3, ((20, 28, 16, 6), (48, 20, 14, 5))),
                                              "soma_23x20x9_r2": (23, 20, 9, 3, 2, ((10, 10, 4, 4),))}.items():
        img = synth.add_somas(synth.synth(w, h, l, seed=seed), somas)
        E = np.zeros_like(img)
        R.ref_imerode_xy(img.copy(), w, h, l, np.float32(rad), E)
        G = E.copy()
        R.ref_imgaussian_u8_xy(G, w, h, l, np.float32(rad))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), img=img, rad=np.int32(rad), eroded=E, blurred=G)
        print(name, "eroded max", E.max(), "blurred max", G.max())
    # eigen KATs: random symmetric matrices incl. degenerate / diagonal / zero cases
    rs = np.random.RandomState(7)
    A = rs.randn(512, 3, 3)
    A = A + A.transpose(0, 2, 1)
    A[0] = 0
    A[1] = np.diag([1.0, 2.0, 3.0])
    A[2] = np.diag([-3.0, 1.0, 1.0])
    A[3] = np.eye(3)
    A[4] = np.array([[2.0, 1, 0], [1, 2, 0], [0, 0, -5]])
    A[5:64] = (A[5:64] * 1e-3).astype(np.float32)  # f32-valued inputs as in frangi3d
    V = np.zeros_like(A)
    d = np.zeros((len(A), 3))
    for i in range(len(A)):
        R.ref_eigen3(np.ascontiguousarray(A[i]), V[i], d[i])
    np.savez_compressed(os.path.join(HERE, "eigen_kat.npz"), A=A, V=V, d=d)
    print("eigen_kat", len(A))


if __name__ == "__main__":
    main()
