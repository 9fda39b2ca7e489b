#!/usr/bin/env python3
"""Numeric diff of two SWC files (SURVEY 8(b) output row; BASELINE configs[4] "SWC diff vs CPU reference").

The reference writes its trees through Vaa3D's writeSWC_file (Advantra_plugin.cpp:480-523): one line `n type x y z r parent` per
node, `#` comment lines for the header.  The text formatting of the floats belongs to the Vaa3D SDK, so two writers are compared
numerically: node ids, types and parents must be identical, x / y / z / r equal within a tolerance (default 2e-3: the %.3f of the
usual writers), line for line in file order.  Prints the first difference and a summary; exit code 0 = equal, 1 = different,
2 = unreadable input.

  python scripts/swc_diff.py a.swc b.swc [--tol 2e-3] [--unordered] [--quiet]

--unordered matches the nodes by id instead of by line (two writers that emit the same tree in different line orders).
Importable: swc_diff.read_swc(path) -> (ids, types, xyzr, parents); swc_diff.diff(a, b, tol, unordered) -> (equal, message).
"""
import argparse
import sys

import numpy as np


def read_swc(path):
    """(ids int64[n], types int64[n], xyzr float64[n, 4], parents int64[n]) of an SWC file; comment and empty lines are skipped"""
    ids, types, xyzr, parents = [], [], [], []
    with open(path) as f:
        for ln_no, ln in enumerate(f, 1):
            s = ln.strip()
            if not s or s[0] == "#":
                continue
            t = s.split()
            if len(t) < 7:
                raise ValueError(f"{path}:{ln_no}: expected 7 fields (n type x y z r parent), found {len(t)}")
            try:
                # ids / types / parents are integers; some writers print them as floats ("3.0")
                i, ty, par = (int(float(t[0])), int(float(t[1])), int(float(t[6])))
                v = [float(t[2]), float(t[3]), float(t[4]), float(t[5])]
            except ValueError as e:
                raise ValueError(f"{path}:{ln_no}: {e}") from None
            ids.append(i); types.append(ty); xyzr.append(v); parents.append(par)
    return (np.asarray(ids, np.int64), np.asarray(types, np.int64), np.asarray(xyzr, np.float64).reshape(-1, 4), np.asarray(parents, np.int64))


def diff(a, b, tol=2e-3, unordered=False):
    """a, b: read_swc() tuples (or paths).  Returns (equal, message)."""
    if isinstance(a, str):
        a = read_swc(a)
    if isinstance(b, str):
        b = read_swc(b)
    ia, ta, va, pa = a
    ib, tb, vb, pb = b
    if len(ia) != len(ib):
        return False, f"node counts differ: {len(ia)} vs {len(ib)}"
    if len(ia) == 0:
        return True, "both files hold no nodes"
    if unordered:
        if len(np.unique(ia)) != len(ia) or len(np.unique(ib)) != len(ib):
            return False, "--unordered needs unique node ids"
        oa, ob = np.argsort(ia, kind="stable"), np.argsort(ib, kind="stable")
        ia, ta, va, pa = ia[oa], ta[oa], va[oa], pa[oa]
        ib, tb, vb, pb = ib[ob], tb[ob], vb[ob], pb[ob]
    names = ("x", "y", "z", "r")
    for what, x, y in (("id", ia, ib), ("type", ta, tb), ("parent", pa, pb)):
        bad = np.nonzero(x != y)[0]
        if len(bad):
            k = int(bad[0])
            return False, f"{what} differs at node #{k} (id {ia[k]}): {x[k]} vs {y[k]} ({len(bad)} of {len(x)} nodes differ in {what})"
    d = np.abs(va - vb)
    nan_mismatch = np.isnan(va) != np.isnan(vb)
    d = np.where(np.isnan(va) & np.isnan(vb), 0.0, d)
    bad = np.nonzero((d > tol) | nan_mismatch)
    if len(bad[0]):
        k, c = int(bad[0][0]), int(bad[1][0])
        return False, (f"{names[c]} differs at node #{k} (id {ia[k]}): {va[k, c]!r} vs {vb[k, c]!r} (|diff| {d[k, c]:.3g} > {tol:g}); "
                       f"{len(np.unique(bad[0]))} of {len(ia)} nodes beyond the tolerance, largest |diff| {np.nanmax(d):.3g}")
    return True, f"{len(ia)} nodes: ids, types and parents identical, x / y / z / r within {tol:g} (largest |diff| {float(d.max()):.3g})"


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--tol", type=float, default=2e-3)
    ap.add_argument("--unordered", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    try:
        ok, msg = diff(args.a, args.b, args.tol, args.unordered)
    except (OSError, ValueError) as e:
        print(f"swc_diff: {e}", file=sys.stderr)
        return 2
    if not args.quiet:
        print(("equal: " if ok else "DIFFERENT: ") + msg)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
