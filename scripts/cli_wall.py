"""What a user of advantra_func sees: advantra_cli on a 1024^3 raw u8 file with the README's parameters
(scales 2,4,6; np 200; ni 200) -> SWC, wall time split into load / context + upload / Frangi / seeds / selection / full trace loop /
reconstruct() / write.  Run on the GPU box:   python scripts/cli_wall.py [size] > gpurun_out/cli_wall.json
The stack is the bench stack (tests/synth.py seed 3), written to /tmp as a raw file first (untimed)."""
import json, os, re, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
path = f"/tmp/pnr_cli_{S}.raw"
img = synth.synth_torch(S, S, S, seed=3).cpu().numpy()
img.tofile(path)
del img
torch.cuda.empty_cache()
cli = os.path.join(R, "pnr_amd", "host", "advantra_cli")
# the 11 parameters of README.md:17: scales, somaradius, tolerance, znccth, kappa, step, ni, np, zdist, nodepervol, vol
cmd = [cli, "--timing", "-d", f"{S},{S},{S}", "-f", "advantra_func", "-i", path, "-p", "2,4,6", "0", "5", "0.3", "3", "2", "200", "200", "2", "4", "1"]
out = []
for rep in range(2):  # (the first run pays the page cache and the driver's first-touch costs)
    t0 = time.time()
    pr = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    for ln in pr.stderr.splitlines():  # (--timing: the stages of reconstruct())
        if ln.startswith("[pnr reconstruct]") or ln.startswith("[pnr trace]") or ln.startswith("[pnr host]"): print(ln, file=sys.stderr)
    m = re.search(r"wall: load ([\d.]+) s, context \+ upload ([\d.]+) s, frangi ([\d.]+) s, seeds ([\d.]+) s, selection ([\d.]+) s, tracing ([\d.]+) s, reconstruct ([\d.]+) s, write ([\d.]+) s \| total ([\d.]+) s", pr.stdout)
    m2 = re.search(r"(\d+) trace nodes, (\d+) traces, (\d+) SMC iterations, (\d+) tree nodes", pr.stdout)
    if not m:
        print(pr.stdout[-2000:], pr.stderr[-2000:], file=sys.stderr)
        sys.exit(1)
    k = ("load", "context_upload", "frangi", "seeds", "selection", "tracing", "reconstruct", "write", "total")
    d = dict(zip(k, map(float, m.groups())))
    d.update(process_wall_s=wall, run=rep, trace_nodes=int(m2.group(1)), traces=int(m2.group(2)), smc_iterations=int(m2.group(3)), tree_nodes=int(m2.group(4)))
    d["Mvox_per_s_file_to_swc"] = S ** 3 / d["total"] / 1e6
    d["share_reconstruct_io"] = (d["load"] + d["reconstruct"] + d["write"]) / d["total"]
    out.append(d)
print(json.dumps({"command": " ".join(cmd), "size": S, "runs": out}, indent=1))
os.remove(path)
for f in (path + "_Advantra.swc",):
    if os.path.exists(f):
        os.remove(f)
