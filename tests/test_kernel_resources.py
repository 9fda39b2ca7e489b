"""CPU (hipcc cross-compiles gfx950 without a GPU): the register and scratch budgets the SMC kernels' co-residency rests on.

Four sampling waves (ph_sample, 1024 threads = 4 waves per SIMD) and one wave of the ordered sums (ph_sums) share a SIMD's 512
VGPRs, and the sampling kernel is held at 96 VGPRs by amdgpu_waves_per_eu(5, 5) at the price of a few spilled dwords.  This test
reads the compiler's own report (-Rpass-analysis=kernel-resource-usage) and the ISA (-S): the budgets hold, and every scratch
access of ph_sample sits outside its sample loops (at most once per work item of 65 / 125 samples, never per sample group)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pnr_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-w"]


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "smc_phased.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", "smc_phased.hip", "-o", out],
                       cwd=SRC, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    usage, name = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]* *Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark: [^ ]* *(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
        if m and name:
            usage[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return usage, open(out).read().splitlines()


def find(usage, frag):
    hit = [k for k in usage if frag in k]
    assert len(hit) == 1, (frag, hit)
    return usage[hit[0]], hit[0]


def test_register_and_scratch_budgets(compiled):
    usage, _ = compiled
    smp, _ = find(usage, "ph_sampleILi54ELb0ELb1EE")   # 3-D stacks (the cube copied from ph_cube's compact copy)
    smp2, _ = find(usage, "ph_sampleILi54ELb1ELb1EE")  # single slice
    shallow, _ = find(usage, "ph_sumsILb0EE")
    deep, _ = find(usage, "ph_sumsILb1EE")
    assert smp["VGPRs"] <= 96 and smp2["VGPRs"] <= 96, (smp, smp2)          # 4 waves x 96 + one ph_sums wave <= 512
    assert smp["ScratchSize"] <= 32 and smp2["ScratchSize"] <= 32, (smp, smp2)  # a handful of dwords, see the ISA test below
    assert shallow["VGPRs"] <= 96 and shallow["ScratchSize"] == 0, shallow   # beside four sampling waves with room to spare
    assert deep["VGPRs"] <= 128 and deep["ScratchSize"] == 0, deep          # 4 x 96 + 128 = 512
    own, _ = find(usage, "ph_sampleILi54ELb0ELb0EE")  # option cube_copy = 0: every work-group stages its cube from the image
    assert own["VGPRs"] <= 96 and own["ScratchSize"] <= 32, own
    for frag in ("ph_predict", "ph_update", "ph_cube"):
        u, _ = find(usage, frag)
        assert u["ScratchSize"] == 0 and u["VGPRs"] <= 128, (frag, u)


def test_ph_sample_scratch_stays_outside_the_sample_loops(compiled):
    usage, asm = compiled
    _, name = find(usage, "ph_sampleILi54ELb0ELb1EE")
    start = next(i for i, ln in enumerate(asm) if ln.startswith(name + ":"))
    end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
    depth, worst, n = 0, 0, 0
    for ln in asm[start:end]:
        if ln.startswith(".LBB"):
            m = re.search(r"Depth=(\d+)", ln)
            depth = int(m.group(1)) if m else 0
        elif "scratch_" in ln:
            n += 1
            worst = max(worst, depth)
    # depth 1 = the loop over work items (one item = five template rows x 64 chains = 65 / 125 samples per lane); the row loop is
    # depth 2 and the five-sample groups depth 3: a spill there would be paid per sample
    assert worst <= 1, f"{n} scratch accesses, deepest at loop depth {worst}"
