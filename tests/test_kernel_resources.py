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


def compile_isa(tmp_path_factory, source):
    out = str(tmp_path_factory.mktemp("isa") / (source + ".s"))
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", source, "-o", out],
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


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    return compile_isa(tmp_path_factory, "smc_phased.hip")


@pytest.fixture(scope="module")
def compiled_frangi(tmp_path_factory):
    return compile_isa(tmp_path_factory, "frangi.hip")


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


def kernel_body(usage, asm, frag):
    _, name = find(usage, frag)
    start = next(i for i, ln in enumerate(asm) if ln.startswith(name + ":"))
    end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
    return asm[start:end]


def test_frangi_kernels_keep_their_loads_in_flight(compiled_frangi):
    """What round 5 found by reading the ISA and what must stay so: hessian_tile at eight waves per SIMD whose wait in front of the ring
    store of an inner plane leaves the newer plane in flight (vmcnt(2) / vmcnt(3), it was vmcnt(0)), with its few spilled dwords
    outside the runs of inner planes; the marching x-y Gaussian without scratch at four waves per SIMD, all of a chunk's loads issued back to back."""
    usage, asm = compiled_frangi
    ht, _ = find(usage, "hessian_tileILb0EE")
    assert ht["VGPRs"] <= 64 and ht["Occupancy"] == 8 and ht["ScratchSize"] <= 32, ht
    body = kernel_body(usage, asm, "hessian_tileILb0EE")
    depth, inner = 0, []
    for i, ln in enumerate(body):  # the waits in front of the ring stores of the inner planes: inside the march (loop depth >= 1)
        if ln.startswith(".LBB"):
            m = re.search(r"Depth=(\d+)", ln)
            depth = int(m.group(1)) if m else 0
        elif depth >= 1 and re.search(r"s_waitcnt vmcnt\((2|3)\)", ln):
            inner.append(i)
    assert len(inner) >= 8, len(inner)  # (a run of six planes, less its last two, in two instantiations: the newer plane stays in flight)
    spills = [i for i, ln in enumerate(body) if "scratch_" in ln]
    assert not [i for i in spills if inner[0] <= i <= inner[-1]], "scratch access inside the runs of inner planes of hessian_tile"
    for L in (6, 12, 18):
        u, _ = find(usage, f"gauss_xy_u8_mILi{L}E")
        assert u["ScratchSize"] == 0 and u["VGPRs"] <= 128 and u["Occupancy"] >= 4, (L, u)
        body = kernel_body(usage, asm, f"gauss_xy_u8_mILi{L}E")
        run = best = 0
        for ln in body:  # the longest run of global loads with no wait between them
            if "global_load_dword" in ln:
                run += 1; best = max(best, run)
            elif "s_waitcnt vmcnt" in ln:
                run = 0
        assert best >= 4, (L, best)
