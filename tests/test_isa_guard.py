"""Build-time guard of the hand-tuned kernels: cross-compiles csrc/eaz_api.hip to gfx950 assembly (no GPU needed) and checks what the
one-launch kernels silently depend on -- no scratch, register budgets of the occupancy each kernel is designed for, the MFMA / buffer-load
streams present, the work-queue claim a single LDS atomic without the atomic optimizer's scan loop around it, the kernarg struct at
offset 0 (the kernels read their arguments through the laundered kernarg segment pointer), and the barrier count of fc16p_body that the
idle head group of subleq16_search_kernel mirrors (kFpBarriers).  A regression here used to show up only as a GPU timeout.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "e-alphazero_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / "eaz.s"
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]  # csrc/Makefile CXXFLAGS
    subprocess.run([HIPCC, *flags, "-o", str(out), "eaz_api.hip"], cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text()


def kernel(asm, stem):
    """-> (body text, metadata dict) of the kernel whose mangled name contains `stem`"""
    m = re.search(r"^(_ZN3eaz\d+%s\w*):" % stem, asm, re.M)
    assert m, f"kernel {stem} not found in the assembly"
    name = m.group(1)
    end = asm.index(f".amdhsa_kernel {name}")
    body = asm[asm.index("\n", m.end()):end]
    i = asm.index(f".name:           {name}")
    j = asm.rfind("- .agpr_count", 0, i)
    k = asm.find("- .agpr_count", i)
    meta_txt = asm[j:k if k > 0 else len(asm)]
    meta = {key: int(v) for key, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", meta_txt, re.M)}
    args = re.findall(r"- \.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", meta_txt)
    meta["explicit_args"] = [(int(o), int(sz), kind) for o, sz, kind in args if not kind.startswith("hidden")]
    return body, meta


def count(body, mnemonic):
    return len(re.findall(r"^\s+%s\b" % re.escape(mnemonic), body, re.M))


# kernel stem -> VGPR budget of the occupancy it is designed for (MI355X_MICROARCH.md register table: 512 / waves per SIMD, granule 8)
BUDGET = {
    "search_persist_kernel": 256,      # 512 threads (4 MFMA + 4 helper waves): two waves per SIMD
    "subleq16_search_kernel": 256,     # 512 threads: two waves per SIMD
    "subleq16_selfplay_kernel": 256,   # 512 threads: two waves per SIMD
    "fc16p_forward_kernel": 168,       # three 256-thread workgroups per CU: three waves per SIMD
    "resnet_forward_kernel": 256,      # one 512-thread workgroup per CU: two waves per SIMD
}


@pytest.mark.parametrize("stem", sorted(BUDGET))
def test_no_scratch_and_register_budget(asm, stem):
    body, meta = kernel(asm, stem)
    assert meta["private_segment_fixed_size"] == 0, f"{stem} uses scratch memory ({meta['private_segment_fixed_size']} bytes per lane)"
    assert meta["vgpr_spill_count"] == 0, f"{stem} spills {meta['vgpr_spill_count']} VGPRs"
    assert count(body, "scratch_load_dword") + count(body, "scratch_store_dword") == 0
    assert meta["vgpr_count"] + meta.get("agpr_count", 0) <= BUDGET[stem], f"{stem}: {meta['vgpr_count']} VGPRs exceed the {BUDGET[stem]} of its occupancy"
    assert count(body, "v_mfma_f32_16x16x4_f32") > 0, f"{stem} has lost its f32 MFMA stream"


def all_kernels(asm):
    """{mangled name: metadata dict} of every kernel of the library"""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)", asm):
        name = m.group(1)
        i = asm.index(f".name:           {name}")
        j = asm.rfind("- .agpr_count", 0, i)
        k = asm.find("- .agpr_count", i)
        txt = asm[j:k if k > 0 else len(asm)]
        out[name] = {key: int(v) for key, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", txt, re.M)}
    return out


# kernels that are compiled but never launched (eaz_api.hip does not dispatch them) and may keep their scratch use
NOT_DISPATCHED = ("subleq_tree_leaf_kernelILi2E",)   # word sizes 65..128: 20 bytes of scratch; those searches stay on tree_step + subleq_wave


def test_no_launched_kernel_uses_scratch_memory(asm):
    """Round 3 shipped two kernels on BASELINE / code-default paths with scratch memory nobody had asked for (subleq_wave_kernel 112 bytes: indexed
    local arrays of the test vectors and the IO rows; connect4_step_kernel 84 bytes: the board as an indexed int8 array).  Both are gone
    (packed compile-time tables / bitboards); this holds it for EVERY kernel of the library, the new ones included."""
    ks = all_kernels(asm)
    assert len(ks) >= 40
    bad = {n: d["private_segment_fixed_size"] for n, d in ks.items() if d.get("private_segment_fixed_size", 0) and not any(x in n for x in NOT_DISPATCHED)}
    assert not bad, f"kernels with scratch memory: {bad}"
    for stem in ("subleq_wave_kernel", "connect4_step_kernel", "connect4_leaf_kernel", "subleq_step_kernel", "subleq16_tree_leaf_kernel",
                 "subleq_tree_leaf_kernelILi1E", "subleq_tree_leaf_kernelILi4E", "sq16_roots_fit_kernel", "sq16_l1_table_kernel"):
        body, _ = kernel(asm, stem)
        assert count(body, "scratch_load_dword") + count(body, "scratch_store_dword") + count(body, "scratch_load_dwordx4") + count(body, "scratch_store_dwordx4") == 0, stem


# occupancy each of the remaining hot kernels is designed for: (VGPRs + AGPRs) budget
WIDE_BUDGET = {
    r"tree_step_kernelILi\d+ELi\d+EE": 256,     # one-wavefront workgroups, two waves per SIMD (memory-level parallelism, not lanes)
    r"fc_forward_kernelILb[01]EE": 512,         # 64-row tiles: 4 x 4 MFMA tiles per wave = 256 accumulator registers (AGPRs) + operands: one wave per SIMD
    r"fc16_forward_kernelILi[012]EE": 168,      # three workgroups per CU
    r"subleq16_tree_leaf_kernel": 128,          # one wavefront per tree, four per SIMD
    r"subleq_tree_leaf_kernelILi[14]E": 128,
    r"connect4_leaf_kernel": 128,
    r"subleq_wave_kernel": 64,
}


@pytest.mark.parametrize("pattern", sorted(WIDE_BUDGET))
def test_register_budgets_of_the_per_simulation_kernels(asm, pattern):
    ks = {n: d for n, d in all_kernels(asm).items() if re.search(pattern, n)}
    assert ks, f"no kernel matches {pattern}"
    for n, d in ks.items():
        total = d["vgpr_count"]  # (the metadata's vgpr_count already includes the AGPRs of a unified register file allocation)
        assert total <= WIDE_BUDGET[pattern], f"{n}: {total} registers exceed the {WIDE_BUDGET[pattern]} of its designed occupancy"
        assert d.get("vgpr_spill_count", 0) == 0, f"{n} spills {d['vgpr_spill_count']} VGPRs"


@pytest.mark.parametrize("stem", ["search_persist_kernel", "subleq16_search_kernel", "subleq16_selfplay_kernel", "fc16p_forward_kernel", "resnet_forward_kernel"])
def test_weight_stream_is_buffer_loads(asm, stem):
    body, _ = kernel(asm, stem)
    # the ResNet's 16-byte loads serve four k-steps each (host-packed operand order, net_resnet.hpp): 2 per group of 8 k-steps, 6 per convolution site
    need = 12 if stem == "resnet_forward_kernel" else 16
    assert count(body, "buffer_load_dwordx4") >= need, "the weight stream (16 k-steps of 16-byte buffer loads in flight) is gone"


@pytest.mark.parametrize("stem", ["subleq16_search_kernel", "subleq16_selfplay_kernel"])
def test_work_queue_claim_is_one_lds_atomic(asm, stem):
    """`atomicAdd` went through the compiler's atomic optimizer, which once turned the claim into a loop that never ended;
    the kernels issue ds_add_rtn_u32 themselves.  One claim site, no lane-scan loop in front of it."""
    body, _ = kernel(asm, stem)
    lines = body.splitlines()
    sites = [i for i, ln in enumerate(lines) if re.match(r"\s+ds_add_rtn_u32\b", ln)]
    assert len(sites) == 1, f"{stem}: {len(sites)} ds_add_rtn_u32 instructions, expected exactly the work-queue claim"
    window = "\n".join(lines[max(0, sites[0] - 40):sites[0] + 8])
    assert "s_ff1_i32_b64" not in window and "v_readlane_b32" not in window.split("ds_add_rtn_u32")[1], \
        "a lane-scan (atomic optimizer) loop surrounds the work-queue claim"
    # issued by ONE lane: exec is narrowed to lane 0 right in front of the atomic and restored behind it (lds_claim_one), so the claim does
    # not rely on the LDS keeping one wavefront's 64 same-address adds together
    before = [ln.strip() for ln in lines[sites[0] - 2:sites[0]]]
    after = [ln.strip() for ln in lines[sites[0] + 1:sites[0] + 4]]
    assert re.match(r"s_mov_b64 exec, 1\b", before[-1]) and re.match(r"s_mov_b64 s\[\d+:\d+\], exec", before[-2]), before
    assert any(re.match(r"s_mov_b64 exec, s\[\d+:\d+\]", ln) for ln in after), after


@pytest.mark.parametrize("stem,size_of", [("subleq16_search_kernel", None), ("subleq16_selfplay_kernel", None)])
def test_argument_struct_sits_at_kernarg_offset_zero(asm, stem, size_of):
    """sq_args() / qp_args() reinterpret the kernarg segment pointer as the argument struct: the struct must be the only explicit parameter."""
    _, meta = kernel(asm, stem)
    assert len(meta["explicit_args"]) == 1, meta["explicit_args"]
    off, size, kind = meta["explicit_args"][0]
    assert off == 0 and kind == "by_value" and size > 256


def test_fc16p_body_barrier_count_matches_kfpbarriers(asm):
    """search_subleq.hpp's idle head group issues kFpBarriers barriers while the other runs fc16p_body: the two must agree or the workgroup
    deadlocks.  fc16p_forward_kernel is fc16p_body and nothing else."""
    src = open(os.path.join(CSRC, "net_fc_pipe.hpp")).read()
    k = int(re.search(r"constexpr int kFpBarriers = (\d+);", src).group(1))
    body, _ = kernel(asm, "fc16p_forward_kernel")
    assert count(body, "s_barrier") == k
    for f in ("search_subleq.hpp",):
        txt = open(os.path.join(CSRC, f)).read()
        assert "__syncthreads(); __syncthreads(); __syncthreads();" not in txt, f"{f} hard-codes a barrier count: use kFpBarriers"
        assert "k < kFpBarriers" in txt


def test_selfplay_kernel_roles_issue_the_same_number_of_barriers():
    """subleq16_selfplay_kernel: inside the forward phase waves 0-3 (MFMA) and 4-7 (helpers) run different code between the same barriers:
    one per head (H1 tile) in both branches, counted in the source (a mismatch is a deadlock on the GPU)."""
    txt = open(os.path.join(CSRC, "search_subleq_persist.hpp")).read()
    f0 = txt.index("// ================= waves 0-3")
    f1 = txt.index("// ================= waves 4-7")
    f2 = txt.index("EAZ_QPSTAMP(7);")
    mfma, helper = txt[f0:f1], txt[f1:f2]
    # the MFMA branch's barrier sits inside `for (int h = 0; h < 3; ++h)`
    assert mfma.count("__syncthreads();") == 1 and "for (int h = 0; h < 3; ++h)" in mfma
    assert helper.count("__syncthreads();") == 3


def test_layer_loops_carry_no_avoidable_vector_alu_work(asm):
    """A VALU instruction of the same wavefront between the MFMAs of a layer loop costs the matrix pipe ~18 clocks: the Subleq
    forward converts its layer-1 operands two at a time from FP8 bytes (one v_cvt_pk_f32_fp8 per two k-steps, none of the v_cvt_f32_ubyte
    that a bit / byte extraction per k-step compiles to), the ResNet's 3x3 convolutions fetch their weights without 64-bit address arithmetic
    (no global_load_dwordx4 left beside the MFMAs of the tower: what remains are the first convolution's and the heads' loads), and the
    DeepSea kernel is a 512-thread workgroup (one helper wavefront per SIMD)."""
    body, _ = kernel(asm, "subleq16_selfplay_kernel")
    assert count(body, "v_cvt_pk_f32_fp8_e32") + count(body, "v_cvt_pk_f32_fp8_sdwa") + count(body, "v_cvt_pk_f32_fp8") >= 24
    body, _ = kernel(asm, "resnet_forward_kernel")
    assert count(body, "buffer_load_dwordx4") >= 12 and count(body, "global_load_dwordx4") <= 48
    _, meta = kernel(asm, "search_persist_kernel")
    assert meta.get("max_flat_workgroup_size", 512) == 512
