"""Register budget of the pipelined cross-view attention kernel (attention_v3.hip), checked at compile time — no GPU needed.

The kernel's schedule assumes two waves per SIMD with nothing in scratch: a change that pushes it over 256 VGPRs still compiles
and still passes parity, but runs ~20 % slower (measured: 1240 -> 1030 TF/s at 32 views when a variant spilled 108 registers).
hipcc's resource remarks are the check.
"""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "hunyuanworld-mirror_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_attention_v3_fits_two_waves_per_simd_without_scratch(tmp_path):
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = line.split("=", 1)[1].split()
    assert flags, "CXXFLAGS not found in the Makefile"
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags if not f.startswith("$(")]
    assert "--offload-arch=gfx950" in flags
    cmd = ["hipcc", *flags, "-x", "hip", "-c", os.path.join(CSRC, "attention_v3.hip"), "-o", str(tmp_path / "a.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # remarks come as blocks: "Function Name: X" followed by that function's figures
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = {}
    for b in blocks:
        name = b.split()[0]
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        seen[name] = {"vgpr": get(r"VGPRs"), "scratch": get(r"ScratchSize \[bytes/lane\]"), "occ": get(r"Occupancy \[waves/SIMD\]"),
                      "spill": get(r"VGPRs Spill")}
    main = [v for k, v in seen.items() if "attn_v3_kernelILi2ELb0E" in k]
    assert len(main) == 1, list(seen)
    m = main[0]
    assert m["scratch"] == 0 and m["spill"] == 0 and m["occ"] >= 2 and m["vgpr"] <= 256, m
    for k, v in seen.items():       # no instantiation may use scratch
        assert v["scratch"] == 0 and v["spill"] == 0, (k, v)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_attention_v4_register_file_split_and_clean_loop(tmp_path):
    """attn_v4_kernel (attention_v4.hip) owns a SIMD: O and Q in 192 AGPRs, everything the VALU touches in arch VGPRs, nothing in
    scratch.  Scratch would not only be slow: spill loads / stores count in vmcnt and would break the kernel's hand-counted
    `s_waitcnt vmcnt(4)` for its LDS-DMA pieces (seen: a spilling f16 variant read K / V tiles before they had landed and flagged
    every unit).  The loop must hold no v_accvgpr_* copy (a first form through MFMA builtins had 44-80 per iteration) and exactly
    the instruction mix the schedule is built on: per 64 MFMAs 128 v_exp, 128 v_add, 64 packs, 24 LDS fragment reads."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = [f.replace("$(ARCH)", "gfx950") for f in line.split("=", 1)[1].split() if not f.startswith("$(")]
    asm = tmp_path / "v4.s"
    r = subprocess.run(["hipcc", *flags, "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "attention_v4.hip"), "-o", str(asm),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    n = 0
    for b in blocks:
        name = b.split()[0]
        if "attn_v4_kernel" not in name:
            continue
        n += 1
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        assert get(r"ScratchSize \[bytes/lane\]") == 0 and get(r"VGPRs Spill") == 0 and get(r"SGPRs Spill") == 0, b[:400]
        assert get(r"AGPRs") == 192 and get(r"VGPRs") <= 256, b[:400]
    assert n == 2
    text = open(asm).read()
    # (the body runs to the function's end label: since round 4 the kernel has an early s_endpgm — a block whose sticky hint is set exits at entry)
    kernels = re.findall(r"^(_ZN\S*attn_v4_kernel\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) == 2
    for name, body in kernels:
        lines = body.split("\n")
        marks = [i for i, l in enumerate(lines) if "Inner Loop Header" in l or "in Loop: Header" in l]
        assert marks, name
        lo, hi = min(marks), max(marks)
        while hi + 1 < len(lines) and not re.match(r"^\.LBB", lines[hi + 1]):   # to the end of the loop's last block
            hi += 1
        loop = lines[lo:hi + 1]              # the tile loop: three tiles per iteration (unrolled over the K / V rings), 64 MFMA gaps each
        count = lambda pat: sum(1 for l in loop if re.search(pat, l))
        assert count(r"v_mfma_f32_32x32x16") == 192, (name, count(r"v_mfma_f32_32x32x16"))
        assert count(r"v_accvgpr") == 0 and count(r"scratch_") == 0, name
        assert count(r"v_exp_f32") == 384 and count(r"v_add_f32") == 384 and count(r"v_cvt_pk_") == 192, name
        assert count(r"ds_read") == 72, (name, count(r"ds_read"))
        assert count(r"s_nop") <= 64, (name, count(r"s_nop"))          # 16 per tile on the fast path + the rare ragged-tile DMA path
        assert count(r"v_mov_b") <= 3, (name, "a register copy in front of a statement would need wait states nobody inserts")
        assert count(r"s_barrier") == 3
        # M0 is written without save / restore by the tile DMA statement: nothing else in the kernel may use it
        assert not [l for l in lines if re.search(r"\bm0\b", l) and not re.search(r"s_mov_b32 (m0, |s\d+, m0)", l)], name
        if "ILi1E" in name:   # f16: the prologue takes the first tile's row maxima from asm MFMA results behind an s_nop fence.  The
            # fence must NAME the score tiles: a "memory"-only fence let the compiler schedule the v_max of q-block 0 right behind its
            # MFMA chain (stale reads on ~1 % of the rows; caught by tests/test_gpu_ops.py test_attention_v4_one_wave_per_simd)
            first_mfma = next(i for i, l in enumerate(lines) if "v_mfma" in l)
            fence = next(i for i, l in enumerate(lines) if "s_nop 15" in l)
            assert fence > first_mfma
            assert not any(re.search(r"v_max", l) for l in lines[first_mfma:fence]), name


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_packed_fp32_instruction_outside_the_gelu_gemm(tmp_path):
    """With two or more HIP queues active a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) can lose a
    half result (profiles/r02_multiqueue_hazard.md).  The forward uses extra queues in two places — the DPT heads + camera head
    beside each other, and the sharded K/V all-gather under the local attention (wm_model.cpp) — so no kernel that can run there may
    hold one.  The build's -fno-slp-vectorize removes the compiler-formed ones and WM_NO_PACKED_FP32 (wm_common.h) the rest where
    vector-typed source still packs; this test is what holds the line: every object is compiled to assembly and scanned per kernel,
    and only the EPI = 2 (GELU) instantiations of gemm.hip — fc1 of the single-queue backbone — may contain packed fp32."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = [f.replace("$(ARCH)", "gfx950") for f in line.split("=", 1)[1].split() if not f.startswith("$(")]
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    assert len(srcs) >= 12
    for src in srcs:
        asm = tmp_path / (src + ".s")
        r = subprocess.run(["hipcc", *flags, "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(asm)],
                           capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, r.stderr[-2000:]
        cur, hits = None, {}
        for line in open(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
            elif cur and re.search(r"v_pk_(?:mul|fma|add)_f32", line):
                hits[cur] = hits.get(cur, 0) + 1
        if src == "gemm.hip":
            # mangled template args: gemm_*_kernelILi<T>ELi<EPI>E...: EPI == 2 is WM_EPI_GELU_T16
            bad = {k: v for k, v in hits.items() if not re.search(r"gemm_\w+_kernelILi\dELi2E", k)}
            assert hits, "the packed GELU epilogue should be there (is this still a device listing?)"
        else:
            bad = hits
        assert not bad, (src, bad)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_unpacked_gelu_build_has_no_packed_fp32_at_all(tmp_path):
    """-DWM_UNPACKED_GELU (wm_common.h) is the build for deployments that put several handles' kernels beside each other on one
    device (wm_share_weights from several threads): with it no kernel of gemm.hip — the one file allowed packed fp32 — holds any."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = [f.replace("$(ARCH)", "gfx950") for f in line.split("=", 1)[1].split() if not f.startswith("$(")]
    asm = tmp_path / "gemm_unpacked.s"
    r = subprocess.run(["hipcc", *flags, "-DWM_UNPACKED_GELU", "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "gemm.hip"), "-o", str(asm)],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    assert not re.search(r"v_pk_(?:mul|fma|add)_f32", open(asm).read())


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_conv3x3_kernels_do_not_spill(tmp_path):
    """The 256-pixel x 256-channel conv tile keeps 128 accumulator registers per lane and prefetches its residual inputs in the
    epilogue: one float4 array too many and hipcc spills (a first form of the LDS-staged epilogue spilled 68 registers)."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = [f.replace("$(ARCH)", "gfx950") for f in line.split("=", 1)[1].split() if not f.startswith("$(")]
    r = subprocess.run(["hipcc", *flags, "-x", "hip", "-c", os.path.join(CSRC, "conv3x3.hip"), "-o", str(tmp_path / "c.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    n = 0
    for b in blocks:
        name = b.split()[0]
        if "conv3x3_rs_kernel" not in name:
            continue
        n += 1
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, name
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, name
    assert n >= 8


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_raster_compositing_streams_records_through_scalar_loads(tmp_path):
    """The compositing pass (raster.hip) is written for 64-wide waves: the Gaussian of a step is wave-uniform, fetched by scalar loads
    into two alternating SGPR sets whose requests are asm statements (the compiler sinks plain loads next to their use).  What that
    needs from the ISA and nothing else would catch: the records really arrive by s_load (not per-lane loads + readfirstlane), there is
    no LDS traffic and no barrier, a set is settled (s_waitcnt) between its request and its first use, and nothing copies a set's
    registers while a load may still own them."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = [f.replace("$(ARCH)", "gfx950") for f in line.split("=", 1)[1].split() if not f.startswith("$(")]
    asm = tmp_path / "raster.s"
    r = subprocess.run(["hipcc", *flags, "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "raster.hip"), "-o", str(asm),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
        if "raster_composite_kernel" in b.split()[0]:
            get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
            assert get(r"ScratchSize \[bytes/lane\]") == 0 and get(r"VGPRs") <= 64 and get(r"Occupancy \[waves/SIMD\]") == 8, b[:300]
    kernels = re.findall(r"^(_ZN\S*raster_composite_kernel\S*):[^\n]*\n(.*?)^\.Lfunc_end", open(asm).read(), flags=re.S | re.M)
    assert len(kernels) == 3      # 1, 2 and 4 pixels per lane
    for name, body in kernels:
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        assert not [l for l in lines if l.startswith("ds_") or l.startswith("s_barrier")], name
        marks = [i for i, l in enumerate(body.split("\n")) if "Loop Header" in l or "in Loop: Header" in l]
        loop = [l.strip() for l in body.split("\n")[min(marks):max(marks) + 400] if l.strip() and not l.strip().startswith(";")]
        x8 = [i for i, l in enumerate(loop) if l.startswith("s_load_dwordx8")]
        assert len(x8) >= 2, (name, "two register sets, each requested in the loop")
        assert not [l for l in loop if l.startswith("global_load") or l.startswith("buffer_load") or l.startswith("flat_load")], name
        for i in x8:   # from a request to the next s_waitcnt lgkmcnt(0): no instruction names a register of the requested tuple
            lo, hi = [int(x) for x in re.search(r"s\[(\d+):(\d+)\]", loop[i]).groups()]
            j = i + 1
            while j < len(loop) and not loop[j].startswith("s_waitcnt lgkmcnt(0)"):
                if not loop[j].startswith("s_load_"):
                    regs = [int(x) for x in re.findall(r"\bs(\d+)\b", loop[j])]
                    for a, b_ in re.findall(r"s\[(\d+):(\d+)\]", loop[j]):
                        regs += list(range(int(a), int(b_) + 1))
                    assert not [x for x in regs if lo <= x <= hi], (name, loop[i], loop[j])
                j += 1
            assert j < len(loop), (name, "no settle behind a request")
