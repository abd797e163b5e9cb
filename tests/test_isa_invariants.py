"""What the hand-scheduled kernels rely on from hipcc, checked on the gfx950 assembly (no GPU needed; ADVICE r1: "check the ISA in
CI").  Each of these was a measured regression at some point: a spilled register, a compiler-inserted
`s_waitcnt vmcnt(0)` that drains a copy ring inside a K loop, a counted wait that silently became a full drain."""
import os
import re
import subprocess

import pytest

from sfd2_amd import build

CSRC = build.CSRC


def _device_asm(tmp_path_factory, src):
    if not build.have_hipcc():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)] + build.SRC_FLAGS.get(src, [])
    subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, src)])
    return open(out).read()


def _kernels(asm):
    """{mangled name: {'body': [lines], 'meta': {...}}} for every kernel of one translation unit"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        out[m.group(1)] = {"body": m.group(2).split("\n")}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)\.wavefront_size", asm, re.S):
        if m.group(1) in out:
            meta = dict(re.findall(r"\.(\w+):\s+(\d+)", m.group(2)))
            out[m.group(1)]["meta"] = {k: int(v) for k, v in meta.items()}
    return out


def _mfma_span(body):
    idx = [i for i, l in enumerate(body) if "v_mfma" in l]
    return body[idx[0]:idx[-1] + 1] if idx else []


def _count(lines, pattern):
    return sum(1 for l in lines if re.search(pattern, l))


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    cache = {}

    def get(src):
        if src not in cache:
            cache[src] = _kernels(_device_asm(tmp_path_factory, src))
        return cache[src]
    return get


@pytest.mark.parametrize("src", ["conv3_kernels.hip", "conv3rf_kernels.hip", "resblock_kernel.hip", "conv1x1_kernels.hip",
                                 "match_mutual_kernel.hip", "fused_stem_kernel.hip", "conv2_kernels.hip", "fused_stem_c_kernel.hip", "rb23_c_kernel.hip",
                                 "conv2b_s2d_kernel.hip"])
def test_no_spills_and_two_waves_per_simd(asm, src):
    ks = asm(src)
    assert ks, src
    for name, k in ks.items():
        meta = k.get("meta")
        assert meta is not None, name
        assert meta["vgpr_count"] <= 256, (name, meta["vgpr_count"])      # two waves per SIMD
        assert _count(_mfma_span(k["body"]), r"\bscratch_") == 0, name    # never a spill inside a K loop
        # conv2a's kernel (64 -> 128 channels, 128 registers for four waves per SIMD) parks one address pair across its
        # loop: one store in the prologue, one reload in the epilogue
        allowed = 2 if "conv_igemm2_kernelILi3ELi1ELi128ELi64E" in name else 0
        assert meta["vgpr_spill_count"] <= allowed, (name, meta)
        # scalar spills go to VGPR lanes (v_writelane / v_readlane); the persistent conv3x3_pp parks a few tile-loop scalars
        # that way, outside the K loop
        # (its compensated instantiations, COMP & 1, have two chunk loops and park a few more between them; the generic
        # compensated kernel and the compensated fused stem keep their tile geometry that way too)
        pp_comp = "conv3x3_pp_kernel" in name and not name.split("EEv")[0].endswith(("ELi0", "ELi2"))
        # (round 4: the range-status pointer and the running maximum are two more tile-loop scalars: 29 -> 33 parked, all outside the K loops --
        #  the assertion below; same-box A/B against a -DSFD2_NO_RANGE build in profiles/r04_range_cost.txt)
        lim = 36 if pp_comp else ((8 if name.split("EEv")[0].endswith("ELi0") else 20) if "conv3x3_pp_kernel" in name else (48 if ("convc_igemm" in name or "fused_stem_c" in name or "gconv_c" in name or "rb23_c" in name or "conv2b_s2d" in name) else 0))
        assert meta["sgpr_spill_count"] <= lim, (name, meta)
        if "convc_igemm" in name or "fused_stem_c" in name or "gconv_c" in name or "conv1a_c" in name or "rb23_c" in name or "conv2b_s2d" in name:
            continue      # (step-list / tile-loop scalars parked in VGPR lanes)
        assert _count(_mfma_span(k["body"]), r"v_readlane|v_writelane") <= (4 if pp_comp else 0), name


def test_conv3x3_pp_loop_has_only_its_own_drains(asm):
    ks = {n: k for n, k in asm("conv3_kernels.hip").items() if "conv3x3_pp_kernel" in n}
    assert ks
    for name, k in ks.items():
        span = _mfma_span(k["body"])
        comp = int(re.search(r"ELi(\d+)EEv", name).group(1))
        if comp & 4:                                                      # f16x3 on planes: the fp16 chunk body twice (second copy: the cross terms)
            assert _count(span, r"v_mfma_f32_32x32x16_f16") == 288 and _count(span, r"v_mfma_scale") == 0, name
            assert _count(span, r"\bscratch_") == 0, name
            continue
        if comp & 1:                                                      # COMP & 1: a second chunk loop on the fp8 MFMA
            assert _count(span, r"v_mfma_f32_32x32x16_f16") == 144 and _count(span, r"v_mfma_scale_f32_32x32x64_f8f6f4") == 72
            # every unit's eight scaled MFMAs sit in their own MFMA section (between the unit's two barriers): instruction
            # selection once sank all 72 below the chunk's last barrier, with the fragments of nine units live
            idx = [i for i, l in enumerate(span) if "v_mfma_scale" in l]
            runs = 1 + sum(1 for a, b in zip(idx, idx[1:]) if any("s_barrier" in l for l in span[a:b]))
            assert runs == 9, (name, runs)
            assert _count(span, r"s_waitcnt.*vmcnt\(0\)") == 11, name
            assert _count(span, r"ds_read_b128") in (60 + 60, 60 + 72)   # the fp8 loop reads like the fp16 loop (row reuse; its first unit's LOAD section lies inside the span)
            continue
        assert _count(span, r"v_mfma_f32_32x32x16_f16") == 144          # 9 units x 16
        # (the span runs from the first to the last MFMA of the unrolled chunk body: the first unit's LOAD section and the
        # last unit's closing wait lie outside it)
        # the two written-out vmcnt(0) of each stage's last unit (3 stages), nothing from hipcc
        assert _count(span, r"s_waitcnt.*vmcnt\(0\)") == 5, name
        assert _count(span, r"ds_read_b128") == 72 - 12                  # 8 fragment reads per unit (filter-column order)
        assert _count(span, r"s_barrier") == 16


def test_conv3x3_rf_loop_keeps_counted_waits(asm):
    ks = {n: k for n, k in asm("conv3rf_kernels.hip").items() if "conv3x3_rf_kernel" in n}
    assert len(ks) == 12                     # stride 1, stride 2, stride 2 with 128-channel blocks, conv2a's resident filters, conv2b compensated (fp8 / fp6 records out),
                                             # f16x3 (COMP = 4 planes out / 12 fp32 out) at stride 1, 2 and 2 with 128-channel blocks: the plain pipeline run three times
    for name, k in ks.items():
        span = _mfma_span(k["body"])
        if name.split("EEv")[0].endswith(("ELi3", "ELi67")):             # COMP = 3 (| 64: fp6 output records): one chunk body, fp16 and fp8 MFMAs behind a wave-uniform flag
            assert _count(span, r"v_mfma_f32_32x32x16_f16") == 36 and _count(span, r"v_mfma_scale_f32_32x32x64_f8f6f4") == 18
            assert _count(span, r"s_waitcnt.*vmcnt\(0\)") == 0, name    # the counted waits survive (no spill reload, no drain)
            assert _count(span, r"\bscratch_") == 0, name
            assert _count(span, r"global_load_dwordx4") == 16 and _count(k["body"], r"global_load_lds") == 0
            assert _count(span, r"s_barrier") == 1
            continue
        if "ELb1E" in name:                                              # resident filters: two unrolled chunks, no loads in the loop
            assert _count(span, r"v_mfma_f32_32x32x16_f16") == 72 and _count(span, r"global_load_dwordx4") == 0
            assert _count(span, r"s_waitcnt.*vmcnt\(0\)") == 0 and _count(span, r"s_barrier") == 2
            continue
        assert _count(span, r"v_mfma_f32_32x32x16_f16") == (36 if "ILi2ELi128E" in name else 72)   # 9 units x 8 (x 4)
        # no full drain inside the chunk loop: filter loads and patch copies are waited for by count
        assert _count(span, r"s_waitcnt.*vmcnt\(0\)") == 0, name
        assert _count(span, r"global_load_dwordx4") == 16                # the filter ring: 2 loads per unit (the last unit's follow its MFMAs)
        assert _count(span, r"buffer_load_dwordx4 .* lds") in (2, 5)     # one patch piece per unit (stride 1 / 2; the first precedes the span)
        assert _count(k["body"], r"global_load_lds") == 0                # (hipcc drains behind the FLAT form)
        assert _count(span, r"s_barrier") == 1


def test_resblock_row_loop_drains(asm):
    ks = {n: k for n, k in asm("resblock_kernel.hip").items() if "resblock_kernel" in n}
    assert ks
    for name, k in ks.items():
        span = _mfma_span(k["body"])
        # typed LDS reads (sfd2_lds_f4 / rb_lds4): no compiler vmcnt(0) in front of the scale / shift reads; what is left are
        # the written-out waits of the row pipeline
        assert _count(span, r"s_waitcnt.*vmcnt\(0\)") <= 3, (name, _count(span, r"s_waitcnt.*vmcnt\(0\)"))


def test_conv1x1_ring_not_drained_in_epilogue(asm):
    ks = {n: k for n, k in asm("conv1x1_kernels.hip").items() if "conv1x1_c256_kernel" in n or "conv1x1_c256_c_kernel" in n}
    assert len(ks) == 7       # fp16: +-residual; compensated: +-residual, plain output (conv1; unit or residual-only input), plain input + residual (conv3)
    for name, k in ks.items():
        span = _mfma_span(k["body"])
        # the residual variants wait for their residual loads, the youngest operations in flight
        limit = 5 if "ILb1E" in name else 0
        assert _count(span, r"s_waitcnt.*vmcnt\(0\)") <= limit, (name, _count(span, r"s_waitcnt.*vmcnt\(0\)"))


def test_matcher_valu_budget(asm):
    ks = {n: k for n, k in asm("match_mutual_kernel.hip").items() if "match_mutual_kernel" in n}
    assert len(ks) == 2                                  # <FWD_IDS = true / false>
    for name, k in ks.items():
        fwd_ids = "ILb1E" in name
        mfma = _count(k["body"], r"v_mfma")
        maxes = _count(k["body"], r"v_max3?_f32")
        packs = _count(k["body"], r"v_and_or_b32")
        # -fno-honor-nans: no canonicalising v_max x, x in front of the packed maxima (it doubled the VALU count).
        # Two stages of two tiles (64 MFMAs): reverse 4 x (32 v_and_or + 16 v_max3), forward 2 x 32 v_max3
        # (+ 2 x 64 v_and_or with the tile ids), a few more in the merges and the final reduction
        assert mfma == 64 and maxes <= 170, (name, mfma, maxes)
        assert packs <= (270 if fwd_ids else 135), (name, packs)
        assert _count(k["body"], r"v_max_f32_e32 (v\d+), \1, \1") == 0


def test_sparse_da3_fragments_stay_in_flight(asm):
    """Round 5: the kernel was 41 us with its filter fragments loaded and waited for tap by tap; the prefetch ring is only worth anything while
    the waits between the MFMAs are COUNTED (a `vmcnt(0)` there is the old tap-by-tap round trip again)."""
    ks = {n: k for n, k in asm("sparse_da3_kernel.hip").items() if "sparse_da3_kernel" in n}
    assert len(ks) == 4                                  # <plain | x3> x <common packing | repacked filters>
    for name, k in ks.items():
        meta = k["meta"]
        x3 = "ILb1E" in name
        assert meta["vgpr_count"] <= 256, (name, meta)   # two blocks per CU
        assert meta["vgpr_spill_count"] <= (8 if x3 else 0), (name, meta)   # (x3 parks seven address registers: measured faster than no ring)
        span = _mfma_span(k["body"])
        assert _count(span, r"v_mfma") == 36 * 8 * (3 if x3 else 1), name
        drains = _count(span, r"s_waitcnt vmcnt\(0\)")
        # per chunk the written drain in front of its barrier (the patch copies) and the compiler's own beside it; x3's spilled reloads add a few; never one per tap (36)
        assert drains <= (16 if x3 else 8), (name, drains)
        assert _count(span, r"s_waitcnt vmcnt\((?:[4-9]|\d\d)\)") >= 30, name      # the counted waits of the ring


def test_pmc_families_name_the_headline_kernels(asm):
    """tools/pmc_to_json.py picks kernels by regex on their mangled names; a renamed template parameter silently drops a family from
    profiles/pmc_traffic.json (it happened twice in round 4: bench.py's roofline.traffic went null).  Every family of the default f16c path
    must match a kernel that the sources actually instantiate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_to_json", os.path.join(os.path.dirname(CSRC), "..", "tools", "pmc_to_json.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fam = {label: rx for label, rx, _ in mod.FAMILIES}
    where = {"conv3x3_pp<comp>": "conv3_kernels.hip", "conv3x3_pp": "conv3_kernels.hip", "conv1x1_c256<comp,plain out>": "conv1x1_kernels.hip",
             "rb23_c_kernel": "rb23_c_kernel.hip", "conv2b_s2d_kernel": "conv2b_s2d_kernel.hip", "fused_stem_c_kernel": "fused_stem_c_kernel.hip",
             "match_mutual_kernel": "match_mutual_kernel.hip"}
    for label, src in where.items():
        names = list(asm(src))
        assert any(re.search(fam[label], n) for n in names), (label, fam[label])
    # the instantiations the default path launches for the dominant family: conv2a (fp6 in, s2d out), conv3a (fp6 in / out), conv3b (fp6 in, bytes out)
    pp = list(asm("conv3_kernels.hip"))
    for comp in (179, 115, 307):
        hit = [n for n in pp if f"ELi{comp}EEv" in n]
        assert hit and re.search(fam["conv3x3_pp<comp>"], hit[0]), comp
