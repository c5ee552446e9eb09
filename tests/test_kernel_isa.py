"""CPU: properties of the compiled generator-stationary sweep (csrc/dto_sweep_gs.hip) that its design rests on and that only the
ISA shows.  The kernel keeps 32 rows of every generator in REGISTERS for the whole launch (up to 320 per lane) and its MFMA loop must
touch neither global memory nor scratch for them: a compiler that re-loads the fragments instead of holding them (it did, in a build
whose loop it had left partly rolled: 176 global loads per item) turns the kernel back into the generator-streaming form without any
test of values noticing.  hipcc cross-compiles without a GPU; ~20 s."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "directtrajopt.jl_amd", "csrc", "dto_sweep_gs.hip")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "gs.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I", os.path.dirname(SRC), SRC, "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def test_generator_fragments_stay_in_registers(isa):
    names = re.findall(r"^(_ZN3dto12_GLOBAL__N_110k_sweep_gsILi(\d)ELi(\d)ELi(\d)ELb(\d)EEEvNS0_6GsArgsE):", isa, re.M)
    assert len(names) == 16, len(names)                       # KU in {8, 4} x MP in {5, 3} x NT in {1, 2} x with / without source terms
    for nm, ku, mp, nt, src in names:
        ku, mp, nt = int(ku), int(mp), int(nt)
        body = isa[isa.index(nm + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        mf = [k for k, l in enumerate(body) if "v_mfma_f64_16x16x4_f64" in l]
        # the loop is fully unrolled: (generators) x (k-steps per wave: 2 KU) x (two row tiles) x (NT column tiles) MFMAs, once
        assert len(mf) == mp * 2 * ku * 2 * nt, (nm, len(mf))
        inside = body[mf[0]:mf[-1] + 1]
        # between the first and the last MFMA: no global load beyond the collect chunk that rotates there at NT = 2 (its term-0 form
        # reads Z with 16 eight-byte loads; later terms come by buffer loads) ...
        gl = sum(1 for l in inside if re.search(r"\bglobal_load_dword", l))
        assert gl <= (16 if nt == 2 else 0), (nm, gl)
        # ... and no scratch traffic worth the name (a few spilled registers of the largest instance are tolerated, the operand is not)
        sc = sum(1 for l in inside if "scratch_" in l)
        assert sc <= 16, (nm, sc)
        # every stationary fragment reaches the matrix instruction from a register file (VGPR or AGPR operand), 2 KU x MP of them x 2 doubles
        ops = set()
        for k in mf:
            m = re.search(r"v_mfma_f64_16x16x4_f64 \S+ \S+ (\S+),", body[k])
            ops.add(m.group(1))
        assert len(ops) >= 2 * (2 * ku) * min(mp, 5) * 0.9, (nm, len(ops))   # distinct register pairs (zero generators of padded slots may share)


# ---- the 64-state generator-stationary K-split sweep (k_sweep_s64 in csrc/dto_sweep_fused.hip)
SRC_FUSED = os.path.join(ROOT, "directtrajopt.jl_amd", "csrc", "dto_sweep_fused.hip")


@pytest.fixture(scope="module")
def isa_fused(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "fused.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I", os.path.dirname(SRC_FUSED), SRC_FUSED, "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def test_s64_generators_stay_in_registers_and_the_mfma_stream_is_clean(isa_fused):
    """80 (48) stationary generator doubles per lane loaded ONCE, before the Taylor loop; per term the MFMAs come in batches of 32
    (two generators: their B fragments are formed in front of the batch) with no memory instruction and no accumulator shuffling
    between register files inside a batch (a build with a branch per generator moved all 32 accumulator registers VGPR <-> AGPR
    around every block of 16).  The kernel is held to 256 registers (two workgroups per CU hide each other's barriers): what the
    five-generator instances spill are loop-invariant values, stored once before the loop."""
    names = re.findall(r"^(_ZN3dto12_GLOBAL__N_111k_sweep_s64ILi(\d)ELi(\d)EEEvNS0_14FusedSweepArgsE):", isa_fused, re.M)
    assert len(names) == 6, names                              # MP in {5, 3} x NX in {0, 1, 2}
    for nm, mp, nx in names:
        mp = int(mp)
        body = isa_fused[isa_fused.index(nm + ":"):]
        body = body[:body.index(".Lfunc_end")].split("\n")    # (the kernel may hold more than one s_endpgm)
        meta = isa_fused[isa_fused.index(".amdhsa_kernel " + nm):]
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 256, nm      # two waves per SIMD
        mf = [k for k, l in enumerate(body) if "v_mfma_f64_16x16x4_f64" in l]
        assert len(mf) % (16 * mp) == 0 and 1 <= len(mf) // (16 * mp) <= 2, (nm, len(mf))   # (the first term may be peeled)
        first = mf[0]
        # the stationary loads all come before the first MFMA, and so does every spill store ...
        assert sum(1 for l in body[:first] if re.search(r"\bglobal_load_dwordx2", l)) >= 16 * mp, nm
        assert not any("scratch_store" in l for l in body[first:]), nm
        if mp == 3:
            assert not any("scratch_" in l for l in body), nm
        for s in range(0, len(mf), 16 * mp):
            term = mf[s:s + 16 * mp]
            for b in range(0, 16 * mp, 32):
                batch = term[b:b + 32]
                inside = body[batch[0]:batch[-1] + 1]
                # ... and a batch holds nothing that touches global memory or LDS, and no accumulator shuffling
                assert not any(re.search(r"\b(global_|buffer_|ds_|v_accvgpr_write)", l) for l in inside), nm
                assert sum(1 for l in inside if "v_accvgpr_read" in l or "scratch_load" in l) <= 16, nm
