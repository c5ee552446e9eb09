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
