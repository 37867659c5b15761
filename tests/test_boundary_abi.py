"""The drop-in boundary without a GPU: libwcsph_hip.so loads, exports every symbol include/wcsph.h
declares, the ctypes mirror of wcsph_params matches the C layout, and the product path refuses to run
without the HIP library / a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "wcsph.h")).read()
    return sorted(set(re.findall(r"\b(wcsph_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from wcsph_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.ALL_SYMBOLS) == names


def test_params_layout_matches_c(tmp_path):
    from wcsph_amd import _lib
    fields = [f[0] for f in _lib.Params._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "wcsph.h"\nint main(){printf("%zu", sizeof(wcsph_params));\n'
    for f in fields:
        prog += 'printf(" %%zu", offsetof(wcsph_params, %s));\n' % f
    prog += "return 0;}\n"
    c = tmp_path / "p.c"
    c.write_text(prog)
    exe = tmp_path / "p"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == C.sizeof(_lib.Params)
    for f, off in zip(fields, out[1:]):
        assert getattr(_lib.Params, f).offset == int(off), f


def test_enum_ids_match_header():
    from wcsph_amd import _lib
    src = open(os.path.join(ROOT, "include", "wcsph.h")).read()
    body = src[src.index("enum wcsph_field"):]
    body = body[:body.index("};")]
    names = re.findall(r"WCSPH_F_([A-Z_]+)", body)
    names = [n for n in names if n != "_COUNT"]
    want = ["POS", "VEL", "VEL_GUESS", "D_VEL", "OMEGA", "D_OMEGA", "NORMAL", "CG_R", "CG_DIR", "CG_AD", "CG_S", "RHO",
            "ADV_RHO", "PRESSURE", "VEL_MAX", "ALPHA", "KAPPA", "KAPPA_V", "CG_MINV", "NEIGHBOR_COUNT", "KBUF", "D_VEL_PRE",
            "POS_STAR", "VEL_STAR", "D_II", "DIJ_PJ", "A_II", "PRESSURE_PRE", "COLOR", "COLOR_GRAD", "POS_AVR", "G"]
    assert names == want and len(_lib.FIELDS) == len(want)


def test_no_cpu_fallback_without_library(monkeypatch, tmp_path):
    from wcsph_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_fields_refuse_to_work_before_setup():
    from wcsph_amd import ParticleData
    pd = ParticleData(0.025)
    pd.add_liquid_point([0.0, 0.0, 0.0])
    with pytest.raises(RuntimeError):
        pd.rho.to_numpy()
    with pytest.raises(RuntimeError):
        pd.hash_grid.update_grid()
    with pytest.raises(ValueError):
        pd.add_solid_point([1, 1, 1]); pd.add_liquid_point([0, 0, 1])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_create_fails_loudly_without_gpu():
    from wcsph_amd import ParticleData
    from wcsph_amd._lib import WcsphError
    pd = ParticleData(0.025)
    pd.add_liquid_points(np.zeros((4, 3)))
    with pytest.raises(WcsphError):
        pd.setup_data_gpu()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under wcsph_amd/ may reference it."""
    pkg = os.path.join(ROOT, "wcsph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower().replace("# oracle-free", ""), os.path.join(dirpath, f)
