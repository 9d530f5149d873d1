"""The C ABI loads and exports exactly what include/lsf.h declares (no compute without a GPU)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "lsf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lsf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from levelsetfortran_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names  # the Python binding covers the whole header, nothing more
    assert lib.lsf_version() == 106


def test_no_cpu_fallback_without_device():
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import _lib

    lib = _lib.load()
    if lib.lsf_device_count() > 0:
        pytest.skip("a GPU is present")
    phi = np.ones((6, 6, 6), order="F")
    with pytest.raises(lsf.LsfError) as e:
        lsf.reinit(phi, None, None, 5, 5, 5, 0, 0.1, 0.01)
    assert e.value.code == _lib.LSF_ERR_NO_DEVICE
    assert np.all(phi == 1.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "levelsetfortran_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".f90", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in txt and "liblsf_oracle" not in txt and "lsf_oracle.h" not in txt, f


def test_argument_validation_happens_before_the_library():
    import levelsetfortran_amd as lsf

    with pytest.raises(ValueError):
        lsf.reinit(np.ones((6, 6, 5), order="F"), None, None, 5, 5, 5, 0, 0.1, 0.01)
    with pytest.raises(ValueError):
        lsf.reinit(np.ones((6, 6, 6), order="C")[::1, :, ::-1], None, None, 5, 5, 5, 0, 0.1, 0.01)
    with pytest.raises(TypeError):
        lsf.reinit(np.ones((6, 6, 6), dtype=np.float16, order="F"), None, None, 5, 5, 5, 0, 0.1, 0.01)
    with pytest.raises(TypeError):
        lsf.narrowBand(5, 5, 5, 0.1, np.ones((6, 6, 6), order="F"), np.zeros((6, 6, 6), order="F"),
                       np.zeros((6, 6, 6), dtype=np.int32, order="F"))


def test_sweep_report_lines_follow_the_reference_protocol():
    from levelsetfortran_amd import SweepReport

    r = SweepReport(3, [0.5, 0.25, 1e-9], True)
    lines = r.lines(0, "steady")
    assert len(lines) == 3 and lines[-1] == "steady" and "Iteration:  1" in lines[1]
    r = SweepReport(2, [0.5, 0.25], False)
    assert len(r.lines(1, "steady")) == 2 and "Iteration:  2" in r.lines(1, "steady")[1]


def _product_text(src):
    """The text of a source file as the PRODUCT build sees it: what sits between `#ifdef LSF_EXPERIMENTS` (or
    `#if defined(LSF_EXPERIMENTS) ...`) and its `#else` / `#endif` is dropped, what sits behind `#ifndef LSF_EXPERIMENTS` kept."""
    out, stack = [], []  # stack entries: [is an LSF_EXPERIMENTS conditional, currently in its product branch]
    for line in src.splitlines():
        t = line.strip()
        if re.match(r"#\s*if", t):
            if re.match(r"#\s*ifdef\s+LSF_EXPERIMENTS\b", t) or re.match(r"#\s*if\s+defined\s*\(?\s*LSF_EXPERIMENTS\b", t):
                stack.append([True, False])
            elif re.match(r"#\s*ifndef\s+LSF_EXPERIMENTS\b", t) or re.match(r"#\s*if\s+!\s*defined\s*\(?\s*LSF_EXPERIMENTS\b", t):
                stack.append([True, True])
            else:
                stack.append([False, True])
        elif re.match(r"#\s*else", t) and stack and stack[-1][0]:
            stack[-1][1] = not stack[-1][1]
        elif re.match(r"#\s*endif", t) and stack:
            stack.pop()
        elif all(keep for _, keep in stack):
            out.append(line)
    assert not stack, "unbalanced conditionals"
    return "\n".join(out)


def test_product_text_filter():
    src = "a\n#ifdef LSF_EXPERIMENTS\nb\n#if X\nc\n#endif\n#else\nd\n#endif\n#ifndef LSF_EXPERIMENTS\ne\n#endif\n#if Y\nf\n#endif"
    assert _product_text(src).split() == ["a", "d", "e", "f"]


def _csrc_files():
    import glob

    return sorted(glob.glob(os.path.join(ROOT, "levelsetfortran_amd", "csrc", "*.h*")))


def test_every_environment_switch_of_the_library_is_documented():
    """every getenv("LSF_...") of the PRODUCT build of levelsetfortran_amd/csrc has its row in INTEGRATION.md
    (switches that exist in the experiment builds of profiles/micro only -- code behind LSF_EXPERIMENTS -- are theirs to document)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set()
    for f in _csrc_files():
        names |= set(re.findall(r'getenv\("(LSF_[A-Z0-9_]+)"\)', _product_text(open(f).read())))
    assert len(names) > 20
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_experiment_hooks_stay_out_of_the_product_build():
    """probe fields, probe switches and the marching-step hook exist behind LSF_EXPERIMENTS only."""
    for f in _csrc_files():
        txt = _product_text(open(f).read())
        assert not re.search(r'getenv\("LSF_PROBE_', txt), f
        assert "probe_us" not in txt and "probe_early" not in txt and "MidHook" not in txt, f
    # ... and the product library is not an experiment build
    mk = open(os.path.join(ROOT, "levelsetfortran_amd", "csrc", "Makefile")).read()
    assert re.search(r"^EXTRA\s*\?=\s*$", mk, flags=re.M), "the default build passes no -D switches"
