"""BASELINE config 1 through the drop-in executable: the reference's own host (set3d.f90, built where
it lies by levelsetfortran_amd/fortran/Makefile) calling liblsf_hip.so through the iso_c_binding shim.
Skipped when the executable was not built (it needs /root/reference at build time)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "build", "dropin", "set3d_hip.exec")


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in executable not built")
@pytest.mark.parametrize("resident", ["2", "1", "0"])
def test_cube40_as_shipped_through_the_fortran_host(tmp_path, cube40, resident):
    """resident = 2 (default): phi, phiNB, phiSB stay on the device from the inside/outside search to the end, the .vti
    files stream from the device copy (host edits E8, E9); 0: every seam copies in and out, as in round 1."""
    import stl_io

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    stl_io.stl_write(tmp_path / "cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
    env = dict(os.environ, LSF_ARITH="strict", LSF_REINIT2_ITER="0", LSF_RESIDENT=resident)
    p = subprocess.run(f"ulimit -s unlimited; cd {tmp_path}; {EXE} cube40.stl", shell=True, env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout
    assert p.returncode == 0, out[-2000:]
    assert "Grid Size: nx = 61 , ny = 61 ,nz = 61" in out
    assert "Distance function time integration has reached steady state" in out
    assert "Min/max time integration has reached steady state" in out
    its = [int(x) for x in re.findall(r"Iteration:\s+(\d+)", out)]
    # reinit #1 prints 0..2153, min/max prints 1..405, reinit #2 (capped to 1 sweep here) prints 0
    assert its[:2154] == list(range(2154)) and its[2154:2154 + 405] == list(range(1, 406))
    asym = float(re.search(r"Asymptotic Error:\s+(\S+)", out).group(1))
    assert abs(asym - 1.0085048924202963E-02) < 1e-15
    shape = (62, 62, 62)
    assert np.array_equal(stl_io.vti_read_phi(tmp_path / "signedDistanceFunction.vti", shape), cube40["phi_reinit"])
    assert np.array_equal(stl_io.vti_read_phi(tmp_path / "smoothedDistanceFunction.vti", shape), cube40["phi_minmax"])
    assert stl_io.vti_header_count(tmp_path / "smoothedDistanceFunction.vti") == (8 * 62 ** 3, False)  # true byte count
    # the .s3d mesh (set3d.f90:601-612) carries the surface nodes advected on the GPU (advectNodes)
    lines = open(tmp_path / "cube40.s3d").read().split("\n")
    nelem, nnode = (int(v) for v in lines[0].split()[:2])
    nodes = np.array([[float(v) for v in ln.split()] for ln in lines[1 + nelem:1 + nelem + nnode]])
    adv = np.load(os.path.join(GOLDEN, "cube40_advect.npz"))["surfXX"]
    assert nodes.shape == adv.shape == (9140, 3)
    assert np.allclose(nodes, adv, rtol=1e-15, atol=0)  # list-directed output prints 17 significant digits


EXE_SEAMS = os.path.join(ROOT, "build", "dropin", "set3d_hip_seams.exec")


@pytest.mark.skipif(not os.path.exists(EXE_SEAMS), reason="seams-only drop-in executable not built")
def test_cube40_as_shipped_through_the_seams_alone(tmp_path, cube40):
    """SURVEY.md section 8b by itself (`make -C levelsetfortran_amd/fortran seams`): host edits E1-E3 only.  The reference's own
    inside/outside search, gradient and advection loops, diagnostics and VTI writers run as shipped (its INTEGER*4 byte count
    included); reinit, narrowBand and the hoisted min/max loop are the library's, every seam copying its arrays in and out.
    What only these host paths would expose -- a field the reference's loops cannot digest -- shows here and nowhere else."""
    import stl_io

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    stl_io.stl_write(tmp_path / "cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
    env = {k: v for k, v in os.environ.items() if not k.startswith("LSF_")}
    env["LSF_ARITH"] = "strict"
    p = subprocess.run(f"ulimit -s unlimited; cd {tmp_path}; {EXE_SEAMS} cube40.stl", shell=True, env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout
    assert p.returncode == 0, out[-2000:]
    assert "Grid Size: nx = 61 , ny = 61 ,nz = 61" in out
    assert "Distance function time integration has reached steady state" in out
    assert "Min/max time integration has reached steady state" in out
    its = [int(x) for x in re.findall(r"Iteration:\s+(\d+)", out)]
    assert its[:2154] == list(range(2154)) and its[2154:2154 + 405] == list(range(1, 406))
    asym = float(re.search(r"Asymptotic Error:\s+(\S+)", out).group(1))
    assert abs(asym - 1.0085048924202963E-02) < 1e-15
    shape = (62, 62, 62)
    assert np.array_equal(stl_io.vti_read_phi(tmp_path / "signedDistanceFunction.vti", shape), cube40["phi_reinit"])
    assert np.array_equal(stl_io.vti_read_phi(tmp_path / "smoothedDistanceFunction.vti", shape), cube40["phi_minmax"])
    # the reference's own writer: its byte count is the element count of a 3-component array (set3d.f90:330)
    assert stl_io.vti_header_count(tmp_path / "smoothedDistanceFunction.vti") == (3 * 8 * 62 ** 3, False)
    lines = open(tmp_path / "cube40.s3d").read().split("\n")
    nelem, nnode = (int(v) for v in lines[0].split()[:2])
    nodes = np.array([[float(v) for v in ln.split()] for ln in lines[1 + nelem:1 + nelem + nnode]])
    adv = np.load(os.path.join(GOLDEN, "cube40_advect.npz"))["surfXX"]
    assert nodes.shape == adv.shape == (9140, 3)
    assert np.allclose(nodes, adv, rtol=1e-15, atol=0)


def _run_dropin(tmp_path, stl_name, surf_key, env_extra):
    import stl_io

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    stl_io.stl_write(tmp_path / stl_name, s[surf_key + "_surfX"], s[surf_key + "_surfElem"])
    env = dict(os.environ, LSF_ARITH="strict", **env_extra)
    p = subprocess.run(f"ulimit -s unlimited; cd {tmp_path}; {EXE} {stl_name}", shell=True, env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


def _sha_file_payload(path, shape):
    import hashlib

    import stl_io

    a = stl_io.vti_read_phi(path, shape)
    return hashlib.sha256(np.ascontiguousarray(a.ravel(order="F")).tobytes()).hexdigest(), a


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in executable not built")
def test_baseline_config2_cube40_256(tmp_path):
    """BASELINE config 2: cube40.stl, 256^3 fp64, reinit only (min/max cap 0), through the Fortran host with the
    run-time overrides of INTEGRATION.md E4.  Fixed 8 sweeps: the field written to signedDistanceFunction.vti must
    equal, bit for bit, what the reference's own reinit produced from the same phi0 (tests/golden/make_golden_c2.py)."""
    fx = os.path.join(GOLDEN, "cube40_256.npz")
    if not os.path.exists(fx):
        pytest.skip("cube40_256.npz not generated")
    g = np.load(fx)
    out = _run_dropin(tmp_path, "cube40.stl", "cube40",
                      dict(LSF_DX=repr(float(g["dx"])), LSF_REINIT_ITER=str(int(g["sweeps"]) - 1), LSF_MINMAX_ITER="0",
                           LSF_REINIT2_ITER="0"))
    assert "Grid Size: nx = 255 , ny = 255 ,nz = 255" in out
    got_sha, a = _sha_file_payload(tmp_path / "signedDistanceFunction.vti", (256, 256, 256))
    assert np.array_equal(a[::8, ::8, ::8], g["sample"])
    assert got_sha == str(g["sha"])
    rms = [float(x) for x in re.findall(r"RMS Error:\s+(\S+)", out)][: int(g["sweeps"])]
    assert np.allclose(rms, g["rms"], rtol=1e-8, atol=0)


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in executable not built")
def test_baseline_config1_cube40_64_literal(tmp_path):
    """BASELINE config 1 at its LITERAL size: cube40.stl with dx = 2/42 -> nx = ny = nz = 63, a 64^3 grid (SURVEY.md 8b:
    ceiling(2 / (2/42)) + 21 = 63 sits on a rounding knife-edge, so the host's own line is asserted).  The whole program
    through the reference's Fortran host with the run-time dx override: reinit to the reference's
    1e-5 stop (2 066 sweeps by the reference's own `reinit`, tests/golden/make_golden_c1.py), min/max flow to its 1e-7
    stop (384 iterations, pinned oracle), both .vti payloads bit for bit."""
    g = np.load(os.path.join(GOLDEN, "cube40_64.npz"))
    out = _run_dropin(tmp_path, "cube40.stl", "cube40", dict(LSF_DX=repr(float(g["dx"])), LSF_REINIT2_ITER="0"))
    assert "Grid Size: nx = 63 , ny = 63 ,nz = 63" in out
    assert "Distance function time integration has reached steady state" in out
    assert "Min/max time integration has reached steady state" in out
    sweeps, mm = int(g["sweeps"]), int(g["mm_iters"])
    its = [int(x) for x in re.findall(r"Iteration:\s+(\d+)", out)]
    # the stop sweep prints the steady-state line instead of its RMS: reinit prints 0 .. sweeps-2, min/max 1 .. mm-1
    assert its[:sweeps - 1] == list(range(sweeps - 1)) and its[sweeps - 1:sweeps - 1 + mm - 1] == list(range(1, mm))
    rms = [float(x) for x in re.findall(r"RMS Error:\s+(\S+)", out)]
    assert np.allclose(rms[:sweeps - 1], g["rms"], rtol=1e-9, atol=0)
    sha1, a1 = _sha_file_payload(tmp_path / "signedDistanceFunction.vti", (64, 64, 64))
    assert np.array_equal(a1, g["phi_re"]) and sha1 == str(g["phi_re_sha"])
    sha2, a2 = _sha_file_payload(tmp_path / "smoothedDistanceFunction.vti", (64, 64, 64))
    assert np.array_equal(a2[::3, ::3, ::3], g["phi_mm_sample"]) and sha2 == str(g["phi_mm_sha"])


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in executable not built")
def test_baseline_config3_twocube10_512(tmp_path):
    """BASELINE config 3: twoCube10.stl at the 512-point resolution (512 x 63 x 63 with the host's uniform padding),
    128 reinit sweeps (the surface diverges later in the reference itself) + 200 min/max-flow iterations."""
    fx = os.path.join(GOLDEN, "twocube10_512.npz")
    if not os.path.exists(fx):
        pytest.skip("twocube10_512.npz not generated")
    g = np.load(fx)
    out = _run_dropin(tmp_path, "twoCube10.stl", "twocube10",
                      dict(LSF_DX=repr(float(g["dx"])), LSF_REINIT_ITER=str(int(g["sweeps"]) - 1),
                           LSF_MINMAX_ITER=str(200), LSF_REINIT2_ITER="0"))
    assert "Grid Size: nx = 511 , ny = 62 ,nz = 62" in out
    shape = tuple(int(v) + 1 for v in g["n"])
    sha1, a1 = _sha_file_payload(tmp_path / "signedDistanceFunction.vti", shape)
    assert np.array_equal(a1[::8, ::4, ::4], g["reinit_sample"]) and sha1 == str(g["reinit_sha"])
    sha2, a2 = _sha_file_payload(tmp_path / "smoothedDistanceFunction.vti", shape)
    assert np.array_equal(a2[::8, ::4, ::4], g["minmax_sample"]) and sha2 == str(g["minmax_sha"])


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in executable not built")
def test_fortran_host_drives_the_block_decomposed_reinit(tmp_path):
    """order = 'jacobi' with a device list in &lsf_inputs: the reference's host runs both reinit calls block-decomposed
    (lsf_reinit_multi, one block per listed device; here four blocks sharing this box's GPU) between seams that keep
    their arrays on the device (resident = 2).  Both .vti payloads and the printed residuals equal the one-GPU Jacobi
    run, and LSF_DEVICES overrides the namelist.  With order = 'gs' (the default) the same device list runs the reference's own
    ordering over z slabs."""
    import stl_io

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    outs = {}
    for name, devices in (("one", None), ("four", "0, 0, 0, 0"), ("env", None)):
        d = tmp_path / name
        d.mkdir()
        stl_io.stl_write(d / "cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
        nml = "&lsf_inputs\n  order = 'jacobi'\n  reinit_iter = 40\n  minmax_iter = 6\n  reinit2_iter = 9\n"
        if devices:
            nml += f"  devices = {devices}\n"
        (d / "run.nml").write_text(nml + "/\n")
        env = {k: v for k, v in os.environ.items() if not k.startswith("LSF_")}
        if name == "env":
            env["LSF_DEVICES"] = "0,0"
        p = subprocess.run(f"ulimit -s unlimited; cd {d}; {EXE} cube40.stl run.nml", shell=True, env=env, text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:]
        outs[name] = p.stdout
    assert "block-decomposed" not in outs["one"]
    assert outs["four"].count("Reinit block-decomposed over  4  devices") == 2
    assert outs["env"].count("Reinit block-decomposed over  2  devices") == 2
    shape = (62, 62, 62)
    rms_one = [float(x) for x in re.findall(r"RMS Error:\s+(\S+)", outs["one"])]
    assert len(rms_one) == 41 + 6 + 10
    for name in ("four", "env"):
        for f in ("signedDistanceFunction.vti", "smoothedDistanceFunction.vti"):
            assert np.array_equal(stl_io.vti_read_phi(tmp_path / name / f, shape), stl_io.vti_read_phi(tmp_path / "one" / f, shape)), (name, f)
        rms = [float(x) for x in re.findall(r"RMS Error:\s+(\S+)", outs[name])]
        assert np.allclose(rms, rms_one, rtol=1e-10, atol=0)  # block sums added in rank order vs one fixed-order sum
    # the reference's own ordering shards too: z slabs of the exact Gauss-Seidel tile graph (lsf_reinit_multi with LSF_ORDER_GS),
    # same payload as the one-device run of that ordering, bit for bit, in the reference's arithmetic (the shim's default)
    got = {}
    for name, devs in (("gs1", None), ("gs2", "0,0"), ("gs3", "0,0,0"), ("gs2off", "0,0")):
        d = tmp_path / name
        d.mkdir()
        stl_io.stl_write(d / "cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
        env = {k: v for k, v in os.environ.items() if not k.startswith("LSF_")}
        env.update(LSF_REINIT_ITER="70", LSF_MINMAX_ITER="4", LSF_REINIT2_ITER="5")
        if devs:
            env["LSF_DEVICES"] = devs
            if name != "gs2off":
                env["LSF_SLABS"] = "1"  # opt-in (namelist `slabs = 1`): a device list alone keeps this ordering on one GPU
        p = subprocess.run(f"ulimit -s unlimited; cd {d}; {EXE} cube40.stl", shell=True, env=env, text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:]
        got[name] = p.stdout
    assert "z slabs" not in got["gs1"] and "z slabs" not in got["gs2off"]
    assert got["gs2off"].count("Reinit in the reference's ordering on one device") == 2
    assert got["gs2"].count("Reinit in the reference's ordering over  2  z slabs") == 2
    assert got["gs3"].count("Reinit in the reference's ordering over  3  z slabs") == 2
    rms1 = re.findall(r"RMS Error:\s+(\S+)", got["gs1"])
    assert len(rms1) == 71 + 4 + 6
    for name in ("gs2", "gs3", "gs2off"):
        for f in ("signedDistanceFunction.vti", "smoothedDistanceFunction.vti"):
            assert np.array_equal(stl_io.vti_read_phi(tmp_path / name / f, shape), stl_io.vti_read_phi(tmp_path / "gs1" / f, shape)), (name, f)
        assert re.findall(r"RMS Error:\s+(\S+)", got[name]) == rms1  # printed residuals: the same digits
