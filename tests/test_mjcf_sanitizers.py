"""The MJCF compiler inside the C-ABI library (robosuite_amd/csrc/rsim_mjcf.cpp, rsim_model_compile) under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY
section 5 lists a sanitizer build of the host code among the auxiliaries).  The compiler is plain host C++ with no HIP dependency, so it is built here with g++ and driven
by tests/san/mjcf_san_driver.cpp on the hand-written corpus of tests/test_mjcf_cpp.py, the unsupported-MJCF cases, the fixture models and (when the reference checkout is
there) the Lift / Panda MJCF the reference assembles with its mesh files -- each file intact, truncated at evenly spaced points and damaged a few hundred times.
Pass = the process exits 0: no out-of-bounds access, use after free, leak, signed overflow or null dereference, and every rejected input carries a reason."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "robosuite_amd", "csrc", "rsim_mjcf.cpp")
DRIVER = os.path.join(ROOT, "tests", "san", "mjcf_san_driver.cpp")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _build(tmp):
    exe = os.path.join(tmp, "mjcf_san")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", SRC, DRIVER, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_mjcf_compiler_under_asan_and_ubsan(tmp_path):
    from tests.test_mjcf_cpp import BAD, CORPUS, GOLD

    exe = _build(str(tmp_path))
    files = []
    for name, xml in list(CORPUS.items()) + [("bad_" + k.replace(" ", "_"), v) for k, v in BAD.items()]:
        p = tmp_path / f"{name}.xml"
        p.write_text(xml)
        files.append(str(p))
    n_expected_ok = len(CORPUS)
    for f in sorted(os.listdir(GOLD)):
        if f.endswith(".xml"):
            files.append(os.path.join(GOLD, f)); n_expected_ok += 1
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, "24", "300"] + files, capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout.strip())
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    assert f"intact files: {n_expected_ok} of {len(files)}" in r.stdout, r.stdout     # the corpus compiles, the unsupported cases are rejected, nothing else

    # the MJCF the reference assembles for the headline configuration, with its STL / OBJ meshes (absolute paths inside): only where the checkout exists
    if os.path.isdir("/root/reference/robosuite"):
        from tests.test_mjcf_cpp import _DUMP
        d = subprocess.run([sys.executable, "-c", _DUMP % (ROOT, {"lift_panda": ("Lift", "Panda")}), str(tmp_path)], capture_output=True, text=True, timeout=900)
        assert d.returncode == 0, d.stderr[-3000:]
        r = subprocess.run([exe, "12", "60", str(tmp_path / "lift_panda.xml"), str(tmp_path / "gripper_tester_robotiq140.xml")], capture_output=True, text=True, timeout=900, env=env)
        print(r.stdout.strip())
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
        assert "intact files: 2 of 2" in r.stdout, r.stdout


def test_hostile_msh_headers_and_deep_nesting_are_rejected_with_a_reason(tmp_path):
    """Round-5 advisor finding: load_msh trusted signed header counts (a negative count wrapped as size_t and passed the bound: an out-of-bounds read at an
    attacker-chosen offset), and the XML parser and tree walks recursed without a depth limit (a deeply nested document overflowed the stack, which no try / catch
    can turn into rsim_last_error()).  Under ASan + UBSan: a .msh with a negative count, one whose counts exceed the file, one whose faces index beyond the vertices,
    and a document nested 100 000 levels deep all come back as rejections with a reason; an intact .msh compiles."""
    import struct

    exe = _build(str(tmp_path))

    def msh(nv, nn, nt, nf, verts=None, faces=None, cut=None):
        verts = verts if verts is not None else [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
        faces = faces if faces is not None else [(0, 2, 1), (0, 1, 3), (0, 3, 2), (1, 2, 3)]
        b = struct.pack("<4i", nv, nn, nt, nf) + b"".join(struct.pack("<3f", *v) for v in verts) + b"".join(struct.pack("<3i", *f) for f in faces)
        return b if cut is None else b[:cut]

    cases = {"ok": (msh(4, 0, 0, 4), True), "negative_vertices": (msh(-1, 0, 0, 4), False), "negative_normals": (msh(4, -3, 0, 4), False),
             "negative_faces": (msh(4, 0, 0, -2), False), "counts_beyond_file": (msh(4, 0, 0, 400000), False), "huge_normals": (msh(4, 2 ** 31 - 1, 2 ** 31 - 1, 4), False),
             "truncated": (msh(4, 0, 0, 4, cut=40), False), "face_out_of_range": (msh(4, 0, 0, 4, faces=[(0, 2, 1), (0, 1, 7), (0, 3, 2), (1, 2, 3)]), False)}
    xml = """<mujoco><asset><mesh name="m" file="%s"/></asset><worldbody><body name="b" pos="0 0 1"><joint type="free"/><geom type="mesh" mesh="m" mass="1"/></body></worldbody></mujoco>"""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for name, (blob, good) in cases.items():
        (tmp_path / f"{name}.msh").write_bytes(blob)
        f = tmp_path / f"{name}.xml"
        f.write_text(xml % str(tmp_path / f"{name}.msh"))
        r = subprocess.run([exe, "0", "0", str(f)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-3000:])       # no sanitizer finding, no crash
        assert (f"intact files: {1 if good else 0} of 1" in r.stdout) and ("rejected with a reason %d" % (0 if good else 1)) in r.stdout, (name, r.stdout)
    deep = tmp_path / "deep.xml"
    deep.write_text("<mujoco><worldbody>" + "<body>" * 100000 + "</body>" * 100000 + "</worldbody></mujoco>")
    r = subprocess.run([exe, "0", "0", str(deep)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "intact files: 0 of 1" in r.stdout and "rejected with a reason 1" in r.stdout, (r.stdout[-300:], r.stderr[-2000:])
