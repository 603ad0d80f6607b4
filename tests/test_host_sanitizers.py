"""The rewritten host parsers under AddressSanitizer + UndefinedBehaviorSanitizer, the way the reference guards its own (CMakeLists.txt:65-66,
fuzz/target_mesh_parser.cpp:9-33, fuzz/target_image_io_read.cpp:10-21): the reference's fuzz targets, compiled UNCHANGED against this
repository's headers and host sources (src/host/mesh.cpp: mmap'ed, multi-threaded OBJ reader with its own decimal conversion;
src/host/image_io.cpp: PNG framing and filters on zlib), are fed a corpus of malformed inputs and random mutations of it; the
reference's parser tests run in the same sanitized build.  Build container only (the targets are compiled from /root/reference)."""
import os
import subprocess

import pytest

from tests import fuzz_corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = [os.path.join(ROOT, "src", "host", f) for f in ("mesh.cpp", "image_io.cpp", "world.cpp")]
FLAGS = ["-std=c++20", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include")]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "fuzz", "target_mesh_parser.cpp")), reason="reference sources not present")


def _build(out, sources, extra=()):
    subprocess.run(["g++"] + FLAGS + list(extra) + sources + HOST + ["-lz", "-pthread", "-o", out], check=True)
    return out


@pytest.mark.parametrize("target,inputs", [("mesh_parser", fuzz_corpus.obj_inputs), ("image_io_read", fuzz_corpus.png_inputs)])
def test_reference_fuzz_target_under_sanitizers(tmp_path, target, inputs):
    exe = _build(str(tmp_path / target), [os.path.join(REF, "fuzz", "target_%s.cpp" % target), os.path.join(ROOT, "tests", "cpp", "fuzz_driver.cpp")])
    corpus = str(tmp_path / "corpus")
    fuzz_corpus.write(corpus, inputs())
    r = subprocess.run([exe, corpus, "8", "7"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PATHTRACE_LOADER_THREADS="3", PATHTRACE_LOADER_PIECE_BYTES="64"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "no sanitizer report" in r.stdout and int(r.stdout.split()[0]) > 10000, r.stdout


def test_reference_parser_tests_under_sanitizers(tmp_path):
    shim = os.path.join(ROOT, "tests", "cpp", "shim")
    tests = [os.path.join(REF, "test", f) for f in ("main.cpp", "test_utils.cpp", "scene/mesh_test.cpp", "image/image_io_test.cpp", "scene/boundig_box_test.cpp")]
    exe = _build(str(tmp_path / "parser_tests"), tests, extra=["-I" + shim, "-I" + os.path.join(REF, "test")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    for name in ("MeshTest.SimpleMeshTest", "ImageIOTest.EncodeDecodeTest", "AABBTest.IntersectionTest"):
        assert "[       OK ] " + name in r.stdout, r.stdout
