"""CPU-side checks of the product: the C-ABI library builds, loads, exports every symbol include/pt_hip.h declares and
refuses to compute without a GPU (there is no CPU fallback); the restated libm of the device code matches glibc."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from cpupathtrace_amd import binding, build, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    build.build()
    lib = ctypes.CDLL(binding.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(binding.EXPORTS), declared ^ set(binding.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_host_side_helpers_without_gpu():
    # pure host entry points work anywhere
    tiles = binding.job_tiles(132, 68)  # reference src/worker.cpp:398-414: tile_size = clamp(min(w,h)/4, 1, 32) = 17
    assert len(tiles) == 8 * 4 and tiles[0].tolist() == (0, 0, 17, 17) and tiles[-1].tolist() == (119, 51, 13, 17)
    assert len(binding.job_tiles(0, 10)) == 0
    assert len(binding.job_tiles(3, 3)) == 9  # tile_size 1
    assert binding.seed_to_state(1234) == (1234 ^ ((~1234 << 32) & 0xFFFFFFFFFFFFFFFF))
    assert binding.pixel_seed(1, 2, 3) != binding.pixel_seed(1, 3, 2)


def test_no_cpu_fallback():
    if binding.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(binding.PtError) as e:
        binding.Scene(scenes.box_scene()[0])
    assert e.value.code == 2  # PT_ERR_NO_DEVICE


def test_scene_builders_match_numpy_restatement():
    # demo/main.cpp object count without the dragon: 2 + 2 + 2 + 8 + 1 + 12
    sc, cam = scenes.cornell_scene()
    assert len(sc["obj_kind"]) == 27 and len(sc["sph"]) == 1
    box, _ = scenes.box_scene()
    assert len(box["obj_kind"]) == 14
    assert scenes.make_plane((0, 0, 0), (1, 0, 0)).shape == (0, 3, 3)  # degenerate: two coordinates coincide


LIBM_CHECK = r"""
#include "%s"
#include <cmath>
#include <cstdio>
int main(int argc, char **argv) {
    unsigned long bad = 0, n = 0;
    unsigned step = (unsigned)atoi(argv[1]);
    for(uint32_t u = 0; u <= ptm::as_u32(7.0f); u += step, n++) {
        float f = ptm::as_f32(u);
        bad += ptm::as_u32(sinf(f)) != ptm::as_u32(ptm::sinf_glibc(f));
        bad += ptm::as_u32(cosf(f)) != ptm::as_u32(ptm::cosf_glibc(f));
    }
    for(uint32_t u = ptm::as_u32(0x1p-33f); u <= ptm::as_u32(1.0f); u += step, n++) {
        float f = ptm::as_f32(u);
        bad += ptm::as_u32(powf(f, 0.5f)) != ptm::as_u32(ptm::powf_glibc(f, 0.5f));
        bad += ptm::as_u32(powf(f, 1.0f)) != ptm::as_u32(ptm::powf_glibc(f, 1.0f));
        bad += ptm::as_u32(ptm::powf_glibc(f, 1.0f)) != u;
    }
    bad += ptm::as_u32(powf(0.0f, 0.5f)) != ptm::as_u32(ptm::powf_glibc(0.0f, 0.5f));
    for(uint32_t u = 0; u <= 0x3f800000u; u += step, n++) {
        for(int s = 0; s < 2; s++) {
            float f = ptm::as_f32(u | (s ? 0x80000000u : 0u));
            bad += ptm::as_u32(acosf(f)) != ptm::as_u32(ptm::acosf_glibc(f));
        }
    }
    // the general powf (gamma correction): bases over the whole non-negative range incl. subnormals, 0, inf; exponents of both signs
    const float ys[] = {1.0f / 1.8f - 1.0f, 1.0f / 2.2f - 1.0f, 1.0f / 0.1f - 1.0f, 1.0f / 2.0f - 1.0f, 0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 3.0f, -2.5f, 40.0f, -40.0f, 1e-3f};
    for(uint32_t u = 0; u <= 0x7f800000u; u += 16 * step + 7, n++) {
        float f = ptm::as_f32(u);
        for(float y : ys) {
            bad += ptm::as_u32(powf(f, y)) != ptm::as_u32(ptm::powf_glibc_full(f, y));
        }
    }
    {
        uint64_t state = 88172645463325252ULL;
        for(int i = 0; i < 2000000; i++, n++) {
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            float f = ptm::as_f32((uint32_t)state), y = ptm::as_f32((uint32_t)(state >> 32));
            float a = powf(f, y), b = ptm::powf_glibc_full(f, y);
            bad += (a != a) ? !(b != b) : (ptm::as_u32(a) != ptm::as_u32(b));
        }
    }
    printf("%%lu %%lu\n", n, bad);
    return 0;
}
"""


@pytest.mark.parametrize("step", [251])
def test_device_libm_matches_glibc(tmp_path, step):
    """cpupathtrace_amd/csrc/pt_libm.h compiled for the host against the running glibc (2.35 in this image): sinf/cosf on
    [0, 7], powf(x, 0.5 | 1) on [2^-33, 1], the general powf over all non-negative bases and 2 M random bit patterns, acosf on [-1, 1]; every `step`-th float (step 1 = exhaustive, ~30 s)."""
    src = tmp_path / "libm_check.cpp"
    src.write_text("#include <cstdlib>\n" + LIBM_CHECK % os.path.join(ROOT, "cpupathtrace_amd", "csrc", "pt_libm.h"))
    exe = tmp_path / "libm_check"
    subprocess.run(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-o", str(exe), str(src)], check=True)
    n, bad = subprocess.run([str(exe), str(step)], check=True, capture_output=True, text=True).stdout.split()
    assert int(n) > 1000000 and int(bad) == 0
