"""The C-ABI library loads, exports every symbol include/*.h declares, agrees with the headers on
struct sizes, and fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from ilgpu_raytracing_amd import _types as T, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _declared_functions(header):
    src = open(os.path.join(INC, header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hrth?_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(hrt_lib):
    names = _declared_functions("hip_raytrace.h") + _declared_functions("hrt_host.h")
    assert "hrt_render_frame" in names and "hrth_scene_load_mesh_instance" in names and len(names) >= 25
    missing = [n for n in names if not hasattr(hrt_lib, n)]
    assert not missing, "libhip_raytrace.so does not export: %s" % missing


def test_test_hooks_live_only_in_the_test_build(hrt_lib, hooks_lib):
    """include/hrt_test_hooks.h: every hook is exported by libhip_raytrace_test.so and by nothing a production host links against."""
    hooks = [n for n in _declared_functions("hrt_test_hooks.h") if n not in _declared_functions("hip_raytrace.h")]
    assert "hrt_math_probe" in hooks and "hrt_debug_treelets" in hooks and len(hooks) >= 7
    assert not [n for n in hooks if not hasattr(hooks_lib, n)], "libhip_raytrace_test.so misses a declared hook"
    out = subprocess.check_output(["nm", "-D", "--defined-only", engine.LIB_PATH]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    assert not [n for n in hooks if n in exported], "the shipped library exports test hooks"
    assert not [n for n in exported if "debug" in n and n.startswith("hrt")]
    assert b"test-hooks" in hooks_lib.hrt_version() and b"test-hooks" not in hrt_lib.hrt_version()
    # the product ABI is the same in both
    names = _declared_functions("hip_raytrace.h") + _declared_functions("hrt_host.h")
    assert not [n for n in names if not hasattr(hooks_lib, n)]


def test_struct_sizes_match_headers():
    """Compile a C program against include/ and compare sizeof() with the ctypes mirrors."""
    prog = r'''
#include <stdio.h>
#include "hip_raytrace.h"
#include "hrt_host.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(hrt_scene_desc), sizeof(hrt_frame_params), sizeof(hrt_stats),
   sizeof(hrt_outputs), sizeof(hrt_render_opts), sizeof(hrt_device_views), sizeof(hrt_kernel_counters), sizeof(hrt_instance),
   sizeof(hrt_sphere), sizeof(hrt_camera), sizeof(hrt_bvh_update_stats), sizeof(hrt_bvh_node));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-std=c11", "-I", INC, c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(x) for x in (T.SceneDesc, T.FrameParams, T.Stats, T.Outputs, T.RenderOpts, T.DeviceViews,
                                  T.KernelCounters, T.InstanceRecord, T.Sphere, T.Camera, T.BvhUpdateStats, T.BvhNode)]
    assert got == want


def test_headers_are_plain_c_and_cite_the_reference():
    for h in ("hip_raytrace.h", "hrt_host.h", "hrt_types.h"):
        src = open(os.path.join(INC, h)).read()
        assert re.search(r"\.cs:\d+", src), "%s cites no reference file:line" % h
        code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)              # declarations only, comments stripped
        assert "torch" not in code and "std::" not in code and "class " not in code


def test_fails_loudly_without_gpu(hrt_lib):
    """On a machine without a usable HIP device hrt_create must return an error (there is no CPU
    path to fall back to); on the GPU box it must succeed."""
    n = hrt_lib.hrt_device_count()
    if n <= 0:
        h = C.c_void_p()
        rc = hrt_lib.hrt_create(None, 0, C.byref(h))
        assert rc in (-4, -3) and not h.value
        assert b"no HIP device" in hrt_lib.hrt_last_error(None)
        with pytest.raises(engine.HrtError):
            engine.RTRenderer([0])
    else:
        r = engine.RTRenderer([0])
        r.close()


def test_version_string(hrt_lib):
    assert b"gfx950" in hrt_lib.hrt_version()


def _build_c_example(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "ilgpu_raytracing_amd", "csrc")
    exe = str(tmp_path / "hrt_bench")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", INC, os.path.join(root, "examples", "hrt_bench.c"), "-o", exe,
                           "-L", csrc, "-lhip_raytrace", "-Wl,-rpath," + csrc, "-lm"])
    return exe


def test_plain_c_host_builds_and_fails_loudly_without_gpu(hrt_lib, tmp_path):
    """examples/hrt_bench.c: the boundary used from C11 with nothing but the two headers and -lhip_raytrace."""
    exe = _build_c_example(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_plain_c_host_renders")
    out = subprocess.run([exe, "64", "48", "1", "1"], capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr and "no CPU path" in out.stderr


@pytest.mark.gpu
def test_plain_c_host_renders(hrt_lib, tmp_path):
    import json
    exe = _build_c_example(tmp_path)
    runs = [json.loads(subprocess.run([exe, "320", "180", "2", "3"], capture_output=True, text=True, check=True).stdout) for _ in range(2)]
    assert runs[0]["color_checksum"] == runs[1]["color_checksum"] and runs[0]["rays_per_frame"] == runs[1]["rays_per_frame"]
    assert runs[0]["host"] == "C" and runs[0]["frames"] == 3 and runs[0]["mrays_per_s"] > 0 and runs[0]["rays_per_frame"] > 320 * 180
    assert runs[0]["device_tlas_same_picture"] is True and runs[0]["device_tlas_nodes"] > 0
