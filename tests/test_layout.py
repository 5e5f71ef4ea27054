"""Repository contract checks: the product never touches the oracle, nothing that runs on the GPU
box reads /root/reference, no compatibility layers in the kernels."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ilgpu_raytracing_amd")


def _files(base, exts):
    for d, _, fs in os.walk(base):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_uses_the_oracle():
    for f in _files(PKG, (".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
        src = open(f, errors="ignore").read()
        code = re.sub(r'""".*?"""', "", src, flags=re.S) if f.endswith(".py") else re.sub(r"//.*", "", src)
        assert not re.search(r"\b(from|import)\s+oracle\b", code), f
        assert "liborc" not in code and "orc_" not in code, f
    for f in _files(os.path.join(ROOT, "include"), (".h",)):
        assert "orc_" not in open(f).read(), f


def test_nothing_on_the_gpu_box_reads_the_reference_tree():
    for f in list(_files(os.path.join(ROOT, "tests"), (".py",))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + \
            list(_files(PKG, (".py",))):
        if not os.path.exists(f) or f.endswith("test_layout.py"):
            continue
        src = open(f).read()
        assert not re.search(r"open\([^)]*root/reference|listdir\([^)]*root/reference", src), f


def test_no_compat_layers_in_kernels():
    for f in _files(os.path.join(PKG, "csrc"), (".hip", ".hpp", ".cpp")):
        src = open(f).read()
        for banned in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "hipify", "triton"):
            assert banned not in src, (f, banned)


def test_required_files_exist():
    for p in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/hip_raytrace.h", "include/hrt_types.h",
              "oracle/orc_kernels.hpp", "oracle/Makefile", "tests/golden/make_golden.py", "profiles"):
        assert os.path.exists(os.path.join(ROOT, p)), p


def test_tools_and_package_compile():
    """Every script under tools/ and the package byte-compile (they are only run on the GPU box otherwise)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "ilgpu_raytracing_amd", "*.py")) + \
        [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, "exec")
