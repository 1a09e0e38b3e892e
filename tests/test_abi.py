"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/vmc.h declares;
the ctypes table mirrors the header (argument counts).  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|long long|const char\*)\s+(vmc_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from vimo_clip_amd import _lib
    fns = _header_functions()
    assert len(fns) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in fns:
        assert hasattr(lib, name), f"{name} declared in include/vmc.h but not exported by libvmc.so"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vmc_\w+)", nm))
    assert exported == set(fns), f"header/library mismatch: {exported ^ set(fns)}"


def test_ctypes_table_matches_header():
    from vimo_clip_amd import _lib
    fns = _header_functions()
    assert set(_lib.SIGNATURES) == set(fns)
    for name, n in fns.items():
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} args, ctypes table {len(_lib.SIGNATURES[name][1])}"


def test_error_strings_and_version():
    from vimo_clip_amd import _lib
    assert _lib.lib.vmc_abi_version() == 1
    assert b"alignment" in _lib.lib.vmc_error_string(-2)
    assert _lib.lib.vmc_error_string(0) == b"success"


def test_gfx950_code_object_present():
    from vimo_clip_amd import _lib
    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback_for_host_tensors():
    import pytest
    import torch
    from vimo_clip_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))


def test_persistent_gemm_kernels_use_no_scratch():
    """gemm8p_kernel counts the operations in its vmcnt queue by hand (vimo_clip_amd/csrc/gemm8.hip, `g8p_wait`): a register
    spill would add scratch loads / stores -- VMEM operations -- to that queue.  The Makefile keeps hipcc's resource-usage
    remarks of gemm8.hip; every persistent instantiation must report no scratch and no spilled VGPR."""
    import re
    path = os.path.join(ROOT, "vimo_clip_amd", "csrc", "build", "gemm8.usage.txt")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    text = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "gemm8p_kernel" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        spills = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        assert scratch == 0 and spills == 0, (name, scratch, spills)
    assert seen >= 10, seen


def test_training_parameter_structs_mirror_the_header():
    """vmc_tfam_layer_params / vmc_tfam_head_params (include/vmc.h) are filled from Python through ctypes.Structure mirrors
    (vimo_clip_amd/tfam_train.py): same field names in the same order, every field one pointer wide."""
    import ctypes as C

    from vimo_clip_amd import tfam_train as tt
    src = open(os.path.join(ROOT, "include", "vmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, mirror in (("vmc_tfam_layer_params", tt.LayerParams), ("vmc_tfam_head_params", tt.HeadParams)):
        body = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + r"\s*;", src, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            assert re.match(r"(const\s+)?(void|float)\s*\*", decl), decl          # pointers only
            names += [n.strip().lstrip("*").strip() for n in re.sub(r"^(const\s+)?(void|float)\s*", "", decl).split(",")]
        assert names == [f[0] for f in mirror._fields_], (cname, names)
        assert all(f[1] is C.c_void_p for f in mirror._fields_) and C.sizeof(mirror) == 8 * len(names)
