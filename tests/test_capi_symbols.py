"""The C-ABI library loads on a machine without a GPU and exports every symbol include/htk_amd.h declares.
No compute is attempted here; the no-device error path is checked instead (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "htk_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(htkamd_\w+)\s*\(", txt)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("htkamd_model_create", "htkamd_outp_block", "htkamd_fb_prepare", "htkamd_fb_execute",
                 "htkamd_accs_device_vector", "htkamd_model_update"):
        assert must in names


def test_every_declared_symbol_is_exported(native):
    L = native.lib()
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_error_string(native):
    L = native.lib()
    assert L.htkamd_version() >= 100
    assert isinstance(L.htkamd_last_error(), bytes)


def test_fails_loudly_without_a_device(native):
    L = native.lib()
    if L.htkamd_device_count() > 0:
        pytest.skip("a GPU is present")
    from htk_amd import synth
    s = synth.generate(4, 1, 3, 0, 10, 1)
    with pytest.raises(native.HtkAmdError) as e:
        native.Model(s.packed())
    assert "no HIP device" in str(e.value)


def test_parm_stream_argument_checks(native):
    """Buffer-mode qualifiers: _Z is refused (needs the whole utterance), and without a device the open fails loudly."""
    q = native.parm_quals_from_kind("MFCC_E_D_A_Z", 13)
    with pytest.raises(native.HtkAmdError) as e:
        native.ParmStream(q, 16)
    assert "_Z" in str(e.value)
    if native.lib().htkamd_device_count() == 0:
        with pytest.raises(native.HtkAmdError) as e:
            native.ParmStream(native.parm_quals_from_kind("MFCC_E_D_A", 13), 16)
        assert "no HIP device" in str(e.value)


def test_bad_arguments_are_rejected(native):
    L = native.lib()
    assert L.htkamd_model_create(None, None) == -1
    assert L.htkamd_outp_block(None, None, 0, None, 0, None, 0, None) == -1
    assert b"outp_block" in L.htkamd_last_error()


def test_header_is_plain_c_and_the_c_driver_links(tmp_path):
    """include/htk_amd.h compiles as C (gnu11, -Wall -Wextra -pedantic clean) and the C hosts under examples/ -- which use nothing
    but that header -- link against the library (the GPU tests run it)."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    src = tmp_path / "hdr.c"
    src.write_text('#include "htk_amd.h"\nint main(void) { return htkamd_device_count() < -1; }\n')
    lib = os.path.join(root, "htk_amd")
    for args in (["-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", str(src)], [os.path.join(root, "examples", "herest_pass.c"), "-Wall", "-Werror"],
                 [os.path.join(root, "examples", "hvite_decode.c"), "-Wall", "-Werror"]):
        subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(root, "include")] + args + ["-o", str(tmp_path / "a.out"), "-L" + lib, "-lhtk_amd",
                                                                                           "-Wl,-rpath," + os.path.abspath(lib)])
