"""The C-ABI library loads without a GPU and exports every symbol include/rt_photon.h declares.
No compute call is made here (that is what -m gpu is for)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_photon.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("rt_create", "rt_destroy", "rt_upload_optics", "rt_set_scene", "rt_trace", "rt_synchronize",
                 "rt_download_tallies", "rt_reset_tallies", "rt_tally_device_buffer", "rt_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import raytracer_amd
    lib = raytracer_amd.load_library()
    for s in declared_symbols():
        assert hasattr(lib, s), f"librt_photon.so does not export {s}"


def test_argument_errors_need_no_gpu():
    import raytracer_amd
    lib = raytracer_amd.load_library()
    assert lib.rt_trace(None, 0, 0, 1) != 0
    assert b"null context" in lib.rt_last_error()
    assert lib.rt_destroy(None) == 0
    h = C.c_void_p()
    assert lib.rt_create(0, 0, 4, 4, 1.0, 1.0, 1.0, C.byref(h)) != 0      # bad grid is rejected before any HIP call
    assert b"not valid" in lib.rt_last_error()


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/, drivers/ or include/ may
    reference it."""
    bad = []
    for base in ("raytracer_amd", "drivers", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"oracle_lib|liboracle|oracle/|photon_oracle|cpu_ref", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
