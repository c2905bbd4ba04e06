"""The C-ABI shared library loads (no GPU needed) and exports every function include/dsopp_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dsopp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(dsopp_hip_[a-z0-9_]+)\s*\(", text))
    names.discard("dsopp_hip_allreduce_fn")
    return names


def test_header_symbols_exported():
    from dsopp_amd import capi
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(capi.LIB_PATH)
    declared = declared_functions()
    assert len(declared) > 40
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/dsopp_hip.h but not exported: {missing}"
    assert set(capi.SYMBOLS) == declared, (set(capi.SYMBOLS) ^ declared)


def test_header_is_plain_c():
    """include/dsopp_hip.h is the C-ABI: it must compile as C (and every declaration must be at file scope)"""
    import subprocess
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "dsopp_hip.h")], check=True)


def test_no_cpu_fallback():
    """without a GPU every compute entry point must fail loudly (DSOPP_HIP_ERR_HIP), never fall back to the CPU"""
    import pytest
    from dsopp_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.HipError) as e:
        capi.HipWindow(capi.default_pba_options())
    assert "no HIP device" in str(e.value) or "-4" in str(e.value)
    with pytest.raises(capi.HipError):
        capi.Pyramid(64, 48, 1)
    with pytest.raises(capi.HipError):
        capi.HipAligner()
    with pytest.raises(capi.HipError) as e:   # the single-process multi-device form: no worker thread is started without a device
        capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 1])
    assert "no HIP device" in str(e.value) or "-4" in str(e.value)


def test_default_options_are_the_production_values():
    """createPhotometricBundleAdjustment / createPoseAlignment — src/tracker/tracker/src/fabric.cpp:63-79,127-142"""
    from dsopp_amd import capi
    o = capi.default_pba_options()
    assert (o.max_iterations, o.initial_trust_region_radius, o.sigma_huber_loss) == (7, 1e5, 20)
    assert tuple(o.affine_brightness_regularizer) == (1e12, 1e8) and o.fixed_state_regularizer == 1e16
    assert o.force_accept == 1 and o.estimate_uncertainty == 1 and o.first_estimate_jacobians == 1 and o.optimize_idepths == 1
    a = capi.default_align_options()
    assert (a.max_iterations, a.initial_trust_region_radius, a.function_tolerance) == (50, 1e2, 1e-5)
