"""The C-ABI library loads and exports every symbol include/pli_frontend.h declares; struct
layouts agree between the header, the ctypes binding and the oracle's copy.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pli_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "pli_frontend.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pli_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libpli_frontend.so does not export %s" % n
    assert set(capi.exported_symbols()) == set(names), "ctypes prototypes out of sync with the header"


def test_struct_sizes():
    assert capi.KEYPOINT_DT.itemsize == 24 and capi.KEYLINE_DT.itemsize == 68 and capi.PROJ_QUERY_DT.itemsize == 32
    from oracle import pyoracle as po
    assert C.sizeof(capi.Config) == C.sizeof(po.Config)
    c = capi.default_config(752, 480)
    o = po.default_config(752, 480)
    assert bytes(c) == bytes(o), "pli_config_default and the oracle's EuRoC defaults differ"
    assert c.orb_nfeatures == 1200 and c.lsd_nfeatures == 500 and abs(c.bf - 47.90639384423901) < 1e-5
    assert capi.lib().pli_kp_capacity(C.byref(c)) >= 1200 + 3 * 8
    assert capi.lib().pli_kl_capacity(C.byref(c)) == 500


def test_the_product_library_reads_four_environment_variables_and_no_more():
    """The schedules that were measured and shelved, the test switches and the tuning knobs live in the development build
    (libpli_frontend_dev.so, -DPLI_DEV): the product library does not even contain their names, so a stray variable in an integrator's
    environment cannot change a schedule.  Both builds export the whole C ABI."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "pli_slam_amd", "csrc")
    names = {}
    for lib in ("libpli_frontend.so", "libpli_frontend_dev.so"):
        out = subprocess.run(["strings", "-n", "6", os.path.join(csrc, lib)], capture_output=True, text=True, check=True).stdout
        names[lib] = sorted({m.group(0) for l in out.splitlines() for m in [re.match(r"PLI_[A-Z0-9_]+", l)] if m})
    assert set(names["libpli_frontend.so"]) == {"PLI_ROCTX", "PLI_SYNC_DEBUG", "PLI_TX_TAIL", "PLI_LSD_MODE"}, names["libpli_frontend.so"]
    assert len(names["libpli_frontend.so"]) <= 8 and len(names["libpli_frontend_dev.so"]) > 40
    dev = capi.lib(dev=True)
    for name in capi._PROTOS:
        assert hasattr(dev, name), name


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pli_slam_amd.frontend import Frontend
    with pytest.raises(capi.PliError) as e:
        Frontend(capi.default_config(752, 480))
    assert e.value.status == -5      # PLI_ERR_NO_DEVICE: the product path fails loudly without a GPU


def test_invalid_config_rejected_before_touching_the_device():
    h = C.c_void_p()
    bad = capi.default_config(752, 480, lsd_refine=1)
    assert capi.lib().pli_ctx_create(C.byref(bad), 0, C.byref(h)) == -1
    bad = capi.default_config(16, 16)
    assert capi.lib().pli_ctx_create(C.byref(bad), 0, C.byref(h)) == -1
    assert capi.lib().pli_ctx_create(None, 0, C.byref(h)) == -1


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pli_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle/" not in txt.replace("oracle/ocv_prims.hpp for the derivation", "") and "pyoracle" not in txt, \
                    "%s references the oracle" % os.path.join(dirpath, f)
