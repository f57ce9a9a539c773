"""CPU suite: the product library loads and exports every symbol include/smarties_hip.h declares
(no compute calls without a GPU), and fails loudly -- no CPU fallback -- when no device exists."""
import ctypes as C
import os
import re

import pytest

from smarties_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "smarties_hip.h")).read()
    return sorted(set(re.findall(r"HL_API\s+[\w\s\*]+?\b(hl_\w+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("hl_create", "hl_destroy", "hl_append_episode", "hl_initialize", "hl_step", "hl_step_begin",
                 "hl_step_end", "hl_readback", "hl_comm_init", "hl_comm_unique_id", "hl_get_params"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build_hip()
    api = capi.load_hip()
    for s in declared_symbols():
        assert hasattr(api.lib, s), "libsmarties_hip.so does not export %s" % s


def test_oracle_exports_the_same_surface_with_ol_prefix():
    from oracle_api import oracle_api
    api = oracle_api()
    skip = {"hl_comm_init", "hl_comm_unique_id", "hl_xchg_export", "hl_xchg_connect", "hl_timing_enable", "hl_timing_get", "hl_status_string",
            "hl_version", "hl_kernel_profile",
            # file I/O of the replay memory: product only, pinned directly by files the compiled reference wrote
            "hl_save_memory", "hl_restart_memory", "hl_metrics"}
    for s in declared_symbols():
        if s in skip:
            continue
        assert hasattr(api.lib, "ol_" + s[3:]), s


def test_no_cpu_fallback_without_device():
    """hl_create must fail with HL_ERR_NO_DEVICE (2) on a machine without a GPU, never compute."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from smarties_amd import capi\n"
            "import ctypes as C\n"
            "api = capi.load_hip(); h = C.c_void_p(); cfg = capi.make_config()\n"
            "rc = api.fn('create')(C.byref(cfg), C.byref(h)); print('RC', rc)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "RC 2" in out.stdout, out.stdout + out.stderr


def test_bad_arguments_are_rejected():
    api = capi.load_hip()
    h = C.c_void_p()
    assert api.fn("create")(None, C.byref(h)) == 1
    cfg = capi.make_config()
    cfg.struct_size = 4
    assert api.fn("create")(C.byref(cfg), C.byref(h)) == 1
    assert api.fn("num_params")(None) == -1
