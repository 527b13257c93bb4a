"""The C-ABI shared library loads and exports every symbol include/llenv.h declares (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT
from lifelike_agility_and_play_amd import capi


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'llenv.h')).read()
    return sorted(set(re.findall(r'\b(ll_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_symbols() == capi.EXPORTED_SYMBOLS


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build_hip()
    lib = capi.load_library()                 # resolves every name in capi._SIGS or raises
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.ll_abi_version() == capi.LL_ABI_VERSION == 2
    assert lib.ll_model_blob_len() == 788


def test_no_gpu_means_loud_failure(model_blob, mocap_table):
    """Without a HIP device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    cfg = capi.make_config(4, prop_type=['joint_pos'])
    with pytest.raises(capi.LLError) as ei:
        capi.Engine(cfg, model_blob, mocap_table)
    assert ei.value.code == -5                # LL_ENODEV


def test_config_validation(model_blob, mocap_table):
    with pytest.raises(TypeError):
        capi.make_config(4, prop_type='joint_pos')            # PLE:113
    with pytest.raises(KeyError):
        capi.make_config(4, prop_type=['nonsense'])           # PLE:111


def test_policy_header_binding_and_library_agree():
    """include/llenv_policy.h (the fused on-device policy) == pmc_policy_hip._SIGS == what libllenv.so exports."""
    from lifelike_agility_and_play_amd import pmc_policy_hip
    text = open(os.path.join(ROOT, 'include', 'llenv_policy.h')).read()
    declared = sorted(set(re.findall(r'\b(ll_policy_[a-z0-9_]+)\s*\(', text)))
    assert declared == pmc_policy_hip.EXPORTED_SYMBOLS
    import __graft_entry__ as g
    g.build_hip()
    lib = pmc_policy_hip.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert pmc_policy_hip.pack_weights().size == pmc_policy_hip.LLP_N_FLOATS == 358647


def test_sepmc_header_binding_and_library_agree():
    """include/llenv_sepmc.h (the two-robot chase-tag env) == sepmc_capi._SIGS == what libllenv.so exports."""
    from lifelike_agility_and_play_amd import sepmc_capi
    text = open(os.path.join(ROOT, 'include', 'llenv_sepmc.h')).read()
    declared = sorted(set(re.findall(r'\b(ll_sepmc_[a-z0-9_]+)\s*\(', text)))
    assert declared == sepmc_capi.EXPORTED_SYMBOLS
    import __graft_entry__ as g
    g.build_hip()
    lib = sepmc_capi.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_xfer_header_binding_and_library_agree():
    """include/llenv_xfer.h (HIP IPC handles + CU-free pulls: the p2p trajectory hand-off) == xfer._SIGS == what libllenv.so exports."""
    from lifelike_agility_and_play_amd import xfer
    text = open(os.path.join(ROOT, 'include', 'llenv_xfer.h')).read()
    declared = sorted(set(re.findall(r'\b(ll_xfer_[a-z0-9_]+)\s*\(', text)))
    assert declared == xfer.EXPORTED_SYMBOLS and len(declared) == 19
    import __graft_entry__ as g
    g.build_hip()
    lib = xfer.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert xfer.HANDLE_BYTES == 64
