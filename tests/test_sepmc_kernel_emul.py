"""SEPMC kernel logic on the CPU: csrc/sepmc_step.hpp compiled for the host (tests/emul; the two robots of an arena run as two
threads that meet where the GPU rows exchange registers) and driven through the C ABI of include/llenv_sepmc.h."""
import os
import subprocess

import numpy as np
import pytest

import sepmc_parity_common as SC
from oracle import sepmc_oracle as SO

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s'])
    return EMUL_LIB


def test_reset_cases_against_reference_goldens(emul_lib):
    SC.check_engine_reset_cases(emul_lib)


def test_scripted_episodes_against_reference_goldens(emul_lib, model_blob):
    SC.check_engine_episodes(emul_lib, SO.BlobModel(model_blob))


def test_free_running_invariants(emul_lib):
    out = SC.check_free_running(emul_lib, n_arenas=4, steps=80)
    print(out)


def test_flag_handover_by_physical_contact(emul_lib):
    SC.check_flag_handover_physical(emul_lib)


def test_robot_robot_contact(emul_lib):
    print(SC.check_robot_robot_contact(emul_lib))


def test_pair_physics_against_oracle(emul_lib):
    print(SC.check_pair_physics_against_oracle(emul_lib))
