"""SEPMC on the real HIP library (default lib_path): the same checks as tests/test_sepmc_kernel_emul.py, at GPU batch sizes."""
import numpy as np
import pytest

import sepmc_parity_common as SC
from oracle import sepmc_oracle as SO

pytestmark = pytest.mark.gpu


def test_reset_cases_against_reference_goldens_gpu():
    SC.check_engine_reset_cases(None)


def test_scripted_episodes_against_reference_goldens_gpu(model_blob):
    SC.check_engine_episodes(None, SO.BlobModel(model_blob))


def test_free_running_invariants_gpu():
    out = SC.check_free_running(None, n_arenas=201, steps=200)        # 402 rows: a partial last wave (two of four rows), ray traces on
    print(out)
    out = SC.check_free_running_big(None, n_arenas=3000, steps=60)    # occupancy-2 build
    print(out)


def test_flag_handover_by_physical_contact_gpu():
    SC.check_flag_handover_physical(None)


def test_robot_robot_contact_gpu():
    print(SC.check_robot_robot_contact(None))


def test_pair_physics_against_oracle_gpu():
    print(SC.check_pair_physics_against_oracle(None, n_arenas=64, seed=9))


def test_free_running_against_the_oracle_env_gpu():
    print(SC.check_free_running_against_oracle_env(None))
