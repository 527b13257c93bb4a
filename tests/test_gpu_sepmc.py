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


def test_zero_copy_torch_views_gpu():
    """torch tensors over the SEPMC engine's own obs / action buffers on the engine's stream: a device-side policy needs no copies."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('torch.cuda is not available on this box')
    from lifelike_agility_and_play_amd import gather
    E = SC.make_engine(SC.env_config((1, 0, 0)), 64, None, auto_reset=1, seed=2)
    gather.use_engine_stream(E)
    T = gather.engine_tensors(E)
    E.reset()
    assert T['obs'].shape == (128, 965) and T['actions'].shape == (128, 12)
    for t in range(3):
        T['actions'].copy_(torch.randn((128, 12), device='cuda') * 0.1)
        a = T['actions'].clone()
        E.step()
        torch.cuda.current_stream().synchronize()
    np.testing.assert_array_equal(T['obs'].cpu().numpy().reshape(64, 2, 965), E.obs())
    live = ~np.repeat(E.reward_done()[1], 2)
    np.testing.assert_allclose(T['obs'][:, 123:135].cpu().numpy()[live], a.cpu().numpy()[live], rtol=1e-6)      # the newest action frame of prop_a
    E.close()
