"""The env plug-in surface on the real HIP library (default lib_path)."""
import numpy as np
import pytest

import lifelike_agility_and_play_amd as lla
from test_env_api import check_single_env_contract, pmc_config

pytestmark = pytest.mark.gpu


def test_single_env_contract_gpu(golden):
    check_single_env_contract(golden, None)


def test_batched_env_zero_copy_torch(golden):
    import torch
    from lifelike_agility_and_play_amd import gather
    env = lla.create_tracking_game(**pmc_config(num_envs=256, seed=9))
    env.engine.set_stream(torch.cuda.current_stream().cuda_stream)
    env.reset()
    t = gather.engine_tensors(env.engine)
    act = torch.randn((256, 12), device='cuda') * 0.1353
    env.step_device(act.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t['obs'].cpu().numpy(), env.engine.obs())          # torch sees the engine's buffer
    r, d, _ = env.engine.reward_done()
    np.testing.assert_array_equal(t['reward'].cpu().numpy(), r)
    np.testing.assert_allclose(t['obs'][:, 123:135].cpu().numpy(), act.cpu().numpy(), rtol=1e-6)   # newest action in prop_a
    env.close()


def test_trajectory_ring_gpu(model_blob, mocap_table):
    import torch
    import parity_common as pc
    from lifelike_agility_and_play_amd import gather

    def read_ring(addr, shape):
        return gather.device_tensor(addr, shape).cpu().numpy()
    pc.check_trajectory_ring(model_blob, mocap_table, None, read_ring)
