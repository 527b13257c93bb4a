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
    with pytest.raises(ValueError):
        env.engine.set_stream(torch.cuda.default_stream().cuda_stream)             # handle 0 would silently mean "private stream"
    gather.bind_torch_stream(env.engine)
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


def test_trained_policy_on_device_closed_loop(model_blob, mocap_table):
    """The trained reference policy evaluated with torch on the engine's own device buffers (zero copies) agrees with its NumPy
    statement, and drives 1024 environments closed-loop on the GPU with a high tracking reward."""
    import os
    import torch
    from conftest import GOLDEN_DIR, PMC_PROP_TYPE, PMC_REWARD_WEIGHTS
    from lifelike_agility_and_play_amd import capi, gather
    from lifelike_agility_and_play_amd.pmc_policy import PmcPolicy
    from lifelike_agility_and_play_amd.pmc_policy_torch import TorchPmcPolicy
    n = 1024
    cfg = capi.make_config(n, control_freq=50.0, kd=0.5, reward_weights=PMC_REWARD_WEIGHTS, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0,
                           auto_reset=1, seed=2)
    E = capi.Engine(cfg, model_blob, mocap_table)
    gather.bind_torch_stream(E)
    T = gather.engine_tensors(E)
    pol, ref = TorchPmcPolicy(), PmcPolicy(os.path.join(GOLDEN_DIR, 'pmc_policy.npz'))
    E.reset()
    a = pol.act(T['obs']).cpu().numpy()
    np.testing.assert_allclose(a, ref.act(E.obs().astype(np.float64)), rtol=2e-3, atol=2e-3)
    rsum = 0.0
    for t in range(150):
        pol.act(T['obs'], out=T['actions'])
        E.step()
        rsum += float(T['reward'].mean())
    assert rsum / 150 > 0.7, rsum / 150
    E.close()
