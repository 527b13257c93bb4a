"""TEST INFRASTRUCTURE ONLY (never imported by the product; the product's policy is the fused HIP kernel, pmc_policy_hip.py).

NumPy forward pass of the reference's trained PMC policy (networks/legged_robot/pmc_net/pmc_net.py:117-178, :99-114):
running-mean/std normalisation clipped to +-5, VQ encoder 207->256->256->32, nearest code of a (32, 256) codebook,
low-level controller  [relu(prop 135->64) | relu(z 32->32)] -> 256 -> 256 -> 12  (mean action).

Not part of the hot path: it exists for the trained-policy sanity run that validates the simulator's physics
 and as the natural next fusion target (policy inference on device)."""
import numpy as np


class PmcPolicy(object):
    def __init__(self, npz_path):
        z = np.load(npz_path)
        self.w = [z['w%02d' % i].astype(np.float64) for i in range(28)]

    def act(self, obs):
        """obs: [N, 207] = prop 99 | prop_a 36 | future 72  ->  mean action [N, 12]"""
        w = self.w
        relu = lambda x: np.maximum(x, 0.0)
        prop, future = obs[:, :135], obs[:, 135:]                       # pmc_net.py:123-127 (append_hist_a)
        prop_rms = np.clip((prop - w[0]) / (w[1] + 1e-8), -5.0, 5.0)     # layers.py:55 + pmc_net.py:131-135
        future_rms = np.clip((future - w[2]) / (w[3] + 1e-8), -5.0, 5.0)
        ob = np.concatenate([prop_rms, future_rms], axis=1)
        h = relu(relu(ob @ w[10] + w[11]) @ w[12] + w[13])
        ze = h @ w[14] + w[15]                                           # vq_encoder pmc_net.py:41-46
        cb = w[16]                                                       # (32, 256)
        dist = (ze ** 2).sum(1, keepdims=True) - 2 * ze @ cb + (cb ** 2).sum(0, keepdims=True)   # pmc_net.py:155-157
        q = cb.T[np.argmax(-dist, 1)]
        s = np.concatenate([relu(prop_rms @ w[17] + w[18]), relu(q @ w[19] + w[20])], axis=1)     # llc pmc_net.py:99-108
        h = relu(relu(s @ w[21] + w[22]) @ w[23] + w[24])
        return h @ w[25] + w[26]                                         # decoder mean

    def value(self, obs):
        """vf head (pmc_net.py:141-146): tanh(207 -> 256) -> tanh(256) -> 1 on the normalised observation"""
        w = self.w
        prop, future = obs[:, :135], obs[:, 135:]
        ob = np.concatenate([np.clip((prop - w[0]) / (w[1] + 1e-8), -5.0, 5.0), np.clip((future - w[2]) / (w[3] + 1e-8), -5.0, 5.0)], axis=1)
        return (np.tanh(np.tanh(ob @ w[4] + w[5]) @ w[6] + w[7]) @ w[8] + w[9])[:, 0]

    def neglogp(self, obs, a):
        """DiagGaussianPd.neglogp of action a under the policy's head (mean = act(obs), logstd = w27; pmc_net.py:109-113)"""
        logstd = self.w[27].reshape(1, 12)
        return 0.5 * (((a - self.act(obs)) / np.exp(logstd)) ** 2).sum(1) + 0.5 * np.log(2.0 * np.pi) * 12 + logstd.sum()
