"""The reference's trained PMC policy (networks/legged_robot/pmc_net/pmc_net.py:117-178, :99-114) evaluated ON THE DEVICE, straight
on the engine's obs buffer and into its action buffer (zero copies): the caller-side neighbour of the hot path.

Plain library GEMMs (torch.matmul -> hipBLASLt / rocBLAS) -- seven small matrix products per step, all MFMA work; a fused
hand-written kernel is a later step.  `oracle/pmc_policy.py` is the NumPy statement of the same forward pass."""
import os

import numpy as np
import torch

DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'pmc_policy.npz')


class TorchPmcPolicy(object):
    def __init__(self, npz_path=DEFAULT_WEIGHTS, device=None, dtype=torch.float32):
        z = np.load(npz_path)
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.w = [torch.as_tensor(z['w%02d' % i].astype(np.float32), device=dev).to(dtype) for i in range(28)]
        w = self.w
        self.cb = w[16]                                         # (32, 256) codebook, pmc_net.py:148-157
        self.cb_sq = (self.cb.float() ** 2).sum(0, keepdim=True)
        self.cb_t = self.cb.t().contiguous()
        self.dtype = dtype

    @torch.no_grad()
    def act(self, obs, out=None):
        """obs [N, 207] (prop 99 | prop_a 36 | future 72) on the device -> mean action [N, 12] (written into `out` if given)."""
        w = self.w
        x = obs.to(self.dtype)
        prop = torch.clamp((x[:, :135] - w[0]) / (w[1] + 1e-8), -5.0, 5.0)          # layers.py:55 + pmc_net.py:131-135
        fut = torch.clamp((x[:, 135:] - w[2]) / (w[3] + 1e-8), -5.0, 5.0)
        h = torch.relu(torch.addmm(w[11], torch.cat([prop, fut], 1), w[10]))
        h = torch.relu(torch.addmm(w[13], h, w[12]))
        ze = torch.addmm(w[15], h, w[14])                                         # vq_encoder pmc_net.py:41-46
        score = 2.0 * (ze @ self.cb).float() - self.cb_sq                          # argmin |ze - code|^2, pmc_net.py:155-157
        q = self.cb_t[torch.argmax(score, 1)]
        s = torch.cat([torch.relu(torch.addmm(w[18], prop, w[17])), torch.relu(torch.addmm(w[20], q, w[19]))], 1)   # llc pmc_net.py:99-108
        h = torch.relu(torch.addmm(w[22], s, w[21]))
        h = torch.relu(torch.addmm(w[24], h, w[23]))
        a = torch.addmm(w[26], h, w[25]).float()
        if out is not None:
            out.copy_(a)
            return out
        return a
