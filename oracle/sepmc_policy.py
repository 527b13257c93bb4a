"""TEST INFRASTRUCTURE ONLY (never imported by the product).

NumPy forward pass of the reference's trained SEPMC policy (data/models/strategic_level.model, key 'fused_hml'), the actor side of
test_scripts/strategic_level/test_strategic_level_env.py (`agent.step(ob, argmax=True)` for both robots): two robots trained against PyBullet
chase each other in OUR arena -- the only Bullet-facing check of the robot-robot contact model (SURVEY.md 8f-3 for the strategic level).

Restated from networks/legged_robot/sepmc_net/sepmc_net.py: hlc_encoder :120-149 (prop fc | percept convs fc | fc(fc(percept_vec, oppo_info,
flag_info, with_flag)) -> fc 256 -> LSTM(32) -> heading angle, clipped to +-pi; argmax = the mean), :279-291 (target_info = cos, sin of the
heading and the commanded speed), mlc_encoder :174-203 (the EPMC encoder with target_info as its vector feature -> LSTM(32) -> 256-way code),
mapping_z, llc (pmc_net.py:99-114); policy_config of the test script.  Building blocks (conv SAME padding, layer norm, the tpolicies LSTM and
what is and is not confirmed about it): see oracle/epmc_policy.py, whose functions this file uses.
Weights: tests/golden/sepmc_policy.npz = arrays 0, 1 and 51..151 of the checkpoint's 152 (tools/extract_epmc_policy.py); the value branch 2..50
is not needed.  obs rows: prop 99 | prop_a 36 | percept_2d 325 | percept_1d 128 | percept_front 325 | percept_vec 5 | oppo_info 15 |
oppo_info_cheat 15 | flag_info 7 | flag_info_cheat 7 | with_flag 2 | control_spd 1 (CTG:349-363)."""
import numpy as np

from .epmc_policy import conv1d, conv2d, layer_norm, sigmoid


class SepmcPolicy(object):
    def __init__(self, npz_path, n_rows, forget_bias=1.0):
        z = np.load(npz_path)
        self.w = {int(k[1:]): z[k].astype(np.float64) for k in z.files}
        self.fb = forget_bias
        self.c = {k: np.zeros((n_rows, 32)) for k in ('hlc', 'z')}
        self.h = {k: np.zeros((n_rows, 32)) for k in ('hlc', 'z')}
        self.last_heading = np.zeros(n_rows)

    def reset(self, rows=None):
        for d in (self.c, self.h):
            for k in d:
                if rows is None:
                    d[k][:] = 0
                else:
                    d[k][rows] = 0

    def _lstm(self, key, x, k0):
        w = self.w
        zz = layer_norm(x @ w[k0], w[k0 + 3], w[k0 + 4]) + layer_norm(self.h[key] @ w[k0 + 1], w[k0 + 5], w[k0 + 6]) + w[k0 + 2]
        i, f, o, u = np.split(zz, 4, axis=1)
        self.c[key] = sigmoid(f + self.fb) * self.c[key] + sigmoid(i) * np.tanh(u)
        self.h[key] = sigmoid(o) * np.tanh(layer_norm(self.c[key], w[k0 + 7], w[k0 + 8]))
        return self.h[key]

    def _percepts(self, p2d, p1d, pfr, k):
        """mlc_usr_cmd_encoder's three conv stacks, variables k .. k + 23"""
        w = self.w
        n = p2d.shape[0]

        def enc2d(img, kk):
            e = conv2d(img, w[kk], w[kk + 1])
            e = conv2d(e, w[kk + 2], w[kk + 3], stride=2)
            e = conv2d(e, w[kk + 4], w[kk + 5], stride=2)
            return conv2d(e, w[kk + 6], w[kk + 7]).reshape(n, -1)
        pad = np.concatenate([p1d[:, -4:], p1d, p1d[:, :4]], axis=1)[:, :, None]
        e = conv1d(pad, w[k + 8], w[k + 9])[:, 4:-4, :]
        e = conv1d(e, w[k + 10], w[k + 11], stride=2)
        e = conv1d(e, w[k + 12], w[k + 13], stride=2)
        e1d = conv1d(e, w[k + 14], w[k + 15]).reshape(n, -1)
        return enc2d(p2d, k), e1d, enc2d(pfr, k + 16)

    def act(self, obs):
        """obs [n_rows, 965] -> A_LLC mean [n_rows, 12]"""
        w = self.w
        relu = lambda x: np.maximum(x, 0.0)
        obs = np.asarray(obs, np.float64)
        n = obs.shape[0]
        prop = obs[:, :135]
        p2d, p1d, pfr = obs[:, 135:460].reshape(n, 25, 13, 1), obs[:, 460:588], obs[:, 588:913].reshape(n, 25, 13, 1)
        vec, oppo, flag, with_flag, spd = obs[:, 913:918], obs[:, 918:933], obs[:, 948:955], obs[:, 962:964], obs[:, 964:965]
        x = np.clip((prop - w[0]) / (w[1] + 1e-8), -5.0, 5.0)
        # hlc_encoder
        e2d, e1d, efr = self._percepts(p2d, p1d, pfr, 53)
        mlc_embed = relu(np.concatenate([e2d, e1d, efr], axis=1) @ w[77] + w[78])
        hu = relu(relu(np.concatenate([vec, oppo, flag, with_flag], axis=1) @ w[79] + w[80]) @ w[81] + w[82])
        embed = relu(np.concatenate([relu(x @ w[51] + w[52]), mlc_embed, hu], axis=1) @ w[83] + w[84])
        heading = np.clip(self._lstm('hlc', embed, 85) @ w[94] + w[95], -np.pi, np.pi)             # argmax of a diagonal Gaussian: its mean
        self.last_heading = heading[:, 0]
        target = np.concatenate([np.cos(heading), np.sin(heading), spd], axis=1)                    # outer_control_spd
        # mlc_encoder
        e2d, e1d, efr = self._percepts(p2d, p1d, pfr, 99)
        usr = relu(np.concatenate([relu(target @ w[123] + w[124]), e2d, e1d, efr], axis=1) @ w[125] + w[126])
        embed = relu(np.concatenate([relu(x @ w[97] + w[98]), usr], axis=1) @ w[127] + w[128])
        code = np.argmax(self._lstm('z', embed, 129) @ w[138] + w[139], axis=1)
        zq = w[140].T[code]
        s = np.concatenate([relu(x @ w[141] + w[142]), relu(zq @ w[143] + w[144])], axis=1)
        hdn = relu(relu(s @ w[145] + w[146]) @ w[147] + w[148])
        return hdn @ w[149] + w[150]
