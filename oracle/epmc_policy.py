"""TEST INFRASTRUCTURE ONLY (never imported by the product).

NumPy forward pass of the reference's trained EPMC policies (data/models/environmental_level_{hurdle,hole,cube}.model), the actor side
of test_scripts/environmental_level/test_environmental_level_env.py:88-110 (`agent.step(obs, argmax=True)`): SURVEY.md 8f-3 for the
environmental level -- a policy trained against PyBullet on hurdles / gaps / stairs drives OUR terrain contacts and OUR 778 analytic rays.

What is restated, and from where:
  * networks/legged_robot/epmc_net/epmc_net.py:117-178 (usr_cmd_encoder, mlc_encoder), :181-199 (mapping_z), :224-262 (epmc_net, the
    self-fed branch), pmc_net.py:99-114 (llc), layers.py:55 (rms), with the policy_config of the test script (relu, discrete_z, z_len 256,
    z_len_llc 32, nlstm 32, lstm_layer_norm, llc_light);
  * tf.contrib.layers.conv2d / conv1d / fully_connected / layer_norm (TensorFlow 1.15, absent here): cross-correlation with SAME padding
    (pad_total = max((ceil(n / s) - 1) s + k - n, 0), the smaller half in front), relu by default, layer_norm over all but the batch axis
    with variance epsilon 1e-12, variables created beta-then-gamma;
  * `tp_layers.lstm_embed_block` lives in the THIRD-PARTY package `tpolicies` (github.com/tencent-ailab/TPolicies, un-pinned in setup.py,
    absent from /root/reference and not installable here).  Its published ops.lstm is OpenAI baselines' lnlstm with a forget bias:
        z = LN_x(x wx) + LN_h(h wh) + b;  i, f, o, u = split(z, 4);  c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(u);
        h' = sigmoid(o) tanh(LN_c(c'))           state = [c | h], forget_bias = EMPCConfig.forget_bias = 1.0
    restated here from memory of that source.  What the checkpoint itself confirms: the variable order of the block is
    [wx (256,128), wh (32,128), b (128), beta_x, gamma_x, beta_h, gamma_h (128 each), beta_c, gamma_c (32 each)] -- three bias-like vectors
    with IDENTICAL values (b, beta_x, beta_h receive identical gradients exactly when z = LN_x + LN_h + b) and gains near 1 in positions
    5, 7, 9 -- i.e. this structure and no other.  What it cannot confirm: the gate order (i, f, o, u) and the forget bias; they are the
    published ones and are flagged as an assumption wherever results of this file are quoted.
Weights: tests/golden/epmc_policy_<element>.npz = arrays 0, 1 and 47..101 of the checkpoint (rms statistics and the policy branch; the value
branch 2..46 is not needed), written by tools/extract_epmc_policy.py.
"""
import numpy as np


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def conv2d(x, w, b, stride=1, act=True):
    """x [N, H, W, Cin], w [kh, kw, Cin, Cout] (tf.contrib.layers.conv2d: SAME, relu)"""
    kh, kw, ci, co = w.shape
    oh, pt, pb = _same_pad(x.shape[1], kh, stride)
    ow, pl, pr = _same_pad(x.shape[2], kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((x.shape[0], oh, ow, co))
    for di in range(kh):
        for dj in range(kw):
            patch = xp[:, di:di + (oh - 1) * stride + 1:stride, dj:dj + (ow - 1) * stride + 1:stride, :]
            out += patch @ w[di, dj]
    out += b
    return np.maximum(out, 0.0) if act else out


def conv1d(x, w, b, stride=1, act=True):
    """x [N, L, Cin], w [k, Cin, Cout]"""
    return conv2d(x[:, None, :, :], w[None], b, stride, act)[:, 0]


def layer_norm(x, beta, gamma):
    m = x.mean(axis=1, keepdims=True)
    v = ((x - m) ** 2).mean(axis=1, keepdims=True)
    return (x - m) / np.sqrt(v + 1e-12) * gamma + beta


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class EpmcPolicy(object):
    """Stateful (LSTM) policy for a batch of envs.  obs rows are the engine's / the reference's flat layout:
    prop 99 | prop_a 36 | percep_2d 325 | percep_1d 128 | percep_front 325 | target 3."""
    GATES = 'ifou'

    def __init__(self, npz_path, n_envs, forget_bias=1.0, gates='ifou'):
        z = np.load(npz_path)
        self.w = {int(k[1:]): z[k].astype(np.float64) for k in z.files}
        self.forget_bias = forget_bias
        self.gates = gates
        self.c = np.zeros((n_envs, 32)); self.h = np.zeros((n_envs, 32))
        self.last_code = np.zeros(n_envs, int)

    def reset(self, ids=None):
        """agent.reset at an episode start: zero hidden state (test_environmental_level_env.py:87, :100)"""
        if ids is None:
            self.c[:] = 0; self.h[:] = 0
        else:
            self.c[ids] = 0; self.h[ids] = 0

    def act(self, obs):
        w = self.w
        relu = lambda x: np.maximum(x, 0.0)
        obs = np.asarray(obs, np.float64)
        n = obs.shape[0]
        prop = obs[:, :135]
        p2d = obs[:, 135:460].reshape(n, 25, 13, 1)
        p1d = obs[:, 460:588]
        pfr = obs[:, 588:913].reshape(n, 25, 13, 1)
        tgt = obs[:, 913:916]
        x = np.clip((prop - w[0]) / (w[1] + 1e-8), -5.0, 5.0)                         # epmc_net.py:232-236, layers.py:55
        prop_embed = relu(x @ w[47] + w[48])                                         # mlc_encoder :148-149

        def enc2d(img, k):                                                            # percep_2d_encoder :88-96
            e = conv2d(img, w[k], w[k + 1])
            e = conv2d(e, w[k + 2], w[k + 3], stride=2)
            e = conv2d(e, w[k + 4], w[k + 5], stride=2)
            e = conv2d(e, w[k + 6], w[k + 7])
            return e.reshape(n, -1)
        e2d = enc2d(p2d, 49)
        pad = np.concatenate([p1d[:, -4:], p1d, p1d[:, :4]], axis=1)[:, :, None]     # periodic_padding_1d :99-108, percep_1d_encoder :111-121
        e = conv1d(pad, w[57], w[58])[:, 4:-4, :]
        e = conv1d(e, w[59], w[60], stride=2)
        e = conv1d(e, w[61], w[62], stride=2)
        e1d = conv1d(e, w[63], w[64]).reshape(n, -1)
        efr = enc2d(pfr, 65)
        vec = relu(tgt @ w[73] + w[74])                                               # usr_cmd_encoder :124-139
        usr = relu(np.concatenate([vec, e2d, e1d, efr], axis=1) @ w[75] + w[76])
        embed = relu(np.concatenate([prop_embed, usr], axis=1) @ w[77] + w[78])       # :150-152
        # tp_layers.lstm_embed_block (see the module docstring)
        zz = layer_norm(embed @ w[79], w[82], w[83]) + layer_norm(self.h @ w[80], w[84], w[85]) + w[81]
        parts = dict(zip(self.gates, np.split(zz, 4, axis=1)))
        i, f, o, u = sigmoid(parts['i']), sigmoid(parts['f'] + self.forget_bias), sigmoid(parts['o']), np.tanh(parts['u'])
        self.c = f * self.c + i * u
        self.h = o * np.tanh(layer_norm(self.c, w[86], w[87]))
        logits = self.h @ w[88] + w[89]                                               # z_logits :160-161; argmax=True
        code = np.argmax(logits, axis=1)
        self.last_code = code
        zq = w[90].T[code]                                                            # mapping_z :181-190
        s = np.concatenate([relu(x @ w[91] + w[92]), relu(zq @ w[93] + w[94])], axis=1)   # llc pmc_net.py:99-108
        hdn = relu(relu(s @ w[95] + w[96]) @ w[97] + w[98])
        return hdn @ w[99] + w[100]
