"""oracle/sepmc_oracle.py against the reference goldens (tests/golden/sepmc_golden.npz, made by importing the reference's
ChaseTagGameEnv through a fake BulletClient: tests/golden/gen_sepmc_golden.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sepmc_oracle as SO  # noqa: E402
import sepmc_parity_common as SC  # noqa: E402


@pytest.fixture(scope='module')
def g():
    return SC.load_golden()


def test_blob_model_matches_urdf_fields(g, model_blob):
    m = SO.BlobModel(model_blob)
    assert np.allclose(m.wheel_pos[:, 2], -0.2115) and np.allclose(np.abs(m.wheel_pos[:, 1])[:2], 0.0355)
    assert m.handle_pos['joint_front_handle'][0] > 0.1 > -0.2 > m.handle_pos['joint_hind_handle'][0]
    # the golden's getLinkStates answers were made from the parsed URDF; the blob must give the same points
    e = 0
    for d in range(3):
        for r in range(2):
            feet, wheels, handles = SO.link_points(g['e_state'][e][r][d], m)
            np.testing.assert_allclose(np.vstack([feet, wheels, handles]), g['e_points'][e][d][r], atol=1e-12)


def test_link_points_feet_agree_with_c_oracle(model_blob, mocap_table):
    from oracle import oracle as O
    ob = O.OracleBatch(O.make_config(1), model_blob, mocap_table)
    rng = np.random.default_rng(0)
    m = SO.BlobModel(model_blob)
    for _ in range(5):
        s = np.concatenate([rng.normal(size=3), (lambda q: q / np.linalg.norm(q))(rng.normal(size=4)), rng.normal(size=6), rng.uniform(-1, 1, 12), rng.normal(size=12)])
        np.testing.assert_allclose(SO.link_points(s, m)[0], ob.fk_feet(s), atol=1e-12)


@pytest.mark.parametrize('case', range(15))
def test_reset_cases(g, model_blob, case):
    SC.check_oracle_reset_case(g, SO.BlobModel(model_blob), case)


@pytest.mark.parametrize('ep', range(4))
def test_scripted_episodes(g, model_blob, ep):
    SC.check_oracle_episode(g, SO.BlobModel(model_blob), ep)
