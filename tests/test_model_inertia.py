"""Link inertias as the reference's loadURDF call makes Bullet build them (round-3 review, "What's weak" #1).

legged_robot.py:208-220 loads max.urdf WITHOUT URDF_USE_INERTIA_FROM_FILE (tests/test_pmc_config_golden.py pins the flag word).  Bullet's
importer then keeps only mass and inertial frame of <inertial> and takes the inertia diagonal from btCompoundShape::calculateLocalInertia:
the solid-box formula on the AABB of the link's collision shapes in the inertial frame (children grown by their margins, the compound by its
own 1 mm), applied at the URDF COM.  These tests hold `urdf_model.aabb_box_inertia`, the shipped blob and the oracle's rigid-body sums to
numbers computed BY HAND from the URDF's dimensions (typed in below), independent of the code under test."""
import os

import numpy as np
import pytest

from lifelike_agility_and_play_amd import urdf_model as um

URDF = '/root/reference/src/lifelike/sim_envs/pybullet_envs/legged_robot/data/urdf/max.urdf'
RX90 = um.rpy_to_mat([np.pi / 2, 0, 0])
RY90 = um.rpy_to_mat([0, np.pi / 2, 0])


def box(m, lx, ly, lz):
    return m / 12.0 * np.array([ly * ly + lz * lz, lx * lx + lz * lz, lx * lx + ly * ly])


def test_aabb_box_inertia_by_hand():
    """Trunk, hip, front shank, hind wheel: edge lengths written out by hand (URDF sizes + Bullet's margins), then the box formula."""
    I3, z3 = np.eye(3), np.zeros(3)
    # trunk: box 0.283 x 0.205 x 0.11 (URDF:8-30); a btBoxShape's AABB is the box itself; compound margin 1 mm per side
    got = um.aabb_box_inertia(5.76855, [(um.PRIM_BOX, np.array([0.1415, 0.1025, 0.055]), z3, I3)], np.array([-0.00458458, 0.00306684, 0.00947989]), I3)
    np.testing.assert_allclose(np.diag(got), box(5.76855, 0.285, 0.207, 0.112), rtol=1e-12)
    np.testing.assert_allclose(np.diag(got), [0.0266281, 0.0450759, 0.0596439], rtol=2e-6)          # vs 0.0478 / 0.1058 / 0.1210 in the file
    assert np.abs(got - np.diag(np.diag(got))).max() == 0.0
    # hip: cylinder r 0.047, length 0.0255, axis turned onto y (rpy pi/2 0 0): convex hull, margin counted twice (cached AABB + getAabb), + compound
    got = um.aabb_box_inertia(0.60318142, [(um.PRIM_CYL, np.array([0.047, 0.01275, 0]), z3, RX90)], np.array([-0.0028696, 0.00034586, -0.00046692]), I3)
    np.testing.assert_allclose(np.diag(got), box(0.60318142, 0.100, 0.0315, 0.100), rtol=1e-9)
    # front shank: box 0.24 x 0.02 x 0.018 turned so that its long axis is z (rpy 0 pi/2 0), centred at z = -0.105
    got = um.aabb_box_inertia(0.16887830, [(um.PRIM_BOX, np.array([0.12, 0.01, 0.009]), np.array([0, 0, -0.105]), RY90)], np.array([0.0015491, 0.0, -0.1244163]), I3)
    np.testing.assert_allclose(np.diag(got), box(0.16887830, 0.020, 0.022, 0.242), rtol=1e-9)
    assert got[2, 2] < 0.4 * 3.41e-5                                                              # 0.36 x the file's value about the long axis
    # a link without mass keeps zero inertia (the importer skips the call): feet
    assert not um.aabb_box_inertia(0.0, [(um.PRIM_SPHERE, np.array([0.025, 0, 0]), z3, I3)], z3, I3).any()
    # the AABB is taken in the INERTIAL frame: turn that frame by 90 degrees about z and x / y swap
    got = um.aabb_box_inertia(1.0, [(um.PRIM_BOX, np.array([0.1, 0.2, 0.3]), z3, I3)], z3, um.rpy_to_mat([0, 0, np.pi / 2]))
    np.testing.assert_allclose(np.diag(got), box(1.0, 0.202, 0.402, 0.602), rtol=1e-9)               # back in link axes: unchanged, as it must be
    got = um.aabb_box_inertia(1.0, [(um.PRIM_BOX, np.array([0.1, 0.2, 0.3]), z3, I3)], z3, um.rpy_to_mat([0, 0, np.pi / 4]))
    l = 0.3 * np.sqrt(2.0) + 0.002                                                                   # a box seen at 45 degrees has a square AABB
    np.testing.assert_allclose(np.diag(got), box(1.0, l, l, 0.602), rtol=1e-9)


def test_shipped_blob_trunk_is_the_hand_computed_composite():
    """Base fields of assets/max_model.npy = trunk box inertia + the two 1 g handle links welded on (URDF:724-751), by the parallel-axis theorem."""
    b = um.model_blob('collision_aabb')
    c_b = np.array([-0.00458458, 0.00306684, 0.00947989])                       # trunk COM in the root link frame = origin of the base frame F0
    parts = [(5.76855, np.zeros(3), np.diag(box(5.76855, 0.285, 0.207, 0.112)))]
    for xyz in ([0.1415, 0.0, 0.085], [-0.2485, 0.0, 0.085]):                   # sphere r 1 mm: AABB edge 2 mm + compound margin = 4 mm
        parts.append((0.001, np.array(xyz) - c_b, np.diag(box(0.001, 0.004, 0.004, 0.004))))
    m = sum(p[0] for p in parts)
    com = sum(p[0] * p[1] for p in parts) / m
    I = np.zeros((3, 3))
    for (mm, cc, II) in parts:
        d = cc - com
        I += II + mm * (d @ d * np.eye(3) - np.outer(d, d))
    assert abs(b[um.OFF_BASE_MASS] - m) < 1e-12
    np.testing.assert_allclose(b[um.OFF_BASE_COM:um.OFF_BASE_COM + 3], com, atol=1e-12)
    np.testing.assert_allclose(b[um.OFF_BASE_INERTIA:um.OFF_BASE_INERTIA + 9].reshape(3, 3), I, atol=1e-12)
    f = um.model_blob('file')
    ratio = np.diag(b[um.OFF_BASE_INERTIA:um.OFF_BASE_INERTIA + 9].reshape(3, 3)) / np.diag(f[um.OFF_BASE_INERTIA:um.OFF_BASE_INERTIA + 9].reshape(3, 3))
    np.testing.assert_allclose(ratio, [0.557, 0.427, 0.493], atol=2e-3)         # the trunk Bullet builds has about half the file's inertia
    # the two blobs differ in inertia tensors and in nothing else
    same = np.ones(len(b), bool)
    same[um.OFF_BASE_INERTIA:um.OFF_BASE_INERTIA + 9] = False
    same[um.OFF_LINK_INERTIA:um.OFF_LINK_INERTIA + 108] = False
    np.testing.assert_array_equal(b[same], f[same])
    assert os.environ.get('LL_MODEL_INERTIA') or np.array_equal(um.default_model_blob(), b)


def _composite_about_com(blob):
    """Whole-robot mass, COM and inertia at q = 0 from the blob's fields with plain NumPy (no code shared with oracle or kernel)."""
    bodies = [(blob[um.OFF_BASE_MASS], blob[um.OFF_BASE_COM:um.OFF_BASE_COM + 3], blob[um.OFF_BASE_INERTIA:um.OFF_BASE_INERTIA + 9].reshape(3, 3))]
    for l in range(4):
        origin = np.zeros(3)
        for k in range(3):
            i = 3 * l + k
            origin = origin + blob[um.OFF_JOINT_ORIGIN + 3 * i:um.OFF_JOINT_ORIGIN + 3 * i + 3]        # all joint frames are unrotated, q = 0
            bodies.append((blob[um.OFF_LINK_MASS + i], origin + blob[um.OFF_LINK_COM + 3 * i:um.OFF_LINK_COM + 3 * i + 3],
                           blob[um.OFF_LINK_INERTIA + 9 * i:um.OFF_LINK_INERTIA + 9 * i + 9].reshape(3, 3)))
    m = sum(x[0] for x in bodies)
    com = sum(x[0] * x[1] for x in bodies) / m
    I = np.zeros((3, 3))
    for (mm, cc, II) in bodies:
        d = cc - com
        I += II + mm * (d @ d * np.eye(3) - np.outer(d, d))
    return m, com, I


@pytest.mark.parametrize('kind', ['collision_aabb', 'file'])
def test_oracle_momentum_is_composite_inertia_times_omega(orc, mocap_table, kind):
    """The oracle's rigid-body sums see the blob's inertias: a robot with straight legs turning as one body about its base origin has
    angular momentum (about its COM) I_composite . omega and linear momentum m omega x r_com."""
    from conftest import make_oracle_batch
    blob = um.model_blob(kind)
    B = make_oracle_batch(orc, blob, mocap_table)
    m, com, I = _composite_about_com(blob)
    assert abs(m - 13.000210501828224) < 1e-9
    for w in (np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0]), np.array([0.3, -0.7, 0.5])):
        s = np.zeros(37)
        s[6] = 1.0
        s[10:13] = w
        P = B.momentum(s)
        lin = m * np.cross(w, com)
        np.testing.assert_allclose(P[0:3], lin, atol=1e-12)
        np.testing.assert_allclose(P[3:6] - np.cross(com, lin), I @ w, atol=1e-12)



def test_composite_inertia_of_the_two_robots():
    """What the change amounts to for the whole robot (straight legs): roll / pitch / yaw inertia about the COM, Bullet's build vs the file's."""
    Ia = _composite_about_com(um.model_blob('collision_aabb'))[2]
    If = _composite_about_com(um.model_blob('file'))[2]
    r = np.diag(Ia) / np.diag(If)
    print('composite inertia, straight legs: aabb', np.diag(Ia), 'file', np.diag(If), 'ratio', r)
    assert (r < 1.0).all() and (r > 0.6).all()             # legs carry most of it by the parallel-axis terms, which do not change


@pytest.mark.skipif(not os.path.exists(URDF), reason='the reference URDF lives only in the build container')
def test_both_blobs_are_what_the_urdf_compiles_to():
    a, f = um.UrdfModel(URDF, 'collision_aabb'), um.UrdfModel(URDF, 'file')
    np.testing.assert_array_equal(a.blob(), um.model_blob('collision_aabb'))
    np.testing.assert_array_equal(f.blob(), um.model_blob('file'))
    # per URDF link, before the welds: the ratios the round-3 review quotes
    r = lambda n: np.diag(a.urdf_links[n]['inertia']) / np.diag(f.urdf_links[n]['inertia'])
    np.testing.assert_allclose(r('body'), [0.557, 0.426, 0.493], atol=2e-3)
    np.testing.assert_allclose(r('link_FR3'), [0.548, 0.542, 0.365], atol=2e-3)
    np.testing.assert_allclose(r('link_HR3'), [0.682, 0.676, 0.324], atol=2e-3)
    assert (r('link_HRW') > 2.3).all() and (r('link_FR1') > 1.3).all() and (r('link_FR2') > 1.3).all()
    for n in ('link_FR4', 'link_FL4', 'link_HR4', 'link_HL4'):
        assert a.urdf_links[n]['mass'] == 0.0 and not a.urdf_links[n]['inertia'].any()
