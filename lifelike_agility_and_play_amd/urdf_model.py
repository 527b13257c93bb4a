"""URDF -> articulated model tables for the batched MAX-quadruped stepper.

Host-side replacement for the work PyBullet does in ``loadURDF(max.urdf, flags=
URDF_MAINTAIN_LINK_ORDER | URDF_USE_SELF_COLLISION | ...)`` at the reference call
site legged_robot.py:208-220 (dynamic robot) and :267-275 (kinematic ghost):
tree topology, joint frames/axes, link mass/COM/inertia, joint limits/damping and
the primitive collision shapes.

Link inertias come in two flavours (``UrdfModel(path, inertia=...)``):
  * ``'collision_aabb'`` (what the reference's call builds, and the shipped default since round 4): the
    reference does NOT pass ``URDF_USE_INERTIA_FROM_FILE`` (legged_robot.py:212-217), and without it Bullet's
    importer keeps only mass and inertial frame of the URDF ``<inertial>`` and takes the inertia diagonal from
    ``btCompoundShape::calculateLocalInertia``: the *box inertia of the AABB of the link's collision shapes*,
    measured in the link's inertial frame, applied at the URDF COM (see ``aabb_box_inertia``);
  * ``'file'``: the ``<inertia>`` tensors of the URDF (``URDF_USE_INERTIA_FROM_FILE`` behaviour; rounds 1-3).

Differences from Bullet's importer, all dynamically equivalent:
  * fixed joints (feet, wheels, handles) are merged into their parent body
    -> 13 bodies / 12 revolute joints (Bullet keeps 23 links, 10 of them welded);
  * the base frame F0 is the *inertial* frame of the URDF root link, because that is
    what ``getBasePositionAndOrientation`` / ``resetBasePositionAndOrientation``
    (legged_robot.py:77,94) read and write; every base-attached quantity is stored
    relative to it.

The result is a flat float64 blob whose layout is fixed by ``include/llenv.h``
(``ll_model``) so that the same numbers feed the CPU oracle and the HIP kernels.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

LEG_NAMES = ['FR', 'FL', 'HR', 'HL']          # constants.py:175 (also the mocap LegOrder)
N_LEGS, N_LINKS = 4, 3
PRIM_SPHERE, PRIM_BOX, PRIM_CYL = 0, 1, 2

# ---- blob layout (doubles).  Keep in sync with include/llenv.h ------------------
# per-leg contact primitive slots, identical structure on all four legs (asserted)
#   hip   : cyl
#   thigh : box, cyl, cyl, wheel cyl
#   shank : box, foot sphere
LEG_PRIM_SLOTS = [(0, PRIM_CYL), (1, PRIM_BOX), (1, PRIM_CYL), (1, PRIM_CYL), (1, PRIM_CYL),
                  (2, PRIM_BOX), (2, PRIM_SPHERE)]
N_LEG_PRIMS = len(LEG_PRIM_SLOTS)
N_BASE_PRIMS = 3          # body box + two handle spheres
PRIM_STRIDE = 16          # type, size[3], pos[3], rot[9] (row-major, prim->body)

OFF_BASE_MASS = 0
OFF_BASE_COM = 1          # 3   composite COM in F0
OFF_BASE_INERTIA = 4      # 9   about composite COM, F0 axes
OFF_JOINT_ORIGIN = 13     # 12*3  in parent frame (hips: relative to F0 origin)
OFF_JOINT_AXIS = 49       # 12*3
OFF_LINK_MASS = 85        # 12
OFF_LINK_COM = 97         # 12*3
OFF_LINK_INERTIA = 133    # 12*9  about COM, link axes
OFF_Q_LO = 241            # 12
OFF_Q_HI = 253            # 12
OFF_DAMPING = 265         # 12
OFF_FOOT_POS = 277        # 4*3  foot link origin in shank frame (legged_robot.py:199-205 FK target)
OFF_BASE_PRIMS = 289      # N_BASE_PRIMS*PRIM_STRIDE
OFF_LEG_PRIMS = OFF_BASE_PRIMS + N_BASE_PRIMS * PRIM_STRIDE       # 4*N_LEG_PRIMS*PRIM_STRIDE
OFF_BASE_LINK_OFFSET = OFF_LEG_PRIMS + N_LEGS * N_LEG_PRIMS * PRIM_STRIDE   # 3: URDF root link origin in F0
MODEL_BLOB_LEN = OFF_BASE_LINK_OFFSET + 3


def _floats(s, n=None, default=None):
    if s is None:
        return np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None:
        assert v.size == n, s
    return v


def rpy_to_mat(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix (R = Rz(y) Ry(p) Rx(r))."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(elem):
    o = elem.find('origin') if elem is not None else None
    if o is None:
        return np.zeros(3), np.eye(3)
    return _floats(o.get('xyz'), 3, [0, 0, 0]), rpy_to_mat(_floats(o.get('rpy'), 3, [0, 0, 0]))


# Bullet's collision margins as the importer sets them (URDF importer: ``gUrdfDefaultCollisionMargin`` = 0.001 on
# every child shape and on the link's btCompoundShape).
URDF_COLLISION_MARGIN = 0.001


def prim_aabb_half_extents(prim, rot_to_frame):
    """Half extents of one collision primitive's AABB in a frame whose axes are ``rot_to_frame`` @ (primitive axes),
    as Bullet's shape classes report it (recalled from the published source, PyBullet itself is absent here):
      * btBoxShape: the full half extents (its margin lives inside the box);
      * btSphereShape: the radius;
      * URDF cylinder without URDF_USE_IMPLICIT_CYLINDER (the reference's flags): a btConvexHullShape over a 32-gon
        prism (vertices at multiples of 2*pi/32: the extremes lie ON the coordinate axes, so the prism spans the full
        radius) whose cached local AABB already holds the margin and whose getAabb adds it once more: + 2 margins.
    The local half extents are rotated with the absolute value of the rotation (btTransformAabb)."""
    t, size, _pos, rot = prim
    if t == PRIM_BOX:
        h = np.array(size, dtype=np.float64)
    elif t == PRIM_SPHERE:
        h = np.full(3, float(size[0]))
    elif t == PRIM_CYL:
        h = np.array([size[0], size[0], size[1]], dtype=np.float64) + 2.0 * URDF_COLLISION_MARGIN
    else:
        raise ValueError(t)
    return np.abs(rot_to_frame @ rot) @ h


def aabb_box_inertia(mass, prims, inertial_xyz, inertial_rot):
    """``btCompoundShape::calculateLocalInertia`` for a link: children sit at inertial_frame^-1 * collision_frame, the
    compound's AABB is the union of the children's AABBs grown by the compound's own margin, and the inertia is the
    solid-box formula on the AABB's edge lengths -- diagonal in the inertial frame, used AT the URDF COM wherever the
    AABB's centre lies.  A link without mass keeps zero inertia (the importer skips the call: ``if (mass)``).
    Returns the 3x3 tensor in LINK axes (about the COM)."""
    if mass == 0.0 or not prims:
        return np.zeros((3, 3))
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for p in prims:
        c = inertial_rot.T @ (p[2] - inertial_xyz)
        h = prim_aabb_half_extents(p, inertial_rot.T)
        lo, hi = np.minimum(lo, c - h), np.maximum(hi, c + h)
    l = (hi - lo) + 2.0 * URDF_COLLISION_MARGIN
    d = mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])
    return inertial_rot @ np.diag(d) @ inertial_rot.T


class _Link:
    def __init__(self, elem):
        self.name = elem.get('name')
        ine = elem.find('inertial')
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))      # about COM, link axes
        self.inertial_rot = np.eye(3)
        if ine is not None:
            xyz, rot = _origin(ine)
            self.mass = float(ine.find('mass').get('value'))
            i = ine.find('inertia')
            g = lambda k: float(i.get(k, 0.0))
            I = np.array([[g('ixx'), g('ixy'), g('ixz')], [g('ixy'), g('iyy'), g('iyz')], [g('ixz'), g('iyz'), g('izz')]])
            self.com = xyz
            self.inertia = rot @ I @ rot.T
            self.inertial_rot = rot
        self.prims = []                        # (type, size[3], pos[3], rot[3x3]) in link frame
        for c in elem.findall('collision'):
            xyz, rot = _origin(c)
            geo = c.find('geometry')
            if geo.find('sphere') is not None:
                self.prims.append((PRIM_SPHERE, np.array([float(geo.find('sphere').get('radius')), 0, 0]), xyz, rot))
            elif geo.find('box') is not None:
                self.prims.append((PRIM_BOX, _floats(geo.find('box').get('size'), 3) * 0.5, xyz, rot))
            elif geo.find('cylinder') is not None:
                cy = geo.find('cylinder')
                self.prims.append((PRIM_CYL, np.array([float(cy.get('radius')), float(cy.get('length')) * 0.5, 0]), xyz, rot))
            else:
                raise ValueError('unsupported collision geometry on ' + self.name)
        self.inertia_file = self.inertia
        self.inertia_aabb = aabb_box_inertia(self.mass, self.prims, self.com, self.inertial_rot)


def _merge_inertia(m1, c1, I1, m2, c2, I2):
    """Composite of two rigid bodies given (mass, com, inertia-about-com) in one frame."""
    m = m1 + m2
    if m == 0.0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = (m1 * c1 + m2 * c2) / m

    def shift(mm, cc, II):
        d = cc - c
        return II + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, shift(m1, c1, I1) + shift(m2, c2, I2)


class UrdfModel:
    """Compiled 13-body model.  Attribute names mirror the blob fields."""

    def __init__(self, urdf_path, inertia='collision_aabb'):
        assert inertia in ('file', 'collision_aabb'), inertia
        self.inertia_source = inertia
        root = ET.parse(urdf_path).getroot()
        links = {l.get('name'): _Link(l) for l in root.findall('link')}
        for l in links.values():                  # per link, BEFORE the fixed joints are merged (Bullet keeps 23 links)
            l.inertia = l.inertia_aabb if inertia == 'collision_aabb' else l.inertia_file
        self.urdf_links = {n: dict(mass=l.mass, com=l.com.copy(), inertia=l.inertia.copy()) for n, l in links.items()}
        joints = []
        for j in root.findall('joint'):
            xyz, rot = _origin(j)
            lim = j.find('limit')
            dyn = j.find('dynamics')
            joints.append(dict(
                name=j.get('name'), type=j.get('type'), parent=j.find('parent').get('link'),
                child=j.find('child').get('link'), xyz=xyz, rot=rot,
                axis=_floats(j.find('axis').get('xyz'), 3) if j.find('axis') is not None else np.array([1., 0, 0]),
                lo=float(lim.get('lower')) if lim is not None else 0.0,
                hi=float(lim.get('upper')) if lim is not None else 0.0,
                damping=float(dyn.get('damping', 0.0)) if dyn is not None else 0.0))
        self.joint_names_urdf_order = [j['name'] for j in joints]       # legged_robot.py:221-229 index discovery
        children = {j['child'] for j in joints}
        roots = [n for n in links if n not in children]
        assert len(roots) == 1
        base = links[roots[0]]
        self.total_mass = sum(l.mass for l in links.values())

        # -- merge fixed joints into their parents (repeat until none left) ------------
        fixed = [j for j in joints if j['type'] == 'fixed']
        while fixed:
            progressed = False
            for j in list(fixed):
                if any(k['parent'] == j['child'] for k in joints if k is not j):
                    continue                                  # merge leaves first
                p, c = links[j['parent']], links[j['child']]
                cc = j['xyz'] + j['rot'] @ c.com
                Ic = j['rot'] @ c.inertia @ j['rot'].T
                p.mass, p.com, p.inertia = _merge_inertia(p.mass, p.com, p.inertia, c.mass, cc, Ic)
                for (t, size, pos, rot) in c.prims:
                    p.prims.append((t, size, j['xyz'] + j['rot'] @ pos, j['rot'] @ rot))
                if not hasattr(p, 'merged'):
                    p.merged = {}
                p.merged[j['name']] = (j['xyz'].copy(), j['rot'].copy())
                joints.remove(j)
                fixed.remove(j)
                progressed = True
            assert progressed, 'cyclic fixed joints'
        rev = {j['name']: j for j in joints}
        assert all(j['type'] == 'revolute' for j in joints) and len(joints) == 12

        # -- base: frame F0 = root link inertial frame (PyBullet base pose convention) ---
        root_elem = [l for l in root.findall('link') if l.get('name') == base.name][0]
        c_b, rot_b = _origin(root_elem.find('inertial'))
        assert np.allclose(rot_b, np.eye(3)), 'rotated root inertial frame not supported'
        self.base_link_offset = -c_b                          # URDF root link origin expressed in F0
        self.base_mass = base.mass
        self.base_com = base.com - c_b
        self.base_inertia = base.inertia
        self.handle_pos = {k: v[0] - c_b for k, v in getattr(base, 'merged', {}).items()}

        self.joint_origin = np.zeros((12, 3))
        self.joint_axis = np.zeros((12, 3))
        self.link_mass = np.zeros(12)
        self.link_com = np.zeros((12, 3))
        self.link_inertia = np.zeros((12, 3, 3))
        self.q_lo = np.zeros(12)
        self.q_hi = np.zeros(12)
        self.damping = np.zeros(12)
        self.foot_pos = np.zeros((4, 3))
        self.wheel_pos = np.zeros((4, 3))                     # origin of the passive wheel link (joint_<leg>W) in the thigh frame
        self.joint_names = []
        leg_prims = []
        for l, leg in enumerate(LEG_NAMES):
            parent = base.name
            for k in range(3):
                j = rev['joint_%s%d' % (leg, k + 1)]
                assert j['parent'] == parent, (j['name'], j['parent'], parent)
                assert np.allclose(j['rot'], np.eye(3)), 'rotated revolute joint frames not supported'
                i = 3 * l + k
                ln = links[j['child']]
                self.joint_names.append(j['name'])
                self.joint_origin[i] = j['xyz'] - (c_b if k == 0 else 0.0)
                ax = j['axis'] / np.linalg.norm(j['axis'])
                assert abs(abs(ax).max() - 1.0) < 1e-12, 'joint axes must be coordinate axes'
                self.joint_axis[i] = ax
                self.link_mass[i], self.link_com[i], self.link_inertia[i] = ln.mass, ln.com, ln.inertia
                self.q_lo[i], self.q_hi[i], self.damping[i] = j['lo'], j['hi'], j['damping']
                parent = j['child']
            self.foot_pos[l] = links['link_%s3' % leg].merged['joint_%s4' % leg][0]
            self.wheel_pos[l] = links['link_%s2' % leg].merged['joint_%sW' % leg][0]
            # order this leg's primitives into the fixed slot structure
            slots = []
            pools = [list(links['link_%s%d' % (leg, k + 1)].prims) for k in range(3)]
            for (k, t) in LEG_PRIM_SLOTS:
                idx = [n for n, p in enumerate(pools[k]) if p[0] == t]
                assert idx, 'leg %s link %d lacks primitive type %d' % (leg, k, t)
                slots.append(pools[k].pop(idx[0]))
            assert all(len(p) == 0 for p in pools), 'unexpected extra collision primitives on leg ' + leg
            leg_prims.append(slots)
        self.leg_prims = leg_prims
        bp = [(t, s, p - c_b, r) for (t, s, p, r) in base.prims]
        assert [p[0] for p in bp] == [PRIM_BOX, PRIM_SPHERE, PRIM_SPHERE], [p[0] for p in bp]
        self.base_prims = bp

    # ------------------------------------------------------------------------------
    def blob(self):
        b = np.zeros(MODEL_BLOB_LEN, dtype=np.float64)
        b[OFF_BASE_MASS] = self.base_mass
        b[OFF_BASE_COM:OFF_BASE_COM + 3] = self.base_com
        b[OFF_BASE_INERTIA:OFF_BASE_INERTIA + 9] = self.base_inertia.ravel()
        b[OFF_JOINT_ORIGIN:OFF_JOINT_ORIGIN + 36] = self.joint_origin.ravel()
        b[OFF_JOINT_AXIS:OFF_JOINT_AXIS + 36] = self.joint_axis.ravel()
        b[OFF_LINK_MASS:OFF_LINK_MASS + 12] = self.link_mass
        b[OFF_LINK_COM:OFF_LINK_COM + 36] = self.link_com.ravel()
        b[OFF_LINK_INERTIA:OFF_LINK_INERTIA + 108] = self.link_inertia.ravel()
        b[OFF_Q_LO:OFF_Q_LO + 12] = self.q_lo
        b[OFF_Q_HI:OFF_Q_HI + 12] = self.q_hi
        b[OFF_DAMPING:OFF_DAMPING + 12] = self.damping
        b[OFF_FOOT_POS:OFF_FOOT_POS + 12] = self.foot_pos.ravel()

        def put(off, prim):
            t, size, pos, rot = prim
            b[off] = t
            b[off + 1:off + 4] = size
            b[off + 4:off + 7] = pos
            b[off + 7:off + 16] = rot.ravel()
        for n, p in enumerate(self.base_prims):
            put(OFF_BASE_PRIMS + n * PRIM_STRIDE, p)
        for l in range(4):
            for n, p in enumerate(self.leg_prims[l]):
                put(OFF_LEG_PRIMS + (l * N_LEG_PRIMS + n) * PRIM_STRIDE, p)
        b[OFF_BASE_LINK_OFFSET:OFF_BASE_LINK_OFFSET + 3] = self.base_link_offset
        return b


def model_blob(inertia='collision_aabb'):
    """A compiled MAX model shipped with the package: ``assets/max_model.npy`` (link inertias as the reference's
    ``loadURDF`` flags make Bullet build them) or ``assets/max_model_file_inertia.npy`` (the URDF's ``<inertia>``
    tensors: the ``URDF_USE_INERTIA_FROM_FILE`` robot, shipped through round 3; kept as the A/B leg)."""
    import os
    names = {'collision_aabb': 'max_model.npy', 'file': 'max_model_file_inertia.npy'}
    if inertia not in names:
        raise ValueError("inertia source %r: one of %s (LL_MODEL_INERTIA selects it for default_model_blob())" % (inertia, sorted(names)))
    name = names[inertia]
    return np.load(os.path.join(os.path.dirname(__file__), 'assets', name))


def default_model_blob():
    """The model every env uses unless told otherwise.  ``LL_MODEL_INERTIA=file`` selects the A/B leg.
    ``LL_MODEL_NO_WHEELS=1`` (an EXPERIMENT leg, round 6: the third of the bars policy's budgeted experiments) shrinks the four passive wheel cylinders at the knees to
    a millimetre, so that they never touch anything before the thigh's other shapes do: collision shapes only -- the inertias were compiled before."""
    import os
    b = model_blob(default_inertia_source())
    if os.environ.get('LL_MODEL_NO_WHEELS') == '1':
        b = b.copy()
        for l in range(N_LEGS):
            off = OFF_LEG_PRIMS + (l * N_LEG_PRIMS + 4) * PRIM_STRIDE            # LEG_PRIM_SLOTS[4]: the wheel cylinder on the thigh
            assert b[off] == PRIM_CYL
            b[off + 1:off + 4] = 1.0e-3
    return b


def default_inertia_source():
    """'collision_aabb' (the default) or 'file', as LL_MODEL_INERTIA says -- recorded in the bench line and the rollout tools so that A/B legs cannot be mixed up"""
    import os
    return os.environ.get('LL_MODEL_INERTIA', 'collision_aabb')
