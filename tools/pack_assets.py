"""Build the data assets shipped with the package from the reference's DATA files.

Runs only in the build container (reads /root/reference); outputs:
  lifelike_agility_and_play_amd/assets/max_model.npy               compiled 13-body model (from max.urdf), link inertias as
                                                                   Bullet builds them WITHOUT URDF_USE_INERTIA_FROM_FILE
                                                                   (the reference's loadURDF flags, legged_robot.py:212-217)
  lifelike_agility_and_play_amd/assets/max_model_file_inertia.npy  the same with the URDF's <inertia> tensors (A/B leg)
  lifelike_agility_and_play_amd/assets/mocap_f64.npz   62 clips packed, float64
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lifelike_agility_and_play_amd import mocap, urdf_model  # noqa: E402

REF = '/root/reference'
URDF = os.path.join(REF, 'src/lifelike/sim_envs/pybullet_envs/legged_robot/data/urdf/max.urdf')
MOCAP = os.path.join(REF, 'data/mocap_data')


def main():
    assets = os.path.join(ROOT, 'lifelike_agility_and_play_amd', 'assets')
    os.makedirs(assets, exist_ok=True)
    np.save(os.path.join(assets, 'max_model.npy'), urdf_model.UrdfModel(URDF, 'collision_aabb').blob())
    np.save(os.path.join(assets, 'max_model_file_inertia.npy'), urdf_model.UrdfModel(URDF, 'file').blob())
    frames, lens, step, names = mocap.load_json_clips(MOCAP)
    mocap.save_packed(os.path.join(assets, 'mocap_f64.npz'), frames, lens, step, names)
    print('packed', len(lens), 'clips', frames.shape, 'frame_step', repr(step))


if __name__ == '__main__':
    main()
