"""The env_config dicts of the reference's training scripts (example_epmc_train.sh:90-117, example_sepmc_train.sh:93-117) for the tools."""


def epmc_env_config(element_id):
    return {'arena_id': 'Playground', 'render': False, 'control_freq': 50.0,
            'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
            'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'element_id': element_id, 'height_range': [0.0, 0.0], 'friction_range': [0.4, 3.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
                                     'cmd_vary_freq_range': [9999, 10000], 'target_spd_range': [0.5, 3.0], 'auxiliary_radius': 0.02,
                                     'hole_config': {'min_gap_height': 0.25, 'max_gap_height': 0.25}}}

def sepmc_env_config(elements):
    return {'arena_id': 'CTG', 'render': False, 'control_freq': 50.0,
            'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
            'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'friction_range': [0.4, 3.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]}},
            'element_config': {'rand_cube': bool(elements), 'hurdle': bool(elements), 'hole': bool(elements)}}


