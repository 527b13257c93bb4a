"""lifelike_agility_and_play_amd -- MI355X-native batched rollout engine for the PMC tracking environment of
Tencent-RoboticsX/lifelike-agility-and-play (hot path only).

``--outer_env lifelike_agility_and_play_amd.create_tracking_game`` replaces
``--outer_env lifelike.sim_envs.pybullet_envs.create_tracking_game`` (bin/run_pg_actor.py:81-83); likewise
``create_playground_game`` for the environmental-level (EPMC) env and ``create_chase_tag_game`` for the two-robot strategic-level
(SEPMC) env.
"""
from .envs import BatchedTrackingEnv, TrackingGame, create_tracking_env, create_tracking_game  # noqa: F401
from .playground import BatchedPlaygroundEnv, PlaygroundGame, create_playground_env, create_playground_game  # noqa: F401
from .chase_tag import BatchedChaseTagEnv, ChaseTagGame, create_chase_tag_env, create_chase_tag_game  # noqa: F401

__all__ = ['create_tracking_game', 'create_tracking_env', 'TrackingGame', 'BatchedTrackingEnv',
           'create_playground_game', 'create_playground_env', 'PlaygroundGame', 'BatchedPlaygroundEnv',
           'create_chase_tag_game', 'create_chase_tag_env', 'ChaseTagGame', 'BatchedChaseTagEnv']
