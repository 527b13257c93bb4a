/*
 * llenv_model.h -- layout of the compiled robot-model blob (float64) shared by the URDF compiler
 * (lifelike_agility_and_play_amd/urdf_model.py), the HIP engine and the CPU oracle.
 * Stands for what loadURDF builds inside Bullet at legged_robot.py:208-220.
 *
 * Bodies: 0 = base (URDF `body` + the two welded handles); 1+3*leg+k = leg link k
 * (k = 0 hip, 1 thigh(+wheel), 2 shank(+foot)), legs in LegOrder FR, FL, HR, HL.
 * The base frame F0 sits at the URDF root link's inertial origin (PyBullet base-pose convention).
 */
#ifndef LLENV_MODEL_H
#define LLENV_MODEL_H

#define LLM_PRIM_SPHERE 0
#define LLM_PRIM_BOX 1
#define LLM_PRIM_CYL 2

#define LLM_N_LEGS 4
#define LLM_N_LEG_PRIMS 7   /* hip: cyl | thigh: box, cyl, cyl, wheel-cyl | shank: box, foot sphere */
#define LLM_N_BASE_PRIMS 3  /* body box, front handle sphere, hind handle sphere */
#define LLM_PRIM_STRIDE 16  /* type, size[3], pos[3], rot[9] row-major (prim -> body frame) */
/* size: sphere {r,-,-}; box {hx,hy,hz}; cylinder {r, half-length, -} with its axis along local z */

#define LLM_OFF_BASE_MASS 0
#define LLM_OFF_BASE_COM 1
#define LLM_OFF_BASE_INERTIA 4
#define LLM_OFF_JOINT_ORIGIN 13
#define LLM_OFF_JOINT_AXIS 49
#define LLM_OFF_LINK_MASS 85
#define LLM_OFF_LINK_COM 97
#define LLM_OFF_LINK_INERTIA 133
#define LLM_OFF_Q_LO 241
#define LLM_OFF_Q_HI 253
#define LLM_OFF_DAMPING 265
#define LLM_OFF_FOOT_POS 277
#define LLM_OFF_BASE_PRIMS 289
#define LLM_OFF_LEG_PRIMS (LLM_OFF_BASE_PRIMS + LLM_N_BASE_PRIMS * LLM_PRIM_STRIDE)
#define LLM_OFF_BASE_LINK_OFFSET (LLM_OFF_LEG_PRIMS + LLM_N_LEGS * LLM_N_LEG_PRIMS * LLM_PRIM_STRIDE)
#define LLM_BLOB_LEN (LLM_OFF_BASE_LINK_OFFSET + 3)

/* which body (0 hip, 1 thigh, 2 shank) each leg primitive slot is attached to */
#define LLM_LEG_PRIM_LINKS {0, 1, 1, 1, 1, 2, 2}

/* ---- physics constants of the simulated world (DESIGN.md "physics spec") */
#define LLM_GRAVITY 9.80665            /* LR:260 setGravity(0,0,-9.80665) */
#define LLM_PLANE_FRICTION 0.9         /* plane.urdf:5 lateral_friction */
#define LLM_LINK_FRICTION 0.5          /* Bullet default lateralFriction of every non-foot link */
#define LLM_CONTACT_MARGIN 0.02        /* Bullet contact breaking threshold */
#define LLM_ERP 0.08                   /* ERP of the CONTACT rows: btMultiBodyConstraintSolver::setupMultiBodyContactConstraint takes btContactSolverInfo::m_erp2 ("contactERP"), which
                                          PyBullet's server sets to 0.08 when it creates the world (as recalled; the reference never calls setPhysicsEngineParameter(contactERP=...):
                                          tests/golden/pmc_config_golden.npz).  Rounds 1 - 4: 0.2 with a cap on the push-out speed; round 5 priced both on the five trained policies and
                                          the cap turned out to have been standing in for the softer ERP (profiles/r05_penetration_recovery.md) */
#define LLM_LIMIT_ERP 0.2               /* ERP of the joint-limit rows: btMultiBodyJointLimitConstraint takes btContactSolverInfo::m_erp ("erp", 0.2) while the joint is less than 0.04 rad past its limit */
#define LLM_LIMIT_SPECULATIVE 0         /* LLM_SPEC_LIMIT_SPECULATIVE below: since round 5 a joint-limit row exists only once the limit is passed (btMultiBodyJointLimitConstraint) */
#define LLM_ERP_DEEP (-1.0)             /* LLM_SPEC_ERP_DEEP below; < 0: one ERP at every depth */
#define LLM_LIMIT_ERP_DEEP 0.0          /* LLM_SPEC_LIMIT_ERP_DEEP below: a joint more than 0.04 rad past its limit is stopped, not pushed back (split impulse leaves the positional part unapplied) */
#define LLM_ERP_DEEP_BELOW (-0.04)      /* btContactSolverInfo::m_splitImpulsePenetrationThreshold (see LLM_SPEC_ERP_DEEP) */
#define LLM_MAX_DEPEN_SPEED 1e30       /* m/s: cap on the penetration-recovery part of a contact row's bias.  Bullet has none, and since round 5 neither has this spec (1e30 = off; rounds
                                          1 - 4: 0.5 m/s on top of ERP 0.2 -- it made up for an ERP that was too stiff: profiles/r05_penetration_recovery.md).  The switch stays */
#define LLM_LINK_DAMPING 0.04          /* btMultiBody default linear & angular damping (quirk Q12) */
#define LLM_MAX_CONTACTS_PER_LEG 4     /* contact slots per leg lane */
#define LLM_LIMIT_GATE 20.0             /* a joint-limit row enters the solve iff  s*qd* + bias < this [rad/s] */
#define LLM_SELF_MARGIN 0.01            /* a capsule pair of two legs becomes a (speculative) row within this distance: covers closing
                                           speeds up to 5 m/s per 2 ms substep; the capsules themselves are 35 mm thick */
#define LLM_MAX_SELF 2                  /* self-collision rows per robot */
#define LLM_SELF_FRICTION 0.0           /* LLM_SPEC_SELF_FRICTION below: mu of a leg-leg contact's two tangential rows; 0 = frictionless (Bullet: 0.5 x 0.5 = 0.25) */
#define LLM_PAIR_FRICTION 0.0           /* LLM_SPEC_PAIR_FRICTION below: the same for the robot-robot contacts of a chase-tag arena */
#define LLM_MAX_PAIR 2                  /* LLM_SPEC_MAX_PAIR below: robot-robot rows per robot pair */
#define LLM_LEG_EDGES 0                 /* LLM_SPEC_LEG_EDGES below: 0 = a leg meets terrain with its own points only (vertices, rim points, mid-span spheres); 1 = the XROWS builds' edge rule */
#define LLM_MAX_PAIR_CAP 4              /* ... at most (a persistent manifold holds four points) */
#define LLM_MAX_COORD_VEL 100.0          /* btMultiBody::m_maxCoordinateVelocity (its constructor's value): every generalized velocity -- base twist, joint rates -- is clipped
                                           to +- this after the unconstrained update and after the solve (applyDeltaVeeMultiDof as recalled; a NaN or an infinity is NOT made a bound: it stays
                                           non-finite for the engine's guard, LL_DONE_NONFINITE).  Inert in every gait
                                           (joint rates stay below 35 rad/s); it is what keeps a robot sane that is RESET onto a discontinuity of the mocap data (clip 27 at
                                           7.07 s, clip 8 at 18.90 s: an IK branch flip between two frames = 600 - 750 rad/s by finite differences, ML:48-63): without it
                                           both oracle and engine blow up there (round 4: every non-finite reset of a soak run was one of these) */
#define LLM_FRICTION_MODE 2             /* the two friction rows of a contact are solved together inside the cone |(t1, t2)| <= mu N (LLM_SPEC_FRICTION_MODE
                                           below; btMultiBodyConstraintSolver's published default, resolveConeFrictionConstraintRows).  Rounds 1 - 3 and most
                                           of round 4 shipped the pyramid (mode 0): DESIGN.md 4 has the evidence that moved the default */
#define LLM_SEG_PARALLEL_REG 1e-3        /* closest points of two capsule axes: weight (relative to |d1|^2 |d2|^2) that pulls the parameter of nearly
                                           parallel segments to the middle of their overlap; sin^2(angle) >> this: Ericson's closest point */
#define LLM_OBSTACLE_REACH 1.2          /* m: the jump obstacle of PLE:182-193 takes part in the substeps of a control step that starts with the base
                                           within this horizontal distance of the box centre (robot reach 0.45 m + half box length 0.5 m + slack) */
#define LLM_FLOATING_MIN_Z 0.05          /* m: a terrain box whose bottom is higher than this above the ground FLOATS (the hanging bars of element 2: bullet_static_entities.py:366-412);
                                           the edges it offers the trunk (reverse candidates) are then its BOTTOM edges -- what the flat of the back meets under a bar -- instead of its top edges */
#define LLM_SELECT_EPS 1e-5             /* m: candidates (contact points, capsule pairs) whose depth is within this of the deepest count as equally
                                           deep and the lower index wins -- the deepest-K choice must not hang on float rounding */

/* ---- spec overrides for the deviation study (DESIGN.md 4 "known deviations"): ids of ll_set_spec_param / orc_set_spec_param.
 * Every constant above that is this build's own choice rather than something the reference states can be moved at run time, in the
 * engine and in the oracle alike, so that its effect on the trained reference policy can be measured (tools/deviation_table.py). */
#define LLM_SPEC_LIMIT_GATE 0            /* rad/s   default LLM_LIMIT_GATE; 1e30 = every limit row enters the solve */
#define LLM_SPEC_MAX_DEPEN_SPEED 1       /* m/s     default LLM_MAX_DEPEN_SPEED; 1e30 = uncapped ERP push-out */
#define LLM_SPEC_LINK_DAMPING 2          /* 1/s     default LLM_LINK_DAMPING */
#define LLM_SPEC_MAX_CONTACTS_PER_LEG 3  /* 1..4    default LLM_MAX_CONTACTS_PER_LEG (the deepest-K rule) */
#define LLM_SPEC_SELF_COLLISION 4        /* 0 / 1   leg-leg capsule rows on / off */
#define LLM_SPEC_SELF_MARGIN 5           /* m       default LLM_SELF_MARGIN */
#define LLM_SPEC_MAX_SELF 6              /* 0..2    default LLM_MAX_SELF */
#define LLM_SPEC_ERP 7                   /*         default LLM_ERP */
#define LLM_SPEC_CONTACT_MARGIN 8        /* m       default LLM_CONTACT_MARGIN */
#define LLM_SPEC_SELF_FRICTION 9         /* mu of two tangential rows per leg-leg contact; default LLM_SELF_FRICTION.  Oracle and engine (round 6) */
#define LLM_SPEC_WARM_START 10           /* factor applied to the previous substep's multipliers of persisting rows; default 0 = none.  ORACLE ONLY */
#define LLM_SPEC_TRUNK_EDGES 11         /* 0 / 1   terrain edges under the body box are contact candidates; default 1.  ORACLE ONLY (a test instrument) */
#define LLM_SPEC_SELECT_EPS 12          /* m       default LLM_SELECT_EPS.  ORACLE ONLY (a test instrument: parity cases on the rule's discontinuity) */
/* ---- round 3: Bullet published-algorithm audit.  ORACLE ONLY: each moves the spec towards what btMultiBodyConstraintSolver
 * / btMultiBody do according to their published source, as far as it can be recalled here (PyBullet itself is absent); tools/deviation_table.py
 * prices them with the trained policy, tools/deviation_sepmc.py on chase-tag episodes. */
#define LLM_SPEC_FRICTION_MODE 13       /* default LLM_FRICTION_MODE = 2.  0 (the spec of rounds 1 - 3): all t1 rows, then all t2 rows, box bounds.  1: after all normal rows, the two friction rows of a
                                           contact adjacent (t1_c, t2_c), box bounds.  2: adjacent and solved together from one velocity, clipped to the
                                           cone |(t1, t2)| <= mu * normal (resolveConeFrictionConstraintRows).  3: the spec's rounds with each friction bound
                                           shrunk to what the contact's other row leaves of the cone: the same admissible set in the kernel's round structure.
                                           ENGINE: 2 (the default, LLM_FRICTION_MODE) and 0; every step kernel has both builds (Pmc::gs_cone_round; parity to the
                                           standing bars under either); 1 and 3 exist in the oracle only */
#define LLM_SPEC_ROW_ORDER 14           /* 0 (spec): slot-major (slot 0 of legs 0..3, slot 1, ...).  1: per body pair as a manifold would list them --
                                           contacts sorted by link index, then candidate index */
#define LLM_SPEC_MAX_COORD_VEL 15       /* default LLM_MAX_COORD_VEL = 100 (the spec since round 4; Bullet's value).  1e30: no clip (rounds 1 - 3).  Oracle and engine */
#define LLM_SPEC_LIMIT_ERP 16           /* ERP of the joint-limit rows; < 0 = LLM_SPEC_ERP (btMultiBodyJointLimitConstraint uses the world's m_erp, PyBullet's "erp", 0.2).  Oracle and engine */
#define LLM_SPEC_PAIR_FRICTION 17       /* SEPMC robot-robot rows: mu of two tangential rows per contact; default LLM_PAIR_FRICTION; Bullet: 0.5 x 0.5.  Oracle and engine (round 6) */
#define LLM_SPEC_MAX_PAIR 18            /* SEPMC robot-robot rows per robot pair; default LLM_MAX_PAIR, up to LLM_MAX_PAIR_CAP (a manifold holds four points).  Oracle and engine (round 6) */
#define LLM_SPEC_FRICTION_DIRS 19       /* 0 (spec): friction directions btPlaneSpace1(n), fixed in the world (-y, +x on the ground).  1: the first direction along
                                           the contact point's lateral velocity after the unconstrained update (Bullet's default rule in convertMultiBodyContact
                                           when SOLVER_DISABLE_VELOCITY_DEPENDENT_FRICTION_DIRECTION is not set), the second = t1 x n; btPlaneSpace1 when it
                                           does not slide.  With box bounds a sliding contact then gets at most mu N along its sliding direction instead of
                                           up to sqrt(2) mu N diagonally.  Oracle and engine (ll_set_spec_param) */
#define LLM_SPEC_LIMIT_SPECULATIVE 20   /* 1 (spec of rounds 1 - 4): a joint-limit row exists while the joint is inside its range too, with the free distance d / dt as its bias: the
                                           joint stops AT the limit; rows whose free approach speed exceeds LLM_LIMIT_GATE stay out.  0: a row only once the limit is passed (d <= 0),
                                           bias erp * d / dt, no gate -- what btMultiBodyJointLimitConstraint::createConstraintRows does as recalled ("if (penetration > 0)
                                           continue;"): the joint overshoots by up to qd * dt, is stopped there and walks back by erp per substep.  Oracle and engine
                                           (round 5: every step kernel; a wave none of whose robots is past a limit skips the limit section of the substep) */
#define LLM_SPEC_GYRO 21                /* 1 (spec, btMultiBody::m_useGyroTerm = true as its constructor sets it): the gyroscopic torque w x (I_c w) of every link is part
                                           of the velocity-product forces.  0: left out (Bullet's setUseGyroTerm(false)); the remaining terms -- m w x v_c, the
                                           Coriolis accelerations of the joints -- stay.  ORACLE ONLY (round 4: the third audit item of the bars policy) */
#define LLM_SPEC_FRICTION_KEEP 22       /* 0 (spec): the friction rows of a contact whose normal multiplier is zero are clipped to zero (bound mu * 0).  1: they are
                                           left as they are in that sweep -- btMultiBodyConstraintSolver::solveSingleIteration as recalled guards the friction solve of a
                                           contact with "if (totalImpulse > 0)" --, so friction gathered in earlier iterations of the substep survives a normal row that has
                                           let go.  ORACLE ONLY (round 4, priced in profiles/r04_cone_decision.md) */
#define LLM_SPEC_ERP_DEEP 23            /* ERP of a row whose penetration is deeper than LLM_SPEC_ERP_DEEP_BELOW; < 0 (default) = no second ERP.  Bullet's btContactSolverInfo carries two:
                                           m_erp ("erp", 0.2) and m_erp2 ("contactERP"; PyBullet's server sets 0.08 as recalled); btSequentialImpulseConstraintSolver::setupContactConstraint
                                           and btMultiBodyJointLimitConstraint pick m_erp while penetration > m_splitImpulsePenetrationThreshold (-0.04) and m_erp2 below.  Oracle and engine */
#define LLM_SPEC_ERP_DEEP_BELOW 24      /* m (rad for limit rows): default LLM_ERP_DEEP_BELOW = -0.04 */
#define LLM_SPEC_LIMIT_ERP_DEEP 25      /* ERP of a joint-limit row whose joint is further past its limit than LLM_SPEC_ERP_DEEP_BELOW (0.04 rad); < 0 = LLM_SPEC_ERP_DEEP, or no second ERP.
                                           btMultiBodyJointLimitConstraint::createConstraintRows as recalled: with m_splitImpulse (btContactSolverInfo's default: true) and penetration
                                           below m_splitImpulsePenetrationThreshold the positional part goes to m_rhsPenetration -- which no multibody solver pass applies -- and
                                           m_rhs keeps the velocity part alone: the row stops the joint and does not push it back, i.e. ERP 0.  Oracle and engine */
#define LLM_SPEC_LEG_EDGES 26           /* 1: the top edges of the terrain boxes (bottom edges of floating ones) are contact candidates against the flat faces of the thigh and shank
                                           boxes too, as they always are against the body box; the engine then runs its XROWS builds (cone friction only).  0 (default;
                                           rounds 1 - 5): a leg meets terrain with its own points -- mid-link spheres stand in.  Built and priced in round 6: no policy tells
                                           the two apart, the rule costs 8 - 33 % of a step, and it is ill-conditioned for a robot spawned INTO an arena element.  Oracle and engine */
#define LLM_SPEC_COUNT 27

#endif
