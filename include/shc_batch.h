/*
 * shc_batch.h — C ABI of the MI355X batched leg-control engine (libshc_batch.so).
 *
 * The reference (csiro-robotics/syropod_highlevel_controller, OpenSHC v0.5.11) has NO plugin / FFI
 * boundary: it is one executable whose per-cycle work is the C++ call sequence
 *     PoseController::updateCurrentPose   (src/state_controller.cpp:167)
 *     AdmittanceController::updateStiffness / updateAdmittance   (:177, :179)
 *     WalkController::updateWalk          (:429)
 *     PoseController::updateStance        (:442)
 *     Model::updateModel                  (:445)
 * on ONE robot.  This header is the boundary a maintainer would bind instead: the same
 * quantities, batched over `n_instances` independent robots, plain pointers and sizes only.
 * Each entry point cites the reference interface it replaces.
 *
 * All floating point is IEEE double (the reference is `double` throughout).  All arrays passed
 * through this ABI are HOST or DEVICE pointers as stated per call; layouts are instance-major
 * ("AoS over instances"): element (i, leg, k) of an array with per-leg width K lives at
 * [(i * leg_count + leg) * K + k].  Internally the engine keeps structure-of-arrays state in HBM
 *  and transposes at this boundary.
 */
#ifndef SHC_BATCH_H
#define SHC_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHC_MAX_LEGS 8      /* reference bound: parameters_and_states.h:298 (joint_parameters[8][6]) */
#define SHC_MAX_JOINTS 6    /* reference bound: parameters_and_states.h:298 */
#define SHC_MAX_LINKS 7     /* reference bound: parameters_and_states.h:299 (base link + one per joint) */
#define SHC_MAX_AUTO_POSERS 8
#define SHC_N_BEARINGS 9    /* 0..360 step 45 (model.h:22 BEARING_STEP) */

/* Status codes (the reference has no error returns: ROS_FATAL + shutdown). */
enum {
  SHC_OK = 0,
  SHC_ERR_INVALID_ARG = 1,
  SHC_ERR_NO_DEVICE = 2,      /* no HIP device / kernel image: the product path never falls back to CPU */
  SHC_ERR_HIP = 3,
  SHC_ERR_UNSUPPORTED = 4,    /* outside the accelerated path: gravity_aligned_tips on a robot whose legs differ in DOF, sequences with
                                 own-clock auto posing, resident mode for a batch that does not fit the chip */
  SHC_ERR_UNSTABLE = 5,       /* reserved: the reference aborts when the IMU correction's norm exceeds 100 rad
                                 (pose_controller.cpp:1228-1232); after its own clamps (:1222-1226) that needs
                                 max_rotation > 100 rad, so no entry point returns this code today */
  SHC_ERR_BUSY = 6,           /* the engine is in resident mode (shc_engine_resident_begin): only the shc_engine_resident_*
                                 calls, shc_engine_instances and shc_engine_get_tables are valid until shc_engine_resident_end */
  SHC_ERR_TIMEOUT = 7         /* a bounded wait ran out (resident mode: the device loop stopped by itself or did not answer) */
};

/* parameters_and_states.h:99 */
enum { SHC_WALK_STARTING = 0, SHC_WALK_MOVING = 1, SHC_WALK_STOPPING = 2, SHC_WALK_STOPPED = 3 };
/* parameters_and_states.h:111 */
enum { SHC_STEP_SWING = 0, SHC_STEP_STANCE = 1, SHC_STEP_FORCE_STANCE = 2, SHC_STEP_FORCE_STOP = 3 };
/* parameters_and_states.h:123 */
enum { SHC_POSING = 0, SHC_STOP_POSING = 1, SHC_POSING_COMPLETE = 2 };
/* parameters_and_states.h:150 PoseResetMode */
enum {
  SHC_NO_RESET = 0, SHC_Z_AND_YAW_RESET = 1, SHC_X_AND_Y_RESET = 2, SHC_PITCH_AND_ROLL_RESET = 3,
  SHC_ALL_RESET = 4, SHC_IMMEDIATE_ALL_RESET = 5
};
enum { SHC_VEL_THROTTLE = 0, SHC_VEL_REAL = 1 }; /* default.yaml:89 velocity_input_mode */

/* One joint: config/default.yaml:31 "<LEG>_<joint>_joint_parameters" (model.cpp:1032-1036). */
typedef struct shc_joint_params {
  double min, max, offset, unpacked, max_vel;
} shc_joint_params;

/* One link: config/default.yaml:51 "<LEG>_<link>_link_parameters" (model.cpp:998-1001). */
typedef struct shc_link_params {
  double d, theta, r, alpha;
} shc_link_params;

/*
 * The subset of `struct Parameters` (parameters_and_states.h:271-381) + gait.yaml + auto_pose.yaml
 * that the accelerated path reads.  One morphology + one gait for the whole batch (mixed
 * batches = several engines / bins).
 */
typedef struct shc_params {
  /* control parameters (default.yaml:9-15) */
  double time_delta;
  int32_t manual_posing, auto_posing, rough_terrain_mode, admittance_control, inclination_posing, imu_posing;
  /* model (default.yaml:25-78) */
  int32_t leg_count;
  int32_t leg_dof[SHC_MAX_LEGS];   /* 3..5 per leg; legs may differ: joint arrays of the ABI are then [legs][longest leg's DOF], a shorter leg's
                                      extra entries read 0 and are ignored on input (the engine pads it behind its tip with locked joints) */
  shc_joint_params joint[SHC_MAX_LEGS][SHC_MAX_JOINTS];
  shc_link_params link[SHC_MAX_LEGS][SHC_MAX_LINKS]; /* link[l][0] = base link */
  int32_t clamp_joint_positions, clamp_joint_velocities;
  /* walker (default.yaml:82-106) */
  double body_clearance, step_frequency, swing_height, swing_width, step_depth, stance_span_modifier;
  double touchdown_threshold, liftoff_threshold; /* default.yaml:107-108 (rough_terrain_mode: Leg::touchdownDetection, model.cpp:712) */
  int32_t velocity_input_mode;
  double stance_position[SHC_MAX_LEGS][2];
  int32_t overlapping_walkspaces, force_normal_touchdown, gravity_aligned_tips;
  int32_t leg_manipulation_mode; /* default.yaml:120: SHC_MANIPULATION_TIP_CONTROL / _JOINT_CONTROL (WalkController::updateManual) */
  /* poser (default.yaml:110-118) */
  double time_to_start;
  double rotation_pid_gains[3];  /* p, i, d */
  double max_translation[3];     /* x, y, z */
  double max_rotation[3];        /* roll, pitch, yaw */
  double max_translation_velocity, max_rotation_velocity;
  /* admittance (default.yaml:122-130) */
  int32_t dynamic_stiffness, use_joint_effort;
  double integrator_step_time, virtual_mass, virtual_stiffness, virtual_damping_ratio, force_gain;
  double load_stiffness_scaler, swing_stiffness_scaler;
  /* gait (gait.yaml) */
  int32_t stance_phase, swing_phase, phase_offset;
  int32_t offset_multiplier[SHC_MAX_LEGS];
  /* auto pose (auto_pose.yaml) */
  double pose_frequency;
  int32_t pose_phase_length;
  int32_t n_auto_posers;
  int32_t pose_phase_starts[SHC_MAX_AUTO_POSERS], pose_phase_ends[SHC_MAX_AUTO_POSERS];
  int32_t pose_negation_phase_starts[SHC_MAX_LEGS], pose_negation_phase_ends[SHC_MAX_LEGS];
  double negation_transition_ratio[SHC_MAX_LEGS];
  double roll_amplitudes[SHC_MAX_AUTO_POSERS], pitch_amplitudes[SHC_MAX_AUTO_POSERS],
      yaw_amplitudes[SHC_MAX_AUTO_POSERS];
  double x_amplitudes[SHC_MAX_AUTO_POSERS], y_amplitudes[SHC_MAX_AUTO_POSERS], z_amplitudes[SHC_MAX_AUTO_POSERS],
      gravity_amplitudes[SHC_MAX_AUTO_POSERS];
} shc_params;

/* walk_controller.h:23-33 StepCycle */
typedef struct shc_step_cycle {
  double frequency;
  int32_t period, swing_period, stance_period, stance_end, swing_start, swing_end, stance_start;
} shc_step_cycle;

/*
 * Per-(morphology, gait) tables produced once by the init chain
 * (state_controller.cpp:263-272: directStartup -> updateDefaultConfiguration ->
 *  Model::generateWorkspaces -> WalkController::generateWalkspace -> generateLimits)
 * and consumed every cycle by WalkController::getLimit (walk_controller.cpp:414).
 */
typedef struct shc_tables {
  shc_step_cycle step;
  int32_t phase_offset[SHC_MAX_LEGS];                      /* walk_controller.cpp:277 */
  double default_joint_position[SHC_MAX_LEGS][SHC_MAX_JOINTS]; /* model.cpp:593 after direct start-up */
  double walkspace[SHC_N_BEARINGS];                        /* walk_controller.cpp:57 */
  double max_linear_speed[SHC_N_BEARINGS];                 /* walk_controller.cpp:344-359 */
  double max_angular_speed[SHC_N_BEARINGS];
  double max_linear_acceleration[SHC_N_BEARINGS];
  double max_angular_acceleration[SHC_N_BEARINGS];
  double workspace_radius[SHC_MAX_LEGS][SHC_N_BEARINGS];   /* model.cpp:309 (simple workspace, plane z = 0) */
  int32_t pose_phase_length, pose_normaliser;              /* pose_controller.cpp:62-63 */
  int32_t auto_pose_reference_leg;                         /* pose_controller.cpp:75-78 */
} shc_tables;

typedef struct shc_engine shc_engine; /* opaque */

/* Feature bits of the fused cycle kernel (compile-time specialisations are picked from these). */
enum {
  SHC_FEAT_TIP_FORCE = 1 << 0, /* Leg::calculateTipForce every cycle (model.cpp:938); needs joint_effort input */
  SHC_FEAT_ODOMETRY = 1 << 1,  /* WalkController::odometry_ideal_ integration (walk_controller.cpp:643, :783-791) */
  /* Diagnostic: run the runtime-flag kernel (every feature compiled in, selected per launch) even where a compile-time
   * specialisation for the parameter set exists.  Both produce identical bits (tests/test_gpu_parity.py). */
  SHC_FEAT_GENERIC_KERNEL = 1 << 30,
  /* Diagnostic: resident mode with ONE wavefront per robot group even where the batch is small enough for the two-wavefront
   * (walker / model) pipeline.  Both produce identical bits (tests/test_gpu_resident.py). */
  SHC_FEAT_RESIDENT_ONE_WAVE = 1 << 29,
  /* Diagnostic: launch every step as ONE kernel on the engine's stream even for batches large enough for the two-stream split
   * (see shc_engine_step).  Both produce identical bits. */
  SHC_FEAT_SINGLE_STREAM = 1 << 28,
  /* Diagnostic: shc_engine_step_k runs its K cycles as K single launches (row k through the setters, one shc_engine_step, q / qd into slot k of the
   * output ring) even where the configuration has a batch kernel - the form every configuration WITHOUT one takes (manual-leg kernels; the
   * runtime-flag families unless the library was built with SHC_GENERIC_LOOP_FORMS=1).  Both produce identical bits (tests/test_gpu_step_k.py). */
  SHC_FEAT_STEP_K_SERIAL = 1 << 27,
  SHC_FEAT_ALL = 0x07ffffff
};

/*
 * Library/device introspection.  shc_device_count() returns the number of visible HIP devices
 * (0 when none; never an error) so a caller can fail loudly before creating an engine.
 */
int shc_abi_version(void); /* 6: shc_engine_adjust_parameter (the nine run-time adjustable parameters); 5: shc_engine_step_k / shc_engine_get_step_k_joint_state (K cycles per launch, each with its own inputs); 4: shc_cycle_inputs.direct + shc_engine_resident_bind_inputs (launch-free posts); 3: resident mode, join, auxiliary state */
/* sizeof(shc_params) / sizeof(shc_tables) as compiled into the library: lets a foreign-language binding check its layout */
int64_t shc_sizeof_params(void);
int64_t shc_sizeof_tables(void);
int shc_device_count(void);
const char *shc_last_error(void);
/* Diagnostics: `reps` launches of a plain plane copy (16-byte loads and stores per lane, the cycle kernel's access shape) of
 * `n_doubles` (even) doubles; the known byte count calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM). */
int shc_debug_plane_copy(int device, int64_t n_doubles, int reps);

/*
 * The same init chain for `count` morphologies at once, ON THE GPU: one thread per
 * (morphology, leg) runs the direct start-up solve and the workspace search (thousands of sequential DLS steps each,
 * model.cpp:309-510, pose_controller.cpp:463-517), one thread per morphology the walkspace and limits
 * (walk_controller.cpp:57-361).  params / out are host arrays; status[i] (may be NULL) receives SHC_OK or the reason
 * morphology i was rejected (its tables are zeroed).  Results match shc_generate_tables up to FP contraction.
 */
int shc_generate_tables_batch(const shc_params *params, int64_t count, shc_tables *out, int32_t *status, int device);
/* shc_engine_create with tables computed beforehand (shc_generate_tables / _batch) instead of running the init chain. */
int shc_engine_create_with_tables(const shc_params *params, const shc_tables *tables, int64_t n_instances, int device,
                                  void *stream, shc_engine **out);

/* HIP stream helpers for hosts without their own HIP binding (the ctypes tests, a cgo / JNI host): a non-blocking
 * stream on `device` to pass to shc_engine_create / shc_engine_set_stream.  Engines of different morphology bins run
 * concurrently when each has its own stream (BASELINE.json configs[4]). */
int shc_stream_create(int device, void **stream);
int shc_stream_destroy(int device, void *stream);

/*
 * Host-side init chain = StateController::init + initModel + direct start-up + workspace /
 * walkspace / limit generation (state_controller.cpp:127-153, 263-272).  Pure host function
 * (no device needed); fills `out`.  Product code (shares the device math headers), NOT the oracle.
 */
int shc_generate_tables(const shc_params *params, shc_tables *out);

/*
 * Create an engine for `n_instances` robots on HIP device `device` (replaces
 * StateController::StateController + init(), state_controller.cpp:13-153, for a batch).
 * Every instance starts in the post-start-up RUNNING state: joints at the default
 * configuration, walk state STOPPED.  `stream` is a hipStream_t (0 = default stream).
 */
int shc_engine_create(const shc_params *params, int64_t n_instances, int device, void *stream, shc_engine **out);
int shc_engine_destroy(shc_engine *e);
int shc_engine_set_stream(shc_engine *e, void *stream);
int shc_engine_set_features(shc_engine *e, uint32_t features);
int shc_engine_get_tables(const shc_engine *e, shc_tables *out);
int64_t shc_engine_instances(const shc_engine *e);

/*
 * Inputs that arrive between cycles in the reference (ROS callbacks, state_controller.cpp).
 * `on_device` != 0: pointers are device pointers in the same layouts (no host round trip).
 * NULL pointer = leave that input unchanged.
 */
/* StateController::bodyVelocityInputCallback (state_controller.cpp:1127): lin [n][2], ang [n]. */
int shc_engine_set_velocity(shc_engine *e, const double *linear_xy, const double *angular, int on_device);
/* imuCallback -> Model::setImuData (state_controller.cpp:1552, model.h:146): quat [n][4] (w,x,y,z), gyro [n][3]. */
int shc_engine_set_imu(shc_engine *e, const double *orientation_wxyz, const double *angular_velocity, int on_device);
/* tipStatesCallback -> Leg::setTipForceMeasured (state_controller.cpp:1618): [n][legs][3]. */
int shc_engine_set_tip_force(shc_engine *e, const double *tip_force, int on_device);
/* jointStatesCallback -> Joint::current_effort_ (state_controller.cpp:1590): [n][legs][dof].
 * Until the first effort arrives Leg::calculateTipForce filters zero torques into a zero state (model.cpp:680-707), so the
 * engine runs the kernels without the estimate; the FIRST call (or the injection of a non-zero filter state) switches to the
 * kernels that evaluate it and synchronises the stream once - make it before capturing the per-cycle path in a hipGraph. */
int shc_engine_set_joint_effort(shc_engine *e, const double *joint_effort, int on_device);
/* bodyPoseInputCallback (state_controller.cpp:1142): translation / rotation velocity inputs [n][3] each. */
int shc_engine_set_pose_input(shc_engine *e, const double *translation_velocity, const double *rotation_velocity,
                              int on_device);
/* poseResetCallback -> PoseController::setPoseResetMode (pose_controller.h:105): mode [n], SHC_NO_RESET .. SHC_IMMEDIATE_ALL_RESET. */
int shc_engine_set_pose_reset_mode(shc_engine *e, const int32_t *mode, int on_device);

/*
 * Advance every instance by `n_cycles` control cycles (StateController::loop with robot_state
 * RUNNING, state_controller.cpp:162-193 + runningState :379-447).  Inputs are held for all
 * n_cycles.  Asynchronous on the engine's stream.
 */
int shc_engine_step(shc_engine *e, int n_cycles);
int shc_engine_synchronize(shc_engine *e);
/*
 * Batches of 4 096 wavefronts and more (40 960 hexapods, 32 768 octopods): shc_engine_step launches the two halves of the batch on
 * two INTERNAL streams (one pair per device, shared by every engine of the process on that device) and does NOT join them between consecutive steps (a robot's next cycle depends
 * on its own last cycle only; one half's partly filled last round of wavefronts then overlaps the other half's full rounds:
 * +20 ... 30 % control cycles per second).  Every other shc_engine_* / shc_leg_* entry point orders the engine's stream after both
 * halves before it enqueues anything, so getters, host-array setters and shc_engine_synchronize behave as before.  Setters given
 * DEVICE arrays (on_device) while split steps are in flight do not join: each half's own stream scatters its share of the array (after
 * waiting for the engine's stream, where the caller made the array ready), and the engine's stream is then ordered after both reads by
 * events - so the caller may overwrite or free the array on the engine's stream right after the setter returns, as before, without a
 * host wait and without draining the steps.  Only a caller that enqueues ITS OWN work on the engine's stream right after shc_engine_step
 * (a kernel reading shc_engine_joint_buffer, an event, a graph capture) calls shc_engine_join first: it makes the engine's stream wait
 * for the internal ones (no host wait).
 * SHC_FEAT_SINGLE_STREAM turns the split off.
 */
int shc_engine_join(shc_engine *e);

/*
 * Resident mode: the node's control loop kept on the chip.
 *
 * The reference runs `while (ros::ok()) { state.loop(); ...publish...; ros::spinOnce(); rate.sleep(); }` (src/main.cpp:106-131):
 * every iteration takes whatever the callbacks delivered since the last one (velocity :1127, body pose :1142, IMU :1552, joint
 * states :1566, tip states :1618 of src/state_controller.cpp), runs one control cycle (StateController::loop :162-193) and
 * publishes the desired joint state (:777-805).  shc_engine_step is one iteration per kernel launch: at batch sizes that fit the
 * chip once (<= 8 wavefronts per compute unit, e.g. up to ~20 000 hexapods on one MI355X) launch + state load + state store cost
 * as much as the cycle itself.  In resident mode ONE launch stays on the device for up to `max_cycles` iterations; per-leg state
 * lives in registers and per-robot state in LDS from one cycle to the next, and every iteration
 *   - waits for the loop tick: a cycle counter ("doorbell") that shc_engine_resident_publish advances,
 *   - takes the inputs posted for that cycle (shc_engine_resident_post; groups that were not posted keep their last value,
 *     exactly like a callback that did not fire), and
 *   - writes that cycle's desired joint positions and velocities to an output ring (shc_engine_resident_get_joint_state).
 * Results are bit-identical to the same cycles run through shc_engine_step(e, 1) with the same inputs set in between.
 *
 *   shc_engine_resident_begin(e, ring_depth, max_cycles, idle_timeout_ms)
 *       starts the loop (on a stream of the engine's own, ordered after everything queued on the engine's stream: the kernel does
 *       not end by itself, so it must not sit on a stream others use - the legacy default stream least of all).  ring_depth (2..255): input sets that may be posted ahead of the cycle that
 *       consumes them = cycles whose outputs stay readable; max_cycles (1..2^31-2): hard bound of this launch; idle_timeout_ms
 *       (0 = 5 000): the device loop stops by itself when everything released has run and the doorbell has not moved for this long
 *       (a host that went away cannot leave the GPU spinning; every device-side wait is bounded).  SHC_ERR_UNSUPPORTED: the batch does not fit the chip once, or
 *       the configuration runs on a manual-leg kernel (a leg toggled, planner mode).  Rough terrain mode, tip rotations and the tip-align
 *       pose (gravity_aligned_tips on <= 3-DOF legs, since round 5) have a resident form: one wavefront per robot group, up to ~990 wavefronts.  Until shc_engine_resident_end every other
 *       entry point that touches the engine's state returns SHC_ERR_BUSY.
 *   shc_engine_resident_post(e, inputs, cycle)
 *       the inputs "the callbacks delivered" for the next unposted cycle (*cycle receives its index, counted from 0 at begin):
 *       arrays as for shc_engine_set_velocity / set_imu / set_pose_input / set_pose_reset_mode / set_tip_force /
 *       set_joint_effort, NULL = not received this iteration.  linear_xy + angular, the two IMU arrays and the two pose arrays are
 *       posted in pairs.  Joint efforts / pose inputs can only be posted to an engine that was given one (shc_engine_set_joint_effort /
 *       shc_engine_set_pose_input) before shc_engine_resident_begin - the first one selects the kernels that evaluate them.
 *       Asynchronous (an internal input stream); blocks only when ring_depth input sets of a group are still waiting to be consumed.
 *       Posting does not release the cycle: shc_engine_resident_publish does.
 *   shc_engine_resident_publish(e, n_cycles)
 *       moves the doorbell: the device may run n_cycles more iterations (cycles nothing was posted for run with the inputs held).
 *   shc_engine_resident_wait(e, cycles, timeout_ms)
 *       returns once `cycles` iterations have completed on every instance and their outputs are visible (SHC_ERR_TIMEOUT otherwise).
 *   shc_engine_resident_get_joint_state(e, cycle, q, qd, on_device)
 *       desired joint positions / velocities [n][legs][dof] of iteration `cycle` (completed, and at most ring_depth - 1 iterations
 *       older than the newest published one) - what publishDesiredJointState sends after that loop iteration.  Blocking for host AND
 *       device buffers: q / qd are complete when the call returns (the copy runs on the engine's private input stream, which a
 *       caller cannot order anything after; the stream-ordered form is the _async call below).
 *   shc_engine_resident_get_joint_state_async(e, cycle, q, qd, timeout_ms)
 *       the same into DEVICE buffers, stream-ordered instead of host-ordered: returns at once having queued, on the engine's
 *       stream, a device-side wait for iteration `cycle` (which may still be unpublished; bounded by timeout_ms, 0 = 5 s) and the copy.
 *       Work the caller queues on that stream afterwards - the all-gather of the fleet's joint buffer - starts the moment the
 *       iteration's outputs exist, its launch latency hidden behind the iterations still running.  At most ring_depth - 1 newer
 *       iterations may be published before the copy has run.  A wait that gave up is reported by shc_engine_resident_end (SHC_ERR_TIMEOUT).
 *   shc_engine_resident_status(e, published, completed, running)
 *   shc_engine_resident_end(e, cycles_run)
 *       stops the loop after the published cycles, waits for it, and leaves the engine exactly as the same cycles through
 *       shc_engine_step would have (state planes, held inputs).  SHC_ERR_TIMEOUT: the loop had already stopped by itself
 *       (idle timeout) before everything published had run - the state is that of *cycles_run iterations, consistent across
 *       instances, and the engine is usable again.  If the loop does not ANSWER the stop request in time the call also returns
 *       SHC_ERR_TIMEOUT, but the engine stays in resident mode (the kernel may still be running: nothing else may touch the state,
 *       shc_engine_destroy waits for the kernel before it frees anything); call shc_engine_resident_end again.
 */
typedef struct shc_cycle_inputs {
  const double *linear_xy;                  /* [n][2]            velocity command (state_controller.cpp:1127) */
  const double *angular;                    /* [n]                                                            */
  const double *imu_orientation_wxyz;       /* [n][4]            sensor_msgs/Imu (:1552), normalised on entry */
  const double *imu_angular_velocity;       /* [n][3]                                                         */
  const double *pose_translation_velocity;  /* [n][3]            body pose input (:1142)                      */
  const double *pose_rotation_velocity;     /* [n][3]                                                         */
  const int32_t *pose_reset_mode;           /* [n]               SHC_NO_RESET ..                              */
  const double *tip_force;                  /* [n][legs][3]      tip wrench (:1618)                           */
  const double *joint_effort;               /* [n][legs][dof]    joint states (:1566)                         */
  int32_t on_device;                        /* the arrays are device pointers                                 */
  int32_t publish;                          /* != 0: release this cycle (and unpublished cycles before it) as soon as its inputs are in
                                               place - post + shc_engine_resident_publish(…, 1) in one kernel launch */
  int32_t direct;                           /* k = 1 .. 4: a DIRECT post from bound input set k - 1 (shc_engine_resident_bind_inputs): no kernel
                                               launch, no copy - the non-NULL members above only say WHICH groups are fresh this cycle (velocity,
                                               IMU, tip force, joint effort); the device loop reads them from the set's arrays when the cycle
                                               runs, and the cycle is released at once (publish is implied).  A few host stores per call. */
  int32_t reserved_;
} shc_cycle_inputs;
enum { SHC_RESIDENT_RUNNING = 0, SHC_RESIDENT_STOPPED = 1, SHC_RESIDENT_IDLE_TIMEOUT = 2, SHC_RESIDENT_MAX_CYCLES = 3, SHC_RESIDENT_FAULT = 4 };
int shc_engine_resident_begin(shc_engine *e, int ring_depth, int64_t max_cycles, int idle_timeout_ms);
/* Bind device arrays as input set `set` (0 .. 3) for direct posts; call it while the loop is NOT running (the addresses travel with the launch
 * of the loop kernel); NULL members = the set does not carry that group; arrays as for the setters ([n][2], [n], [n][4], [n][3],
 * [n][legs][3], [n][legs][dof]).  Contract of a direct post of set k for cycle c: the arrays hold the cycle's inputs (visible device-wide)
 * when shc_engine_resident_post is called.
 *  - Velocity and IMU arrays are sampled into the loop's own state when cycle c starts: they may be rewritten once a LATER cycle has
 *    completed (shc_engine_resident_wait(c + 2)).
 *  - Per-leg arrays (tip force, joint effort) are NOT copied: the set's arrays are the input IN FORCE - read again in every cycle, and once
 *    more when the loop ends (carried into the engine's own planes) - until a later post of THAT GROUP (direct from another set, or an
 *    ordinary post) has replaced them and the cycle it was made for has completed.  Until then they must not be written: a write changes
 *    the held input of the cycles in between, and a read that races it may be torn.
 * A host loop therefore alternates between two sets, as the node's callbacks fill one message while the controller reads the other, and
 * refills a set's per-leg arrays only after the other set's post of the same group has run. */
int shc_engine_resident_bind_inputs(shc_engine *e, int set, const shc_cycle_inputs *arrays);
int shc_engine_resident_post(shc_engine *e, const shc_cycle_inputs *inputs, int64_t *cycle);
int shc_engine_resident_publish(shc_engine *e, int64_t n_cycles);
int shc_engine_resident_wait(shc_engine *e, int64_t cycles, int timeout_ms);
int shc_engine_resident_get_joint_state(shc_engine *e, int64_t cycle, double *q, double *qd, int on_device);
int shc_engine_resident_get_joint_state_async(shc_engine *e, int64_t cycle, double *q, double *qd, int timeout_ms);
int shc_engine_resident_status(shc_engine *e, int64_t *published, int64_t *completed, int32_t *running);
int shc_engine_resident_end(shc_engine *e, int64_t *cycles_run);

/*
 * K loop iterations in ONE launch, each with its own inputs - for batches of any size (resident mode needs the whole batch on the chip at
 * once; this does not).  What `for (k = 0; k < K; ++k) { callbacks deliver inputs[k]; StateController::loop(); publish the desired joint state; }`
 * (src/main.cpp:106-131, src/state_controller.cpp:162-193, :777-805) does, with the state loaded once, kept in registers / LDS for the K cycles and
 * stored once: per cycle only that cycle's inputs are read and its q / qd written.
 *   inputs (may be NULL = every input held): DEVICE arrays (on_device = 1) of K rows each - linear_xy [K][n][2] + angular [K][n],
 *     imu_orientation_wxyz [K][n][4] + imu_angular_velocity [K][n][3], tip_force [K][n][legs][3], joint_effort [K][n][legs][dof]; a NULL member
 *     is held at what the engine has.  Row k is what shc_engine_set_velocity / set_imu / set_tip_force / set_joint_effort would have been given
 *     before cycle k (in rough terrain mode Leg::touchdownDetection runs on the fresh tip force inside the loop, model.cpp:712-722).  Pose
 *     inputs / reset modes are not carried (set them before the call; they are held).  The arrays are read while the launch runs: keep them
 *     valid and unchanged until shc_engine_join (or any synchronising call: shc_engine_synchronize, shc_engine_get_step_k_joint_state, the next
 *     shc_engine_step / step_k) has been ISSUED after this one and the engine's stream has passed that point.  An event recorded on the engine's
 *     stream right after shc_engine_step_k is NOT enough for batches of >= 4 096 wavefronts: those launch on the engine's two internal half-streams
 *     and the engine's own stream is ordered behind them only by the next join (the same holds for a caching allocator that would reuse the
 *     arrays' memory).  After the call the last row is the engine's held input.
 *   The result is bit-identical to K x { setters with row k; shc_engine_step(e, 1) } - state record and the q / qd of every cycle.
 *   shc_engine_get_step_k_joint_state(e, k, q, qd, on_device): q / qd [n][legs][dof] of cycle k (0 .. K - 1) of the latest launch
 *     (stream-ordered; the ring is overwritten by the next shc_engine_step_k).  shc_engine_get_joint_state returns cycle K - 1 as usual.
 * Configurations without a batch kernel - a manual-leg kernel (a leg toggled, planner mode: manual inputs and the plan are held for the K cycles like
 * the pose inputs) and, unless the library was built with SHC_GENERIC_LOOP_FORMS=1, the runtime-flag kernel families (any posing set other than
 * default.yaml's / BASELINE config 3's) - run the same K cycles as K single launches inside the call: same results, same output ring, no launch saved.
 * 1 <= K <= 4096, and K x n x legs x dof x 16 B must stay below 2 GiB.
 */
int shc_engine_step_k(shc_engine *e, int n_cycles, const shc_cycle_inputs *inputs);
int shc_engine_get_step_k_joint_state(shc_engine *e, int k, double *q, double *qd, int on_device);

/*
 * One process per GPU: the exchange of the final joint buffer as peer copies over xGMI, without a collective library (the alternative to the
 * RCCL all-gather the one-process-per-GPU host runs; shc_fleet_all_gather_joints is the same exchange inside one process).  Every rank
 * allocates its gathered buffer with shc_peer_alloc (which also exports it: a 64-byte handle the ranks exchange by whatever means they have),
 * opens every peer's buffer (shc_peer_open) and, after its last step, writes its shard into every buffer at its own offset:
 * shc_peer_scatter(device, shard, bytes, destinations, n, stream) - one copy per destination on a stream of its own (the N - 1 links of a GPU
 * carry N - 1 copies at once), ordered after `stream`, and `stream` ordered after them.  A barrier of the caller's - every rank's copies have
 * completed - closes the exchange.  Nothing orders an exchange after the peers' READS of the previous one: a rank that gathers again while another rank is
 * still consuming its gathered buffer overwrites a slot under that reader, so a barrier of the caller's belongs BEFORE every exchange as well (or one
 * gathered buffer per exchange in flight).  shc_peer_close(device, ptr, opened): opened != 0 for a peer's buffer, 0 frees one's own and releases the
 * copy streams / events shc_peer_scatter keeps for that device.
 */
int shc_peer_alloc(int device, int64_t bytes, void **device_ptr, unsigned char *handle64);
int shc_peer_open(int device, const unsigned char *handle64, void **device_ptr);
int shc_peer_close(int device, void *device_ptr, int opened);
int shc_peer_scatter(int device, const void *src, int64_t bytes, void *const *dst, int n_dst, void *stream);

/*
 * Outputs read after the cycle (state_controller.cpp:777-805 publishDesiredJointState).
 * q/qd: [n][legs][dof] desired joint position / velocity.  Either may be NULL.
 */
int shc_engine_get_joint_state(shc_engine *e, double *q, double *qd, int on_device);
/*
 * Device view of the engine's own joint-position planes (paired structure-of-arrays):
 * the raw buffer a multi-GPU caller may all-gather without a transpose.  `*n_doubles` = 2 * ceil(dof / 2) * n_slots, n_slots =
 * 64 per wavefront of ⌊64 / legs⌋ robots + 192 slots of plane-stride padding
 * (for an odd DOF the last plane also carries the first joint velocity); use shc_engine_joint_index to address it.
 */
int shc_engine_joint_buffer(shc_engine *e, double **device_ptr, int64_t *n_doubles);
/* Map instance-major (i, leg, j) to an index into the buffer above. */
int64_t shc_engine_joint_index(const shc_engine *e, int64_t instance, int leg, int joint);

/* LegState-style per-leg outputs (state_controller.cpp:809-893): any pointer may be NULL.
 *   walker_tip   [n][legs][3]  LegStepper::current_tip_pose_.position_
 *   poser_tip    [n][legs][3]  LegPoser::current_tip_pose_.position_ (posed, body frame)
 *   model_tip    [n][legs][3]  Leg::current_tip_pose_.position_ (FK)
 *   tip_force    [n][legs][3]  Leg::tip_force_calculated_
 *   admittance   [n][legs][3]  Leg::admittance_delta_
 *   leg_status   [n][legs]     packed: bits 0-1 step state, bit 2 ik failure (model.cpp:921), bits 8.. phase
 */
int shc_engine_get_leg_state(shc_engine *e, double *walker_tip, double *poser_tip, double *model_tip,
                             double *tip_force, double *admittance, int32_t *leg_status, int on_device);
/* Per-robot outputs: body pose [n][7] (x,y,z,qw,qx,qy,qz) = Model::current_pose_ (state_controller.cpp:911),
 * desired velocity [n][3] (vx,vy,omega), walk_state [n]. */
int shc_engine_get_body_state(shc_engine *e, double *pose, double *velocity, int32_t *walk_state, int on_device);
/*
 * StateController::changeGait (state_controller.cpp:513-538; gaitSelectionCallback :1206): the gait members of `new_gait`
 * (stance_phase, swing_phase, phase_offset, offset_multiplier and, with auto_posing, the auto-pose tables) replace the
 * engine's; step cycle, velocity / acceleration limits and auto-pose phases are regenerated as generateStepCycle +
 * generateLimits + setAutoPoseParams do.  A batch shares one gait, so the change happens only when EVERY instance is
 * STOPPED; otherwise the velocity inputs of all instances are zeroed (the reference "forces the Syropod to stop") and
 * *still_walking receives the number of instances not yet STOPPED: step on and call again, as the reference retries on
 * every loop while gait_change_flag_ is set.  Synchronises the engine's stream.
 */
int shc_engine_change_gait(shc_engine *e, const shc_params *new_gait, int64_t *still_walking);
/*
 * StateController::adjustParameter (state_controller.cpp:451-509), reached from parameterSelectionCallback / parameterAdjustCallback (:1419-1463, the
 * adjustable_map of :1884-1892) and from dynamicParameterCallback (:1467-1548): one of the nine run-time adjustable parameters takes `value` (the callbacks'
 * clamping to the parameter's min / max is the caller's - the node's - as are adjust_step and the selection).  `which` numbers them as enum
 * ParameterSelection does (parameters_and_states.h:165-178).  A batch shares its parameters: the change is for every instance.
 *   The eight parameters the cycle reads as they are (swing_height, swing_width, step_depth, stance_span_modifier, virtual_mass, virtual_stiffness,
 *   virtual_damping_ratio, force_gain) are in force from the next control cycle: a new launch-uniform block, no table regenerated, no state touched
 *   (stance_span_modifier moves the default tips when calculateStanceSpanChange next runs - at a stop, or at swing / stance start in rough terrain).
 *   step_frequency: as in the reference the new value is stored at once (:454: sequence / transition timings read it from then on) and the maximum-speed maps
 *   and the legs' phase offsets of the NEW step cycle replace the current ones at once (:458-463, generateLimits' setPhaseOffset walk_controller.cpp:277) -
 *   the walker slows down to them.  The step cycle itself and all four limit maps are regenerated (generateStepCycle + generateLimits, :491-492) only when the
 *   desired body velocity is inside what the velocity input maps to under the new limits (:464-489) - here: for EVERY instance; otherwise *pending (may be
 *   NULL) receives the number of instances still outside and the caller calls again after the next cycle, as runningState retries on every loop while
 *   parameter_adjust_flag_ is set (:411-414).  On acceptance (*pending = 0) the legs of walking robots are mapped onto the new cycle
 *   (LegStepper::updatePhase, walk_controller.cpp:862-867) INSIDE the next control cycle, between its posing part and updateWalk, where the reference's
 *   loop has it; that cycle runs alone in its launch on the runtime-flag kernels.  Until it has run, shc_engine_step_k and resident mode map the phases
 *   before they start (the one-loop ordering against the posing part is then not kept).  The auto-pose phase tables keep the old step period
 *   (setAutoPoseParams is not called by adjustParameter).  SHC_ERR_UNSUPPORTED for step_frequency in rough_terrain_mode, with gravity_aligned_tips or with a
 *   non-zero stance_span_modifier (the posing part / the limit generation read stepper state there that this ordering cannot reproduce); the other eight
 *   parameters have no such restriction.  Synchronises the engine's stream; SHC_ERR_BUSY in resident mode.
 */
enum {
  SHC_PARAM_STEP_FREQUENCY = 1, SHC_PARAM_SWING_HEIGHT = 2, SHC_PARAM_SWING_WIDTH = 3, SHC_PARAM_STEP_DEPTH = 4, SHC_PARAM_STANCE_SPAN_MODIFIER = 5,
  SHC_PARAM_VIRTUAL_MASS = 6, SHC_PARAM_VIRTUAL_STIFFNESS = 7, SHC_PARAM_VIRTUAL_DAMPING = 8, SHC_PARAM_FORCE_GAIN = 9
};
int shc_engine_adjust_parameter(shc_engine *e, int which, double value, int64_t *pending);
/* WalkController::getOdometryIdeal() (walk_controller.h:112; integrated at walk_controller.cpp:643 from
 * calculateOdometry :783-791): [n][7] (x,y,z,qw,qx,qy,qz).  Needs SHC_FEAT_ODOMETRY (on by default). */
int shc_engine_get_odometry(shc_engine *e, double *pose, int on_device);
/* Leg::getVirtualStiffness() (model.h:264) as left by AdmittanceController::updateStiffness
 * (admittance_controller.cpp:96-134; only published, state_controller.cpp:889): [n][legs].  Updated while
 * admittance_control && dynamic_stiffness && walk state != STOPPED (state_controller.cpp:172-178); 0 before the
 * first update (the reference leaves the member uninitialised).  SHC_ERR_UNSUPPORTED without admittance_control. */
int shc_engine_get_virtual_stiffness(shc_engine *e, double *stiffness, int on_device);

/*
 * ROS message surface.  Numeric payload of syropod_highlevel_controller/LegState.msg as
 * StateController::publishLegState fills it (msg/LegState.msg; state_controller.cpp:809-893), one record per leg of ONE
 * instance: the node copies the fields into its message and adds stamps, frame ids and the leg name.
 * Tip orientations of the walker / poser tips are not part of the payload (UNDEFINED (0,0,0,0) unless gravity_aligned_tips).
 */
typedef struct shc_leg_state_msg {
  double walker_tip_position[3]; /* walker_tip_pose.pose.position   :822-824 (frame walk_plane) */
  double target_tip_position[3]; /* target_tip_pose.pose.position   :826-828 */
  double poser_tip_position[3];  /* poser_tip_pose.pose.position    :830-832 (frame base_link) */
  double model_tip_position[3];  /* model_tip_pose.pose.position    :834-836 */
  double actual_tip_pose[7];     /* actual_tip_pose.pose :837-839 = Leg::applyFK(false, true): FK of the MEASURED joint positions
                                    (x,y,z,qw,qx,qy,qz); until shc_engine_set_joint_states_msg supplies them these are the initial default
                                    positions Leg::init(true) copied (model.cpp:292-296) */
  double model_tip_velocity[3];  /* model_tip_velocity.twist.linear :845-849: always 0 - publishLegState calls applyFK() on
                                    unchanged joints just before (:840), which resets Leg::current_tip_velocity_ to
                                    (tip - tip) / dt (model.cpp:980) */
  double joint_positions[SHC_MAX_JOINTS];  /* Joint::desired_position_ :854 */
  double joint_velocities[SHC_MAX_JOINTS]; /* Joint::desired_velocity_ :855 */
  double joint_efforts[SHC_MAX_JOINTS];    /* Joint::desired_effort_   :856 = the last measured effort (jointStatesCallback :1590) */
  double stance_progress, swing_progress;  /* :860-861, -1 when not in that state (walk_controller.cpp:871-897) */
  double time_to_swing_end;                /* :862-874 */
  double pose_delta[7];                    /* calculateOdometry(time_to_swing_end) :875: x,y,z,qw,qx,qy,qz */
  double auto_pose[7];                     /* LegPoser::auto_pose_ :877-880 (x,y,z,qw,qx,qy,qz): the identity pose when
                                              auto_posing is off (never assigned, pose_controller.cpp:1716); with auto posing
                                              the per-leg pose of the last cycle (negation applied), re-derived from the
                                              stored poser latches and master phase */
  double tip_force[3];                     /* tip_force_calculated_ * force_gain :883-885 */
  double admittance_delta[3];              /* :886-888 */
  double virtual_stiffness;                /* :889 */
} shc_leg_state_msg;
/* Fills legs[0 .. leg_count) for `instance`.  Synchronises the engine's stream. */
int shc_engine_read_leg_state_msg(shc_engine *e, int64_t instance, shc_leg_state_msg *legs);

/*
 * The other ROS messages of the path, batched: payloads in, payloads out; the node adds names,
 * stamps and frame ids.
 */
/* jointStatesCallback (state_controller.cpp:1566-1594), sensor_msgs/JointState as the motors report it: position rows
 * [n][legs][dof] are RAW motor positions - the callback stores position - Joint::offset_ as Joint::current_position_ (:1581; used by
 * Leg::applyFK(.., use_actual) for LegState.actual_tip_pose) - and effort rows [n][legs][dof] become Joint::current_effort_ /
 * desired_effort_ (:1589-1590, what Leg::calculateTipForce and LegState.joint_efforts read).  velocity is stored by the
 * reference but never read on this path: accepted for symmetry, ignored.  Any pointer may be NULL. */
int shc_engine_set_joint_states_msg(shc_engine *e, const double *position, const double *velocity, const double *effort, int on_device);
/* tipStatesCallback (state_controller.cpp:1617-1679), syropod_highlevel_controller/TipState: wrench force rows [n][legs][3]
 * (= shc_engine_set_tip_force: stores the force, switches touchdown detection on, runs Leg::touchdownDetection) and / or the
 * range sensors' step_plane rows [n][legs][3] = (x, y, z): z = distance to the step surface along the tip's x axis, or
 * UNASSIGNED_VALUE (2147483647) when the sensor lost contact -> Leg::step_plane_pose_ (:1661-1672).  Either may be NULL. */
int shc_engine_set_tip_states_msg(shc_engine *e, const double *wrench_force, const double *step_plane, int on_device);
/* publishDesiredJointState (state_controller.cpp:777-805): the combined sensor_msgs/JointState payload (Leg::
 * generateDesiredJointStateMsg, model.cpp:605-617: desired_position_, desired_velocity_, desired_effort_) and the per-joint
 * std_msgs/Float64 position commands desired_position_ + offset_ (:798).  All [n][legs][dof]; any pointer may be NULL. */
int shc_engine_get_joint_commands(shc_engine *e, double *position, double *velocity, double *effort, double *position_command, int on_device);

/*
 * Per-leg methods of class Leg (model.h:448-492), batched.  The reference's cold paths call them on their own - workspace
 * search (model.cpp:309-510), start-up / shut-down sequences (pose_controller.cpp:145-805), leg manipulation
 * (state_controller.cpp:590-700); the fused cycle runs the same arithmetic in registers.  They act on the engine's joint state:
 * a cycle launched afterwards continues from the joints these calls left.
 * Selection: instances [first, first + count), leg = -1 for every leg or one leg id.  Arrays hold one row per selected
 * (instance, leg) in instance-major order: [count][legs or 1][K].  Host pointers unless on_device != 0.
 */
/* Leg::setDesiredTipPose(tip_pose, apply_delta) (model.h:448, model.cpp:653): tip_pose rows are (x,y,z,qw,qx,qy,qz), an
 * all-zero quaternion = UNDEFINED_ROTATION (position-only IK); tip_pose == NULL = the default argument Pose::Undefined() =
 * "take the poser's tip pose"; apply_delta adds Leg::admittance_delta_. */
int shc_leg_set_desired_tip_pose(shc_engine *e, int64_t first, int64_t count, int leg, const double *tip_pose, int apply_delta, int on_device);
/* Leg::solveIK(delta, solve_rotation) (model.h:470, model.cpp:726): delta rows are the 6-vector (position delta, rotation
 * delta) in the leg (joint 1) frame; joint_delta rows [dof].  Reads the joint state, changes nothing. */
int shc_leg_solve_ik(shc_engine *e, int64_t first, int64_t count, int leg, const double *delta, int solve_rotation, double *joint_delta,
                     int on_device);
/* Leg::updateJointPositions(delta, simulation) (model.h:477, model.cpp:799): integrates joint_delta (velocity clamp unless
 * simulation, position clamp per parameters); limit_proximity rows [1] (may be NULL) receive the returned minimum proximity. */
int shc_leg_update_joint_positions(shc_engine *e, int64_t first, int64_t count, int leg, const double *joint_delta, int simulation,
                                   double *limit_proximity, int on_device);
/* Leg::applyIK(simulation) (model.h:485, model.cpp:861) towards the stored desired tip pose: position solve, rotation solve
 * when the desired rotation is defined, unconstrained retry, 5 mm deviation check, calculateTipForce.  ik_result rows [1]
 * (may be NULL): limit proximity, 0 on failure. */
int shc_leg_apply_ik(shc_engine *e, int64_t first, int64_t count, int leg, int simulation, double *ik_result, int on_device);
/* Leg::applyFK(set_current, use_actual) (model.h:492, model.cpp:945): tip pose rows (x,y,z,qw,qx,qy,qz) in the robot frame from
 * the desired joint positions, or from joint_position rows [dof] when given (use_actual: the measured positions of
 * jointStatesCallback, which the engine does not store - this is what LegState.actual_tip_pose needs, state_controller.cpp:839). */
int shc_leg_apply_fk(shc_engine *e, int64_t first, int64_t count, int leg, const double *joint_position, double *tip_pose, int on_device);

/*
 * Sequences: the LegPoser building blocks of PoseController's start-up / shut-down / stance-change
 * procedures, batched with the same (first, count, leg) selection as the per-leg methods, and the direct start-up built on them.
 */
/* LegPoser::stepToPosition(target_tip_pose, target_pose, lift_height, time_to_step, apply_delta) (pose_controller.h:506,
 * pose_controller.cpp:1571-1712), ONE iteration: the tip follows two quartic Bezier curves from where the leg stood when the
 * sequence began to target_tip_pose (rows (x,y,z,qw,qx,qy,qz); NULL = Pose::Undefined() = stay, rotation undefined) while the
 * body pose eases from the identity to target_pose ([count][7], one row per instance).  tip_pose rows receive
 * LegPoser::current_tip_pose_ - hand them to shc_leg_set_desired_tip_pose + shc_leg_apply_ik as stepToNewStance (:521) and
 * directStartup (:463) do; progress rows (int32, may be NULL) the returned percentage (100 = complete, the next call starts a
 * new sequence). */
int shc_leg_step_to_position(shc_engine *e, int64_t first, int64_t count, int leg, const double *target_tip_pose, const double *target_pose,
                             double lift_height, double time_to_step, int apply_delta, double *tip_pose, int32_t *progress, int on_device);
/* LegPoser::transitionConfiguration(transition_time) towards desired_configuration rows [dof] (pose_controller.h:498,
 * pose_controller.cpp:1476-1567), ONE iteration: every joint follows a cubic Bezier from the configuration it had when the
 * transition began. */
int shc_leg_transition_configuration(shc_engine *e, int64_t first, int64_t count, int leg, const double *desired_configuration,
                                     double transition_time, int32_t *progress, int on_device);
/* PoseController::directStartup (pose_controller.cpp:463-517) for the batch.  shc_engine_begin_direct_startup puts every
 * instance where StateController::init + initModel(true) leave a robot (joints at clamped(0, min, max), model.cpp:286-305,
 * :1038; robot state PACKED "undefined", state_controller.cpp:213-222); each shc_engine_direct_startup call is one
 * StateController::loop() of the PACKED -> READY transition: every joint advances along its transition to the default
 * configuration (the joints the init chain's simulated start-up solve ended on).  *progress = 1..100; the call that returns
 * 100 also runs the loop that enters RUNNING, after which the engine is exactly where shc_engine_create leaves it and the
 * joints published on the way are the reference's start-up trajectory. */
int shc_engine_begin_direct_startup(shc_engine *e);
int shc_engine_direct_startup(shc_engine *e, int32_t *progress);
/*
 * Start-up / shut-down SEQUENCES (start_up_sequence: true): PoseController::executeSequence (pose_controller.cpp:145-459) and
 * PoseController::stepToNewStance (:521-557) for the batch, every instance with its own PoseController state (transition step,
 * leg group, the transition poses its first START_UP learns, pose_controller.h:273-304, :591-593).
 *   shc_engine_begin_sequence_startup  StateController::init() + initModel(false) (main.cpp:99-100, model.cpp:286-305): fresh
 *       walker / poser state, every joint where the motors report it.  joint_positions: [legs][dof] rows shared by all instances,
 *       or [n][legs][dof] with per_instance = 1 (host array, offsets removed); NULL = the READY configuration (every joint at its
 *       `unpacked` position, state_controller.cpp:217).
 *   shc_engine_execute_sequence        ONE call of executeSequence(sequence) per instance.  progress rows [n] (host, may be NULL):
 *       -1 while a first START_UP is still generating its sequence, 0..99, 100 = complete, -2 = gave up (more than
 *       TRANSITION_STEP_THRESHOLD transitions: the reference shuts down).  An instance that has completed `sequence` is left
 *       alone by further calls with the same `sequence` (robots that learnt different sequences finish after different numbers
 *       of calls; the caller loops until every row reads 100).  The node calls it where
 *       transitionRobotState does (:301, :334: once per loop for START_UP, twice per loop for SHUT_DOWN through runningState :384-388).
 *       The body pose is Model::current_pose_ as the last control cycle left it (the sequences run with the robot stopped).
 *   shc_engine_finish_sequence_startup what follows a completed START_UP (:305-313): walker_->init(), the configuration the
 *       sequence ended on becomes the default configuration, workspaces / walkspace / limits are regenerated from it (instance 0
 *       stands for the batch: the tables belong to the engine), robot state RUNNING and the first control cycle of the same loop.
 *       Auto posing on its own clock (pose_frequency != -1) keeps posing through the sequence; the PoseController's phase counter, poser latches and
 *       Model::current_pose_ carry over into RUNNING and the workspace is searched at the pose of the completing loop (model.cpp:338) - which every
 *       instance must share (SHC_ERR_UNSUPPORTED when they completed START_UP in different phases of the auto pose, with IMU posing on top, or on
 *       tip-align robots).  The sequence calls of such robots run their posing part as a pose-only pass of the manual-leg cycle kernels; that
 *       leaves no trace: shc_engine_step keeps its kernels, shc_engine_step_k and resident mode stay available afterwards.
 *   shc_engine_step_to_new_stance      ONE call of stepToNewStance per instance (progress as the reference returns it).
 *   shc_engine_pack_legs / unpack_legs PoseController::packLegs / unpackLegs (:615-707), ONE call per loop: every joint follows its
 *       cubic Bezier (LegPoser::transitionConfiguration) to the packed positions of the current pack step / back to the previous
 *       step's and finally the `unpacked` positions.  packed_positions = Joint::packed_positions_ as [n_pack_steps][legs][dof]
 *       (default.yaml "packed" may list several steps per joint).  *progress as the reference returns it (0 between pack steps,
 *       100 when done); the pack step and "transition executing" flag are PoseController members, kept per engine (every
 *       instance runs the same number of iterations).
 */
enum { SHC_MANIPULATION_TIP_CONTROL = 0, SHC_MANIPULATION_JOINT_CONTROL = 1 };
enum { SHC_SEQUENCE_START_UP = 0, SHC_SEQUENCE_SHUT_DOWN = 1 }; /* enum SequenceSelection (parameters_and_states.h:183-188) */
int shc_engine_begin_sequence_startup(shc_engine *e, const double *joint_positions, int per_instance);
int shc_engine_execute_sequence(shc_engine *e, int sequence, int32_t *progress);
int shc_engine_finish_sequence_startup(shc_engine *e);
int shc_engine_step_to_new_stance(shc_engine *e, int32_t *progress);
/*
 * Manual leg manipulation: StateController::legStateToggle (state_controller.cpp:541-646) with PoseController::
 * poseForLegManipulation (pose_controller.cpp:561-611), and WalkController::updateManual (walk_controller.cpp:652-744, both overloads)
 * inside the control cycle.
 *   shc_engine_toggle_leg_state   ONE StateController::loop() for EVERY instance, with a toggle request pending for the leg
 *       leg_selection[i] designates (host [n]; -1 = no request).  result rows [n]: 1 = transition complete (WALKING <-> MANUAL:
 *       the node clears its toggle flag), 0 = in progress, 2 = refused (MAX_MANUAL_LEGS = 2 already manual), -1 = the robot is
 *       still walking: its velocity inputs are zeroed (:641-645) and its loop is one ordinary control cycle, -3 = no request: one
 *       ordinary control cycle (the call launches the cycle kernel for those two groups only).  While a robot has a leg that is not WALKING its walker
 *       is frozen (updateWalk returns at walk_controller.cpp:503); MANUAL / WALKING_TO_MANUAL legs are not posed (updateStance).
 *       Every posing mode (walk-plane, manual, inclination, IMU, auto) keeps running for the robots that stand during the call - the
 *       posing part of their loop (state_controller.cpp:165-181) is a pose-only pass of the cycle kernel.  With the experimental
 *       tip-align pose (gravity_aligned_tips on <= 3-DOF legs): SHC_ERR_UNSUPPORTED.
 *   shc_engine_set_manual_inputs  primary / secondary leg selection [n] (-1 = LEG_UNDESIGNATED) with their tip velocity inputs
 *       [n][3] (updateManual(.., tip_velocity_input, ..): tip_control moves the tip, joint_control the coxa / tibia joints of
 *       3-DOF legs - whose stepper then holds the FK tip pose WITH its rotation, so that applyIK runs rotation-constrained on them
 *       (walk_controller.cpp:677-690) - and nothing on longer legs, params.leg_manipulation_mode) and tip position inputs [n][3] (updateManual(.., Pose, ..); a zero vector
 *       = none).  Host arrays, any may be NULL.
 *   shc_engine_get_leg_manipulation_state  enum LegState per (instance, leg): 0 WALKING, 1 MANUAL, -1 WALKING_TO_MANUAL,
 *       -2 MANUAL_TO_WALKING.
 */
int shc_engine_toggle_leg_state(shc_engine *e, const int32_t *leg_selection, int32_t *result);
int shc_engine_set_manual_inputs(shc_engine *e, const int32_t *primary_leg, const double *primary_tip_velocity, const double *primary_tip_position,
                                 const int32_t *secondary_leg, const double *secondary_tip_velocity, const double *secondary_tip_position);
int shc_engine_get_leg_manipulation_state(shc_engine *e, int32_t *states);
int shc_engine_pack_legs(shc_engine *e, const double *packed_positions, int n_pack_steps, double time_to_pack, int32_t *progress);
int shc_engine_unpack_legs(shc_engine *e, const double *packed_positions, int n_pack_steps, double time_to_unpack, int32_t *progress);

/*
 * Full controller state of one instance (checkpoint / restore, state injection).  Everything the next control cycle reads
 * that is not an input set through the shc_engine_set_* calls above: restoring a snapshot and replaying the same inputs
 * reproduces the run bit for bit.  Members are named after the reference members they hold.
 */
typedef struct shc_leg_snapshot {
  double joint_position[SHC_MAX_JOINTS];   /* Joint::desired_position_  (model.h:635) */
  double joint_velocity[SHC_MAX_JOINTS];   /* Joint::desired_velocity_  (model.h:636) */
  double walker_tip[3];                    /* LegStepper::current_tip_pose_.position_ (walk_controller.h:520) */
  double walker_tip_velocity[3];           /* LegStepper::current_tip_velocity_       (:527) */
  double swing_origin_tip[3];              /* swing_origin_tip_position_              (:529) */
  double swing_origin_tip_velocity[3];     /* swing_origin_tip_velocity_              (:530) */
  double stance_origin_tip[3];             /* stance_origin_tip_position_             (:531) */
  double default_tip[3];                   /* default_tip_pose_.position_             (:519) */
  double target_tip[3];                    /* target_tip_pose_.position_              (:522) */
  double stride_vector[3];                 /* stride_vector_                          (:512) */
  /* Tip rotations (gravity_aligned_tips, > 3 DOF): every consumer of LegStepper::current_tip_pose_.rotation_ /
   * origin_tip_pose_.rotation_ reads only the rotated x axis (walk_controller.cpp:1217-1224, pose_controller.cpp:129-130,
   * model.cpp:884-893), so the state is that unit vector + whether the current rotation is defined (!= UNDEFINED_ROTATION) */
  double walker_tip_direction[3];
  double origin_tip_direction[3];
  double admittance_state[2];              /* Leg::admittance_state_    (model.h:518) */
  double admittance_delta[3];              /* Leg::admittance_delta_    (model.h:365) */
  double virtual_stiffness;                /* Leg::virtual_stiffness_   (model.h:264) */
  double tip_force_calculated[3];          /* Leg::tip_force_calculated_ (model.cpp:706, low-pass filter state) */
  double swing_progress, stance_progress;  /* LegStepper::swing_progress_ / stance_progress_ (walk_controller.h:498-499) */
  int32_t step_state;                      /* SHC_STEP_* */
  int32_t phase;                           /* LegStepper::phase_ */
  int32_t at_correct_phase, completed_first_step; /* walk_controller.h:493-494 */
  int32_t negate_auto_pose;                /* LegPoser::negate_auto_pose_ (pose_controller.h:575) */
  int32_t ik_failed;                       /* the 5 mm deviation warning of the last applyIK (model.cpp:921) */
  int32_t tip_rotation_defined;            /* current_tip_pose_.rotation_ != UNDEFINED_ROTATION */
  int32_t step_plane_defined;              /* Leg::step_plane_pose_ != Pose::Undefined() (touchdown detection, model.cpp:712-722) */
  double step_plane_position[3];           /* Leg::step_plane_pose_.position_ (robot frame; only its position is read, walk_controller.cpp:1088) */
  /* LegStepper::target_tip_pose_.rotation_ (> 3 DOF): the identity tip rotation with gravity_aligned_tips, UNDEFINED otherwise - until an
   * externally requested target (rough terrain mode) assigns its own, which then stays (walk_controller.cpp:1070 assigns the whole
   * pose, :1044 only the position).  Kept, like the other tip rotations, as the rotated x axis + "defined". */
  double target_tip_direction[3];
  int32_t target_rotation_defined;
  int32_t pad_;
} shc_leg_snapshot;

typedef struct shc_instance_state {
  double desired_linear_velocity[2], desired_angular_velocity; /* walk_controller.h:257-258 */
  double walk_plane[3], walk_plane_normal[3];                   /* WalkController::walk_plane_ / walk_plane_normal_ (:255-256) */
  double stepper_walk_plane[3], stepper_walk_plane_normal[3];   /* the copies the stepping LegSteppers took last cycle (:509-510) */
  double origin_walk_plane_pose[7];   /* PoseController::origin_walk_plane_pose_ (x,y,z,qw,qx,qy,qz) */
  double manual_pose[7];              /* PoseController::manual_pose_ */
  double translation_velocity_input[3], rotation_velocity_input[3]; /* pose_controller.h:285-286 (rewritten by the reset modes) */
  double rotation_absement_error[3], rotation_velocity_error[3];    /* IMU PID state (pose_controller.h:315-317) */
  double auto_pose_rotation[4];       /* PoseController::auto_pose_.rotation_ of the last cycle (w,x,y,z) */
  double current_pose[7];             /* Model::current_pose_ */
  double odometry[7];                 /* WalkController::odometry_ideal_ */
  double tip_align_pose[7], origin_tip_align_pose[7]; /* PoseController::tip_align_pose_ / origin_tip_align_pose_ (pose_controller.h:286-287) */
  int32_t walk_state;                 /* SHC_WALK_* */
  int32_t legs_at_correct_phase, legs_completed_first_step, return_to_default_attempted; /* walk_controller.h:266-268 */
  int32_t auto_posing_state;          /* SHC_POSING .. SHC_POSING_COMPLETE */
  int32_t pose_phase;                 /* PoseController::pose_phase_ (auto posing on its own clock) */
  /* AutoPoser latches, one word per poser: bit 0 start_check_, bit 1 end_check_.first, bit 2 end_check_.second, bit 3 allow_posing_ */
  int32_t auto_poser_flags[SHC_MAX_AUTO_POSERS];
  int32_t touchdown_detection;        /* LegStepper::touchdown_detection_ (walk_controller.h:495): tip-state messages have arrived */
  int32_t pad_;                       /* explicit: the record has no implicit padding (byte-comparable) */
  shc_leg_snapshot leg[SHC_MAX_LEGS];
} shc_instance_state;

/*
 * Auxiliary state: what only the calls AROUND the control cycle keep and shc_instance_state therefore does not carry - the pose
 * reset mode in force, manual leg manipulation (per-leg LegState, the leg selections and their inputs, Leg::desired_tip_pose_ as
 * the next WalkController::updateManual reads it back), externally requested targets / default poses / planner targets
 * (ExternalTarget records with their defined flags), the PoseController / LegPoser members of the start-up, shut-down and planner
 * sequences (transition steps, learnt transition poses, plan step, target configuration / body pose), the LegPoser origin poses of
 * stepToPosition / transitionConfiguration, and the measured joint positions of the last joint-state message.
 * One opaque, versioned blob per instance (its layout is private to the library and tied to the engine's legs x joints;
 * shc_engine_aux_state_bytes gives the size).  A COMPLETE checkpoint of an engine - restore, or migration into another engine of
 * the same parameters - is shc_engine_get_state + shc_engine_get_aux_state (+ the engine-wide scalars the host chose itself:
 * planner mode, pack step); restoring only the first leaves a robot with a MANUAL leg, a pending external target or a half-run
 * sequence without that state.  Held inputs (velocity, IMU, forces, efforts) are inputs: the caller applies them again.
 */
int64_t shc_engine_aux_state_bytes(const shc_engine *e);
int shc_engine_get_aux_state(shc_engine *e, int64_t first, int64_t count, void *blobs);
int shc_engine_set_aux_state(shc_engine *e, int64_t first, int64_t count, const void *blobs);

/*
 * Externally requested tip targets / default stance poses of rough terrain mode: struct ExternalTarget (walk_controller.h:38-46)
 * as targetTipPoseCallback (state_controller.cpp:1706-1767) fills it from syropod_highlevel_controller/TargetTipPose and
 * generateExternalTargetTransforms (:703-773) refreshes its transform_ from the tf tree every loop.  While a leg swings, a
 * defined target replaces the stepper's default target: target_tip_pose_ = pose_.removePose(transform_), the swing clearance
 * takes the requested height, and a target given in the "odom_ideal" frame is led by the body's ideal odometry over the rest
 * of the swing (walk_controller.cpp:1068-1079); the request is dropped at the start of the next stance period (:1159).  A
 * defined default replaces the terrain-following default tip pose at every swing / stance start (:988-990) until another one
 * arrives.  Only rough_terrain_mode reads either.  The tf lookup itself stays with the node.
 */
typedef struct shc_external_target {
  double pose[7];          /* ExternalTarget::pose_ (x,y,z,qw,qx,qy,qz); an all-zero quaternion = UNDEFINED_ROTATION */
  double transform[7];     /* ExternalTarget::transform_; the callback stores the identity (:1732), the tf refresh updates it */
  double swing_clearance;  /* ExternalTarget::swing_clearance_ (0 for a default pose) */
  int32_t frame_is_odom_ideal; /* frame_id_ == "odom_ideal" (:1073) */
  int32_t defined;         /* ExternalTarget::defined_; 0 withdraws the request */
} shc_external_target;
enum { SHC_EXTERNAL_TARGET = 0, SHC_EXTERNAL_DEFAULT = 1, SHC_EXTERNAL_PLANNER_TARGET = 2 /* LegPoser::external_target_, see planner mode */ };
/* LegStepper::setExternalTarget / setExternalDefault for legs `leg` (-1 = all, rows [count][legs]) of instances [first, first +
 * count); rows is a HOST array.  As in the callback (:1734-1757), a LegStepper takes a request only while its robot is not
 * STOPPED; the TARGET of a robot that stands goes to its LegPoser for planner mode (and raises target_tip_pose_acquired_, see
 * shc_engine_execute_plan), a DEFAULT for it is dropped.  Dropped rows are counted in *ignored (may be NULL): defaults for
 * standing robots and stepper requests without rough_terrain_mode (no stepper reads them).  The stepper keeps only the x axis of tip rotations (see
 * shc_leg_snapshot), and legs with <= 3 joints none at all.  On > 3-DOF legs a requested target rotation becomes
 * LegStepper::target_tip_pose_.rotation_ (walk_controller.cpp:1070 assigns the whole pose): updateTipRotation blends the tip towards
 * it over the back half of the swing, Leg::applyIK solves for it (rotation-constrained pass + unconstrained retry) - and, as in the
 * reference, it STAYS the leg's target rotation after the request is dropped (:1044 re-assigns only the position), replacing the
 * identity tip rotation of gravity_aligned_tips; a request with UNDEFINED_ROTATION clears it.
 * which = SHC_EXTERNAL_PLANNER_TARGET addresses the LegPoser's record directly (set: whatever the walk state). */
int shc_engine_set_external_target(shc_engine *e, int which, int64_t first, int64_t count, int leg, const shc_external_target *rows,
                                   int64_t *ignored);
/* generateExternalTargetTransforms: new transform_ rows (x,y,z,qw,qx,qy,qz) for requests that are currently defined. */
int shc_engine_set_external_transform(shc_engine *e, int which, int64_t first, int64_t count, int leg, const double *transform);
/* The requests as the steppers hold them now (defined_ cleared where a swing has consumed the target). */
int shc_engine_get_external_target(shc_engine *e, int which, int64_t first, int64_t count, int leg, shc_external_target *rows);

/*
 * Planner mode: StateController::executePlan (state_controller.cpp:653-698), the branch of runningState (:401-405) that replaces
 * the walking cycle while planner_mode_ is on.  The planner answers the node's request for plan step N with either a joint
 * configuration (target_configuration -> PoseController::transitionConfiguration(5.0), pose_controller.cpp:710-763: every named
 * leg moves on LegPoser::transitionConfiguration's cubic Bezier) or tip targets and / or a body pose (TargetTipPose for a robot that
 * stands, target_body_pose -> PoseController::transitionStance(5.0), :767-807: LegPoser::stepToPosition to the target while the
 * body eases to the pose, Leg::setDesiredTipPose + applyIK every iteration).  Requires what manual leg manipulation requires of
 * the posing modes; gravity_aligned_tips is SHC_ERR_UNSUPPORTED.
 */
enum { SHC_PLAN_WALKING = -1, SHC_PLAN_WAITING = -2 };
/* plannerModeCallback (:1262-1281): switching it on resets plan_step_ of every instance. */
int shc_engine_set_planner_mode(shc_engine *e, int on);
/* targetConfigurationCallback (:1683-1687) for instances [first, first + count): rows [count][legs][dof] (HOST); a leg whose first
 * entry is NaN is not named in the message and stays where it is.  (A leg the message names partly takes UNASSIGNED_VALUE for the
 * joints it omits in the reference; name whole legs.) */
int shc_engine_set_target_configuration(shc_engine *e, int64_t first, int64_t count, const double *configuration);
/* targetBodyPoseCallback (:1691-1702): rows [count][7] (x,y,z,qw,qx,qy,qz), HOST.  (The reference leaves target_body_pose_
 * uninitialised until the first plan step has completed; here it starts as the identity.) */
int shc_engine_set_target_body_pose(shc_engine *e, int64_t first, int64_t count, const double *pose);
/* One StateController::loop() in planner mode for EVERY instance.  A robot that stands runs executePlan: progress[i] = 0 .. 100 of
 * the plan step being executed (100: plan_step_ advances, the acquired flags and the target body pose are reset), or
 * SHC_PLAN_WAITING when nothing has been acquired (the node republishes its request for plan_step[i]; Model::updateModel runs,
 * :666).  A robot that is still walking gets SHC_PLAN_WALKING, its velocity inputs are zeroed (:691-697) and its loop is one
 * ordinary control cycle, launched by this call for those robots only.  progress / plan_step: HOST [n], may be NULL. */
int shc_engine_execute_plan(shc_engine *e, int32_t *progress, int32_t *plan_step);

int64_t shc_sizeof_instance_state(void);
/* Instances [first, first + count) -> states[0 .. count) (host array).  Synchronises the engine's stream. */
int shc_engine_get_state(shc_engine *e, int64_t first, int64_t count, shc_instance_state *states);
/* states[0 .. count) (host array) -> instances [first, first + count).  Inputs (velocity, IMU, forces, efforts, pose
 * inputs' reset mode) are not part of the state and stay as set. */
int shc_engine_set_state(shc_engine *e, int64_t first, int64_t count, const shc_instance_state *states);

/*
 * Fleets: mixed morphologies (BASELINE.json configs[4]) and several GPUs of one node (configs[3]) behind one handle, for hosts
 * that drive all devices from ONE process (a one-process-per-GPU host creates one engine per rank and exchanges with RCCL, see
 * bench.py).  morph_id[i] in [0, n_morphologies) assigns instance i to params[morph_id[i]] (NULL = all 0).  Instances are binned
 * by morphology - the cycle kernel keeps one morphology's tables in LDS and maps one leg to one lane - and every bin is split
 * into contiguous shards over device_ids[0 .. n_devices) (NULL / 0 = device 0); each (bin, device) part is an engine on its own
 * HIP stream.  Nothing is exchanged while stepping.  All arrays are HOST arrays in the CALLER's instance order; per-leg arrays
 * are padded to [n][max_legs][max_k] (shc_fleet_shape), outputs are NaN where a morphology has no such leg / joint.
 */
typedef struct shc_fleet shc_fleet;
int shc_fleet_create(const shc_params *params, int n_morphologies, const int32_t *morph_id, int64_t n_instances, const int *device_ids,
                     int n_devices, shc_fleet **out);
int shc_fleet_destroy(shc_fleet *f);
int64_t shc_fleet_instances(const shc_fleet *f);
int shc_fleet_shape(const shc_fleet *f, int *max_legs, int *max_dof);
/* The parts (one engine per (morphology bin, device)): for device-resident I/O through the shc_engine_* calls. */
int shc_fleet_part_count(const shc_fleet *f);
int shc_fleet_part(const shc_fleet *f, int k, shc_engine **engine, int *morphology, int *device, int64_t *n_instances);
int shc_fleet_part_instances(const shc_fleet *f, int k, int64_t *ids /* [n_instances of part k]: caller's instance ids, ascending */);
int shc_fleet_set_velocity(shc_fleet *f, const double *linear_xy /* [n][2] */, const double *angular /* [n] */);
int shc_fleet_set_imu(shc_fleet *f, const double *orientation_wxyz /* [n][4] */, const double *angular_velocity /* [n][3] */);
int shc_fleet_set_pose_input(shc_fleet *f, const double *translation_velocity /* [n][3] */, const double *rotation_velocity /* [n][3] */);
int shc_fleet_set_tip_force(shc_fleet *f, const double *tip_force /* [n][max_legs][3] */);
int shc_fleet_set_joint_effort(shc_fleet *f, const double *joint_effort /* [n][max_legs][max_dof] */);
int shc_fleet_step(shc_fleet *f, int n_cycles);
int shc_fleet_synchronize(shc_fleet *f);
int shc_fleet_get_joint_state(shc_fleet *f, double *q, double *qd /* [n][max_legs][max_dof], either may be NULL */);
int shc_fleet_get_walk_state(shc_fleet *f, int32_t *walk_state /* [n] */);
/* The exchange step of a sharded batch: every device ends up with the desired joint positions of ALL instances
 * ([n][max_legs][max_dof], caller's order, NaN padded) in its own HBM, copied device to device (hipMemcpyPeerAsync: xGMI on an
 * MI355X node).  device_buffers[d] (may be NULL) receives device_ids[d]'s buffer; the buffers belong to the fleet. */
int shc_fleet_all_gather_joints(shc_fleet *f, double **device_buffers);

#ifdef __cplusplus
}
#endif
#endif /* SHC_BATCH_H */
