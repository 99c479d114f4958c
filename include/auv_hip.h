/* auv_hip.h — C ABI of the MI355X-native batched gym-auv step() path.
 *
 * The reference (krisbrud/gym-auv) is pure Python and has NO FFI layer of its own; the
 * boundary it offers for this path is the gym.Env surface
 *   reset()            gym_auv/environment.py:176-245
 *   step(action)       gym_auv/environment.py:292-366
 *   observe()          gym_auv/environment.py:247-290
 * so every entry point below cites the reference method whose per-environment work it
 * performs for a whole batch of N independent environments.  Plain pointers and sizes
 * only (no torch types); the Python side binds it with ctypes.
 *
 * Conventions
 *   - return 0 on success, negative AUV_E* on failure; text via auv_last_error().
 *   - "dev" pointers are device (HBM) pointers owned by the CALLER (e.g. torch tensors);
 *     the handle owns environment state, world bank and scratch.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  auv_step and the
 *     per-kernel entry points are stream-ordered: no allocation, no host sync, hipGraph
 *     capturable.
 *   - one handle per GPU, one host thread per handle (the reference env is single-threaded
 *     and non-re-entrant; nothing stronger is promised).
 *   - all arithmetic is IEEE fp64 ("f64"); obs/reward are additionally emitted as fp32
 *     because observation_space.dtype is float32 (environment.py:139-143).
 */
#ifndef AUV_HIP_H
#define AUV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUV_ABI_VERSION 4

enum {
  AUV_OK = 0,
  AUV_EINVAL = -1,   /* bad argument / shape mismatch            */
  AUV_EHIP = -2,     /* a HIP runtime call failed                */
  AUV_ESTATE = -3,   /* call order violated (e.g. no worlds yet) */
  AUV_ENOMEM = -4
};

enum { AUV_REWARD_COLAV = 0, AUV_REWARD_PATHFOLLOW = 1 };       /* rewarder.py:143, :56 */
enum { AUV_CULL_REFERENCE = 0, AUV_CULL_EXACT = 1 };            /* sensor.py:74-97 vs :100-137 */
enum { AUV_OBS_RING = 0, AUV_OBS_FILLED = 1, AUV_OBS_MOVER = 2 };
enum { AUV_F32 = 0, AUV_F64 = 1 };

/* Mirrors the fields of gym_auv/config.py that step() reads. */
typedef struct auv_config {
  double dt;                      /* SimulationConfig.t_step_size        config.py:28  */
  double min_goal_distance;       /* EpisodeConfig.min_goal_distance     config.py:20  */
  double min_path_progress;       /* EpisodeConfig.min_path_progress     config.py:23  */
  double min_cumulative_reward;   /* EpisodeConfig.min_cumulative_reward config.py:16  */
  double sensor_range;            /* VesselConfig.sensor_range           config.py:65  */
  double vessel_width;            /* VesselConfig.vessel_width           config.py:41  */
  double look_ahead_distance;     /* VesselConfig.look_ahead_distance    config.py:45  */
  double thrust_max;              /* VesselConfig.thrust_max_auv         config.py:39  */
  double moment_max;              /* VesselConfig.moment_max_auv         config.py:40  */
  int32_t max_timesteps;          /* EpisodeConfig.max_timesteps         config.py:19  */
  int32_t n_sensors;              /* n_sectors * n_sensors_per_sector    config.py:75  */
  int32_t sensor_interval_load_obstacles;                             /* config.py:56  */
  int32_t use_lidar;              /* VesselConfig.use_lidar              config.py:52  */
  int32_t sensor_log_transform;   /* VesselConfig.sensor_log_transform   config.py:66  */
  int32_t rewarder;               /* AUV_REWARD_*                                      */
  int32_t test_mode;              /* BaseEnvironment(test_mode=)   environment.py:32   */
  int32_t cull_mode;              /* AUV_CULL_*                                        */
  int32_t auto_reset;             /* VecEnv semantics: a done env restarts on its next
                                     world of the bank inside the same step            */
  int32_t obs_channels;           /* 1: closeness only; 3: + two velocity channels, which the
                                     reference hard-wires to zero (sensor.py:159, config.py:80-91) */
} auv_config_t;

/* World bank: W pre-generated scenario instances in CSR form (host pointers; copied to HBM
 * by auv_load_worlds).  Built by gym_auv_amd.world from WorldSpec, i.e. from what the
 * reference's _generate() hands to Path/Vessel/obstacles (envs/movingobstacles.py:28-95).
 *
 *  path      dense polyline of Path._points (path.py:38-40) + cumulative arclength (the
 *            measure GEOS LineString.project walks), and the final PCHIP as per-interval
 *            power-basis coefficients (path.py:26-27).
 *  obstacles per obstacle {kind, seg_off, nseg, mover_idx} + static cull circle
 *            (obstacles.py:108-113, :125-127); static boundary segments (obstacles.py:
 *            101-106, :122-123) as (ax, ay, bx, by).
 *  movers    VesselObstacle parameters and reset-time state (obstacles.py:144-215).
 */
typedef struct auv_world_bank {
  int32_t n_worlds;
  /* path */
  const int64_t* poly_off;    /* [W+1]  vertex offsets                                  */
  const double* poly_xy;      /* [sum P][2]                                             */
  const double* poly_cum;     /* [sum P]   cum[j] = arclength at vertex j               */
  const int64_t* knot_off;    /* [W+1]  knot offsets (n_k knots -> n_k-1 intervals)     */
  const double* knot_s;       /* [sum n]                                                */
  const double* knot_coef;    /* [sum n][8] x:c0..c3 (highest power first), y:c0..c3;
                                  row i describes interval [s_i, s_{i+1}); last row of
                                  each world unused                                     */
  const double* world_scalar; /* [W][8] L, end_x, end_y, init_x, init_y, init_psi, 0, 0 */
  /* obstacles */
  const int64_t* obs_off;     /* [W+1]                                                  */
  const int32_t* obs_meta;    /* [sum K][4] kind, seg_off (into seg, absolute), nseg,
                                  mover index within the world (or -1)                  */
  const double* obs_cull;     /* [sum K][3] cx, cy, rho (static obstacles)              */
  int64_t n_seg;
  const double* seg;          /* [n_seg][4] ax, ay, bx, by                              */
  /* movers */
  const int64_t* mv_off;      /* [W+1]                                                  */
  const double* mv_param;     /* [sum M][4] width, pos0_x, pos0_y, n_vel                */
  const double* mv_init;      /* [sum M][4] pos_x, pos_y, heading, counter at reset     */
  const int64_t* mv_vtab_off; /* [sum M + 1] offsets into mv_vtab                       */
  const double* mv_vtab;      /* [.][2] per-tick velocities (length 1 = constant)       */
} auv_world_bank_t;

typedef struct auv_handle auv_handle_t;

/* Field ids for auv_read / auv_write (handle-owned device buffers, all fp64 unless noted):
 * copies between the handle and a caller buffer, stream-ordered.  Test/debug surface. */
enum {
  AUV_FIELD_STATE = 0,       /* [6][N]  x, y, psi, u, v, r   (Vessel._state, SoA)          */
  AUV_FIELD_LIDAR_D = 1,     /* [N][S]  Vessel._last_sensor_dist_measurements              */
  AUV_FIELD_OBS64 = 2,       /* [N][6+S] observation before the float32 cast               */
  AUV_FIELD_REWARD64 = 3,    /* [N]                                                        */
  AUV_FIELD_INFO64 = 4,      /* [N][8] collision, reached_goal, goal_distance, progress,
                                 cumulative_reward, max_progress, vessel_arclength, sum of |cross-track
                                 error| [m] over the steps of the episode so far               */
  AUV_FIELD_WORLD_IDX = 5,   /* [N] int32                                                  */
  AUV_FIELD_COUNTERS = 6,    /* [N][4] int32: t_step, vessel step_counter, episodes, pad   */
  AUV_FIELD_MOVER_STATE = 7, /* [N][Mmax][4] pos_x, pos_y, heading, counter                */
  AUV_FIELD_NEARBY = 8,      /* [N][Kmax] uint8  cached Vessel._nearby_obstacles mask      */
  AUV_FIELD_EPISODE = 9,     /* [N][4] last finished episode: return, length, collision,
                                 reached_goal (fp64)                                        */
  AUV_FIELD_CULL_LIMITS = 10,/* [N][Kmax][2] int32 idx_min_ray, idx_max_ray (sensor.py:58-69)
                                 of the last sweep; INT32_MIN where not evaluated           */
  AUV_FIELD_NAV64 = 11,      /* [N][8] unclipped Vessel.navigate outputs: u, v, r,
                                 look_ahead_heading_error, heading_error, cross_track_error/100,
                                 path_direction, target_arclength        (vessel.py:518-536) */
  AUV_FIELD_COLLISION = 12,  /* [N] uint8  Vessel._collision                                */
  AUV_FIELD_STAMPS = 13,     /* [N][16] uint64 per-phase cycle counts; zeros unless the library
                                 was built with STAMPS=1 (diagnostic)                       */
  AUV_FIELD_BROKEN = 15,     /* [N] uint8  environments whose step a wave that gave up polling has left unfinished
                                 (all zero except between a hand-over time-out and the call that recovers)     */
  AUV_FIELD_FW_STATE = 16,   /* [W] int32  fresh-world mode only: 0 READY (unseen), 1 IN_USE, 2 STALE (queued / being rebuilt) */
  AUV_FIELD_FW_SERIAL = 17,  /* [W] int32  fresh-world mode only: the slot holds world number `serial` of its environment    */
  AUV_FIELD_STEP_INFO = 14   /* [N][4] the `info` dict of the last step() as the reference
                                 returns it (environment.py:336-340): collision, reached_goal,
                                 goal_distance, progress -- of the step that was taken, i.e. the
                                 terminal values for an env that has just been auto-reset      */
};

/* Create an environment batch of n_envs on device `device_id`.            (environment.py:29-164) */
int auv_create(const auv_config_t* cfg, int32_t n_envs, int32_t device_id, auv_handle_t** out);
int auv_destroy(auv_handle_t* h);

/* Upload a world bank (replaces any previous one).                 (BaseEnvironment._generate) */
int auv_load_worlds(auv_handle_t* h, const auv_world_bank_t* bank);

/* reset(): for every env whose mask byte is non-zero (mask_dev == NULL: all) bind it to
 * world_idx_dev[e] (NULL: keep current binding; initial binding is e % W), restore vessel /
 * mover state and counters, then compute the first observation.          (environment.py:176-245) */
int auv_reset(auv_handle_t* h, const uint8_t* mask_dev, const int32_t* world_idx_dev,
              float* obs_dev, void* stream);

/* step(): _update -> vessel.step -> observe -> reward -> done, for all N envs.
 * actions_dev: [N][2] (thrust, rudder) of action_dtype AUV_F32/AUV_F64, 8- / 16-byte aligned.
 * obs_dev [N][6+S] f32, reward_dev [N] f32, done_dev [N] u8.             (environment.py:292-366) */
int auv_step(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev,
             float* reward_dev, uint8_t* done_dev, void* stream);

/* step() of the environments [e0, e0 + ne) only, enqueued on `stream`; the buffers are the same full-batch
 * [N][..] buffers auv_step takes (only the slice's rows are read / written).  Environments are independent
 * (environment.py:86-89), so sub-batches of one handle may be stepped on DIFFERENT streams concurrently: one
 * sub-batch's LiDAR sweeps then run under another's dynamics chain and navigation tail -- the batched analogue
 * of the reference's SubprocVecEnv workers stepping at their own pace (scripts/run.py:293-296; VecEnv
 * step_async / step_wait).  Results are bit-identical to auv_step over the same environments.  Ordering between
 * sub-batches and with the consumer of obs / reward / done is the CALLER's (stream order, events).  Eager only
 * (a captured graph steps the whole batch).                                                                  */
int auv_step_slice(auv_handle_t* h, int32_t e0, int32_t ne, const void* actions_dev, int32_t action_dtype,
                   float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
/* The same for ALL sub-batches with one call: slice i = [bounds[i], bounds[i + 1]) goes to streams[i]
 * (bounds[0] = 0, bounds[n_slices] = N; HOST arrays).  One step of the whole batch as n_slices independent
 * launch chains -- what a VecEnv.step_async() of this handle enqueues.                                        */
int auv_step_pipelined(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams,
                       const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
                       uint8_t* done_dev);

/* SEVERAL steps of every slice in ONE launch per slice (open-loop stretches: the actions are resident).  actions_dev is a ring of
 * n_slots consecutive [N][2] buffers; step k of the call reads slot (first_slot + k) % n_slots.  The launch is the one-launch step's
 * grid n_steps times over, step-major: an environment's step k + 1 starts when ITS finish wave of step k is through -- no barrier
 * over the slice between two steps, no launch turn-around.  What the kernel boundary between two launches did is done by a
 * per-environment carry record (agent-scope stores / loads, checksummed, every mark carrying the step's number): csrc/
 * k_step_fused.hip, k_step_multi.  obs / reward / done hold the LAST step's values afterwards; everything else -- state, counters,
 * episode log, auto-reset -- is bit for bit what n_steps calls of auv_step_pipelined leave (tests/test_gpu_multi.py).  Needs the
 * one-launch shape for every slice and at most 64 obstacles per world; refused with a fresh world per reset.  Polls are bounded and
 * report like the one-launch step's (auv_health).                                                                     */
int auv_step_multi(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev,
                   int32_t action_dtype, int32_t n_slots, int32_t first_slot, int32_t n_steps, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev);

/* Workgroup order of auv_step_multi's launches (same results either way).  order 0: step-major (all of step t, role by role, then
 * step t + 1).  order 1 (default): cohort-pipelined -- cohorts of 64 environments; the sweeps of a cohort-step are dispatched `lead`
 * cohort positions behind its dynamics and its finish waves `lag` positions behind the sweeps, so a wave finds its inputs instead of
 * holding a wave slot while it polls; lead + lag are cut down to (cohorts of the slice) - 1, which keeps every producer ahead of its
 * consumer in dispatch order; slices that are not a multiple of 64 environments, or of fewer than 3 cohorts, use order 0.
 * Defaults: order 1, lead 16, lag 30 (measured at 4096 x 180: tools/lead_lag_grid.sh).                                           */
int auv_set_multi_order(auv_handle_t* h, int32_t order, int32_t lead, int32_t lag);
/* How many obstacle boundary segments the LiDAR wave stages in LDS per batch of its pair sweep (sensor.py:140-159 is evaluated per
 * (segment, ray) pair; a crowded environment takes several batches).  Picked when a bank is loaded or generated: the largest of
 * 32 .. 96 that leaves the one-launch step its best occupancy -- every role of that launch is charged the sweep's LDS slice, so at
 * 256 beams + 47 obstacles + 17 movers a 96-segment stage (12.7 KB) means 12 waves per CU, a 34-segment one 16.  Results do not
 * depend on it (tested bit for bit).  segments = 0: only report; 32 .. 96 (even): override (a new bank picks again; graphs captured
 * before keep the stage they were captured with).  *out_segments (may be NULL) <- the value in force.              */
int auv_lidar_stage(auv_handle_t* h, int32_t segments, int32_t* out_segments);

/* VecEnv.step_async / step_wait (scripts/run.py:293-296: SubprocVecEnv sends the actions to its workers and collects their
 * results) with the ordering between the caller's stream and the chains done INSIDE the library, one call each:
 *   auv_step_async  the actions were produced on `caller_stream`; slice i is stepped on streams[i] behind them.  A slice whose
 *                   stream IS caller_stream is simply enqueued there (stream order, no hand-over at all).
 *   auv_step_wait   work enqueued on `caller_stream` after this call sees obs / reward / done of every slice.
 * Neither call blocks the host.  `rendezvous` picks the mechanism for the slices on other streams:
 *   AUV_RDV_EVENTS  one hipEventRecord behind the actions + one hipStreamWaitEvent per chain; one record + wait per chain back.
 *   AUV_RDV_DEVICE  one-wave kernels and two words in device memory: the caller's stream publishes "actions of step t ready",
 *                   a one-wave kernel in front of each chain's launch polls for it (bounded), a one-wave kernel behind it counts
 *                   the chain off, ONE polling kernel on the caller's stream waits for all of them (csrc/k_step_fused.hip:
 *                   k_rdv_*).  Every waiter waits for something submitted before it.  Needs streams whose kernels really
 *                   execute side by side: where dispatches are serialised -- a counter-collecting profiler -- a polling
 *                   kernel would sit in front of the kernel it waits for.  So (i) the first call on a set of streams runs a
 *                   TRIAL of the pattern with nothing at stake (50 ms limit; synchronises those streams once): if it runs out,
 *                   and whenever a slice is not stepped in the one-launch shape, the handle uses AUV_RDV_EVENTS instead, for
 *                   good (auv_health [7]); (ii) a real wait that runs out (auv_set_rendezvous_limit, default 10 s: raise it if
 *                   the caller's stream can be busy for longer between two steps) is reported by the next call (AUV_ESTATE,
 *                   once), and a chain whose gate ran out does NOT step: the gate raises the abort flag and the launch behind
 *                   it -- and every launch queued behind that -- hands out ABORT packets (environments untouched).
 *   AUV_RDV_CP      the same words written and awaited by the command processors (hipStreamWriteValue64 / WaitValue64).
 * Results are bit-identical to auv_step.  What a full rendezvous per step costs, and why K chains cannot beat ONE launch
 * when every step waits for all of them, is in DESIGN.md section 4.                                                      */
enum { AUV_RDV_EVENTS = 0, AUV_RDV_DEVICE = 1, AUV_RDV_CP = 2 };
int auv_step_async(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev,
                   int32_t action_dtype, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* caller_stream,
                   int32_t rendezvous);
int auv_step_wait(auv_handle_t* h, void* caller_stream);
int auv_set_rendezvous_limit(auv_handle_t* h, double seconds);

/* The same with every sub-batch's launch stamped with its own start / stop HIP event on ITS stream (the kernel's own
 * duration while the other chains run beside it, as a kernel trace reports it): out_ms[n_slices].  Waits for the
 * step's launches (not for earlier work on other streams).  One-launch shape only.                            */
int auv_step_pipelined_timed(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams,
                             const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
                             uint8_t* done_dev, float* out_ms);

/* Episode log: what the reference appends to env.history when an episode ends (save_latest_episode,
 * environment.py:466-489), for all environments of the batch, in completion order.  Row = 8 doubles: environment,
 * return (cumulative reward), timesteps, collision, reached_goal, progress, mean |cross-track error| in metres over the
 * episode's steps (_save_latest_step, environment.py:460-464), world index.  Copies rows [first, first + max_rows) of
 * the log (as far as they exist) to dst_dev on `stream` and returns the number of episodes logged so far in
 * *out_total (this read synchronises `stream`; the count is 64-bit and never wraps).  The device keeps the newest
 * 2^k >= max(65536, 4 N) rows.  A reader that has fallen further behind than that is not refused: the copy starts at the
 * oldest row still held, *out_first (nullable) tells which row that is (= first when nothing was lost; *out_first - first
 * rows were overwritten before they were read), and the caller carries on from *out_first + rows copied.  The log
 * restarts (count 0) whenever a bank is loaded or generated.                                                        */
int auv_episode_log(auv_handle_t* h, double* dst_dev, int64_t max_rows, int64_t first, int64_t* out_total,
                    int64_t* out_first, void* stream);

/* Do kernels on these two streams run side by side?  HIP multiplexes streams onto a few hardware queues (four by
 * default); two streams that land on the same one run their kernels one after the other, and sub-batch chains on them
 * do not overlap.  Launches a 300 us do-nothing wave on each and reports (wall time until both have ended) / 300 us:
 * about 1 when they overlap, about 2 when they do not.  Synchronises both streams.  For choosing the streams of
 * auv_step_pipelined (BatchedAuvEnv.set_sub_batches does).                                                        */
int auv_streams_overlap(auv_handle_t* h, void* stream_a, void* stream_b, float* out_ratio);

/* The three kernels of step(), individually launchable (per-kernel parity tests):        */
/* K1  Vessel.step: clip -> RKF45 of the 3-DOF model -> wrap psi.  (vessel.py:226-247,561-578) */
int auv_step_dynamics(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, void* stream);
/* K2  _update (movers) + Vessel.perceive: LiDAR sweep.   (environment.py:386-392, vessel.py:249-368)
 *     advance_movers = 0 skips the mover update (observe() at reset time).                 */
int auv_lidar(auv_handle_t* h, int32_t advance_movers, void* stream);
/* K3  Vessel.navigate + rewarder.calculate + _isdone + observation assembly.
 *     (vessel.py:461-541, rewarder.py:78-241, environment.py:263-280, 325-347, 375-384)
 *     mode 0: full; 1: navigation + observation only (reset path), no reward/done;
 *     2: reward/done only, from NAV64 / INFO64 / LIDAR_D / COLLISION as they stand (test hook). */
int auv_nav_reward(auv_handle_t* h, int32_t mode, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev, void* stream);

/* Stream-ordered copies between handle-owned buffers and caller DEVICE buffers.           */
int auv_read(auv_handle_t* h, int32_t field, void* dst_dev, size_t bytes, void* stream);
int auv_write(auv_handle_t* h, int32_t field, const void* src_dev, size_t bytes, void* stream);
/* Size in bytes of a field for this handle (0 if unknown).                                 */
size_t auv_field_bytes(const auv_handle_t* h, int32_t field);

/* Capture one auv_step into a hipGraph bound to the given I/O pointers and replay it.     */
int auv_graph_capture(auv_handle_t* h, const void* actions_dev, int32_t action_dtype,
                      float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
int auv_graph_launch(auv_handle_t* h, void* stream);
/* The same with n_steps consecutive steps in ONE graph, so that the fixed cost of a replay is paid once
 * per n_steps steps (open-loop stretches: the action ring feeds step k of the replay from slot
 * (position + k) % n_slots; without a ring every step reads the same buffer).  obs / reward / done
 * hold the LAST step's values after a replay; per-env episode statistics keep accumulating.          */
int auv_graph_capture_steps(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                            float* reward_dev, uint8_t* done_dev, int32_t n_steps, void* stream);

/* Captured CHAINS (BASELINE configs[4] names a hipGraph-captured step): n_steps consecutive steps of every slice in the
 * one-launch shape, each slice a chain with a position of its own in the action ring (so the chains may drift apart like
 * eager chains do; all of them read the one ring of auv_set_action_ring).
 *   one_graph = 0   one LINEAR graph per slice; auv_graph_launch_chains replays slice i's graph on streams[i] -- which
 *                   hardware queue a chain runs on stays the caller's choice (auv_streams_overlap), as for eager chains.
 *   one_graph = 1   ONE graph whose n_slices branches are the chains (fork behind the root, join at the end), replayed
 *                   with auv_graph_launch; the runtime places the branches.
 * obs / reward / done hold the last step's values after a replay.                                                    */
/* (A slice stepped in the three-launch shape -- set_step_mode, no LiDAR, hand-overs disabled -- is captured in that shape and
 * advances its own ring position just the same.)                                                                      */
int auv_graph_capture_chains(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, const void* actions_dev,
                             int32_t action_dtype, float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t n_steps,
                             int32_t one_graph);
int auv_graph_launch_chains(auv_handle_t* h, int32_t n_slices, void* const* streams);

/* Action ring (captured graphs only): after this call `actions_dev` of auv_graph_capture is a ring
 * of n_slots consecutive [N][2] buffers; replayed step k reads slot k % n_slots (the position lives
 * on the device and is advanced by the step itself), so a policy can fill slot k+1 while step k
 * runs and a replayed hipGraph needs no per-step argument update.  n_slots = 1 restores the plain
 * buffer.  Invalidates a captured graph.  EAGER auv_step / auv_step_timed never use the ring: they
 * read `actions_dev` as one plain [N][2] buffer and leave the ring position alone (a caller that
 * launches eagerly passes whatever pointer it likes per step).  Loading or generating a bank
 * resets the ring to one slot.                                                                */
int auv_set_action_ring(auv_handle_t* h, int32_t n_slots);

/* How auv_step / auv_step_slice / auv_graph_capture run a step (same results, bit for bit):
 *   AUV_STEP_AUTO (default)          AUV_STEP_ONE_LAUNCH for launches of fewer than 65536 environments,
 *                                    AUV_STEP_SIDE_BY_SIDE from there on (beyond the sizes the one launch was
 *                                    measured ahead at).
 *   AUV_STEP_ONE_LAUNCH              the whole step in ONE launch of one-wave workgroups with four roles: the
 *                                    first n / 8 integrate the dynamics (K1, eight environments per wave), the
 *                                    next n sweep the LiDAR of one environment each (K2), the next n search the
 *                                    nearest point of one environment's path each (the wide part of K3-nav), the
 *                                    last n / 8 -- eight environments per wave -- evaluate the navigation's scalar
 *                                    tail and run reward / done / auto-reset (K3-reward).  The roles hand their
 *                                    results on inside the launch through per-environment words and checksummed
 *                                    64-byte records stored and loaded coherently (csrc/k_step_fused.hip:
 *                                    k_step_roles, roles_finish_wave); waves that need a result poll for it, bounded.
 *   AUV_STEP_SIDE_BY_SIDE            K1 -> [K2 and K3-nav side by side in one launch] -> K3-reward: nothing is
 *                                    handed over inside a launch.  Inside a captured graph of several steps
 *                                    K3-reward of step t and K1 of step t + 1 share a launch (also what a graph
 *                                    of several steps captured in the modes above uses: it replays faster).
 * The in-launch hand-overs of ONE_LAUNCH assume that the workgroups of a launch are dispatched in index
 * order (a polling wave's producer has a smaller index; true on gfx950, not promised by HIP).  Every bank load
 * therefore runs a probe launch of the same structure (more one-wave workgroups than the chip has slots, three
 * generations polling each other); if any of its polls runs out, and whenever the LiDAR is off, the handle steps in
 * AUV_STEP_SIDE_BY_SIDE whatever mode is set.  Should a poll of a real step ever run out (bounded: seconds), the
 * wave marks the environments it leaves unfinished and raises a device-wide flag on which the launches queued behind
 * it do nothing (their dynamics role hands out ABORT packets); the NEXT call on the handle returns AUV_ESTATE once,
 * having reset exactly the marked environments (auv_health tells which launch reported and how many), cleared the
 * hand-over words and switched the handle to AUV_STEP_SIDE_BY_SIDE for good; later calls succeed.  A step
 * replayed from a captured graph (hipGraphLaunch, or a torch CUDAGraph around auv_step) is not checked per replay:
 * poll auv_health() once per rollout there.
 * (Removed in round 3, measured slower: the whole step as one kernel, [K1 + K3-nav] -> [K2 + K3-reward], K3-nav
 * forked onto a second stream, and K1 -> [K2 | K3-nav + K3-reward]; their enum values 1 .. 4 are rejected.)      */
enum { AUV_STEP_SIDE_BY_SIDE = 0, AUV_STEP_ONE_LAUNCH = 5, AUV_STEP_AUTO = 6 };
int auv_set_step_mode(auv_handle_t* h, int32_t mode);
/* The shape a launch of n_envs_per_launch environments (<= 0: the whole batch) is really stepped in: AUV_STEP_*. */
int auv_effective_step_mode(auv_handle_t* h, int32_t n_envs_per_launch);
/* out8: [0] 1 = in-launch hand-overs in use, [1] polls of the last dispatch-order probe that ran out (-1: no bank yet),
 *       [2] hand-over time-outs over the life of the handle, [3] 1 = a time-out is pending (the next step call
 *       will recover and return AUV_ESTATE), [4] / [5] e0 / ne of the launch that reported the last time-out (-1: none),
 *       [6] environments the last recovery reset, [7] 0 while AUV_RDV_DEVICE is in use, else the number of rendezvous
 *       waits (trial included) that ran out -- events from then on.  Reads host memory only: no synchronisation.
 * A recovery writes the reset observation of the environments it resets into OBS64 and into the obs buffer of the CALL THAT
 * RECOVERS (if that call has one) -- never through a pointer remembered from an earlier call.                     */
int auv_health(auv_handle_t* h, int32_t* out8);
/* The dispatch-order probe on the streams production uses: one probe launch per stream, all in flight together, a foreign
 * kernel behind each, two rounds (BatchedAuvEnv.set_sub_batches runs it on the chain streams it has chosen).  Updates
 * what auv_health reports in [0] / [1]; a probe that fails is not an error -- the handle then steps in the
 * three-launch shape.  Synchronises the device.                                                                   */
int auv_probe_streams(auv_handle_t* h, int32_t n_streams, void* const* streams);

/* One step with every dispatch stamped with its own start / stop HIP event on `stream` (the kernel's
 * own duration, as a kernel trace reports it).  out_ms[0..3] by effective mode:
 *   AUV_STEP_ONE_LAUNCH    the one launch, 0, 0, whole step
 *   AUV_STEP_SIDE_BY_SIDE  K1, [K2 + K3-nav], K3-reward, whole step. */
int auv_step_timed(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                   float* reward_dev, uint8_t* done_dev, void* stream, float* out_ms4);

/* Feasibility pooling (optional post-kernel on the current LiDAR ranges):
 * for every env and sector k (sensors sector_start[k] .. sector_start[k+1]-1) the feasible
 * distance of LidarPreprocessor._feasibility_pooling (sensor.py:251-296) with opening width
 * `width` (= vessel_width * feasibility_width_multiplier) and sensor spacing 2*pi/S.
 * sector_start_dev: [n_sectors+1] int32; out_dist_dev: [N][n_sectors] fp64 (nullable);
 * out_closeness_dev: [N][n_sectors] f32 closeness of those distances (nullable).            */
int auv_feasibility_pooling(auv_handle_t* h, const int32_t* sector_start_dev, int32_t n_sectors, double width,
                            double* out_dist_dev, float* out_closeness_dev, void* stream);

/* On-device scenario generation: build n_worlds MovingObstacles-type worlds on
 * the GPU, one workgroup per world, straight into fixed-capacity world slots, then compute their
 * reset rows and reset every environment -- the device-side replacement of
 * MovingObstacles._generate (envs/movingobstacles.py:28-95: RandomCurveThroughOrigin path with its
 * three PCHIP passes, objects/path.py:19-40,96-120; vessel start; helpers.generate_obstacle
 * placements, utils/helpers.py:5-35; VesselObstacle / CircularObstacle tables,
 * objects/obstacles.py:90-113,144-215).  Replaces any previous bank, like auv_load_worlds.
 *   draws_dev [n_worlds][n_draws] fp64 DEVICE memory, n_draws = 11 + n_moving*(3*C+2) + n_static*3*C
 *   with C = AUV_GEN_CAND: row = u_nwaypoints, u_angle, 6 waypoint jitters, start ux, uy, upsi, then
 *   per mover C x (z ~ N(0,1), u, Poisson(10)) + u_direction + u_speed, then per circle
 *   C x (z, u, Poisson(30)); u ~ U[0,1).  The unbounded rejection loop of generate_obstacle is
 *   a pool of C candidates (first accepted wins); if the whole pool is rejected (~1e-13 per obstacle)
 *   up to 56 more come from a counter-based generator keyed by the pool (gym_auv_amd/devgen.py,
 *   extra_candidate), so that no obstacle is placed on the vessel or the goal.
 *   ring_unit [65][2], nseg_by_radius [n_radius] HOST tables: unit ring of the GEOS point buffer
 *   and the number of segments Douglas-Peucker(0.3) leaves for an integer radius (4..64, power
 *   of two); needed on the first call of a given shape, ignored afterwards.
 * A second call with the same (n_worlds, n_moving, n_static) regenerates in place without
 * allocating.  Synchronous.                                                                   */
#define AUV_GEN_CAND 8
#define AUV_GEN_POLY_CAP 16384   /* polyline vertices per generated world slot (path <= 1638 m) */
int auv_generate_worlds(auv_handle_t* h, int32_t n_worlds, int32_t n_moving, int32_t n_static,
                        const double* draws_dev, int32_t n_draws, const double* ring_unit,
                        const int32_t* nseg_by_radius, int32_t n_radius);

/* Read a table of a GENERATED bank back (device-to-device copy; slot layout: world w owns
 * [w*cap, (w+1)*cap) of each table, counts in AUV_B_POLY_CNT / OBS_META).  For tests and for
 * checkpointing generated worlds.                                                              */
enum {
  AUV_B_POLY_CNT = 0,     /* [W] int32                      */
  AUV_B_POLY_XY = 1,      /* [W][AUV_GEN_POLY_CAP][2]       */
  AUV_B_POLY_CUM = 2,     /* [W][AUV_GEN_POLY_CAP]          */
  AUV_B_KNOT_S = 3,       /* [W][1000]                      */
  AUV_B_KNOT_COEF = 4,    /* [W][1000][8]                   */
  AUV_B_WORLD_SCALAR = 5, /* [W][8]                         */
  AUV_B_OBS_META = 6,     /* [W][K][4] int32                */
  AUV_B_OBS_CULL = 7,     /* [W][K][3]                      */
  AUV_B_SEG = 8,          /* [W][64*n_static][4]            */
  AUV_B_MV_PARAM = 9,     /* [W][M][4]                      */
  AUV_B_MV_INIT = 10,     /* [W][M][4]                      */
  AUV_B_MV_VTAB = 11,     /* [W][M][2]                      */
  AUV_B_CHUNK_BOUND = 12  /* [W][AUV_GEN_POLY_CAP/64][4]    */
};
size_t auv_bank_bytes(const auv_handle_t* h, int32_t table);
int auv_read_bank(auv_handle_t* h, int32_t table, void* dst_dev, size_t bytes, void* stream);

/* A FRESH WORLD ON EVERY RESET (both halves joined): the reference builds a new scenario whenever an episode
 * ends (environment.py:176-218 reset() -> _generate(); envs/movingobstacles.py:28-95).  auv_generate_worlds gives a bank that
 * auto-reset cycles through ((w + N) % W): fine for throughput, not what the reference's learner sees.  This mode does:
 *   - the bank is `depth` slots per environment (slot e + j N belongs to environment e; depth >= 2); the world of environment e's
 *     k-th episode is the world of (seed, env_index_base + e, serial k): draws from a counter-based generator on the device
 *     (k5_generate.hip: k5_draws; gym_auv_amd/devgen.py: counter_draws is the host mirror), so what an environment meets does not
 *     depend on timing, on the sub-batch chains, or on how the batch is sharded over GPUs (env_index_base = the shard's first
 *     GLOBAL environment index);
 *   - an environment whose episode ends (auto-reset, or auv_reset after it has stepped) moves to its next slot and queues the one
 *     it leaves; a refill pass -- ONE graph launch on a side stream, enqueued by the step calls themselves every `period` calls,
 *     paced in GPU time behind the first chain's stream and never waited for -- rebuilds exactly the queued slots (up to
 *     `batch_cap` per pass): tables (k5_generate), reset rows (the step's own kernels on batch_cap shadow environments nobody
 *     steps), and its last kernel makes them bindable.  No host round trip: the host may be thousands of launches ahead.
 *     A finish wave that binds a slot reads its tables with agent-scope loads in that launch (the pass may have completed
 *     while the launch was running); every later launch starts behind a kernel boundary.  Running environments are never touched.
 *   - should an episode end before its environment's next slot is ready (the pass fell behind an episode of a few steps) the
 *     environment starts over in the world it has just finished, and the event is COUNTED (auv_fresh_worlds_stats [2]); size
 *     `depth` / `period` / `batch_cap` so that the count stays 0 (bench.py --fresh-worlds reports it).
 * The episode log's `world` column then holds e + N * serial (what the world's index would be in a bank that never repeats).
 * auv_fresh_worlds_create replaces the bank like auv_generate_worlds (same ring tables; synchronous) and resets every
 * environment; auv_load_worlds / auv_generate_worlds leave the mode.  auv_reset refuses an explicit world_idx in this mode.
 *   auv_fresh_worlds_refill   the step calls that cover the whole batch (auv_step, _pipelined, _async, auv_policy_rollout, graph
 *                             launches) tick by themselves; a caller that drives slices one by one (auv_step_slice) calls this
 *                             with the chains' slices and streams.  flush != 0: synchronises those streams, then runs passes
 *                             until the queue is empty and everything is published (tests; a deterministic hand-over point).
 *   auv_fresh_worlds_stats    out8: [0] mode on, [1] worlds rebuilt, [2] episodes that re-used their world (see above), [3] slots
 *                             queued now, [4] passes enqueued, [5] passes completed, [6] depth, [7] batch_cap.  One small D2H copy.
 *   auv_fresh_worlds_draws    the draws row of (environment, serial) for n_rows pairs (HOST arrays of environment indices
 *                             relative to this handle) -> dst_dev [n_rows][n_draws]: what the host mirror rebuilds a world from.
 * (The pass's stream is a plain stream: creating one with a non-default priority halves the throughput of four sub-batch chains
 * on this stack even while it idles -- tools/side_stream_ab.sh.)                                                       */
int auv_fresh_worlds_create(auv_handle_t* h, int32_t depth, int32_t n_moving, int32_t n_static, uint64_t seed, int64_t env_index_base,
                            int32_t batch_cap, int32_t period, const double* ring_unit, const int32_t* nseg_by_radius, int32_t n_radius);
int auv_fresh_worlds_refill(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, int32_t flush);
/* The stream the refill passes run on (default: a plain stream the library creates).  This GPU runs at most FOUR kernels side by
 * side and HIP multiplexes streams onto four hardware queues (FIFO each): a pass on a stream that shares a queue with a chain
 * stalls that chain for the pass's duration, and beside four busy chains a pass takes a slot from one of them whatever its
 * queue (tools/side_queue_probe.py: 153 -> 70 / 95 M env-steps/s).  So: three chains plus a pass stream chosen to run side by
 * side with all three (auv_streams_overlap) -- what BatchedAuvEnv.set_sub_batches does in this mode.  The stream stays the
 * caller's; it must outlive the mode.                                                                              */
int auv_fresh_worlds_set_stream(auv_handle_t* h, void* stream);
int auv_fresh_worlds_stats(auv_handle_t* h, int64_t* out8);
int auv_fresh_worlds_draws(auv_handle_t* h, const int32_t* envs_host, const int32_t* serials_host, int32_t n_rows, double* dst_dev,
                           void* stream);

/* ---- the policy in the loop (scripts/run.py:332-357: PPO2 with MlpPolicy, net_arch [256, 128, 64] for
 * policy and value function, tanh, diagonal Gaussian over the two actions) --------------------------------------------
 * auv_policy_act evaluates actor and critic for the environments [e0, e0 + ne) with ONE launch on `stream`, straight from
 * the observation rows the step has written: obs -> [256, 128, 64] tanh -> (mu[2] | value), f32 on the matrix cores
 * (v_mfma_f32_16x16x4_f32: exact f32), then in the epilogue a = mu + exp(log_std) eps (counter-based generator keyed by
 * seed, step counter, environment), log pi(a), the action mapped into the action space and written into the
 * environment's action buffer, and the transition stored at position t of the rollout:
 *   O[t] = obs, A[t] = a, LP[t] = log pi(a), V[t] = value, and -- of the step the environment completed BEFORE this
 *   call -- R[t - 1] = reward_scale * clip(reward), Dn[t - 1] = done.  A rollout row holds `ld` environments; environment e
 *   sits in column e - env_base of it (one [T][N] buffer shared by all sub-batches: ld = N, env_base = 0).
 * t lives on the device (ctr[0]; the host zeroes it before a rollout), is advanced by the launch itself, and a call
 * with t == T only stores R[T - 1] / Dn[T - 1] (the flush behind a rollout's last step); so the call can sit in a
 * captured graph.  params: auv_policy_param_floats(obs_dim) floats, 16-byte aligned -- the policy net then the value
 * net, each W1 [256][K0p] b1 [256] W2 [128][256] b2 [128] W3 [64][128] b3 [64] W4 [16][64] b4 [16], then log_std[2]
 * and two floats of padding.  A weight matrix is torch's Linear.weight [out][in], columns padded with zeros to a multiple
 * of 32 (K0p = obs_dim rounded up), the last layer's rows to 16, and stored in MFMA fragment order: element [n][k] at
 *   ((((n / 16) * (K / 32) + k / 32) * 2 + (k % 8) / 4) * 64 + ((k % 32) / 8) * 16 + n % 16) * 4 + k % 4
 * (gym_auv_amd/policy.py: FusedActorCritic.refresh does it with one permuted copy per layer).  All pointers are device
 * pointers.                                                                                                        */
typedef struct auv_policy_io {
  const float* obs;          /* [N][obs_dim]  the environment's observation buffer                          */
  const float* params;       /* packed weights (above)                                                       */
  int64_t* ctr;              /* [4] device: t, generator step counter (monotonic), scratch, -                */
  const float* reward_in;    /* [N]  the environment's reward buffer                                         */
  const uint8_t* done_in;    /* [N]  the environment's done buffer                                           */
  float* actions_out;        /* [N][2] the action buffer auv_step* reads (AUV_F32)                           */
  float* O;                  /* [T][ld][obs_dim] (nullable)                                                  */
  float* A;                  /* [T][ld][2]                                                                   */
  float* LP;                 /* [T][ld]                                                                      */
  float* V;                  /* [T][ld]                                                                      */
  float* R;                  /* [T][ld]                                                                      */
  float* Dn;                 /* [T][ld]                                                                      */
  float* mu_out;             /* [ne][2] (nullable: tests)                                                    */
  float* eps_out;            /* [ne][2] (nullable: tests)                                                    */
  uint64_t seed;
  int32_t obs_dim, T;
  int32_t ld, env_base;      /* environments per rollout row; the environment in column 0                   */
  float act_mid[2], act_half[2], clip_lo[2], clip_hi[2];   /* action = mid + half * clip(a, lo, hi)          */
  float reward_scale, reward_clip;                         /* reward_clip <= 0: no clipping                  */
  const void* params_bf16;   /* NULL: exact f32 (the default).  Else the eight weight matrices as bf16 -- policy net W1 W2
                                W3 W4, then the value net's, padded like the f32 ones, element [n][k] of a matrix at
                                (((n / 16) * (K / 32) + k / 32) * 64 + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8 -- and the
                                launch runs v_mfma_f32_16x16x32_bf16 (activations rounded to bf16 into the MFMA, f32
                                accumulation / bias / tanh): ~1e-2 on the means, NOT the reference's arithmetic; biases and
                                log_std still come from `params`.  16-byte aligned.                            */
} auv_policy_io_t;
size_t auv_policy_param_floats(int32_t obs_dim);
int auv_policy_act(auv_handle_t* h, int32_t e0, int32_t ne, const auv_policy_io_t* io, void* stream);
/* n_steps transitions of every slice with ONE call: per step and slice the policy launch and the environment's step of
 * that slice (actions = ios[i].actions_out), back to back on streams[i]; `flush` != 0: a final policy call per slice
 * stores the last step's reward / done.  t0 / gstep0 (host arrays, one entry per slice; both or neither): the rollout
 * position and the generator's step counter of the FIRST launch of each slice -- the host then names them for every launch
 * (position t0 + k, generator step gstep0 + k) and the launches skip their count-off on io.ctr (~4 us per launch; the
 * device copies are still kept in step); NULL: the launches use and advance io.ctr like auv_policy_act.  The chains are
 * not ordered against each other or the caller's stream.                                                           */
int auv_policy_rollout(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams,
                       const auv_policy_io_t* ios, float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t n_steps,
                       int32_t flush, const int64_t* t0, const int64_t* gstep0);

/* Generalised advantage estimation over a rollout of T steps x N environments (row-major [T][N] device buffers; V and
 * last_v in the same units as R): adv and ret = adv + V, one launch on `stream`.                                       */
int auv_gae(auv_handle_t* h, const float* R, const float* V, const float* Dn, const float* last_v, float gamma, float lam,
            float* adv_out, float* ret_out, int32_t T, int32_t N, void* stream);

int32_t auv_abi_version(void);
const char* auv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* AUV_HIP_H */
