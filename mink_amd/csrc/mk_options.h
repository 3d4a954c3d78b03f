// Process-wide switches of the library (mk_set_option / mk_get_option in include/mink_amd.h):
// which kernel form serves a batch, and the diagnostic switches of the tests and tuning
// tools.  They are plain atomics, read with relaxed loads at launch time: changing one while
// another thread launches affects only which (equivalent) path that launch takes.
#pragma once

enum MkOption {
  MK_OPT_STAGED = 0,       // 1: problems with collision limits take the staged solve (read at mk_problem_create)
  MK_OPT_LANE_ROWS,        // 1: the staged solve runs its lane-per-problem rows stage
  MK_OPT_GENERAL_STAGE,    // 1: the staged solve ends with the general kernel; 0 (diagnostic): problems it would solve report status 1
  MK_OPT_STAGED_LOOP,      // 1: the iterated entry runs staged problems as passes of the staged kernels
  MK_OPT_STREAM_BLOCKS,    // -1: lane streaming by batch size; 0: never; n > 0: always, with n persistent waves
  MK_OPT_STREAM_REFILL,    // lanes that must wait before the streaming kernels hand out new problems
  MK_OPT_COOP,             // -1: the cooperative (16 lanes per problem) kernels by batch size; 0: never; 1: always
  MK_OPT_COOP_MAX_BATCH,   // largest batch the cooperative kernels take when MK_OPT_COOP is -1
  MK_OPT_HANDOFF,          // loops with an exit test on the block-per-64 kernel: -1 a wave hands its last 4 live problems to the cooperative kernel; 0 never; n > 0: its last n
  MK_OPT_F32,              // 1: the step, loop and rollout entries run the fused kernels in float32 arithmetic (problems without such kernels are refused)
  MK_OPT_COUNT
};

long mk_opt(MkOption o);
