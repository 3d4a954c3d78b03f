"""The IK part of the reference's interactive_scripts/teleop_mouse.py (lines 29-121) on
mink_amd: one TidyBot, a mocap body as the end-effector target, the <= 20-step IK loop per
control tick.  No viewer and no input device: the target follows a small circle.

    python examples/teleop_loop.py
"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mink_amd as mink  # noqa: E402

model = mink.load_builtin("tidybot_cube")          # interactive_scripts/stanford_tidybot/cube.xml
configuration = mink.Configuration(model)
configuration.update_from_keyframe("home")

end_effector_task = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
posture_cost = np.zeros(model.nv)
posture_cost[3:] = 1e-3
posture_task = mink.PostureTask(model, cost=posture_cost)
posture_task.set_target_from_configuration(configuration)
tasks = [end_effector_task, posture_task]

# the mocap body starts on the end effector (mink.move_mocap_to_frame, teleop_mouse.py:88)
mink.move_mocap_to_frame(model, configuration.data, "pinch_site_target", "pinch_site", "site")
centre = configuration.data.mocap_pos[0].copy()

dt, max_iters, pos_threshold, ori_threshold = 1.0 / 200.0, 20, 1e-4, 1e-4
ticks, solves = 200, 0
for k in range(-1, ticks):                    # tick -1 builds and caches the device problem
    if k == 0:
        t0 = time.perf_counter()
    a = 2.0 * math.pi * k / ticks
    configuration.data.mocap_pos[0] = centre + 0.05 * np.array([math.cos(a) - 1.0, math.sin(a), 0.0])
    end_effector_task.set_target(mink.SE3.from_mocap_name(model, configuration.data, "pinch_site_target"))
    # the reference's inner loop (teleop_mouse.py:105-121), as one launch
    res = mink.solve_ik_iterate(configuration, tasks, dt, "quadprog", 1e-3, max_iters=max_iters,
                                pos_threshold=pos_threshold, ori_threshold=ori_threshold)
    solves += res["iters"] if k >= 0 else 0
elapsed = time.perf_counter() - t0
err = end_effector_task.compute_error(configuration)
print(f"{ticks} control ticks, {solves} IK solves, {1e6 * elapsed / ticks:.0f} us per tick; "
      f"final error |pos| = {np.linalg.norm(err[:3]):.2e} m, |rot| = {np.linalg.norm(err[3:]):.2e} rad")
