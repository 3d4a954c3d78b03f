// The fused lane-per-problem kernels in float32 arithmetic: the same source as mk_fused.hip with
// the scalar type switched (mk_device.h, `real`).  Arrays in global memory stay float64; the
// launchers are mk_fused32_launch_step / _iterate / _move_to.
#define MK_F32 1
#include "mk_fused.hip"
