"""MI355X-native SPH particle loop: drop-in for kaegi/adaptive-sph's
``FluidSimulation::single_step_without_adaptivity`` (src/simulation/simulation.rs:1980-2730).

  csrc/                      hand-written HIP kernels for gfx950 + the C ABI (include/sph_ffi.h)
  ffi.py                     ctypes binding of the C ABI
  simulation_parameters.py   SimulationParams mirror (YAML + override rule)
  scene.py                   SceneConfig / add_fluid_block mirror, BASELINE workloads
  simulation.py              FluidSimulation mirror (init_fluid_sim, single_step_without_adaptivity, ...)
  distributed.py             one-process-per-GPU slab launcher glue (RCCL id exchange)
"""
from .ffi import Context, SphError, SphLibrary, load_product  # noqa: F401
from .scene import SceneConfig  # noqa: F401
from .simulation import FluidSimulation, init_fluid_sim, init_simulation_params  # noqa: F401
from .simulation_parameters import SimulationParams  # noqa: F401
