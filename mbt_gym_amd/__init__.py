"""mbt_gym_amd - an MI355X-native (gfx950, HIP) vectorised trading environment.

One hot path, behind the plugin API of JJJerome/mbt_gym: `TradingEnvironment.step()` as a single fused HIP kernel
(csrc/step_kernel.hpp) reached through the C ABI of include/mbt_env.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
