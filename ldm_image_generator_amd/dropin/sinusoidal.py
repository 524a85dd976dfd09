"""Flat-module shim: `import sinusoidal` (as the reference's scripts do) resolves to the MI355X-native classes."""
from ldm_image_generator_amd.sinusoidal import *  # noqa: F401,F403
from ldm_image_generator_amd import sinusoidal as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
