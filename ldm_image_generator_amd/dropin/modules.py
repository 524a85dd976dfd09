"""Flat-module shim: `import modules` (as the reference's scripts do) resolves to the MI355X-native classes."""
from ldm_image_generator_amd.modules import *  # noqa: F401,F403
from ldm_image_generator_amd import modules as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
