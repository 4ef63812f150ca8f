"""playableenvironments_amd - MI355X-native volumetric renderer for Playable Environments.

Public surface (mirrors the reference's modules for the renderer hot path only):
  ObjectComposer            drop-in for model/object_composer.py:ObjectComposer
  configs / synthetic       shipped renderer configurations and seeded synthetic scenes
"""
from . import configs, synthetic  # noqa: F401
from .object_composer import ObjectComposer, ObjectIDsHelper  # noqa: F401

__all__ = ["ObjectComposer", "ObjectIDsHelper", "configs", "synthetic"]
