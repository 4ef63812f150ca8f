"""Test-only shim that makes the ORIGINAL reference importable in the build container.

TEST INFRASTRUCTURE - never imported by the product package.  It only works where
``/root/reference`` exists (the build container); nothing under ``tests -m gpu``, ``bench.py``
or ``__graft_entry__.smoke`` may use it.  It is used by ``oracle/make_golden.py`` and
``oracle/check_against_reference.py`` to pin the oracle restatement against outputs of the
reference itself.

What it patches (all version drift between the reference's torch 1.8 / python 3.7 world and
this image, nothing algorithmic):
  * ``collections.Sequence`` alias (utils/lib_3d/ray_helper.py:217 etc.)
  * ``numpy.bool`` alias (model/object_composer.py:350)
  * ``Tensor.cuda`` / ``Module.cuda`` become identity, ``Tensor.get_device`` returns the device
    (hard coded ``.cuda()`` in utils/lib_3d/ray_helper.py:30,36-37,1248-1253,1275,1371,1380)
  * inert stub modules for optional third-party imports that are absent here.
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import os
import sys

sys.dont_write_bytecode = True     # importing the reference must not leave __pycache__ files in its (read-only) tree
import types

import numpy as np
import torch
import yaml

REFERENCE_ROOT = "/root/reference"

_STUBBED = ("torchvision", "wandb", "cv2", "pyrender", "trimesh", "lpips", "kornia", "imageio",
            "skimage", "matplotlib")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _StubModule(self.__name__ + "." + name)
        sub.__path__ = []
        setattr(self, name, sub)
        return sub

    def __call__(self, *args, **kwargs):
        return _StubModule(self.__name__ + "()")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in _STUBBED:
            try:
                # Prefer the real module when the image has it
                if root in ("matplotlib",):
                    return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Idempotently installs the shim and puts the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present: the shim only works in the build container")
    sys.path.insert(0, REFERENCE_ROOT)
    collections.Sequence = collections.abc.Sequence
    if not hasattr(np, "bool"):
        np.bool = bool
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device
    sys.meta_path.insert(0, _StubFinder())
    _installed = True


def load_reference_config(name: str) -> dict:
    """Loads a shipped YAML of the reference and applies the defaults of
    utils/configuration.py:30-242 that the renderer path reads."""
    install()
    from utils.dict_wrapper import DictWrapper
    cfg_dir = os.path.join(REFERENCE_ROOT, "configs", name)
    yamls = [f for f in os.listdir(cfg_dir) if f.endswith(".yaml")]
    assert len(yamls) == 1, yamls
    with open(os.path.join(cfg_dir, yamls[0])) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    config = DictWrapper(config)
    model = config["model"]
    model.setdefault("apply_activation", True)
    model.setdefault("fix_object_overlaps", True)
    model.setdefault("enable_camera_parameters_offsets", False)
    model.setdefault("camera_parameters_memory_size", 1)
    return config


def build_reference_composer(config):
    install()
    from model.object_composer import ObjectComposer
    return ObjectComposer(config)
