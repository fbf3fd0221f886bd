"""Loader for the *reference* Python modules (only used to GENERATE golden vectors).

Runs only in the build container where /root/reference exists.  Nothing from the
reference is copied: modules are imported in place, by file path, to produce
input/output vectors that are then committed as .npz data.

Third-party packages that the reference imports but does not use on the
fbb path and that are absent from this image (wandb, skimage, torchvision) are
registered as EMPTY placeholder modules so the `import` lines succeed; no
functionality is faked.  Anything that would actually need them (torchvision's
VGG16 weights, ToPILImage) is therefore not pinned -- see DESIGN.md "parity
unpinned" notes.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("GANLEAKS_REFERENCE", "/root/reference")


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_placeholders():
    if "torchvision" not in sys.modules:
        tv_models = _placeholder("torchvision.models")
        tv_tf = _placeholder("torchvision.transforms")
        tv_utils = _placeholder("torchvision.utils")
        _placeholder("torchvision", models=tv_models, transforms=tv_tf, utils=tv_utils)
    if "wandb" not in sys.modules:
        _placeholder("wandb")
    if "skimage" not in sys.modules:
        col = _placeholder("skimage.color")
        tr = _placeholder("skimage.transform")
        _placeholder("skimage", color=col, transform=tr)


def load(relpath, modname):
    """Import /root/reference/<relpath> under the private name <modname>."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present at %s" % REF)
    install_placeholders()
    path = os.path.join(REF, relpath)
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
