"""``supnerf_amd.install()`` -- switch the reference's own modules to the MI355X path with ZERO edited lines in the caller.

The reference has no plugin layer: its drivers bind the hot path by name, ``sys.path.insert(0, 'src')`` (optimize_nuscenes.py:1-3) and
then ``from utils import render_rays, render_rays_v2, render_rays_specified, render_full_img, ...`` (src/optimizer_nuscenes.py:14-18,
src/optimizer_kitti.py:13-16, src/optimizer_waymo.py:13-16), ``from utils import ... prepare_pixel_samples, volume_rendering_batch``
(src/trainer_unified_nuscenes.py:14-15, src/data_nuscenes.py:18-19), ``from model_supnerf import SUPNeRF`` (src/optimizer_nuscenes.py:21),
``from src.renderer import volume_rendering3, render_rays_v3`` (scripts/demo.py:16).  ``install()`` rebinds exactly those names:

  * in the caller's ``utils`` / ``src.utils``: the render functions and their building blocks -> ``supnerf_amd.utils``;
  * in ``renderer`` / ``src.renderer``: ``NeRFRenderer``, ``render_rays_v3``, ``volume_rendering3`` -> ``supnerf_amd.renderer``;
  * in ``model_supnerf`` / ``model_codenerf`` / ``model_autorf``: ``SUPNeRF`` / ``CodeNeRF`` / ``AutoRFMix`` -> the reference's OWN class
    with this package's HIP ``forward`` grafted on (``model.hip_decoder_class``): its constructor builds the stock ``ImgEncoder`` as
    src/model_supnerf.py:168-175 does, the state-dict is the reference's key for key (strict ``load_state_dict`` of
    src/optimizer_nuscenes.py:1796 passes), ``encode_img`` / ``pose_update`` stay stock PyTorch.

Modules that are already imported are patched in place, and every module that did ``from utils import render_rays_v2`` BEFORE the
install is re-pointed too (its global is the original function object: found by identity).  Modules that are imported LATER are patched
as they load, through one ``sys.meta_path`` hook -- so ``python -m supnerf_amd.run optimize_nuscenes.py ...`` (``run.py``) installs first
and then runs the unmodified script.  ``uninstall()`` restores everything.  Nothing of the reference is stored in this package: classes and
functions are looked up in the caller's modules at run time.
"""
import importlib
import importlib.abc
import sys
import threading
import types

from . import model as M
from . import renderer as R
from . import utils as U

# what is rebound, by the KIND of reference module (recognised by content, under whichever name the caller imported it)
UTILS_NAMES = ("render_rays", "render_rays_v2", "render_rays_specified", "render_full_img", "render_virtual_imgs", "prepare_pixel_samples",
               "volume_rendering_batch", "volume_rendering2", "volume_rendering", "get_rays", "get_rays_specified", "get_rays_srn",
               "sample_from_rays", "sample_from_rays_v2", "ray_box_intersection", "ray_box_intersection_tensor")
RENDERER_NAMES = ("NeRFRenderer", "render_rays_v3", "volume_rendering3")
DECODER_CLASSES = ("SUPNeRF", "CodeNeRF", "AutoRFMix")
# module names the reference's scripts use for the four files (top-level after sys.path.insert(0, 'src'), or through the src package)
TARGET_MODULES = tuple(p + n for n in ("utils", "renderer", "model_supnerf", "model_codenerf", "model_autorf") for p in ("", "src."))

_ORIG = "__supnerf_amd_originals__"
_LOCK = threading.RLock()
_state = {"hook": None, "patched": [], "rebound": []}


def _kind(mod):
    d = getattr(mod, "__dict__", {})
    if "render_rays_v2" in d and "volume_rendering2" in d and "get_rays" in d:
        return "utils"
    if "NeRFRenderer" in d and "volume_rendering3" in d:
        return "renderer"
    if any(isinstance(d.get(c), type) and hasattr(d[c], "forward") for c in DECODER_CLASSES):
        return "model"
    return None


def _is_ours(mod):
    return getattr(mod, "__name__", "").split(".")[0] == "supnerf_amd"


def patch_module(mod):
    """Rebind the hot-path names of ONE reference module (see the module docstring).  Returns {name: (original, replacement)}; a module
    that is not one of the reference's four files (or is this package's own) is left alone."""
    if mod is None or _is_ours(mod):
        return {}
    kind = _kind(mod)
    if kind is None:
        return {}
    with _LOCK:
        saved = mod.__dict__.setdefault(_ORIG, {})
        done = {}

        def rebind(name, new):
            old = mod.__dict__[name]
            if old is new:
                return
            saved.setdefault(name, old)
            setattr(mod, name, new)
            done[name] = (saved[name], new)

        if kind == "utils":
            for n in UTILS_NAMES:
                if n in mod.__dict__ and hasattr(U, n):
                    rebind(n, getattr(U, n))
        elif kind == "renderer":
            for n in RENDERER_NAMES:
                if n in mod.__dict__:
                    rebind(n, getattr(R, n))
        else:
            for n in DECODER_CLASSES:
                c = mod.__dict__.get(n)
                # only a class this module DEFINES: optimizer_*.py also hold ``SUPNeRF`` as an imported name (re-pointed by identity below)
                if isinstance(c, type) and c.__module__ == mod.__name__ and not issubclass(c, M._DecoderBase):
                    rebind(n, M.hip_decoder_class(c))
        if done and mod not in _state["patched"]:
            _state["patched"].append(mod)
        return done


def _rebind_importers(pairs):
    """``from utils import render_rays_v2`` executed before the install left the ORIGINAL object in the importer's globals: re-point every
    such global (found by identity, in every loaded module)."""
    by_id = {id(old): (old, new) for old, new in pairs}
    n = 0
    for mod in list(sys.modules.values()):
        d = getattr(mod, "__dict__", None)
        if not isinstance(d, dict) or _is_ours(mod):
            continue
        for name, val in list(d.items()):
            if name == _ORIG:
                continue
            hit = by_id.get(id(val))
            if hit is not None and hit[0] is val:
                d[name] = hit[1]
                _state["rebound"].append((mod, name, val))
                n += 1
    return n


class _PatchLoader(importlib.abc.Loader):
    """Runs the module's real loader, then ``patch_module`` (and re-points earlier importers of what it just replaced)."""

    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec) if hasattr(self._inner, "create_module") else None

    def exec_module(self, module):
        self._inner.exec_module(module)
        done = patch_module(module)
        if done:
            _rebind_importers(list(done.values()))

    def __getattr__(self, name):            # get_source, get_filename, is_package ... of the wrapped loader
        return getattr(self._inner, name)


class _PatchFinder(importlib.abc.MetaPathFinder):
    """First on ``sys.meta_path``: finds the real spec of a target module with the finders behind it and wraps its loader."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in TARGET_MODULES:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None:
                if spec.loader is not None and hasattr(spec.loader, "exec_module") and not isinstance(spec.loader, _PatchLoader):
                    spec.loader = _PatchLoader(spec.loader)
                return spec
        return None


def install(model_module=None, utils_module=None, renderer_module=None, *, extra_modules=(), import_now=False, hook=True):
    """Switch the reference's modules to the HIP path (module docstring).  The three module arguments name the caller's
    ``model_supnerf`` / ``utils`` / ``renderer`` modules explicitly (module objects or names); by default every one of the reference's
    module names that is already in ``sys.modules`` is patched, and -- ``hook=True`` -- every one imported later.  ``import_now=True``
    additionally imports ``utils``, ``renderer``, ``model_supnerf``, ``model_codenerf`` right away (they must be on ``sys.path``).
    Returns a report: {"patched": {module name: [names]}, "rebound": n, "hook": bool}.  Idempotent."""
    with _LOCK:
        mods = []
        for m in (model_module, utils_module, renderer_module, *extra_modules):
            if m is not None:
                mods.append(importlib.import_module(m) if isinstance(m, str) else m)
        if import_now:
            for name in ("utils", "renderer", "model_supnerf", "model_codenerf"):
                if name not in sys.modules and ("src." + name) not in sys.modules:
                    mods.append(importlib.import_module(name))
        mods += [sys.modules[n] for n in TARGET_MODULES if n in sys.modules]
        report, pairs = {}, []
        seen = set()
        for mod in mods:
            if id(mod) in seen or not isinstance(mod, types.ModuleType):
                continue
            seen.add(id(mod))
            done = patch_module(mod)
            if done:
                report[mod.__name__] = sorted(done)
                pairs += list(done.values())
        n = _rebind_importers(pairs) if pairs else 0
        if hook and _state["hook"] is None:
            _state["hook"] = _PatchFinder()
            sys.meta_path.insert(0, _state["hook"])
        return {"patched": report, "rebound": n, "hook": _state["hook"] is not None}


def uninstall():
    """Undo ``install``: restore every rebound name and remove the import hook."""
    with _LOCK:
        if _state["hook"] is not None and _state["hook"] in sys.meta_path:
            sys.meta_path.remove(_state["hook"])
        _state["hook"] = None
        for mod, name, old in reversed(_state["rebound"]):
            mod.__dict__[name] = old
        _state["rebound"].clear()
        for mod in _state["patched"]:
            for name, old in mod.__dict__.get(_ORIG, {}).items():
                setattr(mod, name, old)
            mod.__dict__.pop(_ORIG, None)
        _state["patched"].clear()


def installed():
    """{module name: [rebound names]} of what is currently switched over."""
    return {m.__name__: sorted(m.__dict__.get(_ORIG, {})) for m in _state["patched"]}
