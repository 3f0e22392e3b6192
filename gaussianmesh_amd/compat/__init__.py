"""Compatibility layer for scripts written against the reference's host framework.

    import gaussianmesh_amd.compat as compat
    compat.install()          # afterwards `import jittor as jt`, `from jittor import nn` resolve to the torch-backed subset
                              # in compat/jittor/, and the reference's operator modules resolve to this package:
                              #   gaussian_renderer.diff_gaussian_rasterizater -> gaussianmesh_amd.rasterizer
                              #   scene.simple_knn (distCUDA2)                 -> gaussianmesh_amd.simple_knn
                              #   utils.loss_utils l1_loss / ssim              -> gaussianmesh_amd.loss  (opt-in, see install)
    compat.install(edit_tool=True)   # additionally, for edit.py: `from edittool import ObjectVisualTool, SceneVisualTool`
                              # resolves to gaussianmesh_amd.edittool (the reference's package needs igl, plyfile and
                              # the pyACAP binary), `from render_origin import save_image` (a module missing from the
                              # reference tree) to gaussianmesh_amd.io.save_image, and - when libigl is not installed -
                              # `import igl` (edit.py:7) to compat/igl_subset.py: the four calls the reference makes

Only the ~60 symbols the in-scope reference python uses are provided; anything else raises
AttributeError naming the symbol.  A real Jittor installation is never shadowed unless install(force=True).
"""
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def install(force=False, operators=True, edit_tool=False):
    """Make `import jittor` resolve to compat/jittor (unless a real Jittor is importable and force is False) and, with
    operators=True, pre-register this package's operator modules under the reference's import paths; edit_tool=True also
    registers the file-based edit surface under the names edit.py imports."""
    have_real = False
    if "jittor" in sys.modules:
        have_real = not getattr(sys.modules["jittor"], "__gaussianmesh_compat__", False)
    elif not force:
        have_real = importlib.util.find_spec("jittor") is not None
    if not have_real or force:
        if _HERE not in sys.path:
            sys.path.insert(0, _HERE)
        for k in [k for k in sys.modules if k == "jittor" or k.startswith("jittor.")]:
            del sys.modules[k]
        importlib.import_module("jittor")
    if operators:
        from .. import rasterizer, simple_knn
        sys.modules.setdefault("gaussian_renderer.diff_gaussian_rasterizater", rasterizer)
        sys.modules.setdefault("scene.simple_knn", simple_knn)
    if edit_tool:
        import types
        from .. import edittool, io as gio
        # never swap a module somebody else already imported under these names (the reference's own edittool package, a real
        # render_origin) unless force=True - same rule as for the jittor subset and the operator aliases above
        if force or "edittool" not in sys.modules:
            sys.modules["edittool"] = edittool
        if force or "render_origin" not in sys.modules:
            ro = types.ModuleType("render_origin")
            ro.save_image = gio.save_image
            ro.__gaussianmesh_compat__ = True
            sys.modules["render_origin"] = ro
        have_igl = "igl" in sys.modules or importlib.util.find_spec("igl") is not None
        if force or not have_igl:
            from . import igl_subset
            sys.modules["igl"] = igl_subset
    return sys.modules["jittor"]


def uninstall():
    """Undo install(): restore torch.Tensor's own attributes (numpy() raises again for graph / device tensors) and drop the
    module aliases this package registered."""
    j = sys.modules.get("jittor")
    if j is not None and getattr(j, "__gaussianmesh_compat__", False):
        j._unpatch_tensor()
        for k in [k for k in sys.modules if k == "jittor" or k.startswith("jittor.")]:
            del sys.modules[k]
    for k in ("gaussian_renderer.diff_gaussian_rasterizater", "scene.simple_knn", "edittool", "render_origin", "igl"):
        m = sys.modules.get(k)
        if m is not None and (getattr(m, "__name__", "").startswith("gaussianmesh_amd") or getattr(m, "__gaussianmesh_compat__", False)):
            del sys.modules[k]            # only what install() put there
