"""tiny_renderer_amd -- MI355X-native triangle-fill path of tiny_renderer behind a C ABI.

The product is the shared library built from csrc/ (HIP kernels for gfx950 + C++ host) whose
entry points are declared in include/tiny_renderer.h.  This package is the thin Python mirror
of the reference's `Scene` interface (src/scene.rs:44-269) over that library, used by the
tests, the bench and the headless CLI.  There is no CPU rendering path: creating a Scene
without the built library or without a GPU raises.
"""
from ._lib import build_library, library_path, load_library, TinyRendererError  # noqa: F401
from .scene import Scene, PIPELINES, prepare_uniforms, band_rows  # noqa: F401
from .assets import load_assets, load_obj, load_tga, save_tga, save_png  # noqa: F401
from .synthetic import synthetic_scene, instanced_grid  # noqa: F401
from .sharded import ShardedScene, PeerExchange, launch_ranks  # noqa: F401

__all__ = ["Scene", "PIPELINES", "prepare_uniforms", "band_rows", "load_assets", "load_obj", "load_tga", "save_tga", "save_png",
           "synthetic_scene", "instanced_grid", "build_library", "library_path", "load_library",
           "TinyRendererError"]
