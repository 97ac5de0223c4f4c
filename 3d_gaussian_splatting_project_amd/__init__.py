"""MI355X-native Gaussian-splat majority-vote labeler + rasterizer (host side of libgsx.so).

The directory name starts with a digit, so import it with
    importlib.import_module("3d_gaussian_splatting_project_amd")
"""
from . import _lib
from ._lib import Camera, GsxError, build, lib
from . import dist, scene
from .labeler import Context, assign_labels_from_maps, bind_to_gpu_numa_node, camera_array, load_cameras, project_gaussian

__all__ = ["Camera", "Context", "GsxError", "assign_labels_from_maps", "bind_to_gpu_numa_node", "build", "camera_array", "dist", "lib", "load_cameras",
           "project_gaussian", "scene"]
