"""Input side of the hot path: OpenSfM reconstruction parsing,
world normalisation and render trajectories.  Host-side numpy; see the module docstrings for the
reference lines each function follows."""
from .camera_models import intrinsics as camera_model_intrinsics, load_camera_models
from .normalize import (align_principle_axes, normalize, similarity_from_cameras, transform_cameras,
                        transform_points)
from .opensfm import Dataset, Parser, read_opensfm, read_opensfm_points3D
from .traj import (generate_ellipse_path_y, generate_ellipse_path_z, generate_interpolated_path,
                   generate_spiral_path, viewmatrix)

__all__ = ["Parser", "Dataset", "load_camera_models", "camera_model_intrinsics", "read_opensfm", "read_opensfm_points3D", "normalize",
           "similarity_from_cameras", "align_principle_axes", "transform_cameras", "transform_points",
           "viewmatrix", "generate_ellipse_path_z", "generate_ellipse_path_y", "generate_interpolated_path",
           "generate_spiral_path"]
