"""Host-side mirror of `CameraIntrinsics` (src/pinhole.rs:11-60): a plain parameter record.
The one operation the pose path uses, `unproject` (src/pinhole.rs:88-93), runs inside the pose kernel."""
import math
from dataclasses import dataclass
from typing import Optional


@dataclass
class CameraIntrinsics:
    image_width: int
    image_height: int
    focal_x: float
    focal_y: float
    principal_x: Optional[float] = None
    principal_y: Optional[float] = None

    def __post_init__(self):  # src/pinhole.rs:26-35
        if self.principal_x is None:
            self.principal_x = self.image_width / 2.0
        if self.principal_y is None:
            self.principal_y = self.image_height / 2.0

    @classmethod
    def new(cls, image_width, image_height, focal_x, focal_y, principal_x=None, principal_y=None):
        return cls(image_width, image_height, focal_x, focal_y, principal_x, principal_y)

    @classmethod
    def new_from_fov_horizontal(cls, horizontal_fov_radians, sensor_width_mm, resolution_x, resolution_y):  # src/pinhole.rs:37-60
        aspect = resolution_x / resolution_y
        vfov = horizontal_fov_radians / aspect
        sensor_height_mm = sensor_width_mm / aspect
        fx = (sensor_width_mm * 0.5) / math.tan(horizontal_fov_radians * 0.5)
        fy = (sensor_height_mm * 0.5) / math.tan(vfov * 0.5)
        return cls(resolution_x, resolution_y, fx, fy, resolution_x * 0.5, resolution_y * 0.5)
