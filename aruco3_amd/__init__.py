"""aruco3_amd -- MI355X-native ArUco/AprilTag detection behind the aruco3 crate's API.

Host-side mirror of the reference's public surface (src/lib.rs:6-9):
`Detector`, `DetectorConfig`, `Detection`, `Marker`, `ARDictionary`, `CameraIntrinsics`,
`MarkerPose` and the `pose` module.  All computation happens in hand-written HIP kernels
behind the C ABI declared in include/aruco3_hip.h (csrc/ -> libaruco3_hip.so).
"""
from .dictionaries import ARDictionary  # noqa: F401


def __getattr__(name):
    # GPU-backed names are imported lazily so that table handling works without the .so
    if name in ("Detector", "DetectorConfig", "Detection", "Marker"):
        from . import aruco

        return getattr(aruco, name)
    if name in ("CameraIntrinsics",):
        from . import pinhole

        return getattr(pinhole, name)
    if name in ("MarkerPose",):
        from . import pose

        return getattr(pose, name)
    if name == "pose":
        import importlib

        return importlib.import_module(".pose", __name__)
    raise AttributeError(name)
