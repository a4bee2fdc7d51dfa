"""rgbd_pl_slam_amd -- MI355X-native point+line feature front-end (ORB + LSD/LBD + Hamming matchers)
behind the call shapes of maxee1900/RGBD-PL-SLAM.  The compute path is libplf_hip.so (HIP, gfx950);
this package is the thin host-side mirror used by tests and bench.  No CPU fallback exists."""
from ._lib import PlfError, LIB_PATH  # noqa: F401
from .orb import ORBextractor  # noqa: F401
from .lines import LineSegment  # noqa: F401
from .matcher import Matcher, DescriptorDistance  # noqa: F401
from . import frame  # noqa: F401
