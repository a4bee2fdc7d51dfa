"""The synthetic matching workloads live in the package (rgbd_pl_slam_amd/matchgen.py) so that bench.py does not depend on the test tree;
the tests keep importing them under this name."""
from rgbd_pl_slam_amd.matchgen import *  # noqa: F401,F403
