import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rgbd_pl_slam_amd import Matcher
x = torch.zeros(4, device="cuda")
m = Matcher(max_keypoints=2048, max_mappoints=16384, max_batch=2)
try:
    y = torch.ones(4).cuda(); torch.cuda.synchronize(); print("torch ok after matcher create")
except Exception as e:
    print("torch FAILED:", str(e)[:200])
