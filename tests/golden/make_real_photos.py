"""Real photographs as test / bench inputs (VERDICT r05 "configs on stand-in data": no TUM image exists in this image or on the GPU box).
Copies the CC0 / public-domain sample photographs that scikit-image 0.18 ships (found under /opt/conda of this container) into tests/golden/real/ and writes
MANIFEST.json: file sha256, size, licence line of skimage.data's own docstring, and the sha256 of the DECODED pixels as Pillow reads them -- the pin of the in-tree
PNG decoder (rgbd_pl_slam_amd/png.py, tests/test_real_photos.py).  Run once, here:   python tests/golden/make_real_photos.py
Not copied: the Middlebury stereo pair (no CC0 statement), anything that is not a photograph."""
import hashlib, json, os, shutil
import numpy as np
from PIL import Image

SRC = "/opt/conda/lib/python3.9/site-packages/skimage/data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real")
PHOTOS = {
    "camera.png": "skimage.data.camera: No copyright restrictions. CC0 by the photographer (Lav Varshney).",
    "astronaut.png": "skimage.data.astronaut: NASA photograph of Eileen Collins; no known copyright restrictions, released into the public domain.",
    "coffee.png": "skimage.data.coffee: No copyright restrictions. CC0 by the photographer (Rachel Michetti).",
    "chelsea.png": "skimage.data.chelsea: No copyright restrictions. CC0 by the photographer (Stefan van der Walt).",
    "brick.png": "skimage.data.brick: CC0Textures (Bricks25), Creative Commons CC0 License.",
    "grass.png": "skimage.data.grass: CC0Textures (Ground37), Creative Commons CC0 License.",
    "gravel.png": "skimage.data.gravel: CC0Textures (Gravel04), Creative Commons CC0 License.",
}
os.makedirs(DST, exist_ok=True)
man = {}
for name, lic in PHOTOS.items():
    shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, name))
    raw = open(os.path.join(DST, name), "rb").read()
    im = np.asarray(Image.open(os.path.join(DST, name)))
    man[name] = {"sha256": hashlib.sha256(raw).hexdigest(), "bytes": len(raw), "shape": list(im.shape), "dtype": str(im.dtype),
                 "pixels_sha256": hashlib.sha256(np.ascontiguousarray(im).tobytes()).hexdigest(), "licence": lic}
    print(name, im.shape, im.dtype)
json.dump(man, open(os.path.join(DST, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
