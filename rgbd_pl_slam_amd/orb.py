"""Host-side mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:44-112) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  -- include/ORBextractor.h:51-52.
    `__call__(image)` is operator()(image, mask, keypoints, descriptors) (mask ignored, as in the reference)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, max_width=640,
                 max_height=480, max_batch=1, device=0):
        self._h = C.c_void_p()
        p = L.OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, max_width, max_height, max_batch)
        L.check(L.lib().plf_orb_create(C.byref(p), C.byref(self._h)), "plf_orb_create")
        self.nfeatures, self.nlevels, self.max_batch = nfeatures, nlevels, max_batch
        self.capacity = L.lib().plf_orb_capacity(self._h)
        n = C.c_int32()
        sc = np.zeros(nlevels, np.float32); inv = np.zeros(nlevels, np.float32); s2 = np.zeros(nlevels, np.float32)
        is2 = np.zeros(nlevels, np.float32); per = np.zeros(nlevels, np.int32)
        L.check(L.lib().plf_orb_get_tables(self._h, C.byref(n), L.vp(sc), L.vp(inv), L.vp(s2), L.vp(is2), L.vp(per)), "plf_orb_get_tables")
        self._tables = dict(scale=sc, inv=inv, sigma2=s2, invsigma2=is2, perLevel=per)

    def close(self):
        if self._h:
            L.lib().plf_orb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters of the reference class (include/ORBextractor.h:63-83)
    def GetLevels(self): return self.nlevels
    def GetScaleFactors(self): return self._tables["scale"]
    def GetInverseScaleFactors(self): return self._tables["inv"]
    def GetScaleSigmaSquares(self): return self._tables["sigma2"]
    def GetInverseScaleSigmaSquares(self): return self._tables["invsigma2"]
    def GetFeaturesPerLevel(self): return self._tables["perLevel"]

    def __call__(self, image, mask=None):
        image = np.asarray(image)
        if image.ndim != 2 or image.dtype != np.uint8:
            raise ValueError("8-bit single-channel image expected")  # the reference asserts CV_8UC1
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)                      # rows may be padded (pitch > width), columns may not
        h, w = image.shape
        kps = np.zeros(self.capacity, L.KP_DTYPE); desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int32(0)
        st = L.lib().plf_orb_extract(self._h, L.vp(image), w, h, C.c_ssize_t(image.strides[0]), L.vp(kps), L.vp(desc),
                                     self.capacity, C.byref(n))
        if st == L.PLF_E_EMPTY:
            return kps[:0], desc[:0]
        L.check(st, "plf_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: (B,H,W) uint8 numpy array (host) -> list of (kps, desc)"""
        images = np.ascontiguousarray(images, np.uint8)
        B, h, w = images.shape
        kps = np.zeros((B, self.capacity), L.KP_DTYPE); desc = np.zeros((B, self.capacity, 32), np.uint8)
        n = np.zeros(B, np.int32)
        L.check(L.lib().plf_orb_extract_batch(self._h, L.vp(images), L.MEM_HOST, B, w, h, C.c_ssize_t(w), C.c_ssize_t(w * h),
                                              L.vp(kps), L.vp(desc), L.vp(n), L.MEM_HOST, self.capacity, None), "plf_orb_extract_batch")
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(B)]

    def extract_batch_device(self, d_images, w, h, d_kps, d_desc, d_n, capacity, stream=None):
        """all pointers are device tensors (torch) or raw addresses; asynchronous"""
        B = int(d_images.shape[0])
        L.check(L.lib().plf_orb_extract_batch(self._h, L.vp(d_images), L.MEM_DEVICE, B, w, h, C.c_ssize_t(w), C.c_ssize_t(w * h),
                                              L.vp(d_kps), L.vp(d_desc), L.vp(d_n), L.MEM_DEVICE, capacity,
                                              C.c_void_p(stream) if stream else None), "plf_orb_extract_batch")

    # test hooks
    def pyramid_level(self, frame, level):
        lw, lh = C.c_int32(), C.c_int32()
        L.check(L.lib().plf_orb_get_pyramid_level(self._h, frame, level, None, C.byref(lw), C.byref(lh)), "get_pyramid_level")
        out = np.zeros((lh.value + 38, lw.value + 38), np.uint8)
        L.check(L.lib().plf_orb_get_pyramid_level(self._h, frame, level, L.vp(out), None, None), "get_pyramid_level")
        return out

    def blurred_level(self, frame, level):
        lw, lh = C.c_int32(), C.c_int32()
        L.check(L.lib().plf_orb_get_pyramid_level(self._h, frame, level, None, C.byref(lw), C.byref(lh)), "get_pyramid_level")
        out = np.zeros((lh.value, lw.value), np.uint8)
        L.check(L.lib().plf_orb_get_blurred_level(self._h, frame, level, L.vp(out)), "get_blurred_level")
        return out

    def candidates(self, frame, level):
        n = C.c_int32()
        L.check(L.lib().plf_orb_get_candidates(self._h, frame, level, None, 0, C.byref(n)), "get_candidates")
        out = np.zeros((max(n.value, 1), 3), np.float32)
        L.check(L.lib().plf_orb_get_candidates(self._h, frame, level, L.vp(out), n.value, C.byref(n)), "get_candidates")
        return out[:n.value]
