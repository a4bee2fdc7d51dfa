"""Host-side mirror of ORB_SLAM2::LineSegment (include/ExtractLineSegment.h:29-57) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


class LineSegment:
    """ExtractLineSegment(img, keylines, ldesc, lineFunctions, scale=1.2, numOctaves=1) -- include/ExtractLineSegment.h:38.
    `nlines` is the number of lines kept after the response sort (compile-time constant in the fork)."""

    def __init__(self, nlines=100, max_width=640, max_height=480, max_batch=1, device=0, seed_order=0, lbd_sobel_input=L.LBD_BLURRED, max_ms=0.0):
        """max_ms > 0: time budget of LSD region growing per call (plf_line_params.max_ms; an opt-in deviation from the reference, see include/plf.h)"""
        self._h = C.c_void_p()
        p = L.line_params(nlines, seed_order, device, max_width, max_height, max_batch, lbd_sobel_input, max_ms)
        L.check(L.lib().plf_line_create(C.byref(p), C.byref(self._h)), "plf_line_create")
        self.nlines, self.max_batch = nlines, max_batch

    def close(self):
        if self._h:
            L.lib().plf_line_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ExtractLineSegment(self, img):
        """returns (keylines[KL_DTYPE], ldesc[n,32] uint8, lineFunctions[n,3] float64)"""
        img = np.asarray(img)
        if img.ndim != 2 or img.dtype != np.uint8:
            raise ValueError("8-bit single-channel image expected")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        h, w = img.shape
        kl = np.zeros(self.nlines, L.KL_DTYPE); desc = np.zeros((self.nlines, 32), np.uint8); eq = np.zeros((self.nlines, 3), np.float64)
        n = C.c_int32(0)
        st = L.lib().plf_line_extract(self._h, L.vp(img), w, h, C.c_ssize_t(img.strides[0]), L.vp(kl), L.vp(desc), L.vp(eq),
                                      self.nlines, C.byref(n))
        if st == L.PLF_E_EMPTY:
            return kl[:0], desc[:0], eq[:0]
        L.check(st, "plf_line_extract")
        return kl[:n.value].copy(), desc[:n.value].copy(), eq[:n.value].copy()

    def LineSegmentMathch(self, ldesc1, ldesc2):
        """void LineSegmentMathch(Mat &ldesc1, Mat &ldesc2) -- include/ExtractLineSegment.h:41: BFMatcher(NORM_HAMMING).knnMatch(k = 2) of two host
        descriptor matrices into mvlineMatches; the spreads LineDescriptorMAD() reports come from the same device pass"""
        import torch
        from .matcher import Matcher
        d1 = torch.from_numpy(np.ascontiguousarray(ldesc1, np.uint8)).cuda(); d2 = torch.from_numpy(np.ascontiguousarray(ldesc2, np.uint8)).cuda()
        m = Matcher(max_keypoints=64, max_mappoints=64, max_lines=max(int(d1.shape[0]), int(d2.shape[0]), 2), device=d1.device.index or 0)
        try:
            self.mvlineMatches, self.mnnMad, self.mnn12Mad = m.LineDescriptorMAD(d1, d2)
        finally:
            m.close()
        return self.mvlineMatches

    def LineDescriptorMAD(self):
        """void LineDescriptorMAD() -- include/ExtractLineSegment.h:44: (mnnMad, mnn12Mad) of the last LineSegmentMathch"""
        return self.mnnMad, self.mnn12Mad

    @staticmethod
    def LineSegmentOverlap(spl_obs, epl_obs, spl_proj, epl_proj):
        """double LineSegmentOverlap(spl_obs, epl_obs, spl_proj, epl_proj) -- include/ExtractLineSegment.h:47 (host scalar, see plf.h)"""
        f = L.lib().plf_line_segment_overlap
        f.restype = C.c_double; f.argtypes = [C.c_double] * 4
        return f(spl_obs, epl_obs, spl_proj, epl_proj)

    def extract_batch(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        B, h, w = images.shape
        kl = np.zeros((B, self.nlines), L.KL_DTYPE); desc = np.zeros((B, self.nlines, 32), np.uint8)
        eq = np.zeros((B, self.nlines, 3), np.float64); n = np.zeros(B, np.int32)
        L.check(L.lib().plf_line_extract_batch(self._h, L.vp(images), L.MEM_HOST, B, w, h, C.c_ssize_t(w), C.c_ssize_t(w * h), L.vp(kl),
                                               L.vp(desc), L.vp(eq), L.vp(n), L.MEM_HOST, self.nlines, None), "plf_line_extract_batch")
        return [(kl[f, :n[f]].copy(), desc[f, :n[f]].copy(), eq[f, :n[f]].copy()) for f in range(B)]

    def extract_batch_device(self, d_images, w, h, d_lines, d_desc, d_eq, d_n, capacity, stream=None):
        B = int(d_images.shape[0])
        L.check(L.lib().plf_line_extract_batch(self._h, L.vp(d_images), L.MEM_DEVICE, B, w, h, C.c_ssize_t(w), C.c_ssize_t(w * h),
                                               L.vp(d_lines), L.vp(d_desc), L.vp(d_eq), L.vp(d_n), L.MEM_DEVICE, capacity,
                                               C.c_void_p(stream) if stream else None), "plf_line_extract_batch")

    def tune(self, name, value):
        """change one schedule knob of this handle (plf_line_tune: a tuning / test hook -- every schedule gives the same bits)"""
        f = L.lib().plf_line_tune
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.check(f(self._h, name.encode(), float(value)), "plf_line_tune")

    def last_status(self, stream=None):
        """status of the last extract_batch_device call (waits for the stream): 0, PLF_E_CAPACITY or PLF_E_RECTS"""
        return int(L.lib().plf_line_last_status(self._h, C.c_void_p(stream) if stream else None))

    def truncated(self, n=1):
        """flags of the first n frames of the last batch: 1 = the frame ran out of max_ms and holds only the segments found until then"""
        out = np.zeros(n, np.int32)
        L.check(L.lib().plf_line_truncated(self._h, L.vp(out), n), "plf_line_truncated")
        return out

    def wait_front(self, stream):
        """make `stream` wait until the stages before region growing of the last enqueued batch are done"""
        L.check(L.lib().plf_line_wait_front(self._h, C.c_void_p(stream) if stream else None), "plf_line_wait_front")

    def profile(self, enable=True, reset=False):
        """(accumulated ms, launches) of the region-growing kernel measured with HIP events on its stream"""
        ms = C.c_double(0); n = C.c_int32(0)
        L.check(L.lib().plf_line_profile(self._h, int(enable), int(reset), C.byref(ms), C.byref(n)), "plf_line_profile")
        return ms.value, n.value

    def chain_lengths(self, n):
        """accept steps of the region-growing chain of the first n frames of the last batch (plf_line_chain_lengths)"""
        out = np.zeros(n, np.int32)
        L.check(L.lib().plf_line_chain_lengths(self._h, L.vp(out), n), "plf_line_chain_lengths")
        return out

    def rect_counts(self, n):
        """rectangles the first n frames of the last batch handed to the NFA validation (plf_line_rect_counts)"""
        out = np.zeros(n, np.int32)
        L.check(L.lib().plf_line_rect_counts(self._h, L.vp(out), n), "plf_line_rect_counts")
        return out

    def spec_rounds(self, n):
        """per frame of the last few-frames batch: (bands changed in the last even round, in the last odd round, round of the fixpoint, finished by the serial commit wave)"""
        out = np.zeros(4 * n, np.int32)
        if L.lib().plf_line_debug_spec_rounds(self._h, L.vp(out), n) != 0:
            return None
        return out.reshape(n, 4)

    def nfa_counters(self):
        """rectangles of the last batch that entered each rect_improve stage (plf_line_debug_nfa_counters)"""
        out = np.zeros(16, np.int32)
        L.check(L.lib().plf_line_debug_nfa_counters(self._h, L.vp(out)), "plf_line_debug_nfa_counters")
        return out

    def segments(self, frame=0):
        """test hook: all LSD segments of the last call in detection order"""
        n = C.c_int32()
        L.check(L.lib().plf_line_get_segments(self._h, frame, None, 0, C.byref(n)), "plf_line_get_segments")
        out = np.zeros((max(n.value, 1), 4), np.float32)
        L.check(L.lib().plf_line_get_segments(self._h, frame, L.vp(out), n.value, C.byref(n)), "plf_line_get_segments")
        return out[:n.value]
