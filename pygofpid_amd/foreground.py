"""Steps 1-3 of GOFPID.detect as one GPU object (gof_pipeline in the C ABI).

`ForegroundPipeline` is what `GOFPID.detect` uses when its blur / detector /
morphology configuration is made of this package's own callables, and what
bench.py drives with device-resident frames.  `streams` independent videos of
one geometry are stepped together in each launch (one background model each).
"""
import ctypes as C

import numpy as np

from . import _lib


class ForegroundPipeline:
    """blur -> MOG2 | FD -> erode/dilate chain  [REF pipeline.py:461-476].

    Parameters
    ----------
    shape : (h, w) | (h, w, c)
    streams : number of independent streams stepped per call
    blur : None | (kw, kh, sigma_x[, sigma_y])   -- Gaussian, REFLECT_101
           | ('box', kw, kh)                     -- cv.blur, REFLECT_101
    detector : 'MOG2' | 'FD' | 'KNN' (cv2 defaults)
    morph : list of (op, kernel) with op in {'erode', 'dilate'} and kernel a
        (kh, kw) array (non-zero = member)
    convert : None | 'BGR2GRAY' | 'RGB2GRAY' -- step 0 on the device [REF pipeline.py:458-459]:
        frames arrive with shape's 3 or 4 channels, the detector works on their grey image
    """

    def __init__(self, shape, streams=1, blur=(3, 3, 4.0), detector="MOG2", fd_threshold=5,
                 morph=(("erode", np.ones((5, 5), np.uint8)), ("dilate", np.ones((5, 5), np.uint8))),
                 mog2_params=None, device=None, ctx=None, convert=None):
        if len(shape) == 2:
            h, w, c = shape[0], shape[1], 1
        else:
            h, w, c = shape
        # self.c: channels of the frames handed in; the detector sees 1 when converting
        self.h, self.w, self.c, self.streams = int(h), int(w), int(c), int(streams)
        if convert not in (None, 'BGR2GRAY', 'RGB2GRAY'):
            raise ValueError("convert must be None, 'BGR2GRAY' or 'RGB2GRAY'")
        if convert and self.c not in (3, 4):
            raise ValueError("conversion to grey takes 3- or 4-channel frames")
        lib = _lib.load()
        d = _lib.PipelineDesc()
        d.streams, d.h, d.w, d.c = self.streams, self.h, self.w, (1 if convert else self.c)
        if blur is None:
            d.blur = _lib.BLUR_NONE
        elif blur[0] == 'box':
            d.blur = _lib.BLUR_BOX
            d.blur_kw, d.blur_kh = int(blur[1]), int(blur[2])
        else:
            d.blur = _lib.BLUR_GAUSS
            d.blur_kw, d.blur_kh = int(blur[0]), int(blur[1])
            d.blur_sigma_x = float(blur[2])
            d.blur_sigma_y = float(blur[3]) if len(blur) > 3 else 0.0
        if detector == "MOG2":
            d.detector = _lib.DET_MOG2
        elif detector == "FD":
            d.detector = _lib.DET_FD
        elif detector == "KNN":
            d.detector = _lib.DET_KNN
        else:
            raise ValueError("Unknown method for foreground detection")
        if mog2_params is None:
            lib.gof_mog2_default_params(C.byref(d.mog2))
        else:
            d.mog2 = mog2_params
        self.nmixtures = int(d.mog2.nmixtures)
        d.fd_threshold = float(fd_threshold)
        morph = list(morph or [])
        if len(morph) > _lib.MAX_MORPH:
            raise ValueError("at most %d morphology operators" % _lib.MAX_MORPH)
        d.n_morph = len(morph)
        keep = []
        for i, (op, kernel) in enumerate(morph):
            k = np.ascontiguousarray(np.asarray(kernel) != 0, dtype=np.uint8)
            if k.ndim != 2:
                raise ValueError("kernel must be 2-D")
            keep.append(k)
            d.morph[i].op = _lib.ERODE if op == "erode" else _lib.DILATE
            if op not in ("erode", "dilate"):
                raise ValueError("unknown morphology operator %r" % (op,))
            d.morph[i].kh, d.morph[i].kw = k.shape
            d.morph[i].kernel = _lib.p8(k)
        self._ctx = ctx if ctx is not None else _lib.default_context(device)
        self._h = C.c_void_p()
        _lib.check(lib.gof_pipeline_create(self._ctx.handle, C.byref(d), C.byref(self._h)))
        del keep  # the library copied the structuring elements
        self.model_c = 1 if convert else self.c     # channels the detector (and its model) work on
        self.detector = detector
        if convert:
            _lib.check(lib.gof_pipeline_set_convert(self._h, self.c, int(convert == 'RGB2GRAY')))

    # -- host buffers (what GOFPID.detect hands over) ------------------------
    def apply(self, frames):
        """frames: (h,w[,c]) for streams == 1, else (streams,h,w[,c]); -> uint8 mask(s)."""
        x = _lib.as_u8(frames, "frames")
        single = self.streams == 1 and x.ndim <= 3 and x.shape[:2] == (self.h, self.w)
        want = (self.streams, self.h, self.w, self.c)
        if x.size != int(np.prod(want)):
            raise ValueError("Input shape has changed.")
        out = np.empty((self.streams, self.h, self.w), np.uint8)
        _lib.check(_lib.load().gof_pipeline_apply_host(self._h, _lib.p8(x), _lib.p8(out)))
        return out[0] if single else out

    def apply_many(self, frames):
        """T consecutive frames of every stream in one call (temporal blocking: the
        MOG2 model crosses HBM once per 8 frames).  frames: (T, streams, h, w[, c])
        or (T, h, w[, c]) for one stream; returns the T x streams masks, identical
        to T calls of apply()."""
        x = _lib.as_u8(frames, "frames")
        per = self.streams * self.h * self.w * self.c
        if x.ndim < 3 or x.size % per or x.size // per != x.shape[0]:
            raise ValueError("Input shape has changed.")
        T = x.shape[0]
        out = np.empty((T, self.streams, self.h, self.w), np.uint8)
        _lib.check(_lib.load().gof_pipeline_apply_host_many(self._h, _lib.p8(x), _lib.p8(out), T))
        return out[:, 0] if self.streams == 1 else out

    # -- steps 1-3 + step 4's front end, masks stay on the device ---------------
    def apply_blobs(self, frames, min_area=100.0):
        """Steps 1-3 on `frames` (as apply() / apply_many(): (h,w[,c]), (streams,h,w[,c])
        or (T,streams,h,w[,c])) followed by the GPU blob extraction on the resulting
        masks [REF helpers.py:125-168, pipeline.py:555-572].  Only blob records and the
        contour points of blobs with cv.contourArea >= min_area are downloaded; the
        masks stay on the device until last_masks() asks for them.

        Returns one (records, points) pair per mask, index = frame * streams + stream
        (see blobs.BlobExtractor.result); an entry is None where the device scratch
        overflowed -- the caller then uses the mask itself."""
        from .blobs import BlobExtractor
        x = _lib.as_u8(frames, "frames")
        per = self.streams * self.h * self.w * self.c
        if x.size % per:
            raise ValueError("Input shape has changed.")
        T = x.size // per
        if getattr(self, "_blobs", None) is None or self._blobs.n_images < T * self.streams:
            self._blobs = BlobExtractor(T * self.streams, self.h, self.w, ctx=self._ctx)
        _lib.check(_lib.load().gof_pipeline_apply_host_blobs(self._h, _lib.p8(x), T, self._blobs.handle,
                                                             float(min_area)))
        self._last_T = T
        return [self._blobs.result(i) for i in range(T * self.streams)]

    def apply_blobs_frames(self, frames, min_area=100.0):
        """apply_blobs() for a LIST of frames (each what apply() takes), uploaded straight from
        their own buffers -- no gathering copy on the host.  Returns the same list of results."""
        from .blobs import BlobExtractor
        arrs = [_lib.as_u8(f, "frames") for f in frames]
        per = self.streams * self.h * self.w * self.c
        if not arrs or any(a.size != per for a in arrs):
            raise ValueError("Input shape has changed.")
        T = len(arrs)
        if getattr(self, "_blobs", None) is None or self._blobs.n_images < T * self.streams:
            self._blobs = BlobExtractor(T * self.streams, self.h, self.w, ctx=self._ctx)
        ptrs = (C.c_void_p * T)(*[a.ctypes.data for a in arrs])
        _lib.check(_lib.load().gof_pipeline_apply_host_blobs_ptrs(self._h, ptrs, T, self._blobs.handle,
                                                                  float(min_area)))
        self._last_T = T
        return [self._blobs.result(i) for i in range(T * self.streams)]

    # -- pipelined batches (GOFPID.detect_many) ------------------------------
    def submit_blobs_frames(self, frames, min_area=100.0, slot=0):
        """apply_blobs_frames() without the wait: the frames go up (this call returns when they have
        left their buffers) beside the kernels of the batch submitted into the other slot before;
        wait_blobs(slot) returns the results.  Call it from one thread at a time, alternating slots."""
        from .blobs import BlobExtractor
        arrs = [_lib.as_u8(f, "frames") for f in frames]
        per = self.streams * self.h * self.w * self.c
        if not arrs or any(a.size != per for a in arrs):
            raise ValueError("Input shape has changed.")
        T = len(arrs)
        if not hasattr(self, "_slot_blobs"):
            self._slot_blobs, self._slot_T = [None, None], [0, 0]
        ex = self._slot_blobs[slot]
        if ex is None or ex.n_images < T * self.streams:
            ex = self._slot_blobs[slot] = BlobExtractor(T * self.streams, self.h, self.w, ctx=self._ctx)
        ptrs = (C.c_void_p * T)(*[a.ctypes.data for a in arrs])
        _lib.check(_lib.load().gof_pipeline_submit_host_blobs_ptrs(self._h, ptrs, T, ex.handle,
                                                                   float(min_area), int(slot)))
        self._slot_T[slot] = T
        return T

    def wait_blobs(self, slot=0):
        """Results of the batch submitted into `slot` (one (records, points) pair per mask, None
        where the device scratch overflowed)."""
        T = self._slot_T[slot]
        _lib.check(_lib.load().gof_pipeline_wait_slot(self._h, int(slot)))
        return [self._slot_blobs[slot].result(i) for i in range(T * self.streams)]

    def slot_masks(self, slot=0):
        """The masks of the batch that went through `slot` last, downloaded now: (T, streams, h, w)."""
        T = self._slot_T[slot]
        out = np.empty((T, self.streams, self.h, self.w), np.uint8)
        _lib.check(_lib.load().gof_pipeline_fetch_masks_slot(self._h, _lib.p8(out), T, int(slot)))
        return out

    def last_masks(self):
        """The masks of the last apply_blobs() call, downloaded now: (T, streams, h, w)."""
        T = getattr(self, "_last_T", 0)
        if not T:
            raise RuntimeError("no apply_blobs() call yet")
        out = np.empty((T, self.streams, self.h, self.w), np.uint8)
        _lib.check(_lib.load().gof_pipeline_fetch_masks(self._h, _lib.p8(out), T))
        return out

    # -- device buffers (bench / serving loop) -------------------------------
    def apply_device(self, frames_ptr, masks_ptr, stream=None, n_frames=1):
        """Enqueue one step of n_frames frames per stream on device pointers (ints);
        does not synchronise."""
        _lib.check(_lib.load().gof_pipeline_apply_dev_many(
            self._h, C.c_void_p(frames_ptr), C.c_void_p(masks_ptr), int(n_frames),
            C.c_void_p(stream) if stream else None))

    @property
    def stream(self):
        return self._ctx.stream

    @property
    def device(self):
        """HIP device index of the pipeline's context."""
        return int(self._ctx.device)

    def synchronize(self):
        self._ctx.synchronize()

    def mean_modes(self):
        v = C.c_double()
        _lib.check(_lib.load().gof_pipeline_mean_modes(self._h, C.byref(v)))
        return v.value

    def algorithmic_bytes(self, mean_modes, n_frames=1):
        """(total, detector-kernel) algorithmic HBM bytes of one step of n_frames
        frames per stream."""
        t, k = C.c_double(), C.c_double()
        _lib.check(_lib.load().gof_pipeline_algorithmic_bytes(self._h, float(mean_modes),
                                                              int(n_frames), C.byref(t),
                                                              C.byref(k)))
        return t.value, k.value

    def profile(self, enable, capacity=0, stride=1):
        """Bracket each step's kernels (every stride-th step's) with HIP events on the
        launch stream."""
        _lib.check(_lib.load().gof_pipeline_profile_stride(self._h, int(stride)))
        _lib.check(_lib.load().gof_pipeline_profile_enable(self._h, int(bool(enable)),
                                                           int(capacity)))

    def profile_query(self):
        """-> (sum detector-kernel ms, sum morphology-kernel ms, steps measured)"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        _lib.check(_lib.load().gof_pipeline_profile_query(self._h, C.byref(a), C.byref(b),
                                                          C.byref(n)))
        return a.value, b.value, n.value

    def profile_span(self):
        """ms from the first bracketed step's first event to the last one's last event"""
        v = C.c_double()
        _lib.check(_lib.load().gof_pipeline_profile_span(self._h, C.byref(v)))
        return v.value

    def profile_read(self, capacity):
        """-> (detector-kernel ms, morphology-kernel ms) arrays, one entry per measured step"""
        a, b, n = np.zeros(capacity, np.float32), np.zeros(capacity, np.float32), C.c_int()
        f32 = C.POINTER(C.c_float)
        _lib.check(_lib.load().gof_pipeline_profile_read(self._h, a.ctypes.data_as(f32),
                                                         b.ctypes.data_as(f32), int(capacity),
                                                         C.byref(n)))
        k = min(n.value, capacity)
        return a[:k], b[:k]

    def knn_state(self):
        """Export the KNN model (host copies): see cv.BackgroundSubtractorKNN.export_state."""
        from . import cv
        k = _lib.load().gof_pipeline_knn(self._h)
        if not k:
            raise RuntimeError("not a KNN pipeline")
        return cv.knn_export(C.c_void_p(k), self.streams * self.h * self.w, self.model_c)

    def mog2_state(self):
        """Export the model planes (host copies) -- parity tests / checkpoint."""
        m = _lib.load().gof_pipeline_mog2(self._h)
        if not m:
            raise RuntimeError("not a MOG2 pipeline")
        n, K, c = self.streams * self.h * self.w, self.nmixtures, self.model_c
        wt, var = np.empty((K, n), np.float32), np.empty((K, n), np.float32)
        mean, modes = np.empty((K, c, n), np.float32), np.empty(n, np.uint8)
        f32 = C.POINTER(C.c_float)
        _lib.check(_lib.load().gof_mog2_export(m, wt.ctypes.data_as(f32), var.ctypes.data_as(f32),
                                               mean.ctypes.data_as(f32), _lib.p8(modes)))
        return {"weight": wt, "var": var, "mean": mean, "modes": modes,
                "nframes": _lib.load().gof_mog2_nframes(m)}

    def save_state(self, path):
        """Checkpoint the background model (the reference never serialises it, SURVEY.md
        section 5): an .npz of the SoA planes plus the frame counter and geometry."""
        st = self.mog2_state()
        np.savez_compressed(path, geometry=np.array([self.streams, self.h, self.w, self.model_c]),
                            nmixtures=np.array(self.nmixtures), **st)

    def load_state(self, path):
        """Restore a checkpoint written by save_state() into this pipeline."""
        with np.load(path, allow_pickle=False) as z:
            if z["geometry"].tolist() != [self.streams, self.h, self.w, self.model_c]:
                raise ValueError("checkpoint geometry %r does not match %r"
                                 % (z["geometry"].tolist(), [self.streams, self.h, self.w, self.model_c]))
            m = _lib.load().gof_pipeline_mog2(self._h)
            if not m:
                raise RuntimeError("not a MOG2 pipeline")
            f32 = C.POINTER(C.c_float)
            if "nmixtures" in z.files and int(z["nmixtures"]) != self.nmixtures:
                raise ValueError("checkpoint has nmixtures=%d, this pipeline %d"
                                 % (int(z["nmixtures"]), self.nmixtures))
            st = {k: z[k] for k in ("weight", "var", "mean", "modes")}
            if "nmixtures" not in z.files:
                # a checkpoint of the first format: planes padded to 5 modes, no nmixtures key --
                # the modes this pipeline keeps are the first ones
                K = self.nmixtures
                if st["weight"].shape[0] >= K and st["mean"].shape[0] >= K and int(st["modes"].max(initial=0)) <= K:
                    st.update(weight=st["weight"][:K], var=st["var"][:K], mean=st["mean"][:K])
            wt, var, mean, modes = _lib.mog2_state_arrays(
                st, self.nmixtures, self.model_c, self.streams * self.h * self.w)
            _lib.check(_lib.load().gof_mog2_import(m, wt.ctypes.data_as(f32), var.ctypes.data_as(f32),
                                                   mean.ctypes.data_as(f32), _lib.p8(modes),
                                                   int(z["nframes"])))

    def mog2_device_state(self):
        """(device pointer, bytes) of the whole model: one buffer, RCCL-broadcastable."""
        m = _lib.load().gof_pipeline_mog2(self._h)
        if not m:
            raise RuntimeError("not a MOG2 pipeline")
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.load().gof_mog2_device_state(m, C.byref(p), C.byref(n)))
        return p.value, n.value

    def state_to_buffer(self, device_ptr):
        """Copy the whole model into a caller-owned device buffer (mog2_device_state()[1] bytes)."""
        m = _lib.load().gof_pipeline_mog2(self._h)
        _lib.check(_lib.load().gof_mog2_state_copy(m, C.c_void_p(device_ptr), 0))

    def state_from_buffer(self, device_ptr, nframes):
        m = _lib.load().gof_pipeline_mog2(self._h)
        _lib.check(_lib.load().gof_mog2_state_copy(m, C.c_void_p(device_ptr), 1))
        self.set_nframes(nframes)

    def nframes(self):
        m = _lib.load().gof_pipeline_mog2(self._h)
        return _lib.load().gof_mog2_nframes(m) if m else 0

    def set_nframes(self, nframes):
        m = _lib.load().gof_pipeline_mog2(self._h)
        if m:
            _lib.check(_lib.load().gof_mog2_set_nframes(m, int(nframes)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().gof_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
