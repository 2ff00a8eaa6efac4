"""GOFPID: good old fashioned perimeter intrusion detection, MI355X foreground path.

Drop-in for `pygofpid.pipeline.GOFPID` (reference pygofpid/pipeline.py:22-853):
same constructor arguments and defaults, `initialize()`, `detect(X) -> bool`,
`display(X, display)` (plus `detect_many(frames)`, the batch form), and the attributes `foreground_mask_`, `blobs_`,
`tracked_blobs_`, `input_shape_`.

What runs where
  steps 1-3 (blur, background subtraction, morphology; reference :461-476)
      on the GPU.  When the configured callables are this package's own
      (`cv.GaussianBlur`, 'MOG2' | 'FD', `cv.erode` / `cv.dilate`) the three
      steps are ONE `ForegroundPipeline` step: one upload, fused kernels, one
      download of the mask.  Any other callables are called exactly as the
      reference calls them (`fun(X, ksize, borderType)`, `.apply(X)`,
      `fun(mask, kernel)`).
  step 4's front end (contour retrieval, the area filter and the blob moments;
      reference helpers.py:125-168, pipeline.py:555-572) on the GPU as well when
      steps 1-3 are fused: only blob records and contour points are downloaded,
      `foreground_mask_` is fetched from the device when it is read.
  the rest of steps 4-7 (tracking, post-filter, alarm; reference :478-491) on the
      host, on `pygofpid_amd.hostcv` (no cv2 needed).

Deliberate deviations, all off the hot path:
  * dict / list defaults are copied per instance (the reference shares and
    mutates its mutable defaults);
  * the 'ViBe' detector and the interactive `display_config` windows are not built
    (NotImplementedError); 'KNN' keeps cv2's model and rule
    but draws its refresh slots from a hash instead of rand() (cv.BackgroundSubtractorKNN).
"""
import math

import numpy as np

from . import blobs as gpublobs
from . import cv, hostcv
from .foreground import ForegroundPipeline
from .helpers import (SimpleLinearRegression, find_contours, get_bottom, get_bottom_and_center,
                      get_center, unnormalize_coords)
from .segmentation import FrameDifferencing

_DEFAULT_BLUR = {'fun': cv.GaussianBlur, 'ksize': (3, 3), 'borderType': cv.BORDER_DEFAULT}
_DEFAULT_TRACKING = {'factor': 1.5, 'absence_max': 10}
_DEFAULT_POST_FILTER = {'perimeter': None, 'anchor': 'bottom', 'perspective': None,
                        'perspective_coeff': 0.5, 'presence_min': 3, 'distance_min': 0.25}
_DEFAULT_ALARM = {'keep_alarm': True, 'duration_min': 10}
_UNSET = object()


def _wrap(v, bits):
    """A python int as the two's-complement integer of `bits` bits numpy would have kept."""
    half = 1 << (bits - 1)
    return ((v + half) & ((1 << bits) - 1)) - half


def _argmin(row):
    """np.argmin of a list of floats: the first smallest entry; the first NaN if there is one."""
    for j, v in enumerate(row):
        if v != v:
            return j
    return row.index(min(row))


def _default_morph():
    k = cv.getStructuringElement(cv.MORPH_RECT, (5, 5))
    return [{'fun': cv.erode, 'kernel': k}, {'fun': cv.dilate, 'kernel': k.copy()}]


def _require(mapping, key, name):
    if key not in mapping.keys():
        raise KeyError('Parameter %s has no key "%s".' % (name, key))


class GOFPID():
    """See the module docstring; parameters as in the reference (pipeline.py:37-127)."""

    def __init__(self, convert=None, blur=_UNSET, frg_detect='MOG2', mat_morph=_UNSET,
                 tracking=_UNSET, post_filter=_UNSET, alarm_def=_UNSET, verbose=False):
        self.convert = convert
        self.blur = dict(_DEFAULT_BLUR) if blur is _UNSET else blur
        self.frg_detect = frg_detect
        self.mat_morph = _default_morph() if mat_morph is _UNSET else mat_morph
        self.tracking = dict(_DEFAULT_TRACKING) if tracking is _UNSET else tracking
        self.post_filter = dict(_DEFAULT_POST_FILTER) if post_filter is _UNSET else post_filter
        self.alarm_def = dict(_DEFAULT_ALARM) if alarm_def is _UNSET else alarm_def
        self.verbose = verbose

    # ------------------------------------------------------------------ init
    def initialize(self):
        """Check parameters and set the pipeline up (reference pipeline.py:214-283).
        Nothing is allocated on the GPU here: the frame size is only known at the
        first detect()."""
        if self.blur:
            _require(self.blur, 'fun', 'blur')
            self.blur.setdefault('ksize', (3, 3))
            self.blur.setdefault('borderType', cv.BORDER_DEFAULT)

        if self.frg_detect == 'MOG2':
            self._frg_detect_mth = cv.createBackgroundSubtractorMOG2()
        elif self.frg_detect == 'FD':
            self._frg_detect_mth = FrameDifferencing()
        elif self.frg_detect == 'KNN':
            self._frg_detect_mth = cv.createBackgroundSubtractorKNN()
        elif self.frg_detect == 'ViBe':
            raise NotImplementedError(
                "frg_detect='ViBe' is outside the MI355X foreground path (MOG2, KNN and FD are built): a "
                "per-pixel Python loop driven by one sequential RandomState (reference segmentation.py:73-181)")
        else:
            raise ValueError('Unknown method for foreground detection')

        if self.mat_morph:
            for d in self.mat_morph:
                _require(d, 'fun', 'mat_morph')
                if 'kernel' not in d.keys():
                    d['kernel'] = cv.getStructuringElement(cv.MORPH_RECT, (5, 5))

        _require(self.tracking, 'factor', 'tracking')
        _require(self.tracking, 'absence_max', 'tracking')

        _require(self.post_filter, 'anchor', 'post_filter')
        if self.post_filter['anchor'] not in ['bottom', 'center']:
            raise ValueError('Parameter anchor must be "bottom" or "center".')
        for key in ('perspective_coeff', 'presence_min', 'distance_min'):
            _require(self.post_filter, key, 'post_filter')
        self._check_perimeter()
        self._check_perspective()

        _require(self.alarm_def, 'keep_alarm', 'alarm_def')
        _require(self.alarm_def, 'duration_min', 'alarm_def')

        self.tracked_blobs_ = []
        self.input_shape_ = None
        self._alarm_system = {'already_in_alarm': False, 'duration': 0}
        self._fused = None          # ForegroundPipeline, built at the first frame
        self._fused_checked = False
        self._mask = None           # host copy of the final mask (lazy on the fused path)
        self._mask_on_device = False
        self._mask_index = 0        # which mask of the last fused call is the current frame's
        self._gpu_blobs = None      # (records, points) of the last frame, or None
        return self

    def _check_polygon_param(self, key, default, check):
        _require(self.post_filter, key, 'post_filter')
        value = self.post_filter[key]
        if value is None:
            self.post_filter[key] = np.array(default)
        elif isinstance(value, np.ndarray):
            check(value)
        else:
            raise ValueError('Unknown type for parameter "%s".' % key)
        if self.post_filter.get('display_config'):
            raise NotImplementedError(
                'display_config opens interactive cv2 windows (reference pipeline.py:308-407); '
                'not available in this headless build')

    def _check_perimeter(self):
        def check(p):
            if p.ndim != 2 or p.shape[1] != 2:
                raise ValueError('Parameter perimeter has not the good shape.')
            if p.shape[0] < 3:
                raise ValueError('Parameter perimeter has not enough points.')
        self._check_polygon_param('perimeter', [[0, 0], [0, 1], [1, 1], [1, 0]], check)

    def _check_perspective(self):
        def check(p):
            if p.shape != (4, 2):
                raise ValueError('Parameter perspective has not the good shape.')
        self._check_polygon_param('perspective',
                                  [[0.1, 0.4], [0.2, 0.8], [0.8, 0.2], [0.85, 0.3]], check)

    # ---------------------------------------------------------------- detect
    def detect(self, X):
        """True if an intrusion is detected in frame X (reference pipeline.py:434-493)."""
        if not self.input_shape_:
            self.input_shape_ = X.shape
            self._calib_first_frame()
        elif X.shape != self.input_shape_:
            raise ValueError('Input shape has changed.')

        # step 0 + steps 1-3 on the GPU (and, fused, the contours / blob statistics of step 4)
        self._foreground(X)

        # steps 4-7 on the host
        self._find_blob()
        self._track_blob()
        self._post_filter()
        return self._maintain_alarm(self._detect_blob())

    _CHUNK = 8   # frames per device batch of detect_many (the multi-frame kernels take 8 per launch)

    def detect_many(self, frames):
        """detect() on consecutive frames of the stream, in order: a list of its return values.

        Not in the reference; same results as calling detect() frame by frame (steps 4-7 never
        feed back into steps 1-3).  On the fused path the frames go through the device in batches
        of 8 (one upload from the frames' own buffers, the multi-frame kernels -- the background
        model crosses HBM once per batch -- and one small download of all blob records), pipelined:
        batch k + 1 is uploaded (by a helper thread, beside the kernels of batch k) while this
        thread runs the host steps on the records of batch k.  Any number of frames; memory is
        bounded by two batches.  All frames are validated before the first one is consumed.
        Attributes (`blobs_`, `tracked_blobs_`, `foreground_mask_`) are those of the last frame.
        If a device call fails part-way (RuntimeError / MemoryError), the background model has taken the
        frames of the batches submitted so far (up to 16 beyond the last frame whose host steps ran):
        do not feed those frames again -- continue with the frames that follow, or initialize() anew.
        """
        frames = [np.asarray(f) for f in frames]
        shape = self.input_shape_ or (frames[0].shape if frames else None)
        for f in frames:
            if f.shape != shape:
                raise ValueError('Input shape has changed.')
        out = []
        if frames and not getattr(self, '_fused_checked', False):
            out.append(self.detect(frames[0]))        # decides the path, calibrates on the shape
            frames = frames[1:]
        if self._fused is None or not frames:
            return out + [self.detect(f) for f in frames]
        if getattr(self, '_uploader', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._uploader = ThreadPoolExecutor(max_workers=1, thread_name_prefix='gofpid-upload')
        import queue
        import threading
        chunks = [frames[i:i + self._CHUNK] for i in range(0, len(frames), self._CHUNK)]
        submit = self._fused.submit_blobs_frames
        # The helper thread walks the batches on its own: batch k + 2 goes up as soon as this thread has
        # taken batch k's records out of its slot (free[slot]) -- not when it has finished their host steps --
        # so the uploads follow each other on the link.  done: batch numbers (or the exception) in order.
        free = [threading.Semaphore(1), threading.Semaphore(1)]
        done = queue.SimpleQueue()
        stop = threading.Event()

        def upload_all():
            try:
                for k, chunk in enumerate(chunks):
                    free[k & 1].acquire()
                    if stop.is_set():
                        return
                    submit(chunk, self._AREA_MIN, k & 1)
                    done.put(k)
            except BaseException as e:          # handed to the calling thread
                done.put(e)

        fut = self._uploader.submit(upload_all)
        try:
            for k in range(len(chunks)):
                slot = k & 1
                got = done.get()                          # batch k is on the device (or the upload raised)
                if isinstance(got, BaseException):
                    raise got
                results = self._fused.wait_blobs(slot)
                # (a frame whose device scratch overflowed takes its mask and the host path: fetched before
                # the slot is handed back)
                masks = self._fused.slot_masks(slot) if any(r is None for r in results) else None
                if k + 2 < len(chunks):
                    free[slot].release()
                for i, res in enumerate(results):
                    self._gpu_blobs = res
                    self._mask = masks[i, 0] if res is None else None
                    self._mask_on_device = True
                    self._mask_index = i
                    self._mask_slot = slot
                    self._find_blob()
                    self._track_blob()
                    self._post_filter()
                    out.append(self._maintain_alarm(self._detect_blob()))
        finally:
            stop.set()
            for sem in free:
                sem.release()                             # (a helper waiting for a slot leaves)
            fut.result()
        return out

    def _fusable(self):
        """(blur, detector, morph) arguments of a ForegroundPipeline, or None when
        the configuration contains callables that are not this package's."""
        if self.blur:
            kw, kh = self.blur.get('ksize')
            if kw < 1 or kh < 1 or kw % 2 == 0 or kh % 2 == 0:
                return None
            if self.blur.get('fun') is cv.GaussianBlur:
                # reference call convention: borderType arrives as sigmaX (pipeline.py:463-467)
                blur = (int(kw), int(kh), float(self.blur.get('borderType')))
            elif self.blur.get('fun') is cv.blur:
                # ... and as `dst` for cv.blur: ignored
                blur = ('box', int(kw), int(kh))
            else:
                return None
        else:
            blur = None
        if self.frg_detect not in ('MOG2', 'FD', 'KNN'):
            return None
        morph = []
        for d in self.mat_morph or []:
            if d.get('fun') is cv.erode:
                morph.append(('erode', d.get('kernel')))
            elif d.get('fun') is cv.dilate:
                morph.append(('dilate', d.get('kernel')))
            else:
                return None
        return blur, self.frg_detect, morph

    _AREA_MIN = 100   # the reference's fixed area_min (pipeline.py:555)

    @property
    def foreground_mask_(self):
        """The final mask of the last frame.  On the fused path it lives on the device
        and is downloaded when (and only when) it is read."""
        if self._mask is None and self._mask_on_device:
            slot = getattr(self, '_mask_slot', None)
            masks = self._fused.last_masks() if slot is None else self._fused.slot_masks(slot)
            self._mask = masks[self._mask_index, 0]
        if self._mask is None:
            raise AttributeError("foreground_mask_")
        return self._mask

    @foreground_mask_.setter
    def foreground_mask_(self, value):
        self._mask = value
        self._mask_on_device = False
        self._gpu_blobs = None

    def _foreground(self, X):
        if not self._fused_checked:
            self._fused_checked = True
            args = self._fusable()
            conv = {cv.COLOR_BGR2GRAY: 'BGR2GRAY', cv.COLOR_RGB2GRAY: 'RGB2GRAY'}.get(self.convert) \
                if self.convert else None
            if args is not None and (not self.convert or (conv and X.ndim == 3 and X.shape[2] in (3, 4))):
                shape = X.shape if X.ndim == 3 else (X.shape[0], X.shape[1], 1)
                try:
                    self._fused = ForegroundPipeline(
                        shape, streams=1, blur=args[0], detector=args[1], morph=args[2],
                        fd_threshold=getattr(self._frg_detect_mth, 'threshold', 5), convert=conv)
                except ValueError:
                    self._fused = None      # e.g. kernel larger than the fused path takes
        if self._fused is not None:
            # one upload, the kernels of steps 1-3, the blob extraction, one small download
            self._gpu_blobs = self._fused.apply_blobs(X, min_area=self._AREA_MIN)[0]
            self._mask = None
            self._mask_on_device = True
            self._mask_index = 0
            self._mask_slot = None
            return
        # generic path: the reference's own call sequence
        if self.convert:
            X = cv.cvtColor(X, self.convert)
        if self.blur:
            X = self.blur.get('fun')(X, self.blur.get('ksize'), self.blur.get('borderType'))
        mask = self._frg_detect_mth.apply(X)
        if self.mat_morph:
            for d in self.mat_morph:
                mask = d.get('fun')(mask, d.get('kernel'))
        self.foreground_mask_ = mask

    # ----------------------------------------------------------- calibration
    def _calib_first_frame(self):
        self.post_filter['perimeter'] = unnormalize_coords(
            self.post_filter['perimeter'], self.input_shape_, dtype=np.int32)
        self.post_filter['perspective'] = unnormalize_coords(
            self.post_filter['perspective'], self.input_shape_, dtype=np.float64)
        self._calib_perspective()

    def _calib_perspective(self):
        """Area / height / width of a person as linear functions of the row of its
        bottom point, from the two reference boxes (reference pipeline.py:513-547)."""
        (x0, y0), (x1, y1), (x2, y2), (x3, y3) = self.post_filter['perspective']
        bottoms = [np.max([y1, y0]), np.max([y3, y2])]
        rows = np.arange(0, self.input_shape_[0], 1)
        samples = {
            'area': [abs((y1 - y0) * (x1 - x0)), abs((y3 - y2) * (x3 - x2))],
            'height': [abs(y1 - y0), abs(y3 - y2)],
            'width': [abs(x1 - x0), abs(x3 - x2)],
        }
        slr = SimpleLinearRegression()
        self._perspective = {key: slr.fit(bottoms, val).predict_clip(rows, 0)
                             for key, val in samples.items()}
        # (the same float64 values as python floats: steps 5-7 read a few per blob and frame)
        self._perspective_rows = {key: np.asarray(val, dtype=np.float64).tolist()
                                  for key, val in self._perspective.items()}

    def _predict_perspective(self, key, y):
        rows = self._perspective_rows[key]
        return rows[min(max(int(y), 0), len(rows) - 1)]       # (np.clip of a scalar, without its overhead)

    # ----------------------------------------------------------------- blobs
    def _find_blob(self, area_min=100):
        if self._gpu_blobs is not None and area_min == self._AREA_MIN:
            # contours, area filter and moments came from the device with the mask's header
            recs, pts = self._gpu_blobs
            bottoms, centers = gpublobs.bottoms_and_centers(recs, dtype=np.int16)   # all records at once
            anchors = bottoms if self.post_filter['anchor'] == 'bottom' else centers
            xmin, ymin, xmax, ymax = (recs[k].tolist() for k in ('xmin', 'ymin', 'xmax', 'ymax'))
            self.blobs_ = [{'contour': contour, 'anchor': anchors[i], 'bottom': bottoms[i], 'center': centers[i],
                            # (cv.boundingRect of the contour, from the record's extremes)
                            'bbox': (xmin[i], ymin[i], xmax[i] - xmin[i] + 1, ymax[i] - ymin[i] + 1)}
                           for i, contour in enumerate(gpublobs.contours_of(recs, pts))]
            return
        contours = [c for c in find_contours(self.foreground_mask_)
                    if hostcv.contourArea(c) >= area_min]
        self.blobs_ = [self._create_blob(c) for c in contours]

    def _create_blob(self, contour):
        bottom, center = get_bottom_and_center(contour, dtype=np.int16)
        anchor = bottom if self.post_filter['anchor'] == 'bottom' else center
        return {'contour': contour, 'anchor': anchor, 'bottom': bottom, 'center': center}

    # -------------------------------------------------------------- tracking
    def _track_blob(self):
        """Greedy nearest-centre pairing inside an elliptic gate (reference :590-628)."""
        n_tracked, n_blobs = len(self.tracked_blobs_), len(self.blobs_)
        if n_tracked == 0:
            self.tracked_blobs_.extend(self._create_tracked_blob(i) for i in range(n_blobs))
            return
        if n_blobs == 0:
            for j in reversed(range(n_tracked)):
                self._update_unpaired_tracked_blob(j)
            return

        dist = self._tracking_distances().tolist() if self._slow_host_steps else self._tracking_distance_rows()

        for i in range(n_blobs):
            row = dist[i]
            j_min = _argmin(row)
            if row[j_min] <= 1:
                row[j_min] = -1              # mark the pair
                self._update_paired_tracked_blob(j_min, i)
            else:
                self.tracked_blobs_.append(self._create_tracked_blob(i))

        for j in reversed(range(n_tracked)):
            if all(row[j] >= 0 for row in dist):
                self._update_unpaired_tracked_blob(j)

    _slow_host_steps = False     # True: the numpy forms below (the checkers of the python-scalar forms)

    def _tracking_distances(self):
        """_compute_tracking_distance for all (blob, tracked blob) pairs at once: the same int16 /
        int32 / float64 arithmetic element for element; a tracked blob's gate is evaluated once."""
        gates = [self._compute_tracking_space(t['bottom']) for t in self.tracked_blobs_]
        r_maj = np.array([g[0] for g in gates], dtype=np.int32)
        r_min = np.array([g[1] for g in gates], dtype=np.int32)
        delta = (np.array([b['center'] for b in self.blobs_])[:, None, :]
                 - np.array([t['center'] for t in self.tracked_blobs_])[None, :, :])
        ok = (r_maj > 0) & (r_min > 0)
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            return np.where(ok[None, :], delta[..., 0]**2 / r_maj**2 + delta[..., 1]**2 / r_min**2, 0.0)

    def _tracking_distance_rows(self):
        """_tracking_distances() as a list of rows of python floats, evaluated on python ints: the
        int16 differences and squares and the int32 squared radii wrapped by hand, the float64
        quotients by true division of exact integers (correctly rounded, as float64 / float64 of
        exactly representable operands is).  With ~10 blobs this is several times cheaper than
        the array form, whose cost is numpy's per-call overhead.  Anything the scalar form does
        not cover (a squared radius wrapping to 0, non-finite or out-of-range gates, centres that
        are not int16) goes through the array form."""
        factor = self.tracking['factor']
        heights = self._perspective_rows['height']
        last = len(heights) - 1
        try:
            gates = []
            for t in self.tracked_blobs_:
                c = t['center']
                if c.dtype != np.int16:
                    raise ValueError
                r_maj = heights[min(max(int(t['bottom'][1]), 0), last)] * factor
                if not -2147483648.0 < r_maj < 2147483648.0:
                    raise ValueError
                r_maj = int(r_maj)                       # np.int32(float64): truncation
                r_min = int(r_maj / 2)
                cx, cy = c.tolist()
                if r_maj > 0 and r_min > 0:
                    r_maj, r_min = _wrap(r_maj * r_maj, 32), _wrap(r_min * r_min, 32)
                    if r_maj == 0 or r_min == 0:
                        raise ZeroDivisionError
                    gates.append((cx, cy, r_maj, r_min))
                else:
                    gates.append((cx, cy, 0, 0))
            rows = []
            for b in self.blobs_:
                c = b['center']
                if c.dtype != np.int16:
                    raise ValueError
                bx, by = c.tolist()
                row = []
                for cx, cy, r_maj, r_min in gates:
                    if r_maj:
                        dx, dy = bx - cx, by - cy
                        if not (-32768 <= dx <= 32767 and -32768 <= dy <= 32767):
                            dx, dy = _wrap(dx, 16), _wrap(dy, 16)
                        dx *= dx
                        dy *= dy
                        if dx > 32767:                   # (the int16 square wraps from |d| = 182 on)
                            dx = ((dx + 32768) & 65535) - 32768
                        if dy > 32767:
                            dy = ((dy + 32768) & 65535) - 32768
                        row.append(dx / r_maj + dy / r_min)
                    else:
                        row.append(0.0)
                rows.append(row)
            return rows
        except (ValueError, ZeroDivisionError, OverflowError, AttributeError):
            return self._tracking_distances().tolist()

    def _create_tracked_blob(self, i_blob):
        blob = self.blobs_[i_blob]
        return {
            'contour': blob['contour'].copy(), 'anchor': blob['anchor'], 'bottom': blob['bottom'],
            'center': blob['center'], 'center_first': blob['center'], 'presence': 1, 'absence': 0,
            'filtered': set(['presence', 'distance']), 'already_in_alarm': False, 'bbox': blob.get('bbox'),
        }

    def _compute_tracking_distance(self, i_tracked_blob, i_blob):
        tracked = self.tracked_blobs_[i_tracked_blob]
        radius_maj, radius_min = self._compute_tracking_space(tracked['bottom'])
        if radius_maj <= 0 or radius_min <= 0:
            return 0
        delta = self.blobs_[i_blob]['center'] - tracked['center']   # int16, as the reference
        return delta[0]**2 / radius_maj**2 + delta[1]**2 / radius_min**2

    def _compute_tracking_space(self, bottom):
        height_min = self._predict_perspective('height', bottom[1])
        radius_major = np.int32(height_min * self.tracking['factor'])
        radius_minor = np.int32(radius_major / 2)
        return radius_major, radius_minor

    def _update_paired_tracked_blob(self, i_tracked_blob, i_blob):
        tracked, blob = self.tracked_blobs_[i_tracked_blob], self.blobs_[i_blob]
        tracked['contour'] = blob['contour'].copy()
        for key in ('anchor', 'bottom', 'center'):
            tracked[key] = blob[key]
        tracked['bbox'] = blob.get('bbox')
        tracked['absence'] = 0
        tracked['presence'] += 1
        if tracked['presence'] >= self.post_filter['presence_min']:
            tracked['filtered'].discard('presence')
        if self._compute_total_distance(i_tracked_blob) >= self.post_filter['distance_min']:
            tracked['filtered'].discard('distance')

    def _compute_total_distance(self, i_tracked_blob):
        tracked = self.tracked_blobs_[i_tracked_blob]
        c, f = tracked['center'], tracked['center_first']
        if self._slow_host_steps or c.dtype != np.int16 or f.dtype != np.int16:
            dist = hostcv.norm(c - f)
        else:
            # the int16 difference (wrapping), its float64 norm: sqrt of an exact integer sum
            (cx, cy), (fx, fy) = c.tolist(), f.tolist()
            dx, dy = _wrap(cx - fx, 16), _wrap(cy - fy, 16)
            dist = math.sqrt(float(dx * dx + dy * dy))
        area_min = self._predict_perspective('area', tracked['bottom'][1])
        if area_min <= 0:
            return 0
        return dist / math.sqrt(area_min)

    def _update_unpaired_tracked_blob(self, i_tracked_blob):
        tracked = self.tracked_blobs_[i_tracked_blob]
        tracked['presence'] = 0
        tracked['filtered'].add('presence')
        tracked['absence'] += 1
        if tracked['absence'] >= self.tracking['absence_max']:
            self.tracked_blobs_.pop(i_tracked_blob)

    # ----------------------------------------------------------- post-filter
    def _post_filter(self):
        perimeter = hostcv.polygon_points(self.post_filter['perimeter'])
        coeff = self.post_filter['perspective_coeff']
        for tracked in self.tracked_blobs_:
            anchor, filtered = tracked['anchor'], tracked['filtered']
            inside = hostcv.point_in_polygon(perimeter, float(anchor[0]), float(anchor[1])) >= 0
            (filtered.discard if inside else filtered.add)('perimeter')

            _, _, width, height = tracked.get('bbox') or hostcv.boundingRect(tracked['contour'])
            area_persp = self._predict_perspective('area', tracked['bottom'][1])
            big_enough = width * height > area_persp * coeff
            (filtered.discard if big_enough else filtered.add)('perspective')

    def _detect_blob(self):
        blob_in_alarm = False
        for tracked in self.tracked_blobs_:
            if not tracked['filtered']:
                blob_in_alarm = True
                if self.alarm_def['keep_alarm']:
                    tracked['already_in_alarm'] = True
            elif tracked['already_in_alarm']:
                blob_in_alarm = True
        return blob_in_alarm

    def _maintain_alarm(self, sys_in_alarm):
        state = self._alarm_system
        state['duration'] -= 1
        if sys_in_alarm and not state['already_in_alarm']:
            state['duration'] = self.alarm_def['duration_min']     # beginning of an intrusion
        if not sys_in_alarm and state['duration'] >= 0:
            sys_in_alarm = True
        if self.verbose and sys_in_alarm and not state['already_in_alarm']:
            print("Intrusion detected")
        state['already_in_alarm'] = sys_in_alarm
        return sys_in_alarm

    # --------------------------------------------------------------- display
    def display(self, X, display={'tracking': False, 'perspective': False}):
        """Draw perimeter, blobs and (optionally) tracking / perspective gates on X,
        in place (reference pipeline.py:801-853)."""
        hostcv.drawContours(X, [self.post_filter['perimeter']], 0, (25, 200, 200))
        for tracked in self.tracked_blobs_:
            flt = tracked['filtered']
            if flt == set() or tracked['already_in_alarm']:
                color = (0, 0, 255)
            elif 'perimeter' in flt:
                color = (255, 0, 0)
            elif 'perspective' in flt:
                color = (25, 200, 200)
            elif 'presence' in flt or 'distance' in flt:
                color = (0, 255, 0)
            else:
                raise ValueError(f'Unknown filtering type {flt}')
            hostcv.drawContours(X, [tracked['contour']], 0, color)
            hostcv.circle(X, tracked['anchor'], 4, color, -1)
            if display.get('tracking'):
                r_maj, r_min = self._compute_tracking_space(tracked['bottom'])
                hostcv.ellipse(X, tracked['center'], [r_maj, r_min], 0, 0, 360, color, 1)
            if display.get('perspective'):
                bottom = tracked['bottom']
                height = np.int32(self._predict_perspective('height', bottom[1]))
                width = np.int32(self._predict_perspective('width', bottom[1]))
                hostcv.rectangle(X, [bottom[0] - width // 2, bottom[1]],
                                 [bottom[0] + width // 2, bottom[1] - height], color, 1)


__all__ = ['GOFPID']
