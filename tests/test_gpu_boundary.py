"""GPU tests of the C-ABI boundary conditions (run on a real MI355X through gpurun):
caller streams, caller-owned device buffers of any alignment, and the warm-start /
checkpoint hand-over of the MOG2 model.  Checker: the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

from pygofpid_amd.synth import Clip

pytestmark = pytest.mark.gpu


def test_warm_start_device_path(oracle):
    """pipeline A trains, its model travels through a device buffer
    (state_to_buffer -> state_from_buffer + nframes, the RCCL broadcast's payload) into a
    fresh pipeline B; A and B then give identical masks and identical exported state for
    the next 16 frames, through an 8-frame launch and single-frame launches, and both
    match the oracle that saw the whole clip."""
    import torch
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 2, 72, 128, 3
    clips = [Clip(h, w, c, seed=70 + s, n_objects=3) for s in range(S)]
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    a = ForegroundPipeline((h, w, c), streams=S)
    for t0 in range(0, 24, 8):
        fr = np.stack([np.stack([cl.frame(t0 + i) for cl in clips]) for i in range(8)])
        got = a.apply_many(fr)
        for i in range(8):
            for s in range(S):
                assert np.array_equal(got[i, s], refs[s].apply(fr[i, s]))
    _, nbytes = a.mog2_device_state()
    buf = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", a.device))
    a.state_to_buffer(buf.data_ptr())
    b = ForegroundPipeline((h, w, c), streams=S)
    assert b.nframes() == 0
    b.state_from_buffer(buf.data_ptr(), a.nframes())
    assert b.nframes() == a.nframes() == 24
    fr = np.stack([np.stack([cl.frame(24 + i) for cl in clips]) for i in range(8)])
    ma, mb = a.apply_many(fr), b.apply_many(fr)
    assert np.array_equal(ma, mb)
    for i in range(8):
        for s in range(S):
            assert np.array_equal(ma[i, s], refs[s].apply(fr[i, s]))
    for t in range(32, 40):
        f = np.stack([cl.frame(t) for cl in clips])
        ma, mb = a.apply(f), b.apply(f)
        assert np.array_equal(ma, mb)
        for s in range(S):
            assert np.array_equal(ma[s], refs[s].apply(f[s]))
    sa, sb = a.mog2_state(), b.mog2_state()
    assert sa["nframes"] == sb["nframes"] == 40
    assert np.array_equal(sa["modes"], sb["modes"])
    live = np.arange(a.nmixtures)[:, None] < sa["modes"][None, :]
    for key in ("weight", "var"):
        assert np.array_equal(sa[key][live], sb[key][live])
    assert np.array_equal(sa["mean"][np.broadcast_to(live[:, None, :], sa["mean"].shape)],
                          sb["mean"][np.broadcast_to(live[:, None, :], sb["mean"].shape)])


def test_warm_start_single_process():
    """distributed.warm_start() with no process group: the model makes the round trip
    through the staging buffer on the pipeline's own device and nothing changes."""
    from pygofpid_amd import distributed as gd
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 48, 64, 3
    clip = Clip(h, w, c, seed=5)
    a, b = ForegroundPipeline((h, w, c)), ForegroundPipeline((h, w, c))
    for t in range(6):
        f = clip.frame(t)
        assert np.array_equal(a.apply(f), b.apply(f))
    assert gd.warm_start(a) == 6
    for t in range(6, 12):
        f = clip.frame(t)
        assert np.array_equal(a.apply(f), b.apply(f))


def test_apply_on_caller_stream_right_after_create(oracle):
    """the model's zero-fill runs on the context's stream; an apply enqueued on
    ANOTHER stream immediately after create must see the finished model, and the export
    that follows must see the finished apply."""
    import torch
    from pygofpid_amd.foreground import ForegroundPipeline
    S, h, w, c = 8, 540, 960, 3     # 0.4 GB of model: the fill takes a while
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 255, (2, S, h, w, c), dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    d_masks = torch.empty((S, h, w), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    pipe = ForegroundPipeline((h, w, c), streams=S, device=0)
    refs = [oracle.Pipeline(h, w, c) for _ in range(S)]
    for t in range(2):
        pipe.apply_device(d_frames[t].data_ptr(), d_masks.data_ptr(), stream=side.cuda_stream)
        if t == 1:
            st = pipe.mog2_state()     # waits for `side` itself
        side.synchronize()
        got = d_masks.cpu().numpy()
        for s in range(S):
            assert np.array_equal(got[s], refs[s].apply(frames[t, s])), (t, s)
    assert st["nframes"] == 2
    for s in range(S):
        want = refs[s].export()
        n = h * w
        assert np.array_equal(st["modes"][s * n:(s + 1) * n], want["modes"])


@pytest.mark.parametrize("offset", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(37, 53, 3), (40, 64, 3), (31, 30, 1), (24, 32, 4)])
def test_device_buffers_any_alignment(oracle, offset, shape):
    """*_dev callers owe no alignment and no slack.  The frames sit at
    byte `offset` of a torch allocation and END on its last byte; masks likewise.  Odd
    offsets and ragged rows take the general staging path, whose first / last dword is
    assembled from in-buffer bytes only.  One- and multi-frame calls."""
    import torch
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = shape
    T = 3
    clip = Clip(h, w, c, seed=11 + offset, n_objects=2)
    frames = np.stack([clip.frame(t) for t in range(2 * T)]).reshape(2 * T, h, w, c)
    dev = torch.device("cuda", 0)
    nb, nm = T * h * w * c, T * h * w
    fbuf = torch.empty(offset + nb, dtype=torch.uint8, device=dev)
    mbuf = torch.full((offset + nm,), 7, dtype=torch.uint8, device=dev)
    fview, mview = fbuf[offset:], mbuf[offset:]
    assert fview.data_ptr() % 4 == (fbuf.data_ptr() + offset) % 4
    pipe = ForegroundPipeline((h, w, c), device=0)
    ref = oracle.Pipeline(h, w, c)
    # T frames in one call, then T single-frame calls
    fview.copy_(torch.from_numpy(frames[:T].reshape(-1)).to(dev))
    pipe.apply_device(fview.data_ptr(), mview.data_ptr(), n_frames=T)
    pipe.synchronize()
    got = mview.cpu().numpy().reshape(T, h, w)
    for t in range(T):
        assert np.array_equal(got[t], ref.apply(frames[t].reshape(h, w, c) if c > 1 else frames[t, :, :, 0])), t
    if offset:
        assert (mbuf[:offset].cpu().numpy() == 7).all()      # nothing written before the view
    fview.copy_(torch.from_numpy(frames[T:].reshape(-1)).to(dev))
    for t in range(T):
        pipe.apply_device(fview[t * h * w * c:].data_ptr(), mview[t * h * w:].data_ptr())
    pipe.synchronize()
    got = mview.cpu().numpy().reshape(T, h, w)
    for t in range(T):
        assert np.array_equal(got[t], ref.apply(frames[T + t].reshape(h, w, c) if c > 1 else frames[T + t, :, :, 0])), t


def test_state_shape_validation(tmp_path):
    """a checkpoint written with another nmixtures (or mangled arrays) is refused
    in Python, before the C side would read K * n floats from too small an array."""
    from pygofpid_amd import _lib, cv
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = 24, 32, 3
    p3 = _lib.Mog2Params()
    _lib.load().gof_mog2_default_params(C.byref(p3))
    p3.nmixtures = 3
    a = ForegroundPipeline((h, w, c), mog2_params=p3)
    f = Clip(h, w, c, seed=1).frame(0)
    a.apply(f)
    st = a.mog2_state()
    assert st["weight"].shape == (3, h * w) and st["mean"].shape == (3, c, h * w)
    path = str(tmp_path / "k3.npz")
    a.save_state(path)
    with pytest.raises(ValueError):
        ForegroundPipeline((h, w, c)).load_state(path)           # K = 5 model, K = 3 file
    b = ForegroundPipeline((h, w, c), mog2_params=p3)
    b.load_state(path)
    f1 = Clip(h, w, c, seed=1).frame(1)
    assert np.array_equal(a.apply(f1), b.apply(f1))
    m = cv.createBackgroundSubtractorMOG2()
    with pytest.raises(ValueError):
        m.import_state(st, (h, w, c))                            # K = 3 arrays into K = 5
    bad = dict(a.mog2_state())
    bad["modes"] = bad["modes"][:-1]
    with pytest.raises(ValueError):
        b_ = ForegroundPipeline((h, w, c), mog2_params=p3)
        np.savez(str(tmp_path / "bad.npz"), geometry=np.array([1, h, w, c]), nmixtures=np.array(3), **bad)
        b_.load_state(str(tmp_path / "bad.npz"))
    # a checkpoint of the FIRST format (planes padded to 5 modes, no 'nmixtures' key) written by a
    # 3-mode pipeline still loads into a 3-mode pipeline (the modes it keeps are the first ones)
    st3 = a.mog2_state()
    pad = lambda x: np.concatenate([x, np.zeros((2,) + x.shape[1:], x.dtype)])
    np.savez(str(tmp_path / "v1.npz"), geometry=np.array([1, h, w, c]), weight=pad(st3["weight"]),
             var=pad(st3["var"]), mean=pad(st3["mean"]), modes=st3["modes"], nframes=st3["nframes"])
    d = ForegroundPipeline((h, w, c), mog2_params=p3)
    d.load_state(str(tmp_path / "v1.npz"))
    f2 = Clip(h, w, c, seed=1).frame(2)
    assert np.array_equal(a.apply(f2), d.apply(f2))


@pytest.mark.parametrize("code", ["BGR2GRAY", "RGB2GRAY"])
@pytest.mark.parametrize("cin", [3, 4])
def test_pipeline_fused_convert(oracle, code, cin):
    """Step 0 inside the pipeline (gof_pipeline_set_convert, reference pipeline.py:458-459): colour
    frames in, grey conversion on the device, then steps 1-3 -- masks and model equal the oracle's on
    the oracle-converted frames; host, multi-frame and device-pointer entry points."""
    import torch
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w = 60, 88
    clip = Clip(h, w, 3, seed=33, n_objects=2)
    pipe = ForegroundPipeline((h, w, cin), convert=code)
    ref = oracle.Pipeline(h, w, 1)
    rng = np.random.default_rng(1)

    def frame(t):
        f = clip.frame(t)
        if cin == 4:
            f = np.concatenate([f, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
        return f

    def gray(f):
        return oracle.cvt_gray(np.ascontiguousarray(f[:, :, :3]), rgb_order=(code == "RGB2GRAY"))

    for t in range(4):
        f = frame(t)
        assert np.array_equal(pipe.apply(f), ref.apply(gray(f)))
    fr = np.stack([frame(t) for t in range(4, 12)])
    got = pipe.apply_many(fr)
    for i in range(8):
        assert np.array_equal(got[i], ref.apply(gray(fr[i])))
    d = torch.from_numpy(frame(12)).cuda()
    m = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    pipe.apply_device(d.data_ptr(), m.data_ptr())
    pipe.synchronize()
    assert np.array_equal(m.cpu().numpy(), ref.apply(gray(frame(12))))
    st = pipe.mog2_state()
    assert st["mean"].shape[1] == 1 and np.array_equal(st["modes"], ref.export()["modes"])
    res = pipe.apply_blobs(frame(13), min_area=10.0)
    assert np.array_equal(pipe.last_masks()[0, 0], ref.apply(gray(frame(13)))) and res[0] is not None
    with pytest.raises(ValueError):
        ForegroundPipeline((h, w, 1), convert=code)


def test_warm_start_rccl_one_rank_rehearsal():
    """The RCCL calls of the multi-GPU path executed on hardware: a ONE-rank "nccl" process group
    (GOF_DIST_FORCE=1) in a child process -- init with the device id, barrier, the MAX / SUM
    reductions of bench.py's timing on device tensors, and distributed.warm_start() with its
    all_gather geometry check and the two broadcasts -- after which the model must be unchanged
    (a twin pipeline keeps giving identical masks).  N > 1 on real xGMI is the driver's to run."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from pygofpid_amd import distributed as gd
from pygofpid_amd.foreground import ForegroundPipeline
from pygofpid_amd.synth import Clip
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
assert gd.init(backend="nccl", device=dev) == (0, 1)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
gd.barrier()
assert gd.max_over_ranks(2.5, device=dev) == 2.5 and gd.sum_over_ranks(4.0, device=dev) == 4.0
h, w, c = 64, 96, 3
clip = Clip(h, w, c, seed=9)
a, b = ForegroundPipeline((h, w, c), streams=2), ForegroundPipeline((h, w, c), streams=2)
for t in range(6):
    f = np.stack([clip.frame(t), clip.frame(t + 50)])
    assert np.array_equal(a.apply(f), b.apply(f))
assert gd.warm_start(a) == 6
for t in range(6, 14):
    f = np.stack([clip.frame(t), clip.frame(t + 50)])
    assert np.array_equal(a.apply(f), b.apply(f))
gd.barrier()
torch.distributed.destroy_process_group()
print("RCCL-REHEARSAL-OK")
'''
    from conftest import ROOT
    env = dict(os.environ, GOF_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL-REHEARSAL-OK" in out.stdout, (out.stdout[-800:], out.stderr[-1500:])


@pytest.mark.parametrize("shape,cin", [((60, 88), 3), ((61, 87), 3), ((40, 130), 4), ((1080, 1920), 3)])
def test_convert_and_box_blur_fused_into_the_detector(oracle, shape, cin):
    """Step 0 (cv.cvtColor to grey, reference pipeline.py:458-459) and cv.blur 3x3 (reference
    tests/test_pipeline.py:68-75) inside the detector kernel: colour frames in, ONE detector launch per
    step (the byte model of the pipeline counts no separate grey / blur pass for aligned frames), masks
    equal the oracle's on oracle-converted frames for 1-frame and 8-frame calls; odd geometries take
    the separate kernels and give the same masks."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w = shape
    nfr = 10 if h < 500 else 9
    clip = Clip(h, w, 3, seed=7, n_objects=3)
    rng = np.random.default_rng(2)

    def frame(t):
        f = clip.frame(t)
        return f if cin == 3 else np.concatenate([f, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)

    frames = [frame(t) for t in range(nfr)]
    grays = [oracle.cvt_gray(np.ascontiguousarray(f[:, :, :3]), rgb_order=False) for f in frames]
    for blur, okw in (((3, 3, 4.0), dict(blur_k=3)), (("box", 3, 3), dict(blur_k=-3))):
        pipe = ForegroundPipeline((h, w, cin), convert="BGR2GRAY", blur=blur)
        ref = oracle.Pipeline(h, w, 1, **okw)
        assert np.array_equal(pipe.apply(frames[0]), ref.apply(grays[0]))
        got = pipe.apply_many(np.stack(frames[1:9]))
        for i in range(8):
            assert np.array_equal(got[i], ref.apply(grays[1 + i])), (blur, i)
        if nfr > 9:
            assert np.array_equal(pipe.apply(frames[9]), ref.apply(grays[9]))
        tot, det = pipe.algorithmic_bytes(1.0, 8)
        aligned = (w * cin) % 4 == 0 and (h * w * cin) % 4 == 0
        if aligned:
            assert tot - det == 8 * 2 * h * w        # beyond the detector: the one morphology launch only
            assert det >= 8 * (cin + 1) * h * w      # and the detector reads the colour frames itself
    # the box filter alone (no conversion) on colour frames: fused as well
    pipe = ForegroundPipeline((h, w, 3), blur=("box", 3, 3))
    ref = oracle.Pipeline(h, w, 3, blur_k=-3)
    cf = [clip.frame(t) for t in range(9)]
    assert np.array_equal(pipe.apply(cf[0]), ref.apply(cf[0]))
    got = pipe.apply_many(np.stack(cf[1:9]))
    for i in range(8):
        assert np.array_equal(got[i], ref.apply(cf[1 + i]))
    tot, det = pipe.algorithmic_bytes(1.0, 8)
    assert tot - det == 8 * 2 * h * w


@pytest.mark.parametrize("detector", ["FD", "KNN"])
@pytest.mark.parametrize("shape,cin,order", [((60, 88), 3, "BGR2GRAY"), ((61, 87), 3, "RGB2GRAY"), ((40, 130), 4, "RGB2GRAY"),
                                             ((33, 70), 4, "BGR2GRAY"), ((270, 480), 3, "BGR2GRAY")])
def test_convert_fused_into_fd_and_knn(oracle, detector, shape, cin, order):
    """Step 0 inside the frame-differencing and the KNN kernels as well (colour rows staged, grey tile
    built in LDS): masks equal the oracle's on oracle-converted frames, one frame and eight frames per
    call, both channel orders, 3 and 4 source channels; geometries the staged front end does not take
    (odd widths, 4-channel frames whose grey rows are not dword multiples) go through gray_kernel and give
    the same masks.  The byte model counts no separate grey pass when the frames are aligned."""
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w = shape
    clip = Clip(h, w, 3, seed=13, n_objects=3)
    rng = np.random.default_rng(4)

    def frame(t):
        f = clip.frame(t)
        return f if cin == 3 else np.concatenate([f, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)

    frames = [frame(t) for t in range(18)]
    grays = [oracle.cvt_gray(np.ascontiguousarray(f[:, :, :3]), rgb_order=(order == "RGB2GRAY")) for f in frames]
    for blur, okw in (((3, 3, 4.0), dict(blur_k=3)), (("box", 3, 3), dict(blur_k=-3))):
        pipe = ForegroundPipeline((h, w, cin), convert=order, blur=blur, detector=detector)
        ref = oracle.Pipeline(h, w, 1, detector=detector, **okw)
        assert np.array_equal(pipe.apply(frames[0]), ref.apply(grays[0]))
        got = pipe.apply_many(np.stack(frames[1:9]))
        for i in range(8):
            assert np.array_equal(got[i], ref.apply(grays[1 + i])), (blur, i)
        assert np.array_equal(pipe.apply(frames[9]), ref.apply(grays[9]))
        got = pipe.apply_many(np.stack(frames[10:18]))
        for i in range(8):
            assert np.array_equal(got[i], ref.apply(grays[10 + i])), (blur, i)
        if detector == "KNN":
            a, b = pipe.knn_state(), ref.export()
            assert all(np.array_equal(a[k], b[k]) for k in ("samples", "state"))
        tot, det = pipe.algorithmic_bytes(1.0, 8)
        if (w * cin) % 4 == 0 and (h * w * cin) % 4 == 0 and w % 4 == 0 and detector == "FD":
            assert tot - det == 8 * 2 * h * w        # beyond the detector: the one morphology launch only


@pytest.mark.parametrize("offset", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(33, 71, 3), (40, 64, 3), (17, 36, 4), (5, 5, 3)])
def test_step0_on_device_buffers_of_any_alignment(oracle, offset, shape):
    """gof_cvt_gray_dev (four pixels per thread on dwords when both buffers are dword-aligned, bytewise
    otherwise, the last npix % 4 pixels bytewise) and the converting pipelines (step 0 inside the detector on
    aligned colour frames, its own kernel otherwise) on frames that start at byte `offset` of an allocation and
    end on its last byte: the oracle's grey image / masks, nothing written outside the output view."""
    import torch
    from pygofpid_amd import _lib
    from pygofpid_amd.foreground import ForegroundPipeline
    h, w, c = shape
    rng = np.random.default_rng(offset + h)
    lib, ctx, dev = _lib.load(), _lib.default_context(), torch.device("cuda", 0)
    for n_img in (1, 3):
        src = rng.integers(0, 256, (n_img, h, w, c), dtype=np.uint8)
        sbuf = torch.empty(offset + src.size, dtype=torch.uint8, device=dev)
        dbuf = torch.full((offset + n_img * h * w + 5,), 9, dtype=torch.uint8, device=dev)
        sbuf[offset:].copy_(torch.from_numpy(src.reshape(-1)).to(dev))
        for rgb in (0, 1):
            _lib.check(lib.gof_cvt_gray_dev(ctx.handle, sbuf[offset:].data_ptr(), dbuf[offset:].data_ptr(), n_img, h, w, c,
                                            rgb, None))
            ctx.synchronize()
            got = dbuf.cpu().numpy()
            assert (got[:offset] == 9).all() and (got[offset + n_img * h * w:] == 9).all()
            for i in range(n_img):
                want = oracle.cvt_gray(np.ascontiguousarray(src[i, :, :, :3]), rgb_order=bool(rgb))
                assert np.array_equal(got[offset + i * h * w:offset + (i + 1) * h * w].reshape(h, w), want)
    # the converting pipelines on the same kind of views, 1 frame and 3 frames per call
    T = 3
    clip = Clip(h, w, 3, seed=offset, n_objects=2)

    def frame(t):
        f = clip.frame(t)
        return f if c == 3 else np.concatenate([f, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    frames = np.stack([frame(t) for t in range(2 * T)])
    for det in ("MOG2", "FD", "KNN"):
        pipe = ForegroundPipeline((h, w, c), device=0, convert="BGR2GRAY", detector=det)
        ref = oracle.Pipeline(h, w, 1, detector=det)
        fbuf = torch.empty(offset + T * h * w * c, dtype=torch.uint8, device=dev)
        mbuf = torch.full((offset + T * h * w,), 7, dtype=torch.uint8, device=dev)
        fbuf[offset:].copy_(torch.from_numpy(frames[:T].reshape(-1)).to(dev))
        pipe.apply_device(fbuf[offset:].data_ptr(), mbuf[offset:].data_ptr(), n_frames=T)
        pipe.synchronize()
        got = mbuf[offset:].cpu().numpy().reshape(T, h, w)
        for t in range(T):
            assert np.array_equal(got[t], ref.apply(oracle.cvt_gray(np.ascontiguousarray(frames[t][:, :, :3])))), (det, t)
        assert (mbuf[:offset].cpu().numpy() == 7).all()
        fbuf[offset:].copy_(torch.from_numpy(frames[T:].reshape(-1)).to(dev))
        for t in range(T):
            pipe.apply_device(fbuf[offset + t * h * w * c:].data_ptr(), mbuf[offset + t * h * w:].data_ptr())
        pipe.synchronize()
        got = mbuf[offset:].cpu().numpy().reshape(T, h, w)
        for t in range(T):
            assert np.array_equal(got[t], ref.apply(oracle.cvt_gray(np.ascontiguousarray(frames[T + t][:, :, :3])))), (det, T + t)
