"""Cross-stream ordering of the image pyramids.  dsopp_hip_pyramid_build_device only ENQUEUES work on the pyramid's stream;
the aligner, the window and the depth estimator read the texels on their own streams and order themselves behind the build
through the pyramid's `ready` event (pyramid.hpp).  The test rebuilds a pyramid with a DIFFERENT image and consumes it
immediately, with no host synchronisation in between: the result must be the one of the new image."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def test_build_device_then_align_immediately():
    import torch
    from dsopp_amd import capi
    W, H, L = 1280, 1024, 5
    win = syn.make_window(num_frames=3, num_points=30, width=W, height=H, seed=13)
    fr, fa, fb = win.frames
    pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
    pr.build(fr.image_u8)
    imgs = {k: torch.from_numpy(f.image_u8.copy()).cuda() for k, f in (("a", fa), ("b", fb))}
    torch.cuda.synchronize()
    intr = win.scene.intrinsics
    rng = np.random.default_rng(2)
    n = 3000
    xs, ys = rng.integers(8, W - 8, n), rng.integers(8, H - 8, n)
    idsum, wgt = np.zeros((H, W)), np.zeros((H, W))
    idsum[ys, xs] = 1.0 / fr.depth[ys, xs]
    wgt[ys, xs] = 1.0
    T_ref = syn.mat_to_params(fr.T_w_c_gt)

    def solve(target, sync_first):
        a = capi.HipAligner(capi.default_align_options())
        a.set_lm_path(1)
        a.reset()
        a.push_reference_depth_map(1000, T_ref, pr, 0, intr, idsum, wgt, 1.0, np.zeros(2))
        pt.build_device(imgs[target].data_ptr())       # enqueue only
        if sync_first:
            torch.cuda.synchronize()
        a.push_target(2000, syn.mat_to_params({"a": fa, "b": fb}[target].T_w_c_init), pt, 0, intr, 1.0, np.zeros(2))
        r = a.solve()
        a.close()
        return r

    ref = {k: solve(k, True) for k in ("a", "b")}
    assert np.abs(ref["a"]["T_w_target"] - ref["b"]["T_w_target"]).max() > 1e-3   # the two images give different answers
    for _ in range(5):                        # alternate the image, consume without synchronising
        for k in ("a", "b"):
            r = solve(k, False)
            assert r["iterations"] == ref[k]["iterations"] and r["n_valid"] == ref[k]["n_valid"]
            assert r["energy"] == ref[k]["energy"]
            assert np.array_equal(r["T_w_target"], ref[k]["T_w_target"])
    pr.close()
    pt.close()


def test_build_device_then_push_frame_immediately():
    """the window reads the texels on ITS stream: a frame pushed right behind build_device must see the finished image"""
    import torch
    from dsopp_amd import capi
    W, H = 640, 480
    win = syn.make_window(num_frames=3, num_points=600, width=W, height=H, seed=17)
    intr = win.scene.intrinsics
    devs = [torch.from_numpy(f.image_u8.copy()).cuda() for f in win.frames]
    torch.cuda.synchronize()

    def run(sync_first):
        g = capi.HipWindow(capi.default_pba_options())
        pyrs = []
        for i, f in enumerate(win.frames):
            p = capi.Pyramid(W, H, 1)
            p.build_device(devs[i].data_ptr())
            if sync_first:
                torch.cuda.synchronize()
            pyrs.append(p)
            g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_init), 1.0, np.zeros(2), i == 0, False, pyramid=p)
            g.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
            for j in range(i):
                o = win.frames[j]
                g.set_connection(o.frame_id, f.frame_id, np.zeros(len(o.uv), dtype=np.uint8))
                g.set_connection(f.frame_id, o.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
        g.begin()
        e = g.calculate_energy()
        g.close()
        for p in pyrs:
            p.close()
        return e

    assert run(False) == run(True)


def test_pyramid_rewritten_while_a_window_borrows_it():
    """A pyramid a keyframe was pushed with gets a new mask afterwards (dsopp_hip_pyramid_set_mask).  The linearising sweeps read the
    texels, the residual-only sweeps the intensity plane derived from them: both must see the rewrite (round-3 advisor finding: the
    plane was only rebuilt at the next topology change) — the window then agrees with one that was given the mask from the start."""
    import numpy as np
    from dsopp_amd import capi, synthetic as syn
    win = syn.make_window(num_frames=3, num_points=210, width=320, height=240, seed=19)
    H, W = win.frames[0].pixelinfo.shape[:2]
    rng = np.random.default_rng(2)
    mask = (rng.uniform(size=(H, W)) > 0.15).astype(np.uint8) * 255
    intr = win.scene.intrinsics

    def build(with_mask_from_start):
        g = capi.HipWindow(capi.default_pba_options())
        pyr = []
        for i, f in enumerate(win.frames):
            p = capi.Pyramid(W, H, 1)
            p.set_level(0, f.pixelinfo)
            if with_mask_from_start and i == 1:
                p.set_mask(0, mask)
            pyr.append(p)
            g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False, pyramid=p)
            g.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
            for j in range(i):
                h = win.frames[j]
                for (r, t) in ((h, f), (f, h)):
                    g.set_connection(r.frame_id, t.frame_id, np.zeros(len(r.uv), dtype=np.uint8))
        return g, pyr

    ga, pa = build(True)
    gb, pb = build(False)
    gb.begin()
    e_before, n_before = gb.calculate_energy()          # the unmasked state: sweeps have run, the intensity plane exists
    pb[1].set_mask(0, mask)                              # ... and now the borrowed pyramid changes under the window
    ga.begin()
    gb.begin()
    ea, na = ga.calculate_energy()
    eb, nb = gb.calculate_energy()
    assert nb == na and nb < n_before, (na, nb, n_before)
    assert abs(ea - eb) <= 1e-12 * abs(ea)
    ga.linearize()
    gb.linearize()
    for x, y in zip(ga.get_system(), gb.get_system()):
        assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(x).max())
    ga.close()
    gb.close()
    for p in pa + pb:
        p.close()


def test_windows_solved_concurrently_from_their_own_threads():
    """DESIGN.md section 5, "One GPU, many windows": the throughput form is one host thread per window, each on a stream of its own.  The
    library holds no cross-window state that a concurrent solve could trample on (staging buffers, event pools, the solve launch's
    ticket counters and fault word are per window; error text is per thread): four DIFFERENT windows, each solved three times by its
    own thread while the others run, end bitwise where the same windows end when solved one after the other (deterministic build),
    and — every thread reading its own window back — with the oracle-checked single-window results of the other tests."""
    import threading
    import torch
    from dsopp_amd import capi
    wins = [syn.make_window(num_frames=4 + i, num_points=400 + 150 * i, width=320, height=240, seed=40 + i) for i in range(4)]

    def make(i, stream):
        g = capi.HipWindow(capi.default_pba_options(), stream=stream)
        g.set_deterministic(True)
        syn.load_window(g, wins[i])
        g.snapshot()
        return g

    def result(g):
        out = []
        for _ in range(3):
            g.restore()
            e, it, nv = g.optimize()
            out.append((e, it, nv, np.concatenate([g.get_pose(f.frame_id)[0] for f in wins[gs.index(g)].frames])))
        return out

    streams = [torch.cuda.Stream() for _ in range(4)]
    gs = [make(i, streams[i].cuda_stream) for i in range(4)]
    sequential = [result(g) for g in gs]
    concurrent = [None] * 4
    errors = []
    barrier = threading.Barrier(4)

    def worker(i):
        try:
            barrier.wait()
            concurrent[i] = result(gs[i])
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(4):
        for (e0, it0, nv0, p0), (e1, it1, nv1, p1) in zip(sequential[i], concurrent[i]):
            assert it0 == it1 and nv0 == nv1 and e0 == e1
            assert np.array_equal(p0, p1)
        assert sequential[i][0][1] >= 1
    for g in gs:
        g.close()
