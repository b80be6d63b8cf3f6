"""Batch assembly on the device: LLFFDataset.__getitem__ (data/loader.py:325-356), interpolate_poses and
sample_events with the pose interpolation inside (data/loader_events.py:133-148, 259-304), get_rays / get_rays_pix without the
half-pixel offset (utils/rays.py:8-36) -- through the C ABI, against the reference's goldens G28 / G29 and the oracle at real sizes."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(x):
    return torch.as_tensor(np.ascontiguousarray(x), device=DEV)


def N(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


def test_image_batch_matches_golden_and_oracle(O):
    """golden G28: ids, colours, poses, pixel centres bit-exact, ray directions 1e-6 (measured 0: the kernel rounds like torch's unfused
    float32 ops); the oracle bit for bit on a blurfactory-sized dataset (29 x 400 x 600 images, 1024- and 65 536-ray batches); an empty
    batch; an id outside the dataset raises like the reference's indexing."""
    from evdeblurnerf_amd.loader import ImageBatcher
    from test_oracle_golden import check_image_batch
    g = load_golden("G28_image_batch")
    b = ImageBatcher(g["images"], g["poses"], g["K"])
    assert len(b) == g["images"][..., 0].size
    out = b[T(g["ids"])]
    assert "rgbsf_pts0" not in out
    b.set_pts0_prior(g["pts0"])
    out = {k: N(v) for k, v in b.__getitem__(g["ids"].tolist(), check=True).items()}
    check_image_batch(out, g, exact_rays=True)
    i, y, x = b.unravel_idx_from_rayid(T(g["ids"]))
    assert np.array_equal(N(i), g["out_images_idx"][:, 0]) and np.array_equal(N(x) + 0.5, g["out_rays_x"][:, 0]) and np.array_equal(N(y) + 0.5, g["out_rays_y"][:, 0])
    assert b[torch.empty((0,), dtype=torch.int64, device=DEV)]["rays"].shape == (0, 3, 2)
    with pytest.raises(IndexError):
        b.__getitem__([3, len(b)], check=True)
    # full size
    rs = np.random.RandomState(5)
    n_img, H, Wd = 29, 400, 600
    images = rs.uniform(0, 1, (n_img, H, Wd, 3)).astype(np.float32)
    pts0 = np.ascontiguousarray(images[::-1])
    poses = rs.standard_normal((n_img, 3, 4)).astype(np.float32)
    K = np.array([[433.3, 0, 300.0], [0, 433.3, 200.0], [0, 0, 1]], np.float32)
    big = ImageBatcher(images, poses, K)
    big.set_pts0_prior(pts0)
    for nq in (1024, 65536):
        ids = rs.randint(0, len(big), nq).astype(np.int64)
        got = {k: N(v) for k, v in big[T(ids)].items()}
        ref = O.image_batch(ids, images, poses, K, pts0_images=pts0)
        for k in got:
            assert np.array_equal(got[k], ref[k]), (nq, k)
    # the rays feed the renderer unchanged: same as get_rays_pix of the mirror on the same pixels
    from evdeblurnerf_amd.rays import get_rays_pix
    ids = rs.randint(0, len(big), 512).astype(np.int64)
    out = big[T(ids)]
    o, d = get_rays_pix(torch.cat([out["rays_x"], out["rays_y"]], -1) - 0.5, K, out["poses"])
    assert torch.equal(o, out["rays"][..., 0]) and torch.equal(d, out["rays"][..., 1])


def test_rays_without_the_half_pixel():
    """add_halfpix=False on both stand-alone entries (golden G28 from the reference's get_rays / get_rays_pix)"""
    from evdeblurnerf_amd.rays import get_rays, get_rays_pix
    g = load_golden("G28_image_batch")
    o, d = get_rays_pix(T(g["nohalf_coords"]), g["K"], T(g["nohalf_c2ws"]), add_halfpix=False)
    assert np.array_equal(N(o), g["nohalf_pix_o"]) and np.abs(N(d) - g["nohalf_pix_d"]).max() < 1e-6
    H, Wd = g["images"].shape[1:3]
    o, d = get_rays(H, Wd, g["K"], T(g["poses"][1]), add_halfpix=False)
    assert np.array_equal(N(o), g["nohalf_full_o"]) and np.abs(N(d) - g["nohalf_full_d"]).max() < 1e-6
    o1, d1 = get_rays_pix(T(g["nohalf_coords"]), g["K"], T(g["nohalf_c2ws"]))
    assert np.abs(N(d1) - g["nohalf_pix_d"]).max() > 1e-3


def _track(g, tag):
    from evdeblurnerf_amd.poses import PoseTrack
    rc = g[f"{tag}_recenter_c2w"] if f"{tag}_recenter_c2w" in g else None
    return PoseTrack(g[f"{tag}_key_t"], g[f"{tag}_key_poses"], bd_scale=float(g[f"{tag}_bd_scale"]), recenter=rc is not None, recenter_partial=rc)


def test_pose_track_matches_golden_and_oracle(O):
    """golden G29 (the reference's interpolate_poses / sample_events on scipy): poses within one float32 ulp, polarity sums and ids
    bit-exact, rays 1e-5; the oracle on a 3000-key track, 2 M timestamps: <= 1 float32 ulp, > 99.9 % of the entries identical;
    queries outside the key range clip to the end poses; the track's error behaviour."""
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.events import EventSampler
    from evdeblurnerf_amd.poses import PoseTrack
    from test_oracle_golden import check_pose_track
    g = load_golden("G29_pose_track")
    K = W.synthetic_camera()

    def sample(tag):
        smp = EventSampler(g[f"{tag}_events"], g[f"{tag}_coords"], K=K, pose_track=_track(g, tag), integer_coords=bool(g[f"{tag}_intc"]))
        return {k: N(v) for k, v in smp.sample_events(T(g[f"{tag}_ids"]), check=True).items() if v is not None}
    worst = check_pose_track(lambda tag, t: N(_track(g, tag).interpolate_poses(t)), sample, g)
    print(f"pose track vs the reference's interpolate_poses: max abs {worst:.2e}")
    # the multi-hop branch goes through the same kernel: equal to the per-event-table form fed with this track's own poses
    for tag in ("a", "b"):
        trk = _track(g, tag)
        ev = g[f"{tag}_events"]
        table = trk.interpolate_poses(ev[:, -3])[:, :3, :4].contiguous()
        kw = dict(K=K, integer_coords=bool(g[f"{tag}_intc"]))
        ids = T(g[f"{tag}_ids"])
        hops = torch.randint(0, 4, ids.shape, device=DEV)
        a = EventSampler(ev, g[f"{tag}_coords"], pose_track=trk, **kw).sample_events(ids, hops=hops)
        b = EventSampler(ev, g[f"{tag}_coords"], poses=table, **kw).sample_events(ids, hops=hops)
        for k in a:
            assert a[k] is None or torch.equal(a[k], b[k]), (tag, k)
    # full size against the oracle
    rs = np.random.RandomState(7)
    M, n = 3000, 2_000_000
    key_t = np.cumsum(rs.uniform(5e3, 4e4, M)) + 1.6e15 / 1e6
    from scipy.spatial.transform import Rotation as Rot      # test-side only: random rotations for the key poses
    Rk = Rot.from_rotvec(np.cumsum(rs.standard_normal((M, 3)) * 0.03, 0)).as_matrix()
    Tk = np.cumsum(rs.standard_normal((M, 3)) * 0.02, 0)
    kp = np.concatenate([Rk, Tk[..., None]], -1)
    c2w = np.concatenate([Rot.from_rotvec([0.1, -0.2, 0.05]).as_matrix(), np.array([[0.3], [-0.1], [0.2]])], -1)
    trk = PoseTrack(key_t, kp, bd_scale=0.61, recenter=True, recenter_partial=c2w)
    tq = rs.uniform(key_t[0] - 1e5, key_t[-1] + 1e5, n)
    got = N(trk.interpolate_poses(tq))
    ref = O.interpolate_poses(key_t, kp, tq, 0.61, c2w)
    d = np.abs(got - ref)
    scale = np.maximum(np.abs(ref), 1e-3)
    assert (d / scale).max() < 2.4e-7 and (d == 0).mean() > 0.999, ((d / scale).max(), (d == 0).mean())
    ends = N(trk.interpolate_poses(np.array([key_t[0] - 1e9, key_t[0], key_t[-1], key_t[-1] + 1e9])))
    assert np.array_equal(ends[0], ends[1]) and np.array_equal(ends[2], ends[3])
    assert trk.interpolate_poses(np.empty((0,))).shape == (0, 4, 4)
    with pytest.raises(L.EvdError):
        PoseTrack(key_t[:3], kp[:3], recenter=False)
    with pytest.raises(L.EvdError):
        PoseTrack(key_t[::-1].copy(), kp, recenter=False)
    with pytest.raises(NotImplementedError):
        PoseTrack(key_t, kp, recenter=False, spherify=True)


def test_sample_events_rejects_ids_outside_the_table():
    """an id outside [0, N) or an event without successor (-1) in the single-hop branch must not read outside the tables"""
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.events import EventSampler
    rs = np.random.RandomState(3)
    n = 64
    ev = np.stack([rs.randint(0, 5, n).astype(np.float64), np.sort(rs.uniform(0, 1e6, n)), rs.choice([-1.0, 1.0], n), np.full(n, -1.0)], -1)
    ev[:n - 1, -1] = np.arange(1, n)
    ev[:, 0] = 2                                                   # one pixel: every event's successor is its right neighbour
    coords = rs.uniform(0, 300, (5, 2)).astype(np.float32)
    poses = rs.standard_normal((n, 3, 4)).astype(np.float32)
    smp = EventSampler(ev, coords, poses, W.synthetic_camera())
    ok = smp.sample_events(T(np.array([0, 5, n - 2])), check=True)
    assert np.array_equal(N(ok["events_pos_pol_cumsum"]) + N(ok["events_neg_pol_cumsum"]), ev[[1, 6, n - 1], 2].astype(np.float32))
    for bad in ([n - 1], [-3], [n + 100]):                          # no successor; ids outside the table
        out = smp.sample_events(T(np.array([1] + bad)))
        assert float(out["events_pos_pol_cumsum"][1]) == 0 and float(out["events_neg_pol_cumsum"][1]) == 0
        assert int(smp._mismatch.item()) == 1
        with pytest.raises(L.EvdError):
            smp.sample_events(T(np.array([1] + bad)), check=True)
    # a coordinate id (column 0) outside the coordinate tables is rejected like an id outside the event table
    for bad_pix in (-1.0, 5.0, 1e9):
        ev2 = ev.copy()
        ev2[7, 0] = bad_pix
        smp2 = EventSampler(ev2, coords, poses, W.synthetic_camera(), id_to_color_map=np.eye(3, dtype=bool)[rs.randint(0, 3, 5)])
        out = smp2.sample_events(T(np.array([1, 7])))
        assert int(smp2._mismatch.item()) == 1 and int(out["events_coords_ids"][1]) == -1
        assert float(out["events_rays_start"][1].abs().max()) == 0 and not bool(out["events_color_map"][1].any())
        assert int(out["events_coords_ids"][0]) == 2 and float(out["events_rays_start"][0].abs().max()) > 0


# ------------------------------------------------------------------------------------------------ the once-per-dataset event tables
def _tables_from_golden(g, tag, **kw):
    from evdeblurnerf_amd.events import EventTables
    h, w = (int(v) for v in g[f"{tag}_hw"])
    acc = [int(v) for v in g[f"{tag}_acc"]]
    ev_map = (g["flt_inv_mapx"], g["flt_inv_mapy"]) if tag == "flt" else None
    return EventTables.from_arrays(g[f"{tag}_x"], g[f"{tag}_y"], g[f"{tag}_t"], g[f"{tag}_p"], h, w, g[f"{tag}_key_t"], g[f"{tag}_apb"], img_timestamps=g[f"{tag}_img_t"],
                                   ev_map=ev_map, color_events=True, events_tms_unit="us", events_tms_files_unit="us", event_accumulate_step_range=acc[:2],
                                   event_accumulate_step_range_end=acc[2:], recenter=False, **kw)


@pytest.mark.parametrize("tag", ["int", "flt"])
def test_event_tables_match_golden_G34(tag):
    """EventTables.from_arrays (evd_event_coord_ids / evd_event_filter / evd_event_color_map / evd_compute_successor) against golden G34 =
    the reference's LLFFEventsDataset.load_event_data run on the same arrays: every table bit for bit; the pose track built from
    all_poses_bounds reproduces events_pose_bspl (after the LLFF column change interpolate_poses applies, :137) to float32 rounding; and the
    ready EventSampler draws a batch whose start / end events are the table's."""
    from test_oracle_golden import check_event_tables
    g = load_golden("G34_event_tables")
    tb = _tables_from_golden(g, tag)
    got = {k: N(getattr(tb, k)) for k in ("events", "id_to_coords", "id_to_color_map", "events_num_successors", "events_with_successor_idx")}
    got["intcoords"] = tb.intcoords
    check_event_tables(got, g, tag)
    assert np.array_equal(tb.allknown_poses, g[f"{tag}_allknown_poses"])
    if tag == "int":
        c2i = N(tb.coords_to_id)
        assert np.array_equal(c2i, g["int_coords_to_id"])
    raw = g[f"{tag}_pose_bspl"]                                               # [n, 4, 4] float64, the raw interpolator (:175-182)
    ref = np.concatenate([raw[..., 1:2], -raw[..., 0:1], raw[..., 2:]], -1).astype(np.float32)
    ip = N(tb.interpolate_poses(g[f"{tag}_tq"]))
    assert np.abs(ip - ref).max() < 5e-6
    smp = tb.sampler(W.synthetic_camera())
    ids = tb.events_with_successor_idx[:32]
    out = smp.sample_events(ids, check=True) if tag == "int" else smp.sample_events(ids, hops=torch.zeros_like(ids), check=True)
    ev = g[f"{tag}_events"]
    assert np.array_equal(N(out["events_coords_ids"]), ev[N(ids), 0].astype(np.int64))
    assert np.array_equal(N(out["events_color_map"]).astype(np.uint8), g[f"{tag}_id_to_color_map"][ev[N(ids), 0].astype(np.int64)])


def test_event_tables_at_size_equal_the_oracle(O):
    """2 M events on a 260 x 346 sensor (the CDAVIS size), rectified float coordinates with an ev_map, polarities 0 / 1, a tenth of the
    stream outside the pose range: every table bit-equal to the oracle's array-level restatement of load_event_data."""
    from evdeblurnerf_amd.events import EventTables
    rs = np.random.RandomState(77)
    h, w, n = 260, 346, 2_000_000
    act = rs.rand(h, w) < 0.9
    ys, xs = np.where(act)
    pick = rs.randint(0, ys.shape[0], n)
    rect = lambda X, Y: ((X + 0.31 * np.sin(0.04 * Y) + 0.25).astype(np.float32), (Y + 0.27 * np.cos(0.03 * X) - 0.125).astype(np.float32))
    ex, ey = rect(xs[pick].astype(np.float32), ys[pick].astype(np.float32))
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    ev_map = rect(gx, gy)
    key_t = np.arange(20, dtype=np.float64) * 1e5 + 1e6
    et = np.sort(rs.randint(int(key_t[0]) - 100_000, int(key_t[-1]) + 100_000, n)).astype(np.int64)
    ep = rs.randint(0, 2, n)
    from scipy.spatial.transform import Rotation as Rot
    Rk = Rot.from_rotvec(np.cumsum(rs.standard_normal((20, 3)) * 0.03, 0)).as_matrix()
    apb = np.concatenate([np.concatenate([Rk, rs.standard_normal((20, 3, 1)), np.ones((20, 3, 1))], -1).reshape(20, 15), np.ones((20, 2))], -1)
    tb = EventTables.from_arrays(ex, ey, et, ep, h, w, key_t, apb, ev_map=ev_map, color_events=True, events_tms_unit="us", events_tms_files_unit="us", recenter=False)
    ref = O.event_tables(ex, ey, et, ep, h, w, float(key_t.min()), float(key_t.max()), ev_map=ev_map, color_events=True)
    assert np.array_equal(N(tb.events), ref["events"]) and np.array_equal(N(tb.id_to_coords), ref["id_to_coords"])
    assert np.array_equal(N(tb.noev_coord_ids), ref["noev_coord_ids"]) and np.array_equal(N(tb.id_to_color_map).astype(np.uint8), ref["id_to_color_map"])
    assert np.array_equal(N(tb.events_num_successors), ref["events_num_successors"]) and np.array_equal(N(tb.events_with_successor_idx), ref["events_with_successor_idx"])
    assert not tb.intcoords and 0.85 * n < tb.events.shape[0] < 0.95 * n


def test_event_tables_flag_what_the_reference_asserts():
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.events import EventTables
    g = load_golden("G34_event_tables")
    bad_p = g["int_p"].copy()
    bad_p[len(bad_p) // 2] = 3                                # a polarity outside {0, 1} / {-1, 1}, on an event inside the pose range
    h, w = (int(v) for v in g["int_hw"])
    with pytest.raises(L.EvdError):
        EventTables.from_arrays(g["int_x"], g["int_y"], g["int_t"], bad_p, h, w, g["int_key_t"], g["int_apb"], events_tms_unit="us", recenter=False)
    mx = g["flt_inv_mapx"] + np.float32(0.5)                  # no map entry equals an event coordinate any more
    h, w = (int(v) for v in g["flt_hw"])
    with pytest.raises(L.EvdError):
        EventTables.from_arrays(g["flt_x"], g["flt_y"], g["flt_t"], g["flt_p"], h, w, g["flt_key_t"], g["flt_apb"], ev_map=(mx, g["flt_inv_mapy"]), color_events=True,
                                events_tms_unit="us", recenter=False)
