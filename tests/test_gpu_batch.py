"""-m gpu: mh_icp_linearize_batch — all live factors of the sliding window (src/graph/manager.cpp:585-588) in one
K3 + one K4 launch — against (a) the same factors linearized one call at a time (bit-identical) and (b) the oracle."""
import numpy as np
import pytest

from parity import assert_result_parity, assert_state_parity

pytestmark = pytest.mark.gpu


def _window(ctx, world, n_factors, sizes=None, binary=False):
    """n_factors scans of one room (different subsets / poses), each with a twin for the one-at-a-time run and an
    oracle factor."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu

    gm, rm = capi.VoxelMap(ctx), ref_cpu.Map()
    gm.insert(world["map_xyz"])
    rm.insert(world["map_xyz"])
    cfg, rcfg = capi.make_reg_config(**world["cfg"]), ref_cpu.make_config(**world["cfg"])
    pts = world["pts"]
    fa, fb, fr, poses = [], [], [], []
    for i in range(n_factors):
        sub = pts[i::n_factors] if sizes is None else pts[: sizes[i]]
        fa.append(capi.ICPFactor(ctx, gm, sub, cfg, binary=binary))
        fb.append(capi.ICPFactor(ctx, gm, sub, cfg, binary=binary))
        fr.append(ref_cpu.ICP(rm, sub, rcfg, binary=binary))
        dR = synth.so3_exp(np.array([0.001 * i, -0.0007 * i, 0.002 * i]))
        poses.append((world["R"] @ dR, world["t"] + np.array([0.01 * i, -0.004 * i, 0.002 * i])))
    return gm, fa, fb, fr, poses


def _same(a, b):
    for k in ("H_ss", "b_s", "f", "status_hist", "loc_trans_comp", "loc_rot_comp", "loc_trans_final", "loc_rot_final",
              "eigvec_rot", "eigvec_trans", "n_knn", "mean_candidates", "linearize_count", "H_st", "H_tt", "b_t"):
        assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k


def _same_up_to_summation_order(a, b):
    """A window batch picks K3's launch class from the batch's TOTAL (several lanes per point only while the machine is not full),
    a call of its own from the factor's size: when the two differ the per-point results are identical but the rows are summed
    in another order — last digits of the sums."""
    from parity import eigvec_equal_mod_sign, rel
    for k in ("status_hist", "n_knn", "mean_candidates", "linearize_count"):
        assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k
    for k in ("H_ss", "b_s", "loc_trans_final", "loc_rot_final", "loc_trans_comp", "loc_rot_comp"):
        assert rel(a[k], b[k]) <= 1e-12, k
    assert abs(a["f"] - b["f"]) <= 1e-12 * abs(b["f"])
    for k in ("eigvec_rot", "eigvec_trans"):
        assert eigvec_equal_mod_sign(a[k], b[k], tol=1e-9), k


def test_batch_equals_separate_calls_and_oracle(ctx, room_world):
    from mimosa_amd import capi

    gm, fa, fb, fr, poses = _window(ctx, room_world, 5)
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    got = capi.linearize_batch(fa, Rs, ts)                       # cold: every point of every factor runs k-NN
    for i in range(5):
        one = fb[i].linearize(Rs[i], ts[i])
        # (5 x 13 107 points: the window runs one lane per point, a 13 107-point factor on its own two — same per-point results)
        _same_up_to_summation_order(got[i], one)
        assert_result_parity(got[i], fr[i].linearize(Rs[i], ts[i]))
        assert_state_parity(fa[i].state(), fr[i].state())
        assert np.array_equal(fa[i].state()[0], fb[i].state()[0])
    # the update iterations: re-linearize the whole window at slightly moved poses (DA cache partly hit)
    for it in range(3):
        ts = [t + np.array([0.012, -0.009, 0.003]) for t in ts]
        got = capi.linearize_batch(fa, Rs, ts)
        for i in range(5):
            _same_up_to_summation_order(got[i], fb[i].linearize(Rs[i], ts[i]))
            for x, y in zip(fa[i].state(), fb[i].state()):
                assert np.array_equal(x, y, equal_nan=True)
            assert_result_parity(got[i], fr[i].linearize(Rs[i], ts[i]))
    for f in fa + fb:
        f.destroy()
    gm.release()


def test_batch_ragged_sizes(ctx, small_world):
    """Factors of very different sizes, an EMPTY factor, a single-factor batch, and one cloud past the 256/512-thread
    boundary mixed with small ones."""
    from mimosa_amd import capi

    n = len(small_world["pts"])
    gm, fa, fb, fr, poses = _window(ctx, small_world, 4, sizes=[n, 1, 0, 77])
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    got = capi.linearize_batch(fa, Rs, ts)
    for i in range(4):
        _same(got[i], fb[i].linearize(Rs[i], ts[i]))
    assert got[2]["n_knn"] == 0 and not got[2]["H_ss"].any()
    one = capi.linearize_batch(fa[:1], Rs[:1], ts[:1])[0]
    _same(one, fb[0].linearize(Rs[0], ts[0]))
    with pytest.raises(capi.MhError):
        capi.linearize_batch([fa[0], fa[0]], Rs[:2], ts[:2])    # the same factor twice
    for f in fa + fb:
        f.destroy()
    gm.release()


def test_batch_binary_factors(ctx, small_world):
    from mimosa_amd import capi, synth

    gm, fa, fb, fr, poses = _window(ctx, small_world, 3, binary=True)
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    Rt = [synth.so3_exp(np.array([0.002, 0.001, -0.003])) for _ in range(3)]
    tt = [np.array([0.02, -0.01, 0.005]) for _ in range(3)]
    Rs2 = [Rt[i] @ Rs[i] for i in range(3)]
    ts2 = [Rt[i] @ ts[i] + tt[i] for i in range(3)]
    got = capi.linearize_batch(fa, Rs2, ts2, R_tgts=Rt, t_tgts=tt)
    for i in range(3):
        _same(got[i], fb[i].linearize(Rs2[i], ts2[i], R_tgt=Rt[i], t_tgt=tt[i]))
        assert_result_parity(got[i], fr[i].linearize(Rs2[i], ts2[i], R_tgt=Rt[i], t_tgt=tt[i]), binary=True)
    for f in fa + fb:
        f.destroy()
    gm.release()


def test_batch_mixed_block_size(ctx, room_world):
    """One 65 536-pt cloud + a 70 000-pt... the room scan is 65 536 points: a batch whose largest member needs
    512-thread workgroups runs every member at 512 and still equals the separate calls."""
    from mimosa_amd import capi

    pts = room_world["pts"]
    big = np.concatenate([pts, pts[:4000]])  # 69 536 points
    w = dict(room_world, pts=big)
    gm, fa, fb, fr, poses = _window(ctx, w, 2, sizes=[len(big), 9000])
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    got = capi.linearize_batch(fa, Rs, ts)
    for i in range(2):
        # (78 536 points in the window: the 9 000-point member runs one lane per point here and two on its own)
        (_same if i == 0 else _same_up_to_summation_order)(got[i], fb[i].linearize(Rs[i], ts[i]))
        assert_result_parity(got[i], fr[i].linearize(Rs[i], ts[i]))
    for f in fa + fb:
        f.destroy()
    gm.release()


def test_batch_larger_than_the_inline_window(ctx, small_world):
    """More than 8 factors: the argument blocks go through the staged device copy instead of the kernel-argument segment
    (<= 8 is the other tests' path); both equal the separate calls bit for bit."""
    from mimosa_amd import capi

    n = 11
    gm, fa, fb, fr, poses = _window(ctx, small_world, n)
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    got = capi.linearize_batch(fa, Rs, ts)
    for i in range(n):
        # (the staged launch form runs one lane per point; a 431-point factor on its own runs several: the same points, the
        # same per-point results — the state below — summed in another order)
        _same_up_to_summation_order(got[i], fb[i].linearize(Rs[i], ts[i]))
        for x, y in zip(fa[i].state(), fb[i].state()):
            assert np.array_equal(x, y, equal_nan=True)
    got8 = capi.linearize_batch(fa[:8], Rs[:8], ts[:8])          # exactly the inline capacity, re-linearization
    for i in range(8):
        _same(got8[i], fb[i].linearize(Rs[i], ts[i]))
        fr[i].linearize(Rs[i], ts[i])                             # the oracle through the same call sequence: cold, then warm
        assert_result_parity(got8[i], fr[i].linearize(Rs[i], ts[i]))
    for f in fa + fb:
        f.destroy()
    gm.release()


_CORE = ("H_ss", "b_s", "f", "loc_trans_final", "loc_rot_final", "eigvec_rot", "eigvec_trans", "degen_rot", "degen_trans",
         "degen_eigvec_rot", "degen_eigvec_trans", "n_knn", "mean_candidates", "linearize_count", "H_st", "H_tt", "b_t")


def _core_same(a, b):
    """bit-identical, except the eigenvectors of H_rr / H_tt: with the component pass ON the library reports the bases K4
    projected on (derived on the device), with it OFF the host epilogue's decomposition of the same sums — the same
    eigenvectors to rounding (modulo sign), not necessarily to the bit"""
    from parity import eigvec_equal_mod_sign
    for k in _CORE:
        if k in ("eigvec_rot", "eigvec_trans"):
            assert eigvec_equal_mod_sign(a[k], b[k], tol=1e-9), k
        else:
            assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k


def test_components_switched_off_changes_nothing_else(ctx, room_world):
    """mh_icp_set_components(0): K4 is skipped; H, b, f, final localizabilities, eigenvectors, degeneracy info and the
    per-point state are bit-identical, the components come back NaN / -1, and switching it on again restores them —
    single calls (synchronous and pipelined), a clone, and the batch (all off = no K4 launch, mixed = NaN for the off ones)."""
    from mimosa_amd import capi

    gm, fa, fb, fr, poses = _window(ctx, room_world, 4)
    Rs, ts = [p[0] for p in poses], [p[1] for p in poses]
    for f in fa:
        f.set_components(False)
    for i in range(4):                                    # cold, then a re-linearization, one call at a time
        for R, t in ((Rs[i], ts[i]), (Rs[i], ts[i] + np.array([0.003, 0.0, -0.002]))):
            off, on = fa[i].linearize(R, t), fb[i].linearize(R, t)
            _core_same(off, on)
            assert np.all(np.isnan(off["loc_trans_comp"])) and np.all(np.isnan(off["loc_rot_comp"]))
            assert list(off["status_hist"]) == [-1] * 9
            assert not np.any(np.isnan(on["loc_trans_comp"])) and sum(on["status_hist"]) == len(fa[i].state()[0])
        sa, sb = fa[i].state(), fb[i].state()
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y, equal_nan=True)
    # pipelined calls without components (no flag, the wait synchronises the stream)
    outs, want = [], []
    for j in range(3):
        fb[0].reset()
        want.append(fb[0].linearize(Rs[0], ts[0] + np.array([0.001 * j, 0, 0])))
    for j in range(3):
        fa[0].reset()
        outs.append(fa[0].linearize_async(Rs[0], ts[0] + np.array([0.001 * j, 0, 0])))
    fa[0].wait()
    for a, b in zip(outs, want):
        a = a.as_dict()
        for k in ("H_ss", "b_s", "f", "loc_trans_final"):
            assert np.array_equal(a[k], b[k]), k
        assert np.all(np.isnan(a["loc_trans_comp"]))
    # a clone inherits the switch; switching it on brings the components back
    c_off, c_on, d = fa[1].clone(), fa[1].clone(), fb[1].clone()
    c_on.set_components(True)
    assert np.all(np.isnan(c_off.linearize(Rs[1], ts[1])["loc_rot_comp"]))
    _same(c_on.linearize(Rs[1], ts[1]), d.linearize(Rs[1], ts[1]))
    for f in (c_off, c_on, d):
        f.destroy()
    # batch: all off
    for f in fa + fb:
        f.reset()
    got, want = capi.linearize_batch(fa, Rs, ts), capi.linearize_batch(fb, Rs, ts)
    for i in range(4):
        _core_same(got[i], want[i])
        assert np.all(np.isnan(got[i]["loc_trans_comp"])) and list(got[i]["status_hist"]) == [-1] * 9
    # batch: mixed — K4 runs, the factors that asked for it get their components, the others NaN
    fa[2].set_components(True)
    for f in fa + fb:
        f.reset()
    got, want = capi.linearize_batch(fa, Rs, ts), capi.linearize_batch(fb, Rs, ts)
    _same(got[2], want[2])
    for i in (0, 1, 3):
        _core_same(got[i], want[i])
        assert np.all(np.isnan(got[i]["loc_rot_comp"]))
    for f in fa + fb:
        f.destroy()
    gm.release()


def test_batch_of_unlike_factors(ctx, small_world):
    """Any mix in one call: k = 5 and k = 4, neighbour modes 19 and 7 (two maps), unary and binary — one launch group per
    kernel instantiation, every factor still bit-identical to its own mh_icp_linearize and equal to the oracle."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu

    w = small_world
    pts = w["pts"]
    specs = [dict(k=5, mode=19, binary=False), dict(k=4, mode=19, binary=False), dict(k=5, mode=7, binary=False), dict(k=5, mode=19, binary=True),
             dict(k=8, mode=7, binary=True), dict(k=5, mode=19, binary=False)]
    maps = {}
    for mode in (19, 7):
        gm, rm = capi.VoxelMap(ctx, mode=mode), ref_cpu.Map(mode=mode)
        gm.insert(w["map_xyz"])
        rm.insert(w["map_xyz"])
        maps[mode] = (gm, rm)
    Rt, tt = synth.so3_exp(np.array([0.002, 0.001, -0.003])), np.array([0.02, -0.01, 0.005])
    fa, fb, fr, Rs, ts = [], [], [], [], []
    for i, sp in enumerate(specs):
        cfgd = dict(w["cfg"], num_corres_points=sp["k"])
        gm, rm = maps[sp["mode"]]
        sub = pts[i::len(specs)]
        fa.append(capi.ICPFactor(ctx, gm, sub, capi.make_reg_config(**cfgd), binary=sp["binary"]))
        fb.append(capi.ICPFactor(ctx, gm, sub, capi.make_reg_config(**cfgd), binary=sp["binary"]))
        fr.append(ref_cpu.ICP(rm, sub, ref_cpu.make_config(**cfgd), binary=sp["binary"]))
        R = w["R"] @ synth.so3_exp(np.array([0.001 * i, -0.0007 * i, 0.002 * i]))
        t = w["t"] + np.array([0.01 * i, -0.004 * i, 0.002 * i])
        Rs.append(Rt @ R)   # every factor gets the target pose; the unary ones ignore it
        ts.append(Rt @ t + tt)
    for rnd in range(2):   # cold, then through the data-association caches
        got = capi.linearize_batch(fa, Rs, ts, R_tgts=[Rt] * len(specs), t_tgts=[tt] * len(specs))
        for i, sp in enumerate(specs):
            kw = dict(R_tgt=Rt, t_tgt=tt) if sp["binary"] else {}
            _same(got[i], fb[i].linearize(Rs[i], ts[i], **kw))
            assert_result_parity(got[i], fr[i].linearize(Rs[i], ts[i], **kw), binary=sp["binary"])
    for f in fa + fb:
        f.destroy()
    for gm, _ in maps.values():
        gm.release()
