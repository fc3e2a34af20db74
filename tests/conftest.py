import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def ctx():
    """One libmimosa_hip context on device 0.  GPU tests fail loudly (no skip, no fallback) when
    the extension cannot be loaded or there is no device."""
    from mimosa_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def small_world():
    from mimosa_amd import synth

    m, pts, aux = synth.small_world()
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    return dict(map_xyz=m, pts=pts, aux=aux, R=R, t=t, cfg=synth.enwide_config())


@pytest.fixture(scope="session")
def room_world():
    """config[0]-like: one full room (~0.5 M-pt map) and a 64-row scan (65 536 pts)."""
    from mimosa_amd import synth

    m = synth.make_room(synth.BASE_SEED, 0, 0)
    pts, aux = synth.make_scan(64)
    R, t = synth.query_pose()
    return dict(map_xyz=m, pts=pts, aux=aux, R=R, t=t, cfg=synth.enwide_config())


@pytest.fixture(scope="session")
def big_world():
    """BASELINE configs[1]: the full 128 x 1024 = 131 072-pt scan against the 2 x 5-room map (~4.96 M stored points)."""
    from mimosa_amd import synth

    rooms = [xyz for _, _, xyz in synth.make_map_rooms(2, 5)]
    pts, aux = synth.make_scan(128)
    R, t = synth.query_pose()
    return dict(map_rooms=rooms, pts=pts, aux=aux, R=R, t=t, cfg=synth.enwide_config())


@pytest.fixture(scope="session")
def huge_world():
    """BASELINE configs[2]: the 131 072-pt scan against the 10 x 10-room map (~50 M stored points, ~5.2 M voxels)."""
    from mimosa_amd import synth

    rooms = [xyz for _, _, xyz in synth.make_map_rooms(10, 10)]
    pts, aux = synth.make_scan(128)
    R, t = synth.query_pose()
    return dict(map_rooms=rooms, pts=pts, aux=aux, R=R, t=t, cfg=synth.enwide_config())
