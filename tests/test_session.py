"""The interactive surface: the run() / draw() state machine of the reference (src/vk_engine.cpp:1817-1904,
1774-1815) and the "Update Buffer" buttons, headless. CPU: the frame counters and the camera follow the reference's
arithmetic (with a stand-in renderer). GPU: a scripted session — move, stop, accumulate, edit, accumulate — gives the frames the
oracle gives for the same sequence of push constants and scene edits."""
import numpy as np
import pytest

from ray_tracer_amd import engine
from ray_tracer_amd.session import InteractiveSession

from util import cornell_scene


class _FakeRenderer:
    """Records the dispatches; no GPU."""

    def __init__(self):
        self.calls = []

    def upload_scene(self, scene):
        pass

    def clear_framebuffer(self):
        pass

    def render(self, pc, W, H):
        self.calls.append((pc.frameCount, pc.rayTraceParams.progressive, tuple(pc.camInfo.pos), tuple(pc.camInfo.cameraRotation)))
        return np.zeros((H, W, 4), np.float32)


def test_frame_counters_follow_draw():
    r = _FakeRenderer()
    s = InteractiveSession(r, cornell_scene(False), 32, 16)
    assert s.params.progressive == 0 and s.params.singleRender == 0 and s.params.sampleLimit == 10
    for _ in range(3):                      # idle frames: auto-progressive switches accumulation on, frameCount runs on
        s.frame()
    assert [c[0] for c in r.calls] == [0, 1, 2] and [c[1] for c in r.calls] == [1, 1, 1]
    s.frame(keys="W", frame_time_ms=20.0)   # moving: accumulation off for this frame, the counter falls back to 0 after it
    assert r.calls[-1][:2] == (3, 0) and s._frameNumber == 0
    s.frame()
    assert r.calls[-1][:2] == (0, 1)
    # single render: one dispatch of sampleLimit samples, then nothing until the budget is raised (:1782,1812-1814)
    s.params.singleRender = 1
    n = len(r.calls)
    s.frame(); s.frame(); s.frame()
    assert len(r.calls) == n + 1 and s.totalSamples == s.params.sampleLimit
    s.params.singleRender = 0               # the budget is only reset at the end of a draw(): one frame without a dispatch first
    s.frame()
    assert len(r.calls) == n + 1 and s.totalSamples == 0
    s.frame()
    assert len(r.calls) == n + 2 and s.totalSamples == 0


def test_camera_follows_run():
    r = _FakeRenderer()
    s = InteractiveSession(r, cornell_scene(False), 32, 16)
    p0 = np.array(list(s.pc.camInfo.pos), np.float32)
    s.frame(keys="W", frame_time_ms=100.0)  # 100 ms * 0.001 * cameraSpeed 10 = one unit along the view direction
    p1 = np.array(list(s.pc.camInfo.pos), np.float32)
    d = p1 - p0
    assert abs(np.linalg.norm(d) - 1.0) < 1e-5
    M = np.array(list(s.pc.camInfo.cameraRotation), np.float32).reshape(4, 4).T
    assert np.allclose(d, M[:3, 2] / np.linalg.norm(M[:3, 2]), atol=1e-6)       # cameraRotation * (0, 0, 1, 0), normalised
    s.frame(keys="AD")                      # opposite keys cancel: no movement, accumulation stays on
    assert np.array_equal(np.array(list(s.pc.camInfo.pos), np.float32), p1) and s.params.progressive == 1
    a0 = list(s.cameraAngles)
    s.frame(gesture=(0.50, 0.50))           # first gesture event only sets the anchor (:1845-1848)
    assert s.cameraAngles == a0 and s.params.progressive == 0
    s.frame(gesture=(0.51, 0.48))
    assert abs(s.cameraAngles[0] - (a0[0] + (-0.02) * 100)) < 1e-4 and abs(s.cameraAngles[1] - (a0[1] - 0.01 * 100 * 1.6667)) < 1e-4
    s.frame(finger_up=True)
    assert s.prevMouseScroll == (0.0, 0.0) and s.params.progressive == 1


@pytest.mark.gpu
def test_scripted_session_against_the_oracle(renderer):
    from oracle import pyoracle
    renderer.set_tuning("pipeline", -1)
    s = InteractiveSession(renderer, cornell_scene(True), 96, 64)
    s.params.raysPerPixel = 2

    class _Scene:   # what the oracle renders: the session's edited arrays
        def arrays(self_inner):
            return s.arrays()

    prev = None

    def check(img):
        nonlocal prev
        ref, _ = pyoracle.render(_Scene(), s.pc, s.W, s.H, prev=prev if s.pc.rayTraceParams.progressive else None)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        prev = ref

    check(s.frame(keys="WD", frame_time_ms=30.0))    # moving: frame 0, no accumulation
    for _ in range(3):                               # standing still: frames 0, 1, 2 accumulate
        check(s.frame())
    check(s.frame(gesture=(0.3, 0.3)))
    check(s.frame(gesture=(0.33, 0.31)))             # the view turns: accumulation off
    m = s.material(0)
    m.reflectance = 1.0                              # the white walls become mirrors
    s.set_material(0)
    s.set_sphere(1, (0.4, 0.2, -0.1), 0.3, 5)
    s.set_object(0, placement=engine.placement(position=(-0.4, 0.25, -0.45), scale=0.3, rotation=(0, 20, 0)))
    for _ in range(2):
        check(s.frame(finger_up=True))
    assert s.dispatches == 8
