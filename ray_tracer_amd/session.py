"""The reference's interactive surface without a window: the state machine of VulkanEngine::run() and draw()
(src/vk_engine.cpp:1817-1904, 1774-1815) and the three "Update Buffer" buttons of the ImGui panels (:1536-1618), driven by
calls instead of SDL events. One `frame()` is one iteration of run(): input -> camera -> draw() -> one dispatch through the
C ABI when the sample budget allows it -> the frame counters the reference keeps.

    s = InteractiveSession(renderer, scene, 960, 540)
    s.frame(keys="W", frame_time_ms=16.0)      # a frame with W held: the camera moves, progressive accumulation is off
    s.frame()                                  # keys released: auto-progressive switches accumulation on again
    s.params.raysPerPixel = 4                  # the "Ray Tracer Info" panel
    m = s.material(0); m.reflectance = 1.0; s.set_material(0, m)   # the material editor + its "Update Buffer" button
"""
import ctypes as C

import numpy as np

from . import _capi, engine


class InteractiveSession:
    def __init__(self, renderer, scene, width, height):
        self.r, self.scene, self.W, self.H = renderer, scene, int(width), int(height)
        self.pc = engine.push_constants(self.W, self.H)            # defaults of src/vk_engine.h:145-171
        self.params = self.pc.rayTraceParams
        self.cameraAngles = [4.0, 0.0, 0.0]                        # src/vk_engine.h:325
        self.mouseSensitivity, self.cameraSpeed, self.autoProgressive = 100.0, 10.0, True   # :333-336
        self.clicking = False
        self.prevMouseScroll = (0.0, 0.0)
        self._frameNumber, self.totalSamples = 0, 0                # :322-323
        self.frameTime = 0.0                                       # renderStats.frameTime, milliseconds (:1900-1902)
        self.image = None
        self.dispatches = 0
        renderer.upload_scene(scene)
        renderer.clear_framebuffer()
        self._edited = None
        self._rotation()                                           # cameraInfo.cameraRotation as run_compute leaves it

    # ---- run_compute's camera half (src/vk_engine.cpp:1631-1653)
    def _rotation(self):
        ang = (C.c_float * 3)(*[float(x) for x in self.cameraAngles])
        _capi.lib().rt_camera_rotation(ang, self.pc.camInfo.cameraRotation)

    # ---- one iteration of run() (:1821-1903)
    def frame(self, keys="", gesture=None, finger_up=False, frame_time_ms=None):
        """keys: any of "WASD" held this frame; gesture: (x, y) of a two-finger trackpad gesture (SDL_MULTIGESTURE), or None;
        finger_up: SDL_FINGERUP arrived; frame_time_ms: what the previous iteration took (renderStats.frameTime), default: the
        value this session measured last."""
        f32 = np.float32
        moving_mouse = False
        if gesture is not None and not self.clicking:
            gx, gy = float(gesture[0]), float(gesture[1])
            if self.prevMouseScroll == (0.0, 0.0):
                self.prevMouseScroll = (gx, gy)
            dy, dx = f32(gy) - f32(self.prevMouseScroll[1]), f32(gx) - f32(self.prevMouseScroll[0])
            self.cameraAngles[0] = float(f32(self.cameraAngles[0]) + dy * f32(self.mouseSensitivity))
            self.cameraAngles[1] = float(f32(self.cameraAngles[1]) + -dx * f32(self.mouseSensitivity) * f32(1.6667))
            self.prevMouseScroll = (gx, gy)
            moving_mouse = True
        elif finger_up:
            self.prevMouseScroll = (0.0, 0.0)
        move = np.zeros(3, np.float32)
        for k in keys.upper():
            if k == "W": move[2] += 1
            elif k == "S": move[2] -= 1
            elif k == "A": move[0] -= 1
            elif k == "D": move[0] += 1
        ft = self.frameTime if frame_time_ms is None else float(frame_time_ms)
        if move.any():
            # movement = normalize(cameraInfo.cameraRotation * vec4(movement, 0)); pos += movement * frameTime * 0.001 * cameraSpeed
            M = np.array(list(self.pc.camInfo.cameraRotation), np.float32).reshape(4, 4).T      # column-major
            v = (M[:3, 0] * move[0] + M[:3, 1] * move[1]) + M[:3, 2] * move[2]
            v = v * (f32(1) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=np.float32))
            step = f32(ft) * f32(0.001) * f32(self.cameraSpeed)
            for i in range(3):
                self.pc.camInfo.pos[i] = float(f32(self.pc.camInfo.pos[i]) + v[i] * step)
            self.params.progressive = 0
        elif self.autoProgressive:
            self.params.progressive = 0 if moving_mouse else 1
        return self.draw()

    # ---- draw() (:1774-1815)
    def draw(self):
        import time
        t0 = time.perf_counter()
        img = None
        if self.totalSamples < self.params.sampleLimit:
            self._rotation()
            self.pc.frameCount = self._frameNumber
            img = self.r.render(self.pc, self.W, self.H)            # run_compute: counts filled, one dispatch
            self.image = img
            self.dispatches += 1
        self._frameNumber = self._frameNumber + 1 if self.params.progressive else 0
        self.totalSamples += self.params.sampleLimit if self.totalSamples < self.params.sampleLimit else 0
        if not self.params.singleRender:
            self.totalSamples = 0
        self.frameTime = (time.perf_counter() - t0) * 1e3
        return img

    # ---- the editors and their "Update Buffer" buttons (:1536-1618)
    def _arrays(self):
        if self._edited is None:
            self._edited = _EditedArrays(self.scene)
        return self._edited

    def material(self, i):
        return self._arrays().materials[i]

    def set_material(self, i, m=None):
        a = self._arrays()
        if m is not None:
            C.memmove(C.byref(a.materials[i]), C.byref(m), C.sizeof(_capi.RayMaterial))
        self.r._check(self.r._l.rt_update_materials(self.r._h, a.materials, a.nMaterials), "rt_update_materials")

    def set_sphere(self, i, position, radius, materialIndex):
        a = self._arrays()
        a.spheres[i].position[:] = [float(x) for x in position]
        a.spheres[i].radius, a.spheres[i].materialIndex = float(radius), int(materialIndex)
        self.r._check(self.r._l.rt_update_spheres(self.r._h, a.spheres, a.nSpheres), "rt_update_spheres")

    def set_object(self, i, placement=None, materialIndex=None, samplerIndex=None):
        a = self._arrays()
        if placement is not None:
            _capi.lib().rt_transform_matrix(C.byref(placement), a.objects[i].transformMatrix)   # T*Rx*Ry*Rz*S (:1597-1601)
        if materialIndex is not None:
            a.objects[i].materialIndex = int(materialIndex)
        if samplerIndex is not None:
            a.objects[i].samplerIndex = int(samplerIndex)
        self.r._check(self.r._l.rt_update_objects(self.r._h, a.objects, a.nObjects), "rt_update_objects")

    def arrays(self):
        """The scene as edited so far (RtSceneArrays), e.g. for the oracle."""
        if self._edited is None:
            return self.scene.arrays()
        return self._edited.arrays()

    def counts(self):
        return self.scene.counts()


class _EditedArrays:
    """Host copies of the three arrays the panels edit."""

    def __init__(self, scene):
        self.scene = scene
        a = scene.arrays()
        self.nMaterials, self.nObjects, self.nSpheres = a.materialCount, a.objectCount, a.sphereCount
        self.materials = (_capi.RayMaterial * max(self.nMaterials, 1))()
        self.objects = (_capi.RenderObject * max(self.nObjects, 1))()
        self.spheres = (_capi.Sphere * max(self.nSpheres, 1))()
        C.memmove(self.materials, a.materials, C.sizeof(_capi.RayMaterial) * self.nMaterials)
        C.memmove(self.objects, a.objects, C.sizeof(_capi.RenderObject) * self.nObjects)
        C.memmove(self.spheres, a.spheres, C.sizeof(_capi.Sphere) * self.nSpheres)

    def arrays(self):
        a = self.scene.arrays()
        a.materials = C.cast(self.materials, C.POINTER(_capi.RayMaterial))
        a.objects = C.cast(self.objects, C.POINTER(_capi.RenderObject))
        a.spheres = C.cast(self.spheres, C.POINTER(_capi.Sphere))
        return a
