"""Headless host-side mirror of the reference's application classes for the path-tracing hot path.

Same member names, defaults and call order as the reference so that its call sites read the same:
    PathTracer  <- PathTracer/PathTracer.h:58-135, PathTracer.cpp (LoadScene :115-149, InitRtScene :171-339,
                   RenderRT :659-787, UpdateOutputTexture :834-878, RestartAccumulation :912-915, Update :557-571)
    Sky         <- PathTracer/Sky.h:68-99, Sky.cpp (SetupEarthAtmosphere :32-56, ComputeTranmittanceLut :127-148)
    Camera      <- FANCY Common/Camera.h (absent; semantics inferred from its use at PathTracer.cpp:134-148,686-687,
                   747-750 — SURVEY §8c item 3, "parity unpinned")
Everything below the class surface goes through the C-ABI (context.Context); no window, UI or raster path.
"""
import math

import numpy as np

from . import api
from .context import Context, SceneData


class Camera:
    """Perspective camera, left-handed, +Z forward for the identity orientation (PathTracer.cpp:134-148)."""

    def __init__(self):
        self.myPosition = np.zeros(3, dtype=np.float32)
        # rows: right, up, forward in world space (identity == reference's quat_cast(lookAt((0,0,10),0,+Y)))
        self.myOrientation = np.eye(3, dtype=np.float32)
        self.myFovDeg = 60.0
        self.myNear = 1.0
        self.myFar = 10000.0
        self.myWidth = 1280.0
        self.myHeight = 720.0
        self.myIsOrtho = False

    def LookAt(self, eye, target, up=(0.0, 1.0, 0.0)):
        eye = np.asarray(eye, dtype=np.float64)
        f = np.asarray(target, dtype=np.float64) - eye
        f /= np.linalg.norm(f)
        r = np.cross(np.asarray(up, dtype=np.float64), f)
        r /= np.linalg.norm(r)
        u = np.cross(f, r)
        self.myPosition = eye.astype(np.float32)
        self.myOrientation = np.stack([r, u, f]).astype(np.float32)

    def GetViewProj(self):
        """Camera::myViewProj as PathTracer::CameraHasChanged reads it (PathTracer.cpp:446-453). FANCY's UpdateView /
        UpdateProjection are absent; adopted definition: view = world -> (right, up, forward) camera axes, projection =
        left-handed D3D perspective (vertical FOV, aspect, near, far), viewProj = proj * view, column vectors."""
        R = self.myOrientation.astype(np.float64)
        view = np.eye(4)
        view[:3, :3] = R
        view[:3, 3] = -R @ self.myPosition.astype(np.float64)
        f = 1.0 / math.tan(math.radians(self.myFovDeg) * 0.5)
        n, fr = float(self.myNear), float(self.myFar)
        proj = np.zeros((4, 4))
        proj[0, 0] = f / (self.myWidth / self.myHeight)
        proj[1, 1] = f
        proj[2, 2] = fr / (fr - n)
        proj[2, 3] = -n * fr / (fr - n)
        proj[3, 2] = 1.0
        return (proj @ view).astype(np.float32)

    def GetVerticesOnNearPlane(self):
        """v0 bottom-left, v1 bottom-right, v2 top-right, v3 top-left of the near rectangle (vertical FOV)."""
        right, up, fwd = (self.myOrientation[i].astype(np.float32) for i in range(3))
        half_h = np.float32(self.myNear * math.tan(math.radians(self.myFovDeg) * 0.5))
        half_w = np.float32(half_h * np.float32(self.myWidth / self.myHeight))
        c = self.myPosition.astype(np.float32) + fwd * np.float32(self.myNear)
        return [c - right * half_w - up * half_h, c + right * half_w - up * half_h,
                c + right * half_w + up * half_h, c - right * half_w + up * half_h]


def SetupEarthAtmosphere():
    """Priv_Sky::SetupEarthAtmosphere — PathTracer/Sky.cpp:32-56 (metres, despite the "km" comments)."""
    p = api.PtAtmosphere()
    f = np.float32
    EarthBottomRadius, EarthTopRadius = f(6360000.0), f(6460000.0)
    EarthRayleighScaleHeight, EarthMieScaleHeight = f(8000.0), f(1200.0)
    api.fill3(p.RayleighScattering, (0.000005802, 0.000013558, 0.000033100))
    p.RayleighDensityExpScale = float(f(-1.0) / EarthRayleighScaleHeight)
    api.fill3(p.AbsorptionExtinction, (0.000000650, 0.000001881, 0.0000000085))
    p.BottomRadius = float(EarthBottomRadius)
    api.fill3(p.GroundAlbedo, (0.0, 0.0, 0.0))
    p.TopRadius = float(EarthTopRadius)
    api.fill3(p.MieScattering, (0.000003996,) * 3)
    p.MieDensityExpScale = float(f(-1.0) / EarthMieScaleHeight)
    api.fill3(p.MieExtinction, (0.000004440,) * 3)
    p.MiePhaseG = 0.8
    api.fill3(p.MieAbsorption, [max(0.0, float(f(p.MieExtinction[i]) - f(p.MieScattering[i]))) for i in range(3)])
    p.AbsorptionDensity0LayerWidth = 25000.0
    p.AbsorptionDensity0ConstantTerm = float(f(-2.0) / f(3.0))
    p.AbsorptionDensity0LinearTerm = float(f(1.0) / f(15.0))
    p.AbsorptionDensity1ConstantTerm = float(f(8.0) / f(3.0))
    p.AbsorptionDensity1LinearTerm = float(f(-1.0) / f(15.0))
    return p


class Sky:
    """Sky — PathTracer/Sky.h:68-99. Public fields keep the reference's names and defaults (:80-83)."""

    def __init__(self, ctx):
        self._ctx = ctx
        self.myMultiScatteringFactor = 0.0
        self.mySunDir = np.array([0.0, 1.0, 0.0], dtype=np.float32)
        self.mySunIlluminance = np.array([1000.0, 1000.0, 1000.0], dtype=np.float32)
        self.myAtmosphereParams = SetupEarthAtmosphere()

    def ComputeTranmittanceLut(self):
        """Sky::ComputeTranmittanceLut — Sky.cpp:127-148 (the reference re-runs it every frame, PathTracer.cpp:580;
        here it is called when the atmosphere parameters change)."""
        self._ctx.sky_set(self.myAtmosphereParams)


class PathTracer:
    """PathTracer — PathTracer/PathTracer.h:58-135, hot-path members only."""

    def __init__(self, device=0, window_width=1280, window_height=720, ctx=None):
        # ctx: an existing context.Context (or, in the CPU tests of the accumulation control, a recorder with its methods)
        self.ctx = ctx if ctx is not None else Context(device)
        self.myWindowWidth, self.myWindowHeight = int(window_width), int(window_height)
        # PathTracer.h:112-131
        self.myAoDistance = 1.0
        self.myNumAccumulationFrames = 0
        self.myAccumulationNeedsClear = True
        self.myRenderRaster = False
        self.myRenderAo = False
        self.myAccumulate = True
        self.myHalfResRender = True
        self.mySampleSky = True
        self.mySkyFallbackIntensity = 100.0
        self.myMaxRecursionDepth = 4
        self.myLightInstanceIdx = 4
        self.myLightEnabled = True
        self.myLightColor = np.array([1.0, 1.0, 1.0], dtype=np.float32)
        self.myLightStrength = 100.0
        self.myPhongSpecularPower = 10.0
        self.myCamera = Camera()
        self.myCamera.myWidth, self.myCamera.myHeight = float(window_width), float(window_height)
        self.myScene = None
        self.mySky = None
        self.myOutputWidth = self.myOutputHeight = 0
        self._sky_dirty = True
        self.UpdateOutputTexture()
        self.InitSky()
        self.myLastViewMat = self.myCamera.GetViewProj()       # PathTracer.h:115
        self._last_sky = self._sky_settings()

    # ---- PathTracer.cpp:151-154 ----
    def InitSky(self):
        self.mySky = Sky(self.ctx)
        self._sky_dirty = True

    # ---- PathTracer.cpp:446-453 ----
    def CameraHasChanged(self):
        """Any element of the view-projection matrix moved by more than 1e-4 since the last Update()."""
        return bool(np.any(np.abs(self.myCamera.GetViewProj() - self.myLastViewMat) > np.float32(0.0001)))

    def _sky_settings(self):
        """What Sky_Imgui::HaveSettingsChanged watches (Sky_Imgui.cpp:10-43): sun direction / illuminance + atmosphere."""
        import ctypes
        return (bytes(np.asarray(self.mySky.mySunDir, dtype=np.float32)), bytes(np.asarray(self.mySky.mySunIlluminance, dtype=np.float32)),
                ctypes.string_at(ctypes.addressof(self.mySky.myAtmosphereParams), ctypes.sizeof(self.mySky.myAtmosphereParams)))

    # ---- PathTracer.cpp:557-571 (the per-frame accumulation control; no UI here) ----
    def Update(self):
        if not self.myAccumulate:
            self.RestartAccumulation()          # always render frame 0 only (:562-564)
        sky_now = self._sky_settings()
        if sky_now != self._last_sky:           # mySky_Imgui.HaveSettingsChanged() (:566)
            self._last_sky = sky_now
            self.SkySettingsChanged()
        if self.CameraHasChanged():             # (:566-568)
            self.RestartAccumulation()
        self.myLastViewMat = self.myCamera.GetViewProj()   # (:570)

    # ---- PathTracer.cpp:115-149 ----
    def LoadScene(self, scene, aCamPos=(0.0, 0.0, 0.0)):
        if not isinstance(scene, SceneData):
            from .scenes import load_obj
            scene = load_obj(scene)
        self.InitRtScene(scene)
        self.myScene = scene
        self.myCamera.myPosition = np.asarray(aCamPos, dtype=np.float32)
        self.myCamera.myOrientation = np.eye(3, dtype=np.float32)
        self.myCamera.myFovDeg = 60.0
        self.myCamera.myNear = 1.0
        self.myCamera.myFar = 10000.0
        self.myCamera.myWidth, self.myCamera.myHeight = float(self.myWindowWidth), float(self.myWindowHeight)
        self.RestartAccumulation()

    # ---- PathTracer.cpp:171-339 ----
    def InitRtScene(self, aScene, flags=api.PT_SCENE_DEFAULT):
        self.ctx.scene_create(aScene, flags)

    # ---- PathTracer.cpp:834-878 ----
    def UpdateOutputTexture(self):
        w, h = self.myWindowWidth, self.myWindowHeight
        if self.myHalfResRender:
            w, h = (w + 1) // 2, (h + 1) // 2
        self.myOutputWidth, self.myOutputHeight = w, h
        self.ctx.accum_reset(w, h)
        self.myNumAccumulationFrames = 0
        self.myAccumulationNeedsClear = False

    # ---- PathTracer.cpp:912-915 ----
    def RestartAccumulation(self):
        self.myNumAccumulationFrames = 0
        self.myAccumulationNeedsClear = True

    def OnWindowResized(self, aWidth, aHeight):
        self.myWindowWidth, self.myWindowHeight = int(aWidth), int(aHeight)
        self.myCamera.myWidth, self.myCamera.myHeight = float(aWidth), float(aHeight)
        self.UpdateOutputTexture()
        self.RestartAccumulation()

    # ---- PathTracer.cpp:689-768 ----
    def BuildRtConsts(self):
        c = api.PtRtConsts()
        v = self.myCamera.GetVerticesOnNearPlane()
        api.fill3(c.myNearPlaneCorner, v[0])
        c.myAoDistance = self.myAoDistance
        api.fill3(c.myXAxis, v[1] - v[0])
        api.fill3(c.myYAxis, v[3] - v[0])
        api.fill3(c.myCameraPos, self.myCamera.myPosition)
        c.myLightInstanceId = self.myLightInstanceIdx & 0xFFFFFFFF
        c.myNumHaltonSamples = 8192
        emission = self.myLightColor * np.float32(self.myLightStrength) if self.myLightEnabled else np.zeros(3, np.float32)
        api.fill3(c.myLightEmission, emission)
        c.mySampleSky = 1 if self.mySampleSky else 0
        api.fill3(c.mySkyFallbackEmission, (self.mySkyFallbackIntensity,) * 3)
        c.myPhongSpecularPower = self.myPhongSpecularPower
        c.myFrameRandomSeed = self.myNumAccumulationFrames
        c.myNumAccumulationFrames = self.myNumAccumulationFrames
        c.myMaxRecursionDepth = int(self.myMaxRecursionDepth)
        sc = c.mySkyConsts
        sc.myAtmosphere = self.mySky.myAtmosphereParams
        api.fill3(sc.mySunDirection, self.mySky.mySunDir)
        api.fill3(sc.mySunIlluminance, self.mySky.mySunIlluminance)
        sc.myRayMarchMinMaxSPP[0], sc.myRayMarchMinMaxSPP[1] = 4.0, 14.0   # PathTracer.cpp:707
        return c

    # ---- PathTracer.cpp:659-787; aNumFrames reference frames (1 spp each) in one call ----
    def RenderRT(self, aNumFrames=1):
        if self.myAccumulationNeedsClear or not self.myAccumulate:
            self.ctx.accum_reset(self.myOutputWidth, self.myOutputHeight)   # clear_texture.hlsl + counter reset (:668-678)
            self.myAccumulationNeedsClear = False
            self.myNumAccumulationFrames = 0
        if self.mySampleSky and self._sky_dirty:
            self.mySky.ComputeTranmittanceLut()                              # PathTracer.cpp:580
            self._sky_dirty = False
        consts = self.BuildRtConsts()
        # PathTracer.cpp:680-683: myRenderAo selects the AO pipeline state + shader table for the same dispatch
        render = self.ctx.render_ao if self.myRenderAo else self.ctx.render
        render(consts, self.myOutputWidth, self.myOutputHeight, self.myNumAccumulationFrames, int(aNumFrames))
        self.myNumAccumulationFrames += int(aNumFrames)

    Render = RenderRT

    def SkySettingsChanged(self):
        """Sky_Imgui::HaveSettingsChanged -> RestartAccumulation (PathTracer.cpp:566-568)."""
        self._sky_dirty = True
        self.RestartAccumulation()

    def ReadOutput(self):
        """Mean radiance image, (h, w, 4) float32 — the content of the reference's RGBA32F light texture."""
        return self.ctx.accum_read()

    @staticmethod
    def Tonemap(hdr):
        """tonemap_composit.hlsl:37-44 — Reinhard c/(c+1)."""
        rgb = hdr[..., :3]
        return rgb / (rgb + 1.0)
