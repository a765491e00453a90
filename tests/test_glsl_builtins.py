"""An independent pin for the GLSL built-ins.

include/rt_det_math.h is compiled into the HIP path and into the oracle, so the GPU parity tests compare these functions
with themselves. Here every built-in the shader uses (raytrace.comp:177-193,292-293,318-319,356-365,419-423,466-481,
547-556) is restated in numpy float32 from the GLSL 4.50 specification's formulas (section 8), without reading
rt_det_math.h's code, and compared

  * on the CPU with the oracle's build of the header  (oracle_glsl_probe), and
  * on the GPU with the device's build               (rt_device_math_probe, a kernel that returns raw values),

bit for bit where the specification fixes the arithmetic (+ - * / sqrt in a stated order: reflect, refract, smoothstep,
mix, sign, cross, dot, matrix * vector, radians, min / max, abs, isnan / isinf, the PCG hash), and against float64 within
a stated number of units in the last place where it leaves the precision to the implementation (sin, cos, tan, log2,
exp2, pow, inverse; GLSL 4.50 section 4.7.1). The driver-specific approximations of the author's GPU cannot be pinned.

numpy evaluates one IEEE binary32 operation per ufunc call (no fused multiply-add), so `a * b + c` below rounds twice,
which is the arithmetic the specification's formulas describe.
"""
import numpy as np
import pytest

from oracle import pyoracle

F = np.float32
N_PROBES = 4096


def _inputs(seed=11):
    rng = np.random.default_rng(seed)
    x = np.zeros((N_PROBES, 32), np.float32)
    x[:, 0:3] = rng.normal(size=(N_PROBES, 3))                                   # I
    n = rng.normal(size=(N_PROBES, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    x[:, 3:6] = n                                                                # N (unit)
    x[:, 6] = rng.uniform(0.3, 2.5, N_PROBES)                                    # eta: both sides of total internal reflection
    x[:, 7] = rng.uniform(-1.5, 1.5, N_PROBES)                                   # x
    x[:, 8] = rng.uniform(-0.5, 0.2, N_PROBES)                                   # e0
    x[:, 9] = x[:, 8] + rng.uniform(0.01, 1.0, N_PROBES)                         # e1 > e0
    x[:, 10] = rng.uniform(0, 1, N_PROBES)                                       # a
    x[:, 11] = rng.uniform(-2, 6, N_PROBES)                                      # y
    x[:, 12:15] = rng.normal(size=(N_PROBES, 3))                                 # V
    for i in range(N_PROBES):                                                    # M: rotation * scale + translation, some singular-ish
        m = np.eye(4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m[:3, :3] = q @ np.diag(rng.uniform(0.2, 3.0, 3) * rng.choice([-1, 1], 3))
        m[:3, 3] = rng.normal(size=3) * 2
        x[i, 16:32] = m.T.ravel()                                                # column-major
    # edge cases the shader meets
    x[0, 7] = 0.0; x[1, 7] = -0.0; x[2, 7] = np.nan; x[3, 7] = np.inf; x[4, 7] = -np.inf
    x[5, 0:3] = 0.0                                                              # normalize(0)
    x[6, 7] = x[6, 8]; x[7, 7] = x[7, 9]; x[8, 7] = x[8, 8] - 1; x[9, 7] = x[9, 9] + 1   # smoothstep edges and outside
    x[10, 7] = np.nan; x[10, 11] = 2.0; x[11, 7] = 3.0; x[11, 11] = np.nan       # min / max with one NaN operand
    x[12, 0:3] = -x[12, 3:6]; x[12, 6] = 2.0                                     # head-on refraction
    x[13, 6] = 2.5; x[13, 0:3] = np.cross(x[13, 3:6], [1, 0, 0]) + 0.05 * x[13, 3:6]    # grazing: k < 0
    x[14, 7] = np.radians(25.0)                                                  # tan(fov / 2) of the default camera
    x[15, 7] = 1e-30; x[16, 7] = 1e-40                                           # tiny, subnormal
    x[:, 15] = rng.integers(0, 256, N_PROBES)                                    # a texel byte
    x[17, 7] = 1.0; x[18, 7] = -1.0 / 37; x[19, 7] = 36.0 / 37; x[20, 7] = 5.5; x[21, 7] = -7.25; x[22, 7] = 1e12   # texture coordinates
    return x


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def _expected_exact(x):
    """The outputs whose arithmetic the specification fixes, in numpy float32. Returns {slot: values}."""
    I, N, V = x[:, 0:3], x[:, 3:6], x[:, 12:15]
    eta, xs, e0, e1, a, y = (x[:, k] for k in (6, 7, 8, 9, 10, 11))
    two, one, three, zero = F(2), F(1), F(3), F(0)
    out = {}
    with np.errstate(all="ignore"):
        # reflect(I, N) = I - 2 * dot(N, I) * N
        k = two * _dot(N, I)
        out[0] = I - N * k[:, None]
        # refract: k = 1 - eta^2 (1 - dot(N,I)^2); k < 0 -> 0, else eta * I - (eta * dot(N,I) + sqrt(k)) * N
        d = _dot(N, I)
        kk = one - eta * eta * (one - d * d)
        r = I * eta[:, None] - N * (eta * d + np.sqrt(np.maximum(kk, zero)))[:, None]
        r[kk < 0] = 0
        out[3] = r
        # smoothstep: t = clamp((x - e0) / (e1 - e0), 0, 1); t * t * (3 - 2 t)
        t = (xs - e0) / (e1 - e0)
        t = np.fmin(np.fmax(t, zero), one)   # clamp = min(max(t, 0), 1); for NaN the defined choice of min / max applies (GLSL: undefined)
        sm = t * t * (three - two * t)
        out[6] = sm[:, None]
        out[7] = (xs * (one - a) + y * a)[:, None]                                   # mix
        out[8] = np.where(xs > 0, one, np.where(xs < 0, -one, zero)).astype(F)[:, None]   # sign (sign(NaN) = 0 by this rule)
        out[9] = I * (one / np.sqrt(_dot(I, I)))[:, None]                            # normalize = v * inversesqrt(dot(v, v))
        out[13] = (xs * F(np.pi / 180.0))[:, None]                                   # radians
        out[14] = np.stack([I[:, 1] * V[:, 2] - V[:, 1] * I[:, 2], I[:, 2] * V[:, 0] - V[:, 2] * I[:, 0],
                            I[:, 0] * V[:, 1] - V[:, 0] * I[:, 1]], 1)               # cross
        out[17] = _dot(I, V)[:, None]
        out[18] = np.fmin(xs, y)[:, None]                                            # a NaN operand loses (the defined choice)
        out[19] = np.fmax(xs, y)[:, None]
        out[25] = np.sqrt(np.abs(xs))[:, None]
        out[26] = np.abs(xs)[:, None]
        out[27] = np.isnan(xs).astype(F)[:, None]
        out[28] = np.isinf(xs).astype(F)[:, None]
        out[31] = (one / xs)[:, None]
        # raytrace.comp:158-163 on the bits of x
        s = x[:, 7].view(np.uint32).astype(np.uint64)
        ns = (s * 747796405 + 2891336453) & 0xFFFFFFFF
        rr = (((ns >> ((ns >> 28) + 4)) ^ ns) * 277803737) & 0xFFFFFFFF
        rr = ((rr >> 22) ^ rr) & 0xFFFFFFFF
        out[29] = (rr.astype(np.float32) / F(4294967295.0))[:, None]                 # 4294967295.0 rounds to 2^32 in binary32
        out[30] = ns.astype(np.uint32).view(np.float32)[:, None]
        # M * vec4(V, 0) and M * vec4(V, 1), column-major M
        M = x[:, 16:32].reshape(-1, 4, 4)                                            # M[:, c, r]
        lin = (M[:, 0, :3] * V[:, 0:1] + M[:, 1, :3] * V[:, 1:2]) + M[:, 2, :3] * V[:, 2:3]
        out[32] = lin
        out[35] = lin + M[:, 3, :3]
    return out


def _ulp(ref64):
    r = np.abs(ref64).astype(np.float32)
    return np.maximum(np.spacing(r).astype(np.float64), 1e-45)


def _check_all(got, x):
    """got: [n, 64] from some build of rt_det_math.h."""
    exp = _expected_exact(x)
    for slot, e in exp.items():
        g = got[:, slot:slot + e.shape[1]]
        same = (g.view(np.uint32) == np.ascontiguousarray(e, np.float32).view(np.uint32)) | (np.isnan(g) & np.isnan(e)) | ((g == 0) & (e == 0))
        assert same.all(), f"slot {slot}: {int((~same).sum())} values differ, first at probe {np.argwhere(~same)[0].tolist()}"
    xs = x[:, 7].astype(np.float64)
    ys = x[:, 11].astype(np.float64)
    fin = np.isfinite(xs) & (np.abs(xs) < 100.0)   # the polynomial sin / cos are defined for |x| < 8192; the shader stays within 2 pi
    with np.errstate(all="ignore"):
        # implementation-defined precision: against float64. GLSL 4.50 4.7.1 gives no bound for sin / cos / tan and a few ulp
        # for exp2 / log2 / pow; the bounds below are what the polynomial restatement achieves, with a little headroom.
        for slot, ref, tol_ulp, tol_abs in ((20, np.sin(xs), 2, 6e-8), (21, np.cos(xs), 2, 6e-8), (12, np.tan(xs), 4, 1.2e-7)):
            err = np.abs(got[fin, slot].astype(np.float64) - ref[fin])
            assert (err <= np.maximum(tol_ulp * _ulp(ref[fin]), tol_abs)).all(), f"slot {slot}: worst {err.max()}"
        ax = np.abs(xs)
        pos = fin & (ax > 0)
        ref = np.log2(ax[pos])
        err = np.abs(got[pos, 22].astype(np.float64) - ref)
        assert (err <= np.maximum(2 * _ulp(ref), 1.2e-7)).all(), f"log2: worst {err.max()}"
        ref = np.exp2(xs[fin])
        err = np.abs(got[fin, 23].astype(np.float64) - ref)
        assert (err <= 2 * _ulp(ref)).all(), f"exp2: worst {(err / _ulp(ref)).max()} ulp"
        ref = np.power(ax[pos], ys[pos])
        ok = np.isfinite(ref) & (ref > 1e-30) & (ref < 1e30)
        err = np.abs(got[pos, 24].astype(np.float64)[ok] - ref[ok])
        # pow = exp2(y * log2 x) (GLSL 8.2): the rounding of y * log2 x is amplified by |y * log2 x|
        amp = 1.0 + np.abs(ys[pos][ok] * np.log2(ax[pos][ok]))
        assert (err <= 2 * amp * _ulp(ref[ok])).all(), "pow"
        # inverse(M): against float64
        M = x[:, 16:32].reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)
        inv = np.linalg.inv(M)
        gi = got[:, 38:54].reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)
        scale = np.abs(inv).max(axis=(1, 2), keepdims=True)
        assert (np.abs(gi - inv) <= 2e-6 * scale).all(), "inverse(mat4)"
    # textures (the declared semantics of include/rt_amd.h): the sRGB transfer function and nearest-filter addressing
    b = x[:, 15].astype(np.float64) / 255.0
    lin = np.where(b <= 0.04045, b / 12.92, ((b + 0.055) / 1.055) ** 2.4)
    err = np.abs(got[:, 54].astype(np.float64) - lin)
    assert (err <= 8 * _ulp(lin) + 1e-9).all(), f"sRGB decode: worst {(err / _ulp(lin)).max()} ulp"
    f = x[:, 7].astype(np.float32) * np.float32(37)
    ok = np.isfinite(f) & (np.abs(f) < 2.0**30)
    fl = np.floor(f[ok].astype(np.float64)).astype(np.int64)
    assert np.array_equal(got[ok, 55].astype(np.int64), np.mod(fl, 37))           # VK_SAMPLER_ADDRESS_MODE_REPEAT
    assert np.array_equal(got[ok, 56].astype(np.int64), np.clip(fl, 0, 36))       # CLAMP_TO_EDGE
    assert (got[~ok, 55] == 0).all() and (got[~ok, 56] == 0).all()
    assert got[17, 55] == 0 and got[17, 56] == 36 and got[18, 55] == 36 and got[18, 56] == 0   # u = 1 wraps to 0 / clamps to the last texel
    # named edge cases
    assert got[0, 8] == 0 and got[1, 8] == 0                                     # sign(+-0) = 0
    assert np.isnan(got[5, 9:12]).all()                                          # normalize(0) = NaN
    assert got[6, 6] == 0 and got[7, 6] == 1 and got[8, 6] == 0 and got[9, 6] == 1   # smoothstep at and beyond the edges
    assert got[10, 18] == 2 and got[10, 19] == 2 and got[11, 18] == 3 and got[11, 19] == 3   # the non-NaN operand wins
    assert (got[13, 3:6] == 0).all()                                             # refract past the critical angle
    assert got[2, 27] == 1 and got[3, 28] == 1 and got[4, 28] == 1
    assert abs(float(got[14, 12]) - np.tan(np.radians(25.0))) < 1e-7             # the camera plane's tan(fov / 2)


def test_oracle_build_against_numpy_restatements():
    x = _inputs()
    _check_all(pyoracle.glsl_probe(x), x)


def test_kernel_constants_against_the_shader_text():
    """PI and INV_PI as raytrace.comp:6-7 spells them; the miss sentinel of :272,279."""
    x = np.zeros((1, 32), np.float32)
    x[0, 7] = 180.0
    got = pyoracle.glsl_probe(x)
    assert got[0, 13] == F(3.1415926535897932384)       # radians(180) = PI in binary32


@pytest.mark.gpu
def test_device_build_against_numpy_restatements(renderer):
    """The same comparison for the code the GPU runs — device against numpy, not header against header."""
    x = _inputs()
    got = renderer.math_probe(x)
    _check_all(got, x)
    # and the two builds agree on every bit, including the implementation-defined functions
    cpu = pyoracle.glsl_probe(x)
    same = (got.view(np.uint32) == cpu.view(np.uint32)) | (np.isnan(got) & np.isnan(cpu))
    assert same.all(), f"device and host builds differ in slots {sorted(set(np.argwhere(~same)[:, 1].tolist()))}"
