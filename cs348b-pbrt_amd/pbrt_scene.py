"""Scene-file front end (the subset the BASELINE scenes use): reads a .pbrt file the way the reference's
parser + API layer do (core/pbrtlex.ll, core/pbrtparse.yy, core/api.cpp:521-1300) and returns the flattened scene this
package's C ABI takes -- the same dictionary `blob.load("scene_*.bin")` gives, so `abi.SceneHolder(d)`, `abi.params_from_blob(d)`
and everything downstream work unchanged.

Covered: Film "image" (resolution), Sampler "lowdiscrepancy" (pixelsamples), PixelFilter, SurfaceIntegrator "photonmap" and
VolumeIntegrator "photonvolume" parameters (incl. what CreatePhotonShooter reads from both, core/photonshooter.cpp:529-548),
Camera "perspective", the transform directives (Identity, Translate, Scale, Rotate, LookAt, Transform, ConcatTransform,
TransformBegin/End, AttributeBegin/End, ReverseOrientation), WorldBegin/End, LightSource "point" / "spot" / "distant", Material
"matte" / "glass" (with the fork's "Vn"), Shape "trianglemesh" / "sphere", Volume "homogeneous" / "rainbow" / "volumegrid", Include.
Anything else raises Unsupported with the directive's name and line: nothing is skipped silently.

Arithmetic is float32 in the reference's operation order (Matrix4x4::Mul, Transform::operator(), Rotate, LookAt, the
Gauss-Jordan Inverse of core/transform.cpp:76-135; transforms carry (m, mInv) pairs like core/transform.h so an inverse is the
product of the analytic inverses, not a numerical inversion), and "color" parameters go through a restatement of
SampledSpectrum::FromRGB (core/spectrum.cpp:154-241: ParamSet::AddRGBSpectrum converts every colour as a REFLECTANCE) over the
reference's own tables (data/spectral_tables.bin: numbers captured from the compiled reference, see tools/make_spectral_tables.py).
tests/test_pbrt_scene.py holds it against the scenes the reference itself built (tests/golden/scene_*.bin)."""
import math
import os

import numpy as np

from . import blob as _blob

F = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
NB = 30                       # nSpectralSamples (core/spectrum.h:46)
LAMBDA0, LAMBDA1 = 400, 700   # sampledLambdaStart / End


class Unsupported(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ spectra
_tables = None


def tables():
    global _tables
    if _tables is None:
        _tables = _blob.load(os.path.join(_HERE, "data", "spectral_tables.bin"))
    return _tables


def _lerp(t, a, b):
    return (F(1) - t) * a + t * b


def average_spectrum_samples(lam, vals, l0, l1):
    """core/spectrum.cpp:58-94: mean of the piecewise-linear curve (lam, vals) over [l0, l1]."""
    lam, vals = np.asarray(lam, F), np.asarray(vals, F)
    n = len(lam)
    l0, l1 = F(l0), F(l1)
    if l1 <= lam[0]:
        return vals[0]
    if l0 >= lam[n - 1]:
        return vals[n - 1]
    if n == 1:
        return vals[0]
    s = F(0)
    if l0 < lam[0]:
        s = F(s + vals[0] * (lam[0] - l0))
    if l1 > lam[n - 1]:
        s = F(s + vals[n - 1] * (l1 - lam[n - 1]))
    i = 0
    while l0 > lam[i + 1]:
        i += 1

    def interp(w, i):
        return _lerp(F((w - lam[i]) / (lam[i + 1] - lam[i])), vals[i], vals[i + 1])
    while i + 1 < n and l1 >= lam[i]:
        a, b = max(l0, lam[i]), min(l1, lam[i + 1])
        s = F(s + F(F(0.5) * F(interp(a, i) + interp(b, i))) * F(b - a))
        i += 1
    return F(s / F(l1 - l0))


_curves = {}


def _resampled(kind):
    """The seven Smits curves averaged onto the 30 bins (SampledSpectrum::Init, core/spectrum.h:384-420)."""
    if kind not in _curves:
        t = tables()
        lam = np.asarray(t["rgb2spect.lambda"], F)
        raw = np.asarray(t["rgb2spect." + kind], F).reshape(7, -1)
        out = np.zeros((7, NB), F)
        for b in range(NB):
            w0 = _lerp(F(F(b) / F(NB)), F(LAMBDA0), F(LAMBDA1))
            w1 = _lerp(F(F(b + 1) / F(NB)), F(LAMBDA0), F(LAMBDA1))
            for c in range(7):
                out[c, b] = average_spectrum_samples(lam, raw[c], w0, w1)
        _curves[kind] = out
    return _curves[kind]


# which curves follow white, by (smallest channel, order of the other two): (middle curve, top curve, lo, mid, hi channel)
_WHITE, _CYAN, _MAGENTA, _YELLOW, _RED, _GREEN, _BLUE = range(7)


def from_rgb(rgb, illuminant=False):
    """SampledSpectrum::FromRGB (core/spectrum.cpp:154-241): white x the smallest channel, a secondary x the gap to the middle
    one, a primary x the gap to the largest; x .94 (reflectance) or .86445 (illuminant); illuminants are clamped at 0."""
    r, g, b = (F(x) for x in rgb)
    cur = _resampled("illum" if illuminant else "refl")
    if r <= g and r <= b:
        lo = r
        plan = (_CYAN, g, _BLUE, b) if g <= b else (_CYAN, b, _GREEN, g)
    elif g <= r and g <= b:
        lo = g
        plan = (_MAGENTA, r, _BLUE, b) if r <= b else (_MAGENTA, b, _RED, r)
    else:
        lo = b
        plan = (_YELLOW, r, _GREEN, g) if r <= g else (_YELLOW, g, _RED, r)
    sec, mid, prim, hi = plan
    s = np.zeros(NB, F)
    s = (s + lo * cur[_WHITE]).astype(F)
    s = (s + F(mid - lo) * cur[sec]).astype(F)
    s = (s + F(hi - mid) * cur[prim]).astype(F)
    s = (s * (F(.86445) if illuminant else F(.94))).astype(F)
    if illuminant:
        s = np.maximum(s, F(0))
    return s


def const_spectrum(v):
    return np.full(NB, F(v), F)


# ------------------------------------------------------------------------------------------------ transforms
def _mul(a, b):
    """Matrix4x4::Mul (core/transform.h:83-92): four products added left to right."""
    r = np.zeros((4, 4), F)
    for i in range(4):
        for j in range(4):
            r[i, j] = F(F(F(a[i, 0] * b[0, j]) + F(a[i, 1] * b[1, j])) + F(a[i, 2] * b[2, j])) + F(a[i, 3] * b[3, j])
    return r


def _inverse(m):
    """Inverse(Matrix4x4), core/transform.cpp:76-135: Gauss-Jordan with full pivoting, float32."""
    minv = np.array(m, F)
    ipiv = [0] * 4
    indxr, indxc = [0] * 4, [0] * 4
    for i in range(4):
        irow = icol = -1
        big = F(0)
        for j in range(4):
            if ipiv[j] != 1:
                for k in range(4):
                    if ipiv[k] == 0:
                        if abs(minv[j, k]) >= big:
                            big = F(abs(minv[j, k]))
                            irow, icol = j, k
                    elif ipiv[k] > 1:
                        raise ValueError("singular matrix")
        ipiv[icol] += 1
        if irow != icol:
            minv[[irow, icol]] = minv[[icol, irow]]
        indxr[i], indxc[i] = irow, icol
        if minv[icol, icol] == 0:
            raise ValueError("singular matrix")
        pivinv = F(F(1) / minv[icol, icol])
        minv[icol, icol] = F(1)
        for j in range(4):
            minv[icol, j] = F(minv[icol, j] * pivinv)
        for j in range(4):
            if j != icol:
                save = F(minv[j, icol])
                minv[j, icol] = F(0)
                for k in range(4):
                    minv[j, k] = F(minv[j, k] - F(minv[icol, k] * save))
    for j in range(3, -1, -1):
        if indxr[j] != indxc[j]:
            minv[:, [indxr[j], indxc[j]]] = minv[:, [indxc[j], indxr[j]]]
    return minv


class Transform:
    """core/transform.h: a matrix and its inverse, carried together."""

    def __init__(self, m=None, minv=None):
        self.m = np.eye(4, dtype=F) if m is None else np.array(m, F).reshape(4, 4)
        self.minv = _inverse(self.m) if minv is None else np.array(minv, F).reshape(4, 4)

    def __mul__(self, t2):   # core/transform.cpp:286-290
        return Transform(_mul(self.m, t2.m), _mul(t2.minv, self.minv))

    def inverse(self):
        return Transform(self.minv, self.m)

    def point(self, p):      # core/transform.h:192-201
        x, y, z = (F(v) for v in p)
        m = self.m
        out = [F(F(F(m[i, 0] * x) + F(m[i, 1] * y)) + F(m[i, 2] * z)) + m[i, 3] for i in range(4)]
        if out[3] == F(1):
            return np.array(out[:3], F)
        return np.array([F(v / out[3]) for v in out[:3]], F)

    def vector(self, v):     # core/transform.h:215-220
        x, y, z = (F(c) for c in v)
        m = self.m
        return np.array([F(F(m[i, 0] * x) + F(m[i, 1] * y)) + F(m[i, 2] * z) for i in range(3)], F)

    def swaps_handedness(self):   # core/transform.cpp:293-302
        m = self.m
        det = F(F(m[0, 0] * F(F(m[1, 1] * m[2, 2]) - F(m[1, 2] * m[2, 1]))) - F(m[0, 1] * F(F(m[1, 0] * m[2, 2]) - F(m[1, 2] * m[2, 0])))) + \
            F(m[0, 2] * F(F(m[1, 0] * m[2, 1]) - F(m[1, 1] * m[2, 0])))
        return det < 0


def _sinf(x):
    return F(math.sin(float(F(x))))


def _cosf(x):
    return F(math.cos(float(F(x))))


def _radians(deg):
    return F(F(math.pi) / F(180)) * F(deg)


def _length(v):
    return F(np.sqrt(F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2]))))


def _normalize(v):
    v = np.asarray(v, F)
    inv = F(F(1) / _length(v))   # Vector::operator/(float): multiply by the reciprocal
    return np.array([F(c * inv) for c in v], F)


def _cross(a, b):   # core/geometry.h:477-484: in double, rounded once
    a, b = [float(x) for x in a], [float(x) for x in b]
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)


def translate(d):
    m, mi = np.eye(4, dtype=F), np.eye(4, dtype=F)
    for i in range(3):
        m[i, 3] = F(d[i])
        mi[i, 3] = F(-F(d[i]))
    return Transform(m, mi)


def scale(x, y, z):
    m, mi = np.eye(4, dtype=F), np.eye(4, dtype=F)
    for i, v in enumerate((x, y, z)):
        m[i, i] = F(v)
        mi[i, i] = F(F(1) / F(v))
    return Transform(m, mi)


def rotate(angle, axis):   # core/transform.cpp:205-233
    a = _normalize(axis)
    s, c = _sinf(_radians(angle)), _cosf(_radians(angle))
    one = F(1)
    m = np.eye(4, dtype=F)
    m[0, 0] = F(a[0] * a[0]) + F(F(one - F(a[0] * a[0])) * c)
    m[0, 1] = F(F(F(a[0] * a[1]) * F(one - c)) - F(a[2] * s))
    m[0, 2] = F(F(F(a[0] * a[2]) * F(one - c)) + F(a[1] * s))
    m[1, 0] = F(F(F(a[0] * a[1]) * F(one - c)) + F(a[2] * s))
    m[1, 1] = F(a[1] * a[1]) + F(F(one - F(a[1] * a[1])) * c)
    m[1, 2] = F(F(F(a[1] * a[2]) * F(one - c)) - F(a[0] * s))
    m[2, 0] = F(F(F(a[0] * a[2]) * F(one - c)) - F(a[1] * s))
    m[2, 1] = F(F(F(a[1] * a[2]) * F(one - c)) + F(a[0] * s))
    m[2, 2] = F(a[2] * a[2]) + F(F(one - F(a[2] * a[2])) * c)
    return Transform(m, m.T.copy())


def look_at(pos, look, up):   # core/transform.cpp:236-272
    pos, look = np.asarray(pos, F), np.asarray(look, F)
    d = _normalize(look - pos)
    left = _cross(_normalize(up), d)
    if _length(left) == 0:
        return Transform()
    left = _normalize(left)
    new_up = _cross(d, left)
    m = np.eye(4, dtype=F)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, pos
    cam_to_world = m
    return Transform(_inverse(cam_to_world), cam_to_world)


def _coordinate_system(v1):   # core/geometry.h:508-518
    if abs(v1[0]) > abs(v1[1]):
        inv = F(F(1) / F(np.sqrt(F(F(v1[0] * v1[0]) + F(v1[2] * v1[2])))))
        v2 = np.array([F(-v1[2] * inv), F(0), F(v1[0] * inv)], F)
    else:
        inv = F(F(1) / F(np.sqrt(F(F(v1[1] * v1[1]) + F(v1[2] * v1[2])))))
        v2 = np.array([F(0), F(v1[2] * inv), F(-v1[1] * inv)], F)
    return v2, _cross(v1, v2)


# ------------------------------------------------------------------------------------------------ tokens and parameter lists
def _tokens(text, path):
    i, n, line = 0, len(text), 1
    while i < n:
        c = text[i]
        if c == "\n":
            line += 1
            i += 1
        elif c.isspace():
            i += 1
        elif c == "#":
            while i < n and text[i] != "\n":
                i += 1
        elif c == '"':
            j = text.index('"', i + 1)
            yield ("str", text[i + 1:j], line)
            line += text.count("\n", i, j)
            i = j + 1
        elif c in "[]":
            yield (c, c, line)
            i += 1
        else:
            j = i
            while j < n and not text[j].isspace() and text[j] not in '[]"#':
                j += 1
            w = text[i:j]
            try:
                yield ("num", float(w), line)
            except ValueError:
                yield ("id", w, line)
            i = j


class _Params(dict):
    """name -> (type, values).  `get_*` mirror ParamSet::FindOne* (core/paramset.cpp): the default when absent."""

    def one(self, name, types, default):
        if name in self and self[name][0] in types:
            return self[name][1]
        return default

    def f(self, name, default):
        v = self.one(name, ("float",), None)
        return F(default) if v is None else F(v[0])

    def i(self, name, default):
        v = self.one(name, ("integer",), None)
        return int(default) if v is None else int(v[0])

    def b(self, name, default):
        v = self.one(name, ("bool",), None)
        return bool(default) if v is None else (v[0] == "true")

    def point(self, name, default):
        v = self.one(name, ("point", "vector", "normal"), None)
        return np.array(default if v is None else v[:3], F)

    def spectrum(self, name, default):
        v = self.one(name, ("color", "rgb"), None)
        return const_spectrum(default) if v is None else from_rgb(v[:3])   # AddRGBSpectrum: FromRGB's default type, REFLECTANCE


def _read_params(toks, k):
    """`"type name" value-or-[list]` pairs starting at toks[k]; returns (_Params, next k)."""
    ps = _Params()
    while k < len(toks) and toks[k][0] == "str":
        decl = toks[k][1].split()
        if len(decl) != 2:
            break
        typ, name = decl
        k += 1
        vals = []
        if toks[k][0] == "[":
            k += 1
            while toks[k][0] != "]":
                vals.append(toks[k][1])
                k += 1
            k += 1
        else:
            vals.append(toks[k][1])
            k += 1
        ps[name] = (typ, vals)
    return ps, k


# ------------------------------------------------------------------------------------------------ the API layer
class _Builder:
    def __init__(self):
        self.ctm = Transform()
        self.stack = []                 # AttributeBegin / TransformBegin
        self.reverse = False
        self.material = ("matte", _Params())
        self.film = [640, 480]
        self.spp = 4
        self.surf = ("photonmap", _Params())
        self.vol = ("photonvolume", _Params())
        self.camera = None
        self.lights, self.tris, self.tri_mat, self.tri_flip, self.mats, self.mat_keys = [], [], [], [], [], []
        self.spheres = []
        self.volume = None
        self.in_world = False

    # materials are numbered in the order shapes first use them
    def _material_index(self):
        name, ps = self.material
        key = (name, tuple(sorted((k, v[0], tuple(v[1])) for k, v in ps.items())))
        if key in self.mat_keys:
            return self.mat_keys.index(key)
        if name == "matte":     # materials/matte.cpp:64-70
            m = {"kind": 0, "kd": ps.spectrum("Kd", 0.5), "kr": np.zeros(NB, F), "kt": np.zeros(NB, F), "ior": F(1), "vn": F(0)}
        elif name == "glass":   # materials/glass.cpp:64-72 (the fork adds "Vn": Abbe number of the dispersive glass)
            m = {"kind": 1, "kd": np.zeros(NB, F), "kr": ps.spectrum("Kr", 1.0), "kt": ps.spectrum("Kt", 1.0), "ior": ps.f("index", 1.5), "vn": ps.f("Vn", 0.0)}
        else:
            raise Unsupported('Material "%s"' % name)
        self.mat_keys.append(key)
        self.mats.append(m)
        return len(self.mats) - 1

    def shape(self, name, ps, line):
        if name == "sphere":    # CreateSphereShape + Sphere::Sphere (shapes/sphere.cpp:217-225, 41-49)
            radius = ps.f("radius", 1.0)
            z0, z1 = ps.f("zmin", -radius), ps.f("zmax", radius)
            clamp = lambda v, lo, hi: F(min(max(F(v), F(lo)), F(hi)))   # noqa: E731
            zmin, zmax = clamp(min(z0, z1), -radius, radius), clamp(max(z0, z1), -radius, radius)
            acosf = lambda x: F(math.acos(float(F(x))))                  # noqa: E731
            self.spheres.append({"o2w": self.ctm, "f": np.array([radius, zmin, zmax, acosf(clamp(F(zmin) / F(radius), -1, 1)),
                                                                  acosf(clamp(F(zmax) / F(radius), -1, 1)),
                                                                  _radians(clamp(ps.f("phimax", 360.0), 0, 360))], F),
                                 "material": self._material_index(), "flip": int(self.reverse ^ bool(self.ctm.swaps_handedness()))})
            return
        if name != "trianglemesh":
            raise Unsupported('Shape "%s" (line %d): only "trianglemesh" and "sphere" are implemented' % (name, line))
        idx = [int(v) for v in ps.one("indices", ("integer",), [])]
        P = np.array(ps.one("P", ("point",), []), F).reshape(-1, 3)
        if len(idx) % 3 or (idx and max(idx) >= len(P)):
            raise ValueError("trianglemesh: bad indices (line %d)" % line)
        mi = self._material_index()
        world = np.array([self.ctm.point(p) for p in P], F)   # TriangleMesh ctor: vertices go to world space (trianglemesh.cpp:73-75)
        flip = int(self.reverse ^ bool(self.ctm.swaps_handedness()))
        for t in range(len(idx) // 3):
            self.tris.append(world[[idx[3 * t], idx[3 * t + 1], idx[3 * t + 2]]].reshape(-1))
            self.tri_mat.append(mi)
            self.tri_flip.append(flip)

    def light(self, name, ps):
        inten = lambda key: (ps.spectrum(key, 1.0) * ps.spectrum("scale", 1.0)).astype(F)   # noqa: E731
        L = {"pos": np.zeros(3, F), "dir": np.zeros(3, F), "cos": np.zeros(2, F)}
        if name == "point":       # lights/point.cpp:78-85
            P = ps.point("from", (0, 0, 0))
            l2w = translate(P) * self.ctm
            L.update(kind=0, l2w=l2w, intensity=inten("I"), pos=l2w.point((0, 0, 0)))
        elif name == "spot":      # lights/spot.cpp:96-117, :40-47
            frm, to = ps.point("from", (0, 0, 0)), ps.point("to", (0, 0, 1))
            cone, delta = ps.f("coneangle", 30.0), ps.f("conedeltaangle", 5.0)
            d = _normalize(to - frm)
            du, dv = _coordinate_system(d)
            m = np.eye(4, dtype=F)
            m[0, :3], m[1, :3], m[2, :3] = du, dv, d
            l2w = self.ctm * translate(frm) * Transform(m).inverse()
            L.update(kind=1, l2w=l2w, intensity=inten("I"), pos=l2w.point((0, 0, 0)),
                     cos=np.array([_cosf(_radians(cone)), _cosf(_radians(F(cone - delta)))], F))
        elif name == "distant":   # lights/distant.cpp:72-80, :40-45
            frm, to = ps.point("from", (0, 0, 0)), ps.point("to", (0, 0, 1))
            L.update(kind=2, l2w=self.ctm, intensity=inten("L"), dir=_normalize(self.ctm.vector(frm - to)))
        else:
            raise Unsupported('LightSource "%s"' % name)
        self.lights.append(L)

    def volume_region(self, name, ps):
        kinds = {"homogeneous": 1, "volumegrid": 2, "rainbow": 3}
        if name not in kinds:
            raise Unsupported('Volume "%s"' % name)
        if self.volume is not None:
            raise Unsupported("more than one Volume (AggregateVolume)")
        v = {"kind": kinds[name], "v2w": self.ctm, "sigma_a": ps.spectrum("sigma_a", 0.0), "sigma_s": ps.spectrum("sigma_s", 0.0),
             "le": ps.spectrum("Le", 0.0), "g": ps.f("g", 0.0), "p0": ps.point("p0", (0, 0, 0)), "p1": ps.point("p1", (1, 1, 1)),
             "dims": np.zeros(3, np.int32), "density": None}
        if name == "volumegrid":   # volumes/volumegrid.cpp:60-84
            v["dims"] = np.array([ps.i("nx", 1), ps.i("ny", 1), ps.i("nz", 1)], np.int32)
            v["density"] = np.array(ps.one("density", ("float",), []), F)
            if len(v["density"]) != int(np.prod(v["dims"])):
                raise ValueError("volumegrid: density has %d values, nx*ny*nz = %d" % (len(v["density"]), int(np.prod(v["dims"]))))
        self.volume = v


def _run(path, b, depth=0):
    if depth > 8:
        raise ValueError("Include nested too deeply")
    with open(path) as fh:
        toks = list(_tokens(fh.read(), path))
    k = 0
    while k < len(toks):
        kind, word, line = toks[k]
        if kind != "id":
            raise ValueError("%s:%d: unexpected token %r" % (path, line, word))
        k += 1

        def nums(n):
            nonlocal k
            out = []
            if toks[k][0] == "[":
                k += 1
                while toks[k][0] != "]":
                    out.append(toks[k][1])
                    k += 1
                k += 1
            else:
                out = [t[1] for t in toks[k:k + n]]
                k += n
            if len(out) != n:
                raise ValueError("%s:%d: %s takes %d numbers" % (path, line, word, n))
            return [F(v) for v in out]

        def named():
            nonlocal k
            name = toks[k][1]
            ps, k2 = _read_params(toks, k + 1)
            k = k2
            return name, ps
        if word == "Identity":
            b.ctm = Transform()
        elif word == "Translate":
            b.ctm = b.ctm * translate(nums(3))
        elif word == "Scale":
            b.ctm = b.ctm * scale(*nums(3))
        elif word == "Rotate":
            v = nums(4)
            b.ctm = b.ctm * rotate(v[0], v[1:])
        elif word == "LookAt":
            v = nums(9)
            b.ctm = b.ctm * look_at(v[0:3], v[3:6], v[6:9])
        elif word in ("Transform", "ConcatTransform"):   # api.cpp:640-668: the file holds the transpose
            m = np.array(nums(16), F).reshape(4, 4).T.copy()
            b.ctm = Transform(m) if word == "Transform" else b.ctm * Transform(m)
        elif word in ("AttributeBegin", "TransformBegin"):
            b.stack.append((word, b.ctm, b.reverse, b.material))
        elif word in ("AttributeEnd", "TransformEnd"):
            if not b.stack:
                raise ValueError("%s:%d: unmatched %s" % (path, line, word))
            w0, ctm, rev, mat = b.stack.pop()
            b.ctm = ctm
            if w0 == "AttributeBegin":
                b.reverse, b.material = rev, mat
        elif word == "ReverseOrientation":
            b.reverse = not b.reverse
        elif word == "WorldBegin":
            b.ctm, b.in_world = Transform(), True
        elif word == "WorldEnd":
            b.in_world = False
        elif word == "Include":
            inc = toks[k][1]
            k += 1
            _run(os.path.join(os.path.dirname(path), inc), b, depth + 1)
        elif word == "Film":
            name, ps = named()
            if name != "image":
                raise Unsupported('Film "%s"' % name)
            b.film = [ps.i("xresolution", 640), ps.i("yresolution", 480)]
        elif word == "Sampler":
            name, ps = named()
            if name != "lowdiscrepancy":
                raise Unsupported('Sampler "%s"' % name)
            b.spp = ps.i("pixelsamples", 4)
        elif word == "PixelFilter":
            name, ps = named()
            if name != "gaussian":
                raise Unsupported('PixelFilter "%s"' % name)
        elif word == "SurfaceIntegrator":
            b.surf = named()
        elif word == "VolumeIntegrator":
            b.vol = named()
            if b.vol[0] != "photonvolume":
                raise Unsupported('VolumeIntegrator "%s"' % b.vol[0])
        elif word == "Camera":
            name, ps = named()
            if name != "perspective":
                raise Unsupported('Camera "%s"' % name)
            if ps.f("lensradius", 0.0) != 0:
                raise Unsupported("Camera lensradius")
            b.camera = {"fov": ps.f("fov", 90.0), "c2w": b.ctm.inverse()}   # api.cpp:857-863
        elif word == "Material":
            b.material = named()
        elif word == "LightSource":
            name, ps = named()
            b.light(name, ps)
        elif word == "Shape":
            name, ps = named()
            b.shape(name, ps, line)
        elif word == "Volume":
            name, ps = named()
            b.volume_region(name, ps)
        else:
            raise Unsupported("%s:%d: directive %s" % (path, line, word))


def _mat16(m):
    return np.array(m, F).reshape(16)


def load(path):
    """Parse `path`; returns the flattened scene dictionary (keys of tests/golden/scene_*.bin)."""
    b = _Builder()
    _run(os.path.abspath(path), b)
    if b.camera is None:
        raise ValueError("no Camera")
    t = tables()
    d = {}
    v = b.volume
    if v is None:
        d.update({"vol.kind": np.array([0], np.int32), "vol.extent": np.zeros(6, F), "vol.w2v": _mat16(np.eye(4)), "vol.v2w": _mat16(np.eye(4)),
                  "vol.sigma_a": np.zeros(NB, F), "vol.sigma_s": np.zeros(NB, F), "vol.le": np.zeros(NB, F), "vol.g": np.zeros(1, F),
                  "vol.dims": np.zeros(3, np.int32)})
    else:
        d.update({"vol.kind": np.array([v["kind"]], np.int32), "vol.extent": np.concatenate([v["p0"], v["p1"]]).astype(F),
                  "vol.w2v": _mat16(v["v2w"].minv), "vol.v2w": _mat16(v["v2w"].m), "vol.sigma_a": v["sigma_a"], "vol.sigma_s": v["sigma_s"],
                  "vol.le": v["le"], "vol.g": np.array([v["g"]], F), "vol.dims": v["dims"]})
        if v["density"] is not None:
            d["vol.density"] = v["density"]
    L = b.lights
    d["lights.kind"] = np.array([x["kind"] for x in L], np.int32)
    d["lights.pos"] = np.concatenate([x["pos"] for x in L]).astype(F) if L else np.zeros(0, F)
    d["lights.dir"] = np.concatenate([x["dir"] for x in L]).astype(F) if L else np.zeros(0, F)
    d["lights.l2w"] = np.concatenate([_mat16(x["l2w"].m) for x in L]) if L else np.zeros(0, F)
    d["lights.w2l"] = np.concatenate([_mat16(x["l2w"].minv) for x in L]) if L else np.zeros(0, F)
    d["lights.intensity"] = np.concatenate([x["intensity"] for x in L]).astype(F) if L else np.zeros(0, F)
    d["lights.cos"] = np.concatenate([x["cos"] for x in L]).astype(F) if L else np.zeros(0, F)
    d["tris.p"] = np.concatenate(b.tris).astype(F) if b.tris else np.zeros(0, F)
    d["tris.material"] = np.array(b.tri_mat, np.int32)
    d["tris.flip"] = np.array(b.tri_flip, np.int32)
    if b.spheres:   # optional keys (abi.SceneHolder): absent for scenes without spheres
        d["spheres.o2w"] = np.concatenate([_mat16(x["o2w"].m) for x in b.spheres])
        d["spheres.w2o"] = np.concatenate([_mat16(x["o2w"].minv) for x in b.spheres])
        d["spheres.f"] = np.concatenate([x["f"] for x in b.spheres]).astype(F)
        d["spheres.material"] = np.array([x["material"] for x in b.spheres], np.int32)
        d["spheres.flip"] = np.array([x["flip"] for x in b.spheres], np.int32)
    M = b.mats
    d["mats.kind"] = np.array([m["kind"] for m in M], np.int32)
    for key in ("kd", "kr", "kt"):
        d["mats." + key] = np.concatenate([m[key] for m in M]).astype(F) if M else np.zeros(0, F)
    d["mats.ior"] = np.array([m["ior"] for m in M], F)
    d["mats.vn"] = np.array([m["vn"] for m in M], F)
    # Scene::WorldBound: the aggregate's bound united with the volume region's (core/scene.cpp:40-45)
    lo, hi = np.full(3, np.inf, F), np.full(3, -np.inf, F)
    if b.tris:
        P = d["tris.p"].reshape(-1, 3)
        lo, hi = np.minimum(lo, P.min(0)), np.maximum(hi, P.max(0))
    for x in b.spheres:   # Shape::WorldBound = ObjectToWorld(ObjectBound()) (core/shape.cpp:52-54, sphere.cpp:52-55)
        r, z0, z1 = x["f"][0], x["f"][1], x["f"][2]
        for cx in (-r, r):
            for cy in (-r, r):
                for cz in (z0, z1):
                    w = x["o2w"].point((cx, cy, cz))
                    lo, hi = np.minimum(lo, w), np.maximum(hi, w)
    if v is not None:   # Transform::operator()(BBox): the eight corners (core/transform.cpp:275-284)
        for cx in (v["p0"][0], v["p1"][0]):
            for cy in (v["p0"][1], v["p1"][1]):
                for cz in (v["p0"][2], v["p1"][2]):
                    w = v["v2w"].point((cx, cy, cz))
                    lo, hi = np.minimum(lo, w), np.maximum(hi, w)
    d["world"] = np.concatenate([lo, hi]).astype(F)
    d["cie.x"], d["cie.y"], d["cie.z"] = (np.asarray(t["cie." + c], F) for c in "xyz")
    d["xyz_scale"] = np.asarray(t["xyz_scale"], F)
    sp, vp = b.surf[1], b.vol[1]
    # PhotonVolumeIntegrator (integrators/photonvolume.cpp:224-229) and CreatePhotonShooter (core/photonshooter.cpp:529-548)
    d["params.f"] = np.array([vp.f("stepsize", 1.0), vp.f("maxdist", 0.1), sp.f("stepsize", 0.1)], F)
    # nused: the VOLUME integrator's default is 250 (photonvolume.cpp:226); 50 is the surface integrator's (photonmap.cpp:323)
    d["params.i"] = np.array([vp.i("nused", 250), vp.i("volumephotons", 0), sp.i("maxphotondepth", 5), sp.i("causticphotons", 20000),
                              sp.i("indirectphotons", 10000), int(sp.b("finalgather", True))], np.int32)
    # the surface integrator's own parameters (integrators/photonmap.cpp:336-363) for pvol_set_surface_integrator
    d["surf.name"] = b.surf[0]
    d["surf.params.i"] = np.array([sp.i("nused", 50), sp.i("maxspeculardepth", 5), int(sp.b("finalgather", True)),
                                   sp.i("finalgathersamples", 32)], np.int32)
    d["surf.params.f"] = np.array([sp.f("maxdist", 0.1), sp.f("gatherangle", 10.0)], F)
    d["camera.c2w"] = _mat16(b.camera["c2w"].m)
    d["camera.fov"] = np.array([b.camera["fov"]], F)
    d["film"] = np.array([b.film[0], b.film[1], b.spp], np.int32)
    return d
