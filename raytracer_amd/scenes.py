"""Scene model on the Python side of the boundary and the synthetic scenes of BASELINE.json's configs.

A :class:`Scene` is the flat (SoA-friendly) form of what the reference builds with ``new Sphere(...)`` /
``new HittablesList(...)`` / ``new LightsList(...)`` / ``Material[]`` (reference: src/postgres.h:54-153,
src/main.cu:69-90,517-530).  It round-trips through the ``rtscene v1`` text format that mirrors the reference's
Postgres tables (src/postgres.h:183-228); numbers are written with ``%.9g`` so float32 values survive exactly.

Hittables keep their list order: the reference's closest-hit loop lets the earlier object win ties
(src/hittables/hittablesList.h:27).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

SPHERE, PLANE, CUBE = 0, 1, 2
DIRLIGHT, POINTLIGHT = 0, 1
_HNAMES = {SPHERE: "sphere", PLANE: "plane", CUBE: "cube"}
_LNAMES = {DIRLIGHT: "dirlight", POINTLIGHT: "pointlight"}

f32 = np.float32


def _f(x) -> float:
    return float(np.float32(x))


@dataclass
class Scene:
    # materials: rows of (r, g, b, roughness, reflection, refraction, er, eg, eb, glow) -- src/material.h:5-11
    materials: List[Tuple[float, ...]] = field(default_factory=list)
    # hittables: (type, a[3], b[3], radius, material)
    hittables: List[Tuple[int, Tuple[float, float, float], Tuple[float, float, float], float, int]] = field(default_factory=list)
    # lights: (type, v[3])
    lights: List[Tuple[int, Tuple[float, float, float]]] = field(default_factory=list)
    fov: float = 60.0
    znear: float = 0.01
    zfar: float = 1000.0
    name: str = "scene"

    # -- builders mirroring the reference constructors ---------------------------------------------------
    def material(self, color, roughness, reflection, refraction, emission=(0.0, 0.0, 0.0), glow=0.0) -> int:
        self.materials.append(tuple(_f(v) for v in (*color, roughness, reflection, refraction, *emission, glow)))
        return len(self.materials) - 1

    def sphere(self, center, radius, material) -> None:          # src/hittables/sphere.h:10
        self.hittables.append((SPHERE, tuple(map(_f, center)), (0.0, 0.0, 0.0), _f(radius), int(material)))

    def plane(self, center, normal, material) -> None:           # src/hittables/plane.h:10
        self.hittables.append((PLANE, tuple(map(_f, center)), tuple(map(_f, normal)), 0.0, int(material)))

    def cube(self, position, size, material) -> None:            # src/hittables/cube.h:10
        self.hittables.append((CUBE, tuple(map(_f, position)), tuple(map(_f, size)), 0.0, int(material)))

    def dirlight(self, direction) -> None:                       # src/lights/directionalLight.h:8
        self.lights.append((DIRLIGHT, tuple(map(_f, direction))))

    def pointlight(self, position) -> None:                      # repo-defined (config C4), see DESIGN.md
        self.lights.append((POINTLIGHT, tuple(map(_f, position))))

    # -- text form ---------------------------------------------------------------------------------------
    def to_text(self) -> str:
        g = lambda v: "%.9g" % v
        out = ["# rtscene v1: %s" % self.name]
        if (self.fov, self.znear, self.zfar) != (60.0, 0.01, 1000.0):
            out.append("camera %s %s %s" % (g(self.fov), g(self.znear), g(self.zfar)))
        for m in self.materials:
            out.append("material " + " ".join(g(v) for v in m))
        for t, a, b, r, m in self.hittables:
            if t == SPHERE:
                out.append("sphere %s %s %d" % (" ".join(map(g, a)), g(r), m))
            else:
                out.append("%s %s %s %d" % (_HNAMES[t], " ".join(map(g, a)), " ".join(map(g, b)), m))
        for t, v in self.lights:
            out.append("%s %s" % (_LNAMES[t], " ".join(map(g, v))))
        return "\n".join(out) + "\n"

    def save(self, path) -> None:
        with open(path, "w") as fh:
            fh.write(self.to_text())

    @staticmethod
    def _table_row(s: "Scene", kw: str, line: str, scene_id: int) -> bool:
        """A row of the reference's Postgres tables (src/postgres.h:183-228) in table-column order, VEC3 columns written
        "(x, y, z)" as readVec3 (src/postgres.h:232-256) takes them; columns separated by blanks, '|' or ','.
            material <id> (color) <roughness> <reflection> <refraction> (emission_color) <glow_strength>
            sphere <id> (position) <radius> <material_id> <scene_id>
            directional_light <id> (direction) <scene_id>
        Rows of another scene are skipped (getWorld / getLights select by scene_id, src/postgres.h:88,113); material
        indices are row order (getMaterials, src/postgres.h:66-77).  False: not such a row."""
        if kw not in ("material", "sphere", "directional_light") or "(" not in line:
            return False
        cols, rest = [], line[len(kw):].replace("|", " ").replace("\t", " ")
        while rest.strip(" ,"):
            rest = rest.lstrip(" ,")
            if rest[0] == "(":
                end = rest.index(")")
                parts = rest[1:end].split(",")
                if len(parts) != 3:
                    raise ValueError
                cols.append(tuple(float(x) for x in parts))          # atof of each part
                rest = rest[end + 1:]
            else:
                tok = rest.split(None, 1)[0].split(",", 1)[0]
                if tok.endswith("::VEC3"):
                    tok = tok[:-6]
                cols.append(float(tok))
                rest = rest[len(tok):]
        shape = [isinstance(c, tuple) for c in cols]
        if kw == "material" and shape == [False, True, False, False, False, True, False]:
            s.materials.append(tuple(_f(x) for x in (*cols[1], cols[2], cols[3], cols[4], *cols[5], cols[6])))
        elif kw == "sphere" and shape == [False, True, False, False, False]:
            if int(cols[4]) == scene_id:
                s.sphere(cols[1], cols[2], int(cols[3]))
        elif kw == "directional_light" and shape == [False, True, False]:
            if int(cols[2]) == scene_id:
                s.dirlight(cols[1])
        else:
            raise ValueError
        return True

    @staticmethod
    def from_text(text: str, name: str = "scene", scene_id: int = 0) -> "Scene":
        s = Scene(name=name)
        for ln, raw in enumerate(text.splitlines(), 1):
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            kw, *rest = line.split()
            try:
                if Scene._table_row(s, kw, line.replace("'", "").replace("::VEC3", ""), scene_id):
                    continue
                v = [float(x) for x in rest]
                if kw == "material":
                    if len(v) != 10:
                        raise ValueError
                    s.materials.append(tuple(_f(x) for x in v))
                elif kw == "sphere":
                    if len(v) != 5:
                        raise ValueError
                    s.sphere(v[0:3], v[3], int(v[4]))
                elif kw in ("plane", "cube"):
                    if len(v) != 7:
                        raise ValueError
                    (s.plane if kw == "plane" else s.cube)(v[0:3], v[3:6], int(v[6]))
                elif kw in ("dirlight", "pointlight"):
                    if len(v) != 3:
                        raise ValueError
                    (s.dirlight if kw == "dirlight" else s.pointlight)(v)
                elif kw == "camera":
                    if len(v) != 3:
                        raise ValueError
                    s.fov, s.znear, s.zfar = map(_f, v)
                else:
                    raise KeyError(kw)
            except KeyError:
                raise ValueError("rtscene line %d: unknown keyword %r" % (ln, kw))
            except ValueError:
                raise ValueError("rtscene line %d: malformed %r record" % (ln, kw))
        return s

    @staticmethod
    def load(path) -> "Scene":
        with open(path) as fh:
            return Scene.from_text(fh.read(), name=str(path))

    # -- flat arrays for the C-ABI -------------------------------------------------------------------------
    def arrays(self):
        mats = np.asarray(self.materials, dtype=f32).reshape(-1, 10)
        n = len(self.hittables)
        htype = np.asarray([h[0] for h in self.hittables], dtype=np.int32).reshape(n)
        ha = np.asarray([h[1] for h in self.hittables], dtype=f32).reshape(n, 3)
        hb = np.asarray([h[2] for h in self.hittables], dtype=f32).reshape(n, 3)
        hr = np.asarray([h[3] for h in self.hittables], dtype=f32).reshape(n)
        hm = np.asarray([h[4] for h in self.hittables], dtype=np.int32).reshape(n)
        ltype = np.asarray([l[0] for l in self.lights], dtype=np.int32).reshape(len(self.lights))
        lv = np.asarray([l[1] for l in self.lights], dtype=f32).reshape(len(self.lights), 3)
        return mats, htype, ha, hb, hr, hm, ltype, lv


# ---------------------------------------------------------------------------------------------------------
# Materials of the reference's default rows (src/postgres.h:207-210 == src/main.cu:522-525)
def _default_materials(s: Scene) -> None:
    s.material((0.8, 0.8, 0.0), 0.0, 0.0, 0.0)
    s.material((0.8, 0.2, 0.1), 0.08, 0.02, 0.0, (1.0, 0.0, 0.0), 4.5)
    s.material((0.8, 0.8, 0.8), 0.9, 0.75, 0.0)
    s.material((0.0, 0.0, 0.0), 0.0, 0.0, 1.85)


def default4() -> Scene:
    """DEFAULT-4 (configs C1-C3): src/postgres.h:224-227 spheres, :191 light."""
    s = Scene(name="default4")
    _default_materials(s)
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    s.sphere((0.0, 1.0, -4.0), 1.0, 1)
    s.sphere((-3.0, 1.0, -4.0), 1.0, 2)
    s.sphere((3.0, 1.0, -4.0), 1.0, 3)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def random10(seed: int = 1) -> Scene:
    """RANDOM-10(seed): the shape of generateRandomScene (src/postgres.h:132-153: x,z ~ U(-20,0), y = 0.25,
    r ~ U(1,3), material ~ floor(U(0,4))) with main.cpp's light (src/main.cpp:244), from a fixed numpy seed
    instead of std::random_device."""
    rs = np.random.RandomState(seed)
    s = Scene(name="random10_%d" % seed)
    _default_materials(s)
    for _ in range(10):
        x, z = rs.uniform(-20.0, 0.0, 2)
        r = rs.uniform(1.0, 3.0)
        m = int(rs.uniform(0.0, 4.0))
        s.sphere((x, 0.25, z), r, m)
    s.dirlight((-0.35, -0.55, -0.5))
    return s


def many_objects(n: int = 2048, n_lights: int = 8, seed: int = 7) -> Scene:
    """Config C4: n jittered spheres/cubes on a ground sphere + a ring of point lights."""
    rs = np.random.RandomState(seed)
    s = Scene(name="many%d_l%d" % (n, n_lights))
    _default_materials(s)
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    nx = int(np.ceil(np.sqrt(2.0 * n)))
    nz = int(np.ceil(n / nx))
    k = 0
    for iz in range(nz):
        for ix in range(nx):
            if k >= n:
                break
            x = -30.0 + 60.0 * (ix + rs.uniform(0.15, 0.85)) / nx
            z = 4.0 - 64.0 * (iz + rs.uniform(0.15, 0.85)) / nz
            size = rs.uniform(0.2, 0.8)
            m = k % 4
            if k % 3 == 2:
                s.cube((x - size / 2, 0.0, z - size / 2), (size, size, size), m)
            else:
                s.sphere((x, size, z), size, m)
            k += 1
    for i in range(n_lights):
        ang = 2.0 * np.pi * i / max(n_lights, 1)
        s.pointlight((25.0 * np.cos(ang), 18.0, -28.0 + 25.0 * np.sin(ang)))
    return s


def glass_mirror() -> Scene:
    """Config C5: glass / perfect-mirror / mixed (reflect + refract) objects, one cube of each kind."""
    s = Scene(name="glass_mirror")
    _default_materials(s)                                    # 0 diffuse, 1 red glow, 2 rough mirror, 3 glass
    mirror = s.material((0.9, 0.9, 0.9), 0.0, 0.9, 0.0)     # 4
    mixed = s.material((0.1, 0.1, 0.1), 0.05, 0.5, 1.5)     # 5
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    s.sphere((0.0, 1.0, -4.0), 1.0, 3)
    s.sphere((-2.5, 1.0, -3.0), 1.0, mirror)
    s.sphere((2.5, 1.0, -3.0), 1.0, mixed)
    s.sphere((-1.2, 0.6, -1.0), 0.6, 3)
    s.sphere((1.4, 0.5, -0.5), 0.5, mirror)
    s.sphere((-4.5, 1.5, -6.5), 1.5, 3)
    s.sphere((4.5, 1.2, -6.0), 1.2, 1)
    s.cube((-0.75, 0.0, -8.0), (1.5, 1.5, 1.5), 3)
    s.cube((2.0, 0.0, -9.0), (1.2, 2.0, 1.2), mirror)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def plane_scene() -> Scene:
    """Plane coverage (src/hittables/plane.h): reproduces the sign quirk -- a y=0 'ground' plane shows up in the sky."""
    s = Scene(name="plane")
    _default_materials(s)
    s.plane((0.0, 0.0, 0.0), (0.0, 2.0, 0.0), 0)
    s.plane((0.0, 0.0, -12.0), (0.3, 0.1, 1.0), 2)
    s.sphere((0.0, 1.0, -4.0), 1.0, 1)
    s.sphere((3.0, 1.0, -4.0), 1.0, 3)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def cube_scene() -> Scene:
    """Cube coverage (src/hittables/cube.h): diffuse, rough-mirror and glass cubes + an axis-aligned view ray case."""
    s = Scene(name="cube")
    _default_materials(s)
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    s.cube((-1.0, 0.0, -5.0), (2.0, 2.0, 2.0), 1)
    s.cube((-4.5, 0.0, -4.0), (2.0, 1.5, 2.0), 2)
    s.cube((2.5, 0.0, -4.0), (1.5, 2.5, 1.5), 3)
    s.sphere((0.5, 0.5, -1.5), 0.5, 2)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def point_light_scene() -> Scene:
    """Small C4-shaped case: mixed primitives under three point lights and one directional light."""
    s = many_objects(n=24, n_lights=3, seed=3)
    s.name = "pointlights"
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def degenerate_scene() -> Scene:
    """Edge cases: a material whose reflection ray can fall below the surface with no refraction (maxDepth left at 0,
    renderer.h:80,128 -> colour / 0), coincident spheres (closest-hit tie: first in list wins), a light-less look."""
    s = Scene(name="degenerate")
    s.material((0.8, 0.8, 0.0), 0.0, 0.0, 0.0)
    s.material((0.5, 0.6, 0.0), 1.0, 0.6, 0.0)      # roughness 1: perturbed direction often leaves the hemisphere
    s.material((0.2, 0.4, 0.9), 0.3, 1.0, 0.0)      # reflection 1: multiplier never shrinks, depth-limited
    s.material((0.1, 0.1, 0.1), 0.5, 0.3, 1.2, (0.2, 0.9, 0.1), 2.0)   # rough mixed, glowing
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    s.sphere((0.0, 1.0, -4.0), 1.0, 1)
    s.sphere((0.0, 1.0, -4.0), 1.0, 2)              # coincident with the previous one
    s.sphere((-3.0, 1.0, -4.0), 1.0, 2)
    s.sphere((3.0, 1.0, -4.0), 1.0, 3)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def nan_scene() -> Scene:
    """Not-a-number rules of the closest-hit loop (hittablesList.h:25: `dist >= best` is false for a NaN on either
    side): a sphere with a NaN radius 'hits' every ray at a NaN distance and replaces the best so far; after it only
    real hits replace, misses do not; a second NaN sphere further down the list takes over from whatever came before."""
    s = Scene(name="nan")
    _default_materials(s)
    nan = float("nan")
    s.sphere((0.0, -1000.0, -4.0), 1000.0, 0)
    s.sphere((0.0, 1.0, -4.0), nan, 1)              # NaN distance for every ray
    s.sphere((-3.0, 1.0, -4.0), 1.0, 2)             # most rays miss it: a miss must not replace the NaN best
    s.sphere((3.0, 1.0, -4.0), 1.0, 3)
    s.cube((-0.5, 0.0, -1.0), (1.0, 1.0, 1.0), 2)
    s.dirlight((-0.25, -0.75, 0.45))
    return s


def nan_tail_scene() -> Scene:
    """nan_scene with another NaN sphere at the end of the list."""
    s = nan_scene()
    s.name = "nan_tail"
    s.sphere((1.0, 2.0, -2.0), float("nan"), 0)
    return s


BUILTIN = {
    "default4": default4, "random10": random10, "many_objects": many_objects, "glass_mirror": glass_mirror,
    "plane": plane_scene, "cube": cube_scene, "pointlights": point_light_scene, "degenerate": degenerate_scene,
    "nan": nan_scene, "nan_tail": nan_tail_scene,
}
