"""MJCF-subset compiler: XML kinematic tree -> flat :class:`Model` table.

The reference hands MJCF files to ``mujoco.MjModel.from_xml_path``
(interactive_scripts/teleop_mouse.py:29, envs/mj_env.py) and then reads the
``MjModel`` arrays (``jnt_type``, ``jnt_qposadr``, ``body_parentid`` ...) in
mink/configuration.py:86-97, mink/limits/configuration_limit.py:41-56 and
mink/limits/collision_avoidance_limit.py:75-115.  MuJoCo is not part of the
reference tree, so this module re-derives the part of its model compiler that
inverse kinematics needs: bodies, joints, sites, geoms, inertials, keyframes,
``<include>``, nested ``<default class>`` / ``childclass`` and ``autolimits``.
Field names follow ``MjModel`` so host code reads like the reference.

Physics-only elements (actuators, tendons, equality, contact, sensors, assets)
are ignored.  Mesh geoms keep their id/body/contype bookkeeping, but their
compiler-recentred frames are not reproduced (no STL processing).
"""

from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# Joint type enum values of MuJoCo (mjtJoint), also assumed by
# mink/constants.py:27-34.
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
_JNT_NAMES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
QPOS_WIDTH = {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}
DOF_WIDTH = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}

# Geom type enum values of MuJoCo (mjtGeom).
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3
GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 4, 5, 6, 7
_GEOM_NAMES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE, "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}

MJ_MAXVAL = 1e10  # mujoco.mjMAXVAL
MJ_MINVAL = 1e-15  # mujoco.mjMINVAL


def _floats(text: str) -> List[float]:
    return [float(t) for t in text.split()]


def _normalize_quat(q) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    n = math.sqrt(float(q @ q))
    if n < MJ_MINVAL:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return q / n


def _quat_mul(a, b) -> np.ndarray:
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def _quat_to_mat(q) -> np.ndarray:
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _z_to_quat(vec) -> np.ndarray:
    """Quaternion of the smallest rotation that takes the z axis to `vec` (what the MuJoCo
    compiler uses for `zaxis` and `fromto`): axis z x vec, angle atan2(|z x vec|, vec_z); a
    vector along -z turns about x."""
    v = np.asarray(vec, dtype=np.float64)
    v = v / np.linalg.norm(v)
    axis = np.cross([0.0, 0.0, 1.0], v)
    n = np.linalg.norm(axis)
    axis = np.array([1.0, 0.0, 0.0]) if n < 1e-10 else axis / n
    ang = np.arctan2(n, v[2])
    return np.concatenate([[np.cos(0.5 * ang)], np.sin(0.5 * ang) * axis])


def _mat_to_quat(m) -> np.ndarray:
    """mju_mat2Quat for a rotation matrix (rows m[0..2]), normalised."""
    m = np.asarray(m, dtype=np.float64)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        w = 0.5 * np.sqrt(1.0 + tr)
        q = [w, 0.25 * (m[2, 1] - m[1, 2]) / w, 0.25 * (m[0, 2] - m[2, 0]) / w, 0.25 * (m[1, 0] - m[0, 1]) / w]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        x = 0.5 * np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
        q = [0.25 * (m[2, 1] - m[1, 2]) / x, x, 0.25 * (m[0, 1] + m[1, 0]) / x, 0.25 * (m[0, 2] + m[2, 0]) / x]
    elif m[1, 1] > m[2, 2]:
        y = 0.5 * np.sqrt(1.0 - m[0, 0] + m[1, 1] - m[2, 2])
        q = [0.25 * (m[0, 2] - m[2, 0]) / y, 0.25 * (m[0, 1] + m[1, 0]) / y, y, 0.25 * (m[1, 2] + m[2, 1]) / y]
    else:
        z = 0.5 * np.sqrt(1.0 - m[0, 0] - m[1, 1] + m[2, 2])
        q = [0.25 * (m[1, 0] - m[0, 1]) / z, 0.25 * (m[0, 2] + m[2, 0]) / z, 0.25 * (m[1, 2] + m[2, 1]) / z, z]
    return _normalize_quat(q)


def _axisangle_quat(axis, angle: float) -> np.ndarray:
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < MJ_MINVAL or angle == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    s = math.sin(0.5 * angle)
    return np.concatenate([[math.cos(0.5 * angle)], s * axis / n])


@dataclass
class Model:
    """Flat kinematic table; attribute names mirror ``mujoco.MjModel``."""

    name: str = ""
    nq: int = 0
    nv: int = 0
    nbody: int = 0
    njnt: int = 0
    ngeom: int = 0
    nsite: int = 0
    nkey: int = 0
    nmocap: int = 0
    # bodies
    body_names: List[str] = field(default_factory=list)
    body_parentid: np.ndarray = None
    body_pos: np.ndarray = None
    body_quat: np.ndarray = None
    body_jntadr: np.ndarray = None
    body_jntnum: np.ndarray = None
    body_dofadr: np.ndarray = None
    body_dofnum: np.ndarray = None
    body_weldid: np.ndarray = None
    body_mocapid: np.ndarray = None
    body_geomadr: np.ndarray = None
    body_geomnum: np.ndarray = None
    body_mass: np.ndarray = None
    body_ipos: np.ndarray = None
    body_inertial_known: np.ndarray = None  # False: mass would come from a mesh
    # joints
    jnt_names: List[str] = field(default_factory=list)
    jnt_type: np.ndarray = None
    jnt_qposadr: np.ndarray = None
    jnt_dofadr: np.ndarray = None
    jnt_bodyid: np.ndarray = None
    jnt_pos: np.ndarray = None
    jnt_axis: np.ndarray = None
    jnt_range: np.ndarray = None
    jnt_limited: np.ndarray = None
    qpos0: np.ndarray = None
    # dofs
    dof_bodyid: np.ndarray = None
    dof_jntid: np.ndarray = None
    dof_parentid: np.ndarray = None
    # sites
    site_names: List[str] = field(default_factory=list)
    site_bodyid: np.ndarray = None
    site_pos: np.ndarray = None
    site_quat: np.ndarray = None
    # geoms
    geom_names: List[str] = field(default_factory=list)
    geom_bodyid: np.ndarray = None
    geom_type: np.ndarray = None
    geom_size: np.ndarray = None
    geom_pos: np.ndarray = None
    geom_quat: np.ndarray = None
    geom_contype: np.ndarray = None
    geom_conaffinity: np.ndarray = None
    geom_dataid: np.ndarray = None       # mesh id of a mesh geom, else -1
    # convex hulls of the meshes that collision geoms use (vertices in the geom frame the
    # MJCF gives, scale applied; MuJoCo itself re-centres meshes on their inertial frame and
    # adjusts geom_pos / geom_quat to match, which leaves the geometry in the world unchanged)
    nmesh: int = 0
    mesh_names: List[str] = field(default_factory=list)
    mesh_vertadr: np.ndarray = None
    mesh_vertnum: np.ndarray = None
    mesh_vert: np.ndarray = None         # [sum vertnum, 3]
    mesh_center: np.ndarray = None       # [nmesh, 3] centre of a bounding sphere
    mesh_radius: np.ndarray = None       # [nmesh]
    # keyframes
    key_names: List[str] = field(default_factory=list)
    key_qpos: np.ndarray = None

    _ARRAYS = (
        "body_parentid body_pos body_quat body_jntadr body_jntnum body_dofadr "
        "body_dofnum body_weldid body_mocapid body_geomadr body_geomnum body_mass "
        "body_ipos body_inertial_known jnt_type jnt_qposadr jnt_dofadr jnt_bodyid "
        "jnt_pos jnt_axis jnt_range jnt_limited qpos0 dof_bodyid dof_jntid "
        "dof_parentid site_bodyid site_pos site_quat geom_bodyid geom_type geom_size "
        "geom_pos geom_quat geom_contype geom_conaffinity geom_dataid mesh_vertadr mesh_vertnum "
        "mesh_center mesh_radius key_qpos"
    ).split()
    _INT_ARRAYS = (
        "body_parentid body_jntadr body_jntnum body_dofadr body_dofnum body_weldid "
        "body_mocapid body_geomadr body_geomnum jnt_type jnt_qposadr jnt_dofadr "
        "jnt_bodyid dof_bodyid dof_jntid dof_parentid site_bodyid geom_bodyid "
        "geom_type geom_contype geom_conaffinity geom_dataid mesh_vertadr mesh_vertnum"
    ).split()
    _SCALARS = "name nq nv nbody njnt ngeom nsite nkey nmocap nmesh".split()
    _LISTS = "body_names jnt_names site_names geom_names key_names mesh_names".split()

    # ---- name lookup (the reference uses mj_name2id / model.joint(name)) ----
    def body_id(self, name: str) -> int:
        return self.body_names.index(name) if name in self.body_names else -1

    def joint_id(self, name: str) -> int:
        return self.jnt_names.index(name) if name in self.jnt_names else -1

    def site_id(self, name: str) -> int:
        return self.site_names.index(name) if name in self.site_names else -1

    def geom_id(self, name: str) -> int:
        return self.geom_names.index(name) if name in self.geom_names else -1

    def key_id(self, name: str) -> int:
        return self.key_names.index(name) if name in self.key_names else -1

    # ---- (de)serialisation: the GPU box has no MJCF, only the JSON table ----
    def to_dict(self) -> dict:
        d = {k: getattr(self, k) for k in self._SCALARS + self._LISTS}
        for k in self._ARRAYS:
            a = getattr(self, k)
            d[k] = None if a is None else np.asarray(a).tolist()
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "Model":
        m = cls()
        for k in cls._SCALARS + cls._LISTS:
            if k in d:                      # tables written before a field existed keep its default
                setattr(m, k, d[k])
        for k in cls._ARRAYS:
            v = d.get(k)
            if v is None:
                continue
            if k in cls._INT_ARRAYS:
                arr = np.asarray(v, dtype=np.int32)
            elif k in ("jnt_limited", "body_inertial_known"):
                arr = np.asarray(v, dtype=bool)
            else:
                arr = np.asarray(v, dtype=np.float64)
            setattr(m, k, arr)
        m._fix_empty_shapes()
        return m

    def _fix_empty_shapes(self) -> None:
        def shape(a, cols, dtype):
            a = np.asarray(a, dtype=dtype)
            return a.reshape(-1, cols) if cols else a.reshape(-1)

        self.site_pos = shape(self.site_pos, 3, np.float64)
        self.site_quat = shape(self.site_quat, 4, np.float64)
        self.geom_pos = shape(self.geom_pos, 3, np.float64)
        self.geom_quat = shape(self.geom_quat, 4, np.float64)
        self.geom_size = shape(self.geom_size, 3, np.float64)
        self.jnt_pos = shape(self.jnt_pos, 3, np.float64)
        self.jnt_axis = shape(self.jnt_axis, 3, np.float64)
        self.jnt_range = shape(self.jnt_range, 2, np.float64)
        self.key_qpos = shape(self.key_qpos, self.nq, np.float64) if self.nq else self.key_qpos
        if self.geom_dataid is None:
            self.geom_dataid = np.full(self.ngeom, -1, dtype=np.int32)
        self.mesh_vertadr = shape(self.mesh_vertadr if self.mesh_vertadr is not None else [], 0, np.int32)
        self.mesh_vertnum = shape(self.mesh_vertnum if self.mesh_vertnum is not None else [], 0, np.int32)
        self.mesh_center = shape(self.mesh_center if self.mesh_center is not None else [], 3, np.float64)
        self.mesh_radius = shape(self.mesh_radius if self.mesh_radius is not None else [], 0, np.float64)
        self.mesh_vert = shape(self.mesh_vert if self.mesh_vert is not None else [], 3, np.float64)

    def save_json(self, path: str) -> None:
        """The table as JSON; hull vertices go to a binary side file <path minus .json>.meshes.npz."""
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=0)
        side = os.path.splitext(path)[0] + ".meshes.npz"
        if self.mesh_vert is not None and len(self.mesh_vert):
            np.savez_compressed(side, mesh_vert=np.asarray(self.mesh_vert, dtype=np.float64))
        elif os.path.exists(side):
            os.remove(side)

    @classmethod
    def load_json(cls, path: str) -> "Model":
        with open(path) as f:
            m = cls.from_dict(json.load(f))
        side = os.path.splitext(path)[0] + ".meshes.npz"
        if os.path.exists(side):
            with np.load(side) as z:
                m.mesh_vert = np.asarray(z["mesh_vert"], dtype=np.float64).reshape(-1, 3)
        return m


def load_stl(path: str) -> np.ndarray:
    """Triangles [n, 3, 3] of a binary or ASCII STL file."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        n = int(np.frombuffer(data[80:84], dtype="<u4")[0])
        if 84 + 50 * n == len(data):
            rec = np.frombuffer(data[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
            return rec["v"].astype(np.float64)
    verts = [[float(x) for x in line.split()[1:4]] for line in data.decode("ascii", "replace").splitlines()
             if line.strip().startswith("vertex")]
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3, 3)


def mesh_volume_and_com(tri: np.ndarray):
    """Signed volume and centre of mass of a closed triangle mesh (divergence theorem)."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    v6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = v6.sum() / 6.0
    com = ((a + b + c) / 4.0 * v6[:, None]).sum(axis=0) / 6.0 / vol if abs(vol) > 0 else tri.reshape(-1, 3).mean(axis=0)
    return abs(vol), com


def convex_hull_vertices(points: np.ndarray) -> np.ndarray:
    """Vertices of the convex hull of a point set, in a reproducible (lexicographic) order."""
    from scipy.spatial import ConvexHull
    pts = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 3), axis=0)
    hull = pts[ConvexHull(pts).vertices]
    return hull[np.lexsort((hull[:, 2], hull[:, 1], hull[:, 0]))]


class _Defaults:
    """Tree of ``<default class=...>`` nodes; lookups walk up to ``main``."""

    def __init__(self):
        self.parent: Dict[str, Optional[str]] = {"main": None}
        self.attrs: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def add(self, elem: ET.Element, parent: Optional[str]) -> None:
        cls = elem.get("class", "main" if parent is None else None)
        if cls is None:
            raise ValueError("nested <default> needs a class attribute")
        if cls not in self.parent:
            self.parent[cls] = parent
            self.attrs[cls] = {}
        for child in elem:
            if child.tag == "default":
                self.add(child, cls)
            else:
                self.attrs[cls].setdefault(child.tag, {}).update(child.attrib)

    def resolve(self, tag: str, cls: str, own: Dict[str, str]) -> Dict[str, str]:
        chain = []
        c: Optional[str] = cls
        while c is not None:
            if c not in self.parent:
                raise ValueError(f"unknown default class '{c}'")
            chain.append(c)
            c = self.parent[c]
        out: Dict[str, str] = {}
        for c in reversed(chain):
            out.update(self.attrs[c].get(tag, {}))
        out.update(own)
        return out


def _expand_includes(elem: ET.Element, base_dir: str) -> None:
    """Splice ``<include file=...>`` children in place (recursively)."""
    i = 0
    while i < len(elem):
        child = elem[i]
        if child.tag == "include":
            path = os.path.join(base_dir, child.get("file"))
            inc_root = ET.parse(path).getroot()
            _expand_includes(inc_root, base_dir)
            elem.remove(child)
            for k, sub in enumerate(list(inc_root)):
                elem.insert(i + k, sub)
            i += len(inc_root)
        else:
            _expand_includes(child, base_dir)
            i += 1


class _Compiler:
    def __init__(self, root: ET.Element):
        self.root = root
        self.angle_scale = math.pi / 180.0  # MJCF default is degrees
        self.autolimits = True
        self.eulerseq = "xyz"
        for comp in root.findall("compiler"):
            if comp.get("angle") == "radian":
                self.angle_scale = 1.0
            elif comp.get("angle") == "degree":
                self.angle_scale = math.pi / 180.0
            if comp.get("autolimits") is not None:
                self.autolimits = comp.get("autolimits") == "true"
            if comp.get("eulerseq") is not None:
                self.eulerseq = comp.get("eulerseq")
        self.meshdir = ""
        for comp in root.findall("compiler"):
            if comp.get("meshdir") is not None:
                self.meshdir = comp.get("meshdir")
        self.defaults = _Defaults()
        for d in root.findall("default"):
            self.defaults.add(d, None)
        self.base_dir = "."
        self.meshes: List[dict] = []      # {"name", "file", "scale"}
        for asset in root.findall("asset"):
            for me in asset.findall("mesh"):
                a = self.defaults.resolve("mesh", me.get("class", "main"), me.attrib)
                fname = a.get("file", "")
                name = a.get("name") or os.path.splitext(os.path.basename(fname))[0]
                scale = (_floats(a.get("scale", "1 1 1")) + [1.0, 1.0, 1.0])[:3]
                self.meshes.append({"name": name, "file": fname, "scale": np.array(scale)})

        self.bodies: List[dict] = []
        self.joints: List[dict] = []
        self.sites: List[dict] = []
        self.geoms_by_body: Dict[int, List[dict]] = {}

    # -- orientation of an element frame from quat / euler / axisangle --
    def _orientation(self, a: Dict[str, str]) -> np.ndarray:
        if "quat" in a:
            return _normalize_quat(_floats(a["quat"]))
        if "euler" in a:
            e = [v * self.angle_scale for v in _floats(a["euler"])]
            q = np.array([1.0, 0.0, 0.0, 0.0])
            for ch, ang in zip(self.eulerseq, e):
                axis = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ch.lower()]
                r = _axisangle_quat(axis, ang)
                # lower case: intrinsic (rotate about the moving axes)
                q = _quat_mul(q, r) if ch.islower() else _quat_mul(r, q)
            return _normalize_quat(q)
        if "axisangle" in a:
            v = _floats(a["axisangle"])
            return _axisangle_quat(v[:3], v[3] * self.angle_scale)
        if "xyaxes" in a:
            # x axis as given, y made orthogonal to it, z = x cross y; the frame's axes are the
            # columns of its rotation matrix
            v = _floats(a["xyaxes"])
            x = np.array(v[:3]) / np.linalg.norm(v[:3])
            y = np.array(v[3:6]) - np.dot(x, v[3:6]) * x
            y = y / np.linalg.norm(y)
            return _mat_to_quat(np.column_stack([x, y, np.cross(x, y)]))
        if "zaxis" in a:
            return _z_to_quat(_floats(a["zaxis"]))
        return np.array([1.0, 0.0, 0.0, 0.0])

    def _visit_body(self, elem: ET.Element, parent_id: int, childclass: str) -> None:
        childclass = elem.get("childclass", childclass)
        bid = len(self.bodies)
        body = {
            "name": elem.get("name", ""),
            "parent": parent_id,
            "pos": np.array(_floats(elem.get("pos", "0 0 0"))),
            "quat": self._orientation(elem.attrib),
            "mocap": elem.get("mocap", "false") == "true",
            "joints": [],
            "inertial": None,
        }
        self.bodies.append(body)
        self.geoms_by_body[bid] = []
        self._visit_body_children(elem, bid, childclass)

    def _visit_body_children(self, elem: ET.Element, bid: int, childclass: str) -> None:
        body = self.bodies[bid]
        for child in elem:
            tag = child.tag
            if tag == "body":
                continue
            cls = child.get("class", childclass)
            if tag in ("joint", "freejoint"):
                a = self.defaults.resolve("joint", cls, child.attrib) if tag == "joint" else dict(child.attrib)
                jtype = JNT_FREE if tag == "freejoint" else _JNT_NAMES[a.get("type", "hinge")]
                rng = _floats(a["range"]) if "range" in a else [0.0, 0.0]
                limited_attr = a.get("limited", "auto")
                if limited_attr == "auto":
                    limited = self.autolimits and "range" in a
                else:
                    limited = limited_attr == "true"
                is_rot = jtype in (JNT_HINGE, JNT_BALL)
                scale = self.angle_scale if is_rot else 1.0
                axis = np.array(_floats(a.get("axis", "0 0 1")))
                nrm = np.linalg.norm(axis)
                axis = axis / nrm if nrm > MJ_MINVAL else axis
                ref = float(a.get("ref", "0")) * (self.angle_scale if jtype == JNT_HINGE else 1.0)
                j = {
                    "name": a.get("name", ""),
                    "type": jtype,
                    "body": bid,
                    "pos": np.array(_floats(a.get("pos", "0 0 0"))),
                    "axis": axis if jtype in (JNT_HINGE, JNT_SLIDE) else np.array([0.0, 0.0, 1.0]),
                    "range": np.array(rng) * scale,
                    "limited": bool(limited),
                    "ref": ref,
                }
                if jtype == JNT_FREE:
                    j["pos"] = np.zeros(3)
                body["joints"].append(len(self.joints))
                self.joints.append(j)
            elif tag == "inertial":
                body["inertial"] = {
                    "mass": float(child.get("mass")),
                    "pos": np.array(_floats(child.get("pos", "0 0 0"))),
                }
            elif tag == "site":
                a = self.defaults.resolve("site", cls, child.attrib)
                self.sites.append({
                    "name": a.get("name", ""),
                    "body": bid,
                    "pos": np.array(_floats(a.get("pos", "0 0 0"))),
                    "quat": self._orientation(a),
                })
            elif tag == "geom":
                a = self.defaults.resolve("geom", cls, child.attrib)
                if "type" in a:
                    gtype = _GEOM_NAMES[a["type"]]
                else:
                    gtype = GEOM_MESH if "mesh" in a else GEOM_SPHERE
                size = _floats(a.get("size", "0 0 0"))
                size = (size + [0.0, 0.0, 0.0])[:3]
                gpos = np.array(_floats(a.get("pos", "0 0 0")))
                gquat = self._orientation(a)
                if "fromto" in a:
                    # a segment instead of pos / quat / length: centre at its midpoint, z axis along
                    # from - to, half-length in the size slot of the geom's long axis (capsule,
                    # cylinder: size[1]; box, ellipsoid: size[2])
                    if gtype not in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID):
                        raise ValueError("fromto needs a capsule, cylinder, box or ellipsoid geom")
                    ft = np.array(_floats(a["fromto"]))
                    vec = ft[:3] - ft[3:]
                    size[1 if gtype in (GEOM_CAPSULE, GEOM_CYLINDER) else 2] = 0.5 * float(np.linalg.norm(vec))
                    gpos = 0.5 * (ft[:3] + ft[3:])
                    gquat = _z_to_quat(vec)
                self.geoms_by_body[bid].append({
                    "name": a.get("name", ""),
                    "body": bid,
                    "type": gtype,
                    "size": np.array(size),
                    "pos": gpos,
                    "quat": gquat,
                    "contype": int(a.get("contype", "1")),
                    "conaffinity": int(a.get("conaffinity", "1")),
                    "mass": float(a["mass"]) if "mass" in a else None,
                    "density": float(a.get("density", "1000")),
                    "mesh": a.get("mesh"),
                })
        for child in elem:
            if child.tag == "body":
                self._visit_body(child, bid, childclass)

    def compile(self) -> Model:
        # World body (id 0); every <worldbody> section contributes to it.
        self.bodies.append({
            "name": "world", "parent": 0, "pos": np.zeros(3),
            "quat": np.array([1.0, 0, 0, 0]), "mocap": False, "joints": [], "inertial": None,
        })
        self.geoms_by_body[0] = []
        # Non-body children of all worldbody sections first (they belong to body 0),
        # then the body subtrees, depth-first in document order.
        worlds = self.root.findall("worldbody")
        for wb in worlds:
            saved = [c for c in wb if c.tag == "body"]
            for c in saved:
                wb.remove(c)
            self._visit_body_children(wb, 0, "main")
            for c in saved:
                wb.append(c)
        for wb in worlds:
            for c in wb:
                if c.tag == "body":
                    self._visit_body(c, 0, "main")
        return self._assemble()

    @staticmethod
    def _geom_mass_and_com(g: dict):
        """Mass and local centre for primitive geoms; None for meshes."""
        s = g["size"]
        t = g["type"]
        if t == GEOM_BOX:
            vol = 8.0 * s[0] * s[1] * s[2]
        elif t == GEOM_SPHERE:
            vol = 4.0 / 3.0 * math.pi * s[0] ** 3
        elif t == GEOM_CYLINDER:
            vol = math.pi * s[0] ** 2 * 2.0 * s[1]
        elif t == GEOM_CAPSULE:
            vol = math.pi * s[0] ** 2 * 2.0 * s[1] + 4.0 / 3.0 * math.pi * s[0] ** 3
        elif t == GEOM_ELLIPSOID:
            vol = 4.0 / 3.0 * math.pi * s[0] * s[1] * s[2]
        elif t == GEOM_PLANE:
            return 0.0, g["pos"]
        else:
            return None
        mass = g["mass"] if g["mass"] is not None else g["density"] * vol
        return mass, g["pos"]

    def _assemble(self) -> Model:
        m = Model(name=self.root.get("model", ""))
        nb = len(self.bodies)
        m.nbody = nb
        m.body_names = [b["name"] for b in self.bodies]
        m.body_parentid = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        m.body_pos = np.array([b["pos"] for b in self.bodies], dtype=np.float64)
        m.body_quat = np.array([b["quat"] for b in self.bodies], dtype=np.float64)

        # Joints are already in body (depth-first) order.
        nj = len(self.joints)
        m.njnt = nj
        m.jnt_names = [j["name"] for j in self.joints]
        m.jnt_type = np.array([j["type"] for j in self.joints], dtype=np.int32)
        m.jnt_bodyid = np.array([j["body"] for j in self.joints], dtype=np.int32)
        m.jnt_pos = np.array([j["pos"] for j in self.joints], dtype=np.float64).reshape(-1, 3)
        m.jnt_axis = np.array([j["axis"] for j in self.joints], dtype=np.float64).reshape(-1, 3)
        m.jnt_range = np.array([j["range"] for j in self.joints], dtype=np.float64).reshape(-1, 2)
        m.jnt_limited = np.array([j["limited"] for j in self.joints], dtype=bool)
        qadr, vadr = [], []
        nq = nv = 0
        for j in self.joints:
            qadr.append(nq)
            vadr.append(nv)
            nq += QPOS_WIDTH[j["type"]]
            nv += DOF_WIDTH[j["type"]]
        m.nq, m.nv = nq, nv
        m.jnt_qposadr = np.array(qadr, dtype=np.int32)
        m.jnt_dofadr = np.array(vadr, dtype=np.int32)

        m.body_jntnum = np.array([len(b["joints"]) for b in self.bodies], dtype=np.int32)
        m.body_jntadr = np.array(
            [b["joints"][0] if b["joints"] else -1 for b in self.bodies], dtype=np.int32)
        m.body_dofnum = np.array(
            [sum(DOF_WIDTH[self.joints[j]["type"]] for j in b["joints"]) for b in self.bodies],
            dtype=np.int32)
        m.body_dofadr = np.array(
            [vadr[b["joints"][0]] if b["joints"] else -1 for b in self.bodies], dtype=np.int32)

        # weld id: nearest ancestor-or-self that owns degrees of freedom (world = 0).
        weld = np.zeros(nb, dtype=np.int32)
        for b in range(1, nb):
            weld[b] = b if m.body_dofnum[b] > 0 else weld[m.body_parentid[b]]
        m.body_weldid = weld

        mocap = np.full(nb, -1, dtype=np.int32)
        k = 0
        for b in range(nb):
            if self.bodies[b]["mocap"]:
                mocap[b] = k
                k += 1
        m.body_mocapid = mocap
        m.nmocap = k

        # qpos0: joint reference; free joints take the body's own frame.
        qpos0 = np.zeros(nq)
        for jid, j in enumerate(self.joints):
            a = qadr[jid]
            if j["type"] == JNT_FREE:
                qpos0[a:a + 3] = self.bodies[j["body"]]["pos"]
                qpos0[a + 3:a + 7] = self.bodies[j["body"]]["quat"]
            elif j["type"] == JNT_BALL:
                qpos0[a:a + 4] = [1.0, 0.0, 0.0, 0.0]
            else:
                qpos0[a] = j["ref"]
        m.qpos0 = qpos0

        # dof tables
        dof_body, dof_jnt = [], []
        for jid, j in enumerate(self.joints):
            for _ in range(DOF_WIDTH[j["type"]]):
                dof_body.append(j["body"])
                dof_jnt.append(jid)
        m.dof_bodyid = np.array(dof_body, dtype=np.int32)
        m.dof_jntid = np.array(dof_jnt, dtype=np.int32)
        # parent dof: previous dof on the same body, else last dof of the nearest
        # ancestor that has any.
        last_dof_of_body = np.full(nb, -1, dtype=np.int32)
        for b in range(nb):
            if m.body_dofnum[b] > 0:
                last_dof_of_body[b] = m.body_dofadr[b] + m.body_dofnum[b] - 1
        dof_parent = np.full(nv, -1, dtype=np.int32)
        for d in range(nv):
            b = dof_body[d]
            if d > m.body_dofadr[b]:
                dof_parent[d] = d - 1
            else:
                p = m.body_parentid[b]
                while p != 0 and last_dof_of_body[p] < 0:
                    p = m.body_parentid[p]
                dof_parent[d] = last_dof_of_body[p] if p != 0 else -1
        m.dof_parentid = dof_parent

        # sites (document order within depth-first body order)
        self.sites.sort(key=lambda s: s["body"])  # stable: keeps order inside a body
        m.nsite = len(self.sites)
        m.site_names = [s["name"] for s in self.sites]
        m.site_bodyid = np.array([s["body"] for s in self.sites], dtype=np.int32)
        m.site_pos = np.array([s["pos"] for s in self.sites], dtype=np.float64).reshape(-1, 3)
        m.site_quat = np.array([s["quat"] for s in self.sites], dtype=np.float64).reshape(-1, 4)

        # geoms grouped by body id
        geoms: List[dict] = []
        gadr = np.full(nb, -1, dtype=np.int32)
        gnum = np.zeros(nb, dtype=np.int32)
        for b in range(nb):
            gl = self.geoms_by_body[b]
            if gl:
                gadr[b] = len(geoms)
                gnum[b] = len(gl)
                geoms.extend(gl)
        m.ngeom = len(geoms)
        m.body_geomadr, m.body_geomnum = gadr, gnum
        m.geom_names = [g["name"] for g in geoms]
        m.geom_bodyid = np.array([g["body"] for g in geoms], dtype=np.int32)
        m.geom_type = np.array([g["type"] for g in geoms], dtype=np.int32)
        m.geom_size = np.array([g["size"] for g in geoms], dtype=np.float64).reshape(-1, 3)
        m.geom_pos = np.array([g["pos"] for g in geoms], dtype=np.float64).reshape(-1, 3)
        m.geom_quat = np.array([g["quat"] for g in geoms], dtype=np.float64).reshape(-1, 4)
        m.geom_contype = np.array([g["contype"] for g in geoms], dtype=np.int32)
        m.geom_conaffinity = np.array([g["conaffinity"] for g in geoms], dtype=np.int32)

        # meshes: triangles (scaled) when the file can be read; convex hulls for the meshes of
        # geoms that can collide; volume / centre of mass for bodies without <inertial>
        names = [me["name"] for me in self.meshes]
        m.nmesh = len(self.meshes)
        m.mesh_names = names
        tris: List[Optional[np.ndarray]] = []
        for me in self.meshes:
            path = os.path.join(self.base_dir, self.meshdir, me["file"])
            tris.append(load_stl(path) * me["scale"] if os.path.isfile(path) else None)
        dataid = np.full(m.ngeom, -1, dtype=np.int32)
        collides = np.zeros(m.nmesh, dtype=bool)
        for gi, g in enumerate(geoms):
            if g["type"] == GEOM_MESH and g.get("mesh") in names:
                dataid[gi] = names.index(g["mesh"])
                if g["contype"] or g["conaffinity"]:
                    collides[dataid[gi]] = True
        m.geom_dataid = dataid
        vadr_m, vnum_m, verts, centers, radii = [], [], [], [], []
        for k in range(m.nmesh):
            vadr_m.append(sum(vnum_m))
            if tris[k] is not None and collides[k]:
                hull = convex_hull_vertices(tris[k])
                lo, hi = hull.min(axis=0), hull.max(axis=0)
                ctr = 0.5 * (lo + hi)
                vnum_m.append(len(hull))
                verts.append(hull)
                centers.append(ctr)
                radii.append(float(np.linalg.norm(hull - ctr, axis=1).max()))
            else:
                vnum_m.append(0)
                centers.append(np.zeros(3))
                radii.append(0.0)
        m.mesh_vertadr = np.array(vadr_m, dtype=np.int32)
        m.mesh_vertnum = np.array(vnum_m, dtype=np.int32)
        m.mesh_vert = np.concatenate(verts, axis=0) if verts else np.zeros((0, 3))
        m.mesh_center = np.array(centers, dtype=np.float64).reshape(-1, 3)
        m.mesh_radius = np.array(radii, dtype=np.float64)
        for gi, g in enumerate(geoms):      # geom_size of a mesh geom: half extents of its hull (as MuJoCo reports an aabb)
            k = dataid[gi]
            if k >= 0 and vnum_m[k] > 0:
                h = m.mesh_vert[vadr_m[k]:vadr_m[k] + vnum_m[k]]
                m.geom_size[gi] = 0.5 * (h.max(axis=0) - h.min(axis=0))

        def mesh_mass(g):
            k = names.index(g["mesh"]) if g.get("mesh") in names else -1
            if k < 0 or tris[k] is None:
                return None
            vol, com = mesh_volume_and_com(tris[k])
            rot = _quat_to_mat(g["quat"])
            mass_g = g["mass"] if g["mass"] is not None else g["density"] * vol
            return mass_g, g["pos"] + rot @ com

        # inertials: explicit <inertial> wins; else the geoms' volumes (meshes: the volume the
        # triangles enclose; unknown when the mesh file is not available).
        mass = np.zeros(nb)
        ipos = np.zeros((nb, 3))
        known = np.ones(nb, dtype=bool)
        for b in range(1, nb):
            inert = self.bodies[b]["inertial"]
            if inert is not None:
                mass[b] = inert["mass"]
                ipos[b] = inert["pos"]
                continue
            tot, acc = 0.0, np.zeros(3)
            for g in self.geoms_by_body[b]:
                mc = mesh_mass(g) if g["type"] == GEOM_MESH else self._geom_mass_and_com(g)
                if mc is None:
                    known[b] = False
                    continue
                tot += mc[0]
                acc += mc[0] * np.asarray(mc[1])
            mass[b] = tot
            ipos[b] = acc / tot if tot > 0 else np.zeros(3)
        m.body_mass, m.body_ipos, m.body_inertial_known = mass, ipos, known

        # keyframes
        keys = [k for kf in self.root.findall("keyframe") for k in kf.findall("key")]
        m.nkey = len(keys)
        m.key_names = [k.get("name", "") for k in keys]
        kq = np.tile(qpos0, (len(keys), 1)) if keys else np.zeros((0, nq))
        for i, k in enumerate(keys):
            if k.get("qpos") is None:
                continue
            vals = _floats(k.get("qpos"))
            if len(vals) != nq:
                raise ValueError(f"key '{k.get('name')}': expected {nq} qpos values, got {len(vals)}")
            kq[i] = vals
            for jid, j in enumerate(self.joints):
                a = qadr[jid]
                if j["type"] == JNT_FREE:
                    kq[i, a + 3:a + 7] = _normalize_quat(kq[i, a + 3:a + 7])
                elif j["type"] == JNT_BALL:
                    kq[i, a:a + 4] = _normalize_quat(kq[i, a:a + 4])
        m.key_qpos = kq
        return m


def load_mjcf(path: str) -> Model:
    """Compile the MJCF file at ``path`` (with its includes) into a Model."""
    tree = ET.parse(path)
    root = tree.getroot()
    _expand_includes(root, os.path.dirname(os.path.abspath(path)))
    comp = _Compiler(root)
    comp.base_dir = os.path.dirname(os.path.abspath(path))
    return comp.compile()


def load_mjcf_string(xml: str, base_dir: str = ".") -> Model:
    """Compile an MJCF document given as text (mujoco.MjModel.from_xml_string, as the
    reference's tests build their small models); files it names are looked up under base_dir."""
    root = ET.fromstring(xml)
    _expand_includes(root, os.path.abspath(base_dir))
    comp = _Compiler(root)
    comp.base_dir = os.path.abspath(base_dir)
    return comp.compile()


_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_builtin(name: str) -> Model:
    """Load a precompiled model table shipped with the package, e.g.
    ``tidybot_cube`` (the reference's interactive_scripts/stanford_tidybot/cube.xml)."""
    return Model.load_json(os.path.join(_DATA_DIR, name + ".json"))
