"""VtkExporter of the reference (platform/desktop/vtk_exporter.rs:31-254): one legacy-VTK POLYDATA file per snapshot plus a
ParaView `<basename>.vtk.series` index.  Host-side observability, nothing here touches the device except field downloads.

Same content as `write_vtk_file` (:82-152) / `write_vtk_file2` (:256-367): particles as VERTICES, the Sdf2D polygon edges as
LINES (two extra points per edge, point data padded with zeros for them), point data in the reference's order and names --
density, density_error, density_error2, pressure, mass, aii, h, ppe_source_term, [distances, lambda], velocity, pressure_accel
(3-component, z = 0), flag_is_fluid_surface, flag_neighborhood_reduced -- legacy version 4.2, BINARY, big-endian, title
"SPH Particles 1.0" (:334-337).  The reference writes through the `vtkio` crate (not vendored): the byte-exact layout of blank
lines between sections is that crate's and is not pinned here; the files load in ParaView / any legacy-VTK reader.

Differences, deliberate: an SdfPlane boundary makes the reference hit `todo!()` (:124-126) -- here it simply contributes no
line cells; `density_error2` is never written on the step path (simulation.rs:305, 1483) and is exported as its default 0.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

from .scene import BoundaryPolygon


def _plane_distance(planes, pos: np.ndarray) -> np.ndarray:
    """distance_to_boundary (boundary_winchenbach2020.rs:320-325) for SdfPlane::probe = dir . x + delta (sdf_plane.rs:36-38)."""
    d = np.full(len(pos), np.inf, np.float32)
    for pl in planes:
        dx, dy, delta = (np.float32(v) for v in pl)
        d = np.minimum(d, pos[:, 0] * dx + pos[:, 1] * dy + delta)
    return d


def _polygon_distance(points: Sequence[Tuple[float, float]], pos: np.ndarray) -> np.ndarray:
    """Sdf2D::probe for one connected component (sdf2d.rs:36-160, 182-210): per edge, in order, the edge itself if the point
    projects inside it, then the edge's start corner; the sign of a corner comes from its pseudo-normal (sum of the left
    normals of the two adjoining edges).  Positive on the air side (air on the left of every edge)."""
    P = np.asarray(points, np.float32)
    n = len(P)
    d = np.roll(P, -1, axis=0) - P
    d = (d / np.sqrt((d * d).sum(1, keepdims=True))).astype(np.float32)          # normalized_line_dir
    left = np.stack([-d[:, 1], d[:, 0]], 1)
    pseudo = np.roll(left, 1, axis=0) + left                                       # prev_line_left + next_line_left
    x, y = pos[:, 0].astype(np.float32), pos[:, 1].astype(np.float32)
    best_sq = np.full(len(pos), np.inf, np.float32)
    best = np.zeros(len(pos), np.float32)
    for i in range(n):
        s, t = P[i], P[(i + 1) % n]
        l = t - s
        ll = np.float32(l[0] * l[0] + l[1] * l[1])
        qx, qy = x - s[0], y - s[1]
        proj = qx * d[i, 0] + qy * d[i, 1]
        dist = qx * left[i, 0] + qy * left[i, 1]
        upd = (proj > 0) & (proj * proj < ll) & (dist * dist < best_sq)
        best_sq = np.where(upd, dist * dist, best_sq)
        best = np.where(upd, dist, best)
        dsq = qx * qx + qy * qy
        upd = dsq < best_sq
        sign = np.where(pseudo[i, 0] * qx + pseudo[i, 1] * qy >= 0, np.float32(1), np.float32(-1))
        best = np.where(upd, np.sqrt(dsq) * sign, best)
        best_sq = np.where(upd, dsq, best_sq)
    return best.astype(np.float32)


def write_vtk_file2(path, positions: np.ndarray, data_ft: List[Tuple[str, np.ndarray]], data_vec: List[Tuple[str, np.ndarray]],
                    data_u8: List[Tuple[str, np.ndarray]], lines: List[Tuple[Tuple[float, float], Tuple[float, float]]]):
    """vtk_exporter.rs:256-367."""
    n = len(positions)
    nl = len(lines)
    pts = np.zeros((n + 2 * nl, 3), ">f4")
    pts[:n, :2] = positions
    for k, (a, b) in enumerate(lines):
        pts[n + 2 * k, :2] = a
        pts[n + 2 * k + 1, :2] = b
    verts = np.empty((n, 2), ">i4")
    verts[:, 0] = 1
    verts[:, 1] = np.arange(n)
    with open(path, "wb") as f:
        f.write(b"# vtk DataFile Version 4.2\nSPH Particles 1.0\nBINARY\nDATASET POLYDATA\n")
        f.write(f"POINTS {n + 2 * nl} float\n".encode())
        f.write(pts.tobytes())
        f.write(f"\nVERTICES {n} {2 * n}\n".encode())
        f.write(verts.tobytes())
        if nl:
            li = np.empty((nl, 3), ">i4")
            li[:, 0] = 2
            li[:, 1] = n + 2 * np.arange(nl)
            li[:, 2] = n + 2 * np.arange(nl) + 1
            f.write(f"\nLINES {nl} {3 * nl}\n".encode())
            f.write(li.tobytes())
        f.write(f"\nPOINT_DATA {n + 2 * nl}\n".encode())
        for name, arr in data_ft:
            a = np.zeros(n + 2 * nl, ">f4")
            a[:n] = arr
            f.write(f"SCALARS {name} float 1\nLOOKUP_TABLE default\n".encode())
            f.write(a.tobytes())
            f.write(b"\n")
        for name, arr in data_vec:
            a = np.zeros((n + 2 * nl, 3), ">f4")
            a[:n, :2] = arr
            f.write(f"SCALARS {name} float 3\nLOOKUP_TABLE default\n".encode())
            f.write(a.tobytes())
            f.write(b"\n")
        for name, arr in data_u8:
            a = np.zeros(n + 2 * nl, np.uint8)
            a[:n] = arr
            f.write(f"SCALARS {name} unsigned_char 1\nLOOKUP_TABLE default\n".encode())
            f.write(a.tobytes())
            f.write(b"\n")


class VtkExporter:
    """`VtkExporter::new(folder, basename)` + `add_snapshot(time, ...)` (:31-80); call `close()` (the reference's Drop, :249-253)."""

    def __init__(self, folder, basename: str):
        self.folder = str(folder)
        self.basename = basename
        os.makedirs(self.folder, exist_ok=True)
        self.snapshot_number = 1
        self.series_file = open(os.path.join(self.folder, f"{basename}.vtk.series"), "w")
        self.series_file.write('{\n"file-series-version": "1.0",\n"files": [')

    def add_snapshot(self, time: float, sim, planes):
        """`sim`: FluidSimulation mirror (simulation.py); `planes`: what scene.boundary_planes returned for it."""
        if self.snapshot_number > 1:
            self.series_file.write(",")
        name = f"{self.basename}-{self.snapshot_number:05d}.vtk"
        p = sim.particles
        pos = p.position
        zeros = np.zeros(len(pos), np.float32)
        data_ft = [("density", p.density), ("density_error", p.density_error), ("density_error2", zeros), ("pressure", p.pressure),
                   ("mass", p.mass), ("aii", p.aii), ("h", p.h2), ("ppe_source_term", p.ppe_source_term)]
        lines = []
        if isinstance(planes, BoundaryPolygon):
            pts = planes.points
            lines = [(pts[i], pts[(i + 1) % len(pts)]) for i in range(len(pts))]      # Sdf2D::draw_lines, sdf2d.rs:167-180
            data_ft.append(("distances", _polygon_distance(pts, pos)))
            data_ft.append(("lambda", p.lambda_sum))
        elif planes:
            data_ft.append(("distances", _plane_distance(planes, pos)))
            data_ft.append(("lambda", p.lambda_sum))
        data_vec = [("velocity", p.velocity), ("pressure_accel", p.pressure_accel)]
        data_u8 = [("flag_is_fluid_surface", p.flag_is_fluid_surface), ("flag_neighborhood_reduced", p.flag_neighborhood_reduced)]
        write_vtk_file2(os.path.join(self.folder, name), pos, data_ft, data_vec, data_u8, lines)
        self.series_file.write(f'\n{{ "name": "{name}", "time": {float(time)} }}')
        self.snapshot_number += 1

    def close(self):
        if self.series_file:
            self.series_file.write("\n]\n}")
            self.series_file.close()
            self.series_file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
