"""Binary STL reader with the reference's node de-duplication (restates subs.f90:17-121 in numpy).

Test/fixture tooling only: the STL reader is host I/O and out of the hot-path scope (SURVEY.md
section 2 row 9); it is needed here to regenerate phi0 from the sample surfaces for the oracle's
phi0 check.
"""
import numpy as np


def stl_read(path):
    """Returns (surfX float64 (nNode,3), surfElem int32 (nTri,3) 1-based) like stlRead."""
    with open(path, "rb") as f:
        f.read(80)  # header, subs.f90:38
        ntri = int(np.frombuffer(f.read(4), dtype="<i4")[0])  # :39
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("pad", "<i2")])  # :48-52
        tris = np.frombuffer(f.read(ntri * 50), dtype=rec)
    verts = tris["v"].reshape(-1, 3)  # REAL*4 triangles(3, ntri*3), :49-51
    # subs.f90:69-93: a vertex is shared if all three REAL*4 coordinates differ by < 1e-13 from an
    # already stored node.  For float32 data that is exact equality except between values both
    # smaller than 1e-13; node numbers follow first occurrence.  (The search window quirk
    # `DO kk = 1,nSurfNode` with nSurfNode updated once per triangle only matters for a triangle
    # that repeats one of its own new vertices, i.e. a degenerate triangle.)
    key = np.where(np.abs(verts) < 1e-13, np.float32(0), verts)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)  # unique rows in order of first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    surfElem = (rank[inv.ravel()] + 1).astype(np.int32).reshape(ntri, 3)
    surfX = verts[first[order]].astype(np.float64)  # promoted to REAL(8), :99-103
    return surfX, surfElem


def grid_from_surface(surfX, dx=0.05, dd=10):
    """set3d.f90:90-157: bounding box, nx/ny/nz, xLo for the as-shipped parameters."""
    mn, mx = surfX.min(axis=0), surfX.max(axis=0)
    n = [int(np.ceil((mx[a] - mn[a]) / dx)) + 1 + 2 * dd for a in range(3)]
    xLo = mn - dd * dx
    return n, xLo, mn, mx


def grid_from_surface_pads(surfX, dx, pad_lo, pad_hi):
    """set3d.f90:90-157 with the per-axis pad cells of host edit E4b: n = ceil(extent/dx) + 1 +
    pad_lo + pad_hi, xLo = min - pad_lo*dx.  pad_lo = pad_hi = (dd, dd, dd) is the reference as shipped."""
    mn, mx = surfX.min(axis=0), surfX.max(axis=0)
    n = [int(np.ceil((mx[a] - mn[a]) / dx)) + 1 + int(pad_lo[a]) + int(pad_hi[a]) for a in range(3)]
    xLo = mn - np.asarray(pad_lo, dtype=np.float64) * dx
    return n, xLo, mn, mx


def stl_write(path, surfX, surfElem):
    """Binary STL from nodes + 1-based connectivity (normals zero: the reference reader ignores them)."""
    tri = np.asarray(surfX, dtype=np.float32)[np.asarray(surfElem) - 1]  # (ntri,3,3)
    rec = np.zeros(tri.shape[0], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("pad", "<i2")]))
    rec["v"] = tri
    with open(path, "wb") as f:
        f.write(b"levelsetfortran_amd test surface".ljust(80, b" "))
        f.write(np.int32(tri.shape[0]).tobytes())
        f.write(rec.tobytes())


def vti_read_phi(path, shape):
    """Payload of a .vti file as the reference's writer lays it out (set3d.f90:336-351): '_' + byte count + raw
    Float64, i fastest.  The count is an int32 in the reference (a wrong one: 3 x too large, overflowing at >= 448^3)
    and the true UInt32 / UInt64 (header_type="UInt64") count in lsf_write_vti's files; it is returned for inspection
    by vti_header_count, never trusted here."""
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    head = bytes(raw[:4096])
    tag = b'<AppendedData encoding="raw">'
    k = head.index(tag) + len(tag) + 1  # + lf
    assert head[k:k + 1] == b"_"
    wide = b'header_type="UInt64"' in head
    n = int(np.prod(shape))
    return np.frombuffer(raw, dtype="<f8", count=n, offset=k + 1 + (8 if wide else 4)).reshape(shape, order="F")


def vti_header_count(path):
    """(byte count stored in front of the payload, True if it is a UInt64)"""
    head = open(path, "rb").read(4096)
    tag = b'<AppendedData encoding="raw">'
    k = head.index(tag) + len(tag) + 2
    wide = b'header_type="UInt64"' in head
    return int(np.frombuffer(head, dtype="<u8" if wide else "<u4", count=1, offset=k)[0]), wide
