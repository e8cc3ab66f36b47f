"""Wire / disk formats at the boundary of the VO path, host side, no OpenCV:

* Middlebury ``.flo`` optical flow -- slam_py/flow_utils.py:10-26, voldor/utils.cpp:23-41: float32 magic 202021.25,
  int32 width, int32 height, float32 [h][w][2], little endian.
* disparity maps -- voldor_slam.py:300-311: ``.flo`` (disparity = -flow_x) or 16-bit PNG (value / 256).
* camera poses as text -- voldor_slam.py:317-329: KITTI (12 numbers of Tcw[:3,:4] per line) and TartanAir
  (tz tx ty qz qx qy qw).

The C entry points ``vk_read_flo`` / ``vk_write_flo`` of libvoldor_hip.so do the same for C++ callers.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

FLO_MAGIC = 202021.25


def load_flow(path):
    """-> float32 [h, w, 2], or None when the magic number is wrong (flow_utils.py:12-21)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12:
            return None
        magic, w, h = struct.unpack("<fii", head)
        if magic != FLO_MAGIC:
            return None
        data = np.frombuffer(f.read(h * w * 8), dtype="<f4")
    if data.size != h * w * 2:
        raise ValueError(f"{path}: truncated .flo ({data.size} of {h * w * 2} floats)")
    return data.reshape(h, w, 2).astype(np.float32)


def save_flow(path, flow):
    flow = np.ascontiguousarray(flow, dtype="<f4")
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError("flow must be [h, w, 2]")
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        f.write(struct.pack("<fii", FLO_MAGIC, w, h))
        f.write(flow.tobytes())


# ---- minimal PNG (8/16-bit grayscale, non-interlaced): all a disparity map needs ----------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def read_png_gray(path):
    """-> uint8 / uint16 [h, w] of a non-interlaced grayscale PNG (what cv2.imread(..., IMREAD_UNCHANGED) returns)."""
    raw = open(path, "rb").read()
    if raw[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG")
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        (n,), typ = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 0 or depth not in (8, 16) or interlace != 0:
        raise ValueError(f"{path}: only non-interlaced 8/16-bit grayscale PNG is supported (type {ctype}, depth {depth})")
    bpp = depth // 8
    stride = w * bpp
    data = zlib.decompress(b"".join(idat))
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = data[y * (stride + 1)]
        line = np.frombuffer(data, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:  # 1 (sub), 3 (average), 4 (Paeth) depend on the reconstructed left neighbour: serial over the line
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                pred = a if ft == 1 else ((a + b) >> 1 if ft == 3 else _paeth(a, b, c))
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 8:
        return out.reshape(h, w)
    return out.reshape(h, w, 2).astype(np.uint16)[..., 0] * 256 + out.reshape(h, w, 2)[..., 1]


def write_png_gray(path, img, filter_type=0):
    """uint8 / uint16 [h, w] -> grayscale PNG (filter_type 0..4 for all scanlines: the test vectors of read_png_gray)."""
    img = np.asarray(img)
    if img.dtype == np.uint16:
        depth, rows = 16, img.astype(">u2").view(np.uint8).reshape(img.shape[0], -1)
    elif img.dtype == np.uint8:
        depth, rows = 8, img
    else:
        raise ValueError("uint8 or uint16 only")
    h, w = img.shape
    bpp = depth // 8
    body = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if filter_type == 0:
            f = cur
        elif filter_type == 1:
            f = cur - a
        elif filter_type == 2:
            f = cur - prev
        elif filter_type == 3:
            f = cur - ((a + prev) >> 1)
        else:
            f = cur - np.array([_paeth(int(x), int(b), int(z)) for x, b, z in zip(a, prev, c)], np.int32)
        body.append(filter_type)
        body += (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(typ, data):
        return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(body), 6)) + chunk(b"IEND", b""))


def load_disparity(path):
    """voldor_slam.py:302-309: '.flo' -> -flow[..., 0]; '.png' -> uint16 / 256; float32 [h, w]."""
    if path.endswith(".flo"):
        fl = load_flow(path)
        if fl is None:
            raise ValueError(f"{path}: bad .flo magic")
        return np.ascontiguousarray(-fl[..., 0])
    if path.endswith(".png"):
        return read_png_gray(path).astype(np.float32) / 256.0
    raise ValueError(f"Unsupported disparity format {path}")


def save_disparity_png(path, disp):
    write_png_gray(path, np.clip(np.rint(np.asarray(disp, np.float64) * 256.0), 0, 65535).astype(np.uint16))


# ---- poses (voldor_slam.py:317-329) --------------------------------------------------------------------------------
def _quat_xyzw(R):
    """Rotation matrix -> unit quaternion (x, y, z, w), w >= 0 branch of scipy's Rotation.as_quat up to sign."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def save_poses(path, Tcw_list, format="KITTI"):
    with open(path, "w") as f:
        for T in Tcw_list:
            T = np.asarray(T)
            if format == "KITTI":
                f.write(" ".join(str(v) for v in T[:3, :4].reshape(-1)) + "\n")
            elif format == "TartanAir":
                q, t = _quat_xyzw(T[:3, :3]), T[:3, 3]
                f.write(f"{t[2]} {t[0]} {t[1]} {q[2]} {q[0]} {q[1]} {q[3]}\n")
            else:
                raise ValueError(f"unknown pose format {format}")


def load_poses_kitti(path):
    """-> float64 [n, 4, 4]"""
    out = []
    for line in open(path):
        v = [float(x) for x in line.split()]
        if not v:
            continue
        if len(v) != 12:
            raise ValueError(f"{path}: expected 12 numbers per line, got {len(v)}")
        T = np.eye(4)
        T[:3, :4] = np.array(v).reshape(3, 4)
        out.append(T)
    return np.stack(out) if out else np.zeros((0, 4, 4))
