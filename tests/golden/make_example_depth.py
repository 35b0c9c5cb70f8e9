#!/usr/bin/env python3
"""Fixture generator (build container only): Blender's depth pass of the reference's example view, as DATA.

    python tests/golden/make_example_depth.py        ->  tests/golden/example_depth.npz

Reads /root/reference/example_data/imgs/r_0_depth_0001.exr (800 x 800, three identical float32 channels, ZIP compression: decoded here
with zlib — no OpenEXR in the image) and stores every 8th pixel of it, rows and columns 4, 12, ..., 796 (`z` [100, 100] float32; 1e10 =
no surface), with the pixel indices, and the same pixels of r_0_normal_0001.exr (`normal` [100, 100, 3]: world-space unit normals of that surface).
The depth values are Blender's Z pass: distance ALONG THE VIEW AXIS to the first surface of the object
whose voxelisation is example_data/voxelize/mesh_4_128_1.5_1.165.obj.  tests/test_example_depth.py checks the camera model, the grid's
placement and axis order and the voxeliser against it.  The file holds pixels only — no text of any reference source."""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/example_data/imgs/r_0_depth_0001.exr"
SRC_NORMAL = "/root/reference/example_data/imgs/r_0_normal_0001.exr"


def read_exr_zip(path):
    """{channel: float32 [H, W]} of a scan-line OpenEXR file with FLOAT channels and ZIP (16-line) compression."""
    b = open(path, "rb").read()
    assert b[:4] == bytes.fromhex("762f3101"), "not an OpenEXR file"
    p, hdr = 8, {}
    while True:
        e = b.index(b"\0", p); name = b[p:e].decode(); p = e + 1
        if name == "":
            break
        e = b.index(b"\0", p); typ = b[p:e].decode(); p = e + 1
        n = struct.unpack("<I", b[p:p + 4])[0]; p += 4
        hdr[name] = (typ, b[p:p + n]); p += n
    x0, y0, x1, y1 = struct.unpack("<4i", hdr["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    val, q, chans = hdr["channels"][1], 0, []
    while val[q] != 0:
        e = val.index(b"\0", q); chans.append(val[q:e].decode())
        assert struct.unpack("<I", val[e + 1:e + 5])[0] == 2, "FLOAT channels only"
        q = e + 1 + 16
    assert hdr["compression"][1][0] == 3, "ZIP compression only"
    nb = (H + 15) // 16
    img = np.zeros((len(chans), H, W), np.float32)
    for o in struct.unpack("<%dQ" % nb, b[p:p + 8 * nb]):
        y, size = struct.unpack("<ii", b[o:o + 8])
        d = np.frombuffer(zlib.decompress(b[o + 8:o + 8 + size]), np.uint8).astype(np.int64)
        d = (np.cumsum(np.concatenate([[d[0]], d[1:] - 128])) % 256).astype(np.uint8)          # the predictor: t[i] = t[i-1] + d[i] - 128
        half = (len(d) + 1) // 2
        out = np.empty(len(d), np.uint8); out[0::2] = d[:half]; out[1::2] = d[half:]           # the two byte halves, interleaved again
        rows = min(16, H - (y - y0))
        img[:, y - y0:y - y0 + rows, :] = out.view(np.float32).reshape(rows, len(chans), W).transpose(1, 0, 2)   # per line: channels in name order
    return dict(zip(chans, img))


def main():
    ch = read_exr_zip(SRC)
    z = ch["R"]
    assert z.shape == (800, 800) and np.array_equal(z, ch["G"]) and np.array_equal(z, ch["B"])
    sel = np.arange(4, 800, 8)
    nm = read_exr_zip(SRC_NORMAL)                                       # Blender's normal pass: world-space unit normals, channels R, G, B = x, y, z
    normal = np.stack([nm["R"], nm["G"], nm["B"]], -1)
    out = os.path.join(HERE, "example_depth.npz")
    np.savez_compressed(out, z=z[sel][:, sel].copy(), normal=normal[sel][:, sel].copy(), rows=sel, cols=sel)
    hit = z < 1e9
    print(out, os.path.getsize(out), "bytes; surface on", float(hit.mean()), "of the pixels, depth", float(z[hit].min()), "..", float(z[hit].max()))


if __name__ == "__main__":
    main()
